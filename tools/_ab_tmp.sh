cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_train.py -x -q -m gpu -k "linear or train" 2>&1 | tail -3
for r in 1 2 3; do python tools/bench_train.py --steps 100 2>&1 | grep -v amdgpu; done
python tools/bench_train.py --steps 40 --batch 128 2>&1 | grep -v amdgpu
