cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_train.py -x -q -m gpu 2>&1 | tail -3
for r in 1 2; do
echo "== one stream"; SUBREG_TRAIN_ONE_STREAM=1 python tools/bench_train.py --steps 30 2>&1 | grep -v amdgpu
echo "== two streams"; python tools/bench_train.py --steps 30 2>&1 | grep -v amdgpu
done
echo "== one stream 128"; SUBREG_TRAIN_ONE_STREAM=1 python tools/bench_train.py --steps 20 --batch 128 2>&1 | grep -v amdgpu
echo "== two streams 128"; python tools/bench_train.py --steps 20 --batch 128 2>&1 | grep -v amdgpu
echo "== one stream 8"; SUBREG_TRAIN_ONE_STREAM=1 python tools/bench_train.py --steps 20 --batch 8 2>&1 | grep -v amdgpu
echo "== two streams 8"; python tools/bench_train.py --steps 20 --batch 8 2>&1 | grep -v amdgpu
