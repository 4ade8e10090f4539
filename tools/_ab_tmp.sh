cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_train.py -x -q -m gpu 2>&1 | tail -3
B=$GRAFT_REPO_ROOT/subspace-reg_amd/build/libsubreg_hip_base.so
for r in 1 2; do
echo "== base"; SUBREG_LIB=$B python tools/bench_train.py --steps 30 2>&1 | grep -v amdgpu
echo "== new"; python tools/bench_train.py --steps 30 2>&1 | grep -v amdgpu
done
echo "== base 128"; SUBREG_LIB=$B python tools/bench_train.py --steps 20 --batch 128 2>&1 | grep -v amdgpu
echo "== new 128"; python tools/bench_train.py --steps 20 --batch 128 2>&1 | grep -v amdgpu
