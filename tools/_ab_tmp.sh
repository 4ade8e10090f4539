cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "conv" 2>&1 | tail -3
B=$GRAFT_REPO_ROOT/subspace-reg_amd/build/libsubreg_hip_base.so
for r in 1 2; do
echo "== base"; SUBREG_LIB=$B python tools/bench_conv.py --batch 700 --only L3.1 2>&1 | grep -v amdgpu; SUBREG_LIB=$B python tools/bench_conv.py --batch 700 --only L4.1 2>&1 | grep -v amdgpu
echo "== new"; python tools/bench_conv.py --batch 700 --only L3.1 2>&1 | grep -v amdgpu; python tools/bench_conv.py --batch 700 --only L4.1 2>&1 | grep -v amdgpu
done
echo "== base fwd"; SUBREG_LIB=$B python tools/bench_forward.py --lanes 2 --batches 500,750 2>&1 | grep -v amdgpu
echo "== new fwd"; python tools/bench_forward.py --lanes 2 --batches 500,750 2>&1 | grep -v amdgpu
echo "== base fwd"; SUBREG_LIB=$B python tools/bench_forward.py --lanes 2 --batches 500,750 2>&1 | grep -v amdgpu
echo "== new fwd"; python tools/bench_forward.py --lanes 2 --batches 500,750 2>&1 | grep -v amdgpu
