cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
echo "== host count"; SUBREG_MASK_HOST_COUNT=1 python tools/bench_train.py --steps 100 2>&1 | grep -v amdgpu
echo "== device count"; python tools/bench_train.py --steps 100 2>&1 | grep -v amdgpu
done
