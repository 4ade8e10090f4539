cd $GRAFT_REPO_ROOT
for r in 1 2; do
for k in 0 1 2 3 4 6; do echo "== bwd fork from block $k (fwd 0)"; SUBREG_TRAIN_FORK_FROM=$k python tools/bench_train.py --steps 80 2>&1 | grep -v amdgpu; done
for k in 1 2 3 6; do echo "== fwd fork from block $k (bwd 0)"; SUBREG_TRAIN_FORK_FWD_FROM=$k python tools/bench_train.py --steps 80 2>&1 | grep -v amdgpu; done
done
for k in 0 2 6; do echo "== B=128 bwd fork from $k"; SUBREG_TRAIN_FORK_FROM=$k python tools/bench_train.py --steps 40 --batch 128 2>&1 | grep -v amdgpu; done
for k in 2 6; do echo "== B=128 fwd fork from $k"; SUBREG_TRAIN_FORK_FWD_FROM=$k python tools/bench_train.py --steps 40 --batch 128 2>&1 | grep -v amdgpu; done
