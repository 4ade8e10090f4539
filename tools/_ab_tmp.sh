cd $GRAFT_REPO_ROOT
for r in 1 2; do for wv in 512 768 1024 1280; do echo "== WGRAD_WAVES=$wv"; SUBREG_WGRAD_WAVES=$wv python tools/bench_train.py --steps 80 2>&1 | grep -v amdgpu; done; done
for wv in 512 768 1024; do echo "== B=128 WGRAD_WAVES=$wv"; SUBREG_WGRAD_WAVES=$wv python tools/bench_train.py --steps 40 --batch 128 2>&1 | grep -v amdgpu; done
