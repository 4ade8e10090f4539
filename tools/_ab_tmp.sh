cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
echo "== no split"; SUBREG_NO_SPLITK=1 python tools/bench_train.py --steps 100 2>&1 | grep -v amdgpu
echo "== split-K"; python tools/bench_train.py --steps 100 2>&1 | grep -v amdgpu
done
echo "== one stream, no split"; SUBREG_TRAIN_ONE_STREAM=1 SUBREG_NO_SPLITK=1 python tools/bench_train.py --steps 100 2>&1 | grep -v amdgpu
echo "== one stream, split"; SUBREG_TRAIN_ONE_STREAM=1 python tools/bench_train.py --steps 100 2>&1 | grep -v amdgpu
for b in 128 512 8; do echo "== split B=$b"; python tools/bench_train.py --steps 40 --batch $b 2>&1 | grep -v amdgpu; done
