#!/bin/bash
# The driver's round-end sequence in ONE lease, with a device health probe and a wall-clock stamp after every step:
#   gpurun -- 'bash tools/round_end_sequence.sh [exact|perfile] [repeat]'
#   exact   : pytest -m gpu (whole suite), smoke(), bench.py --gpus 1 --steps 20 --warmup 5   (what the driver runs)
#   perfile : the same, but pytest file by file with a probe after each (to locate a step that leaves the GPU slow)
# Log: gpurun_out/round_end_sequence.txt (copy to profiles/rNN_round_end_sequence.txt).
set -u
MODE=${1:-exact}
REPEAT=${2:-2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
LOG=$O/round_end_sequence.txt
H=$R/tools/probes/health
cd $R
[ -x $H ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/probes/health.hip -o $H
: > $LOG
say() { echo "$@" | tee -a $LOG; }
probe() {
  local t0=$(date +%s%3N)
  timeout 120 $H >> $LOG 2>&1
  local rc=$?
  local t1=$(date +%s%3N)
  say "  probe after [$1]: rc=$rc wall=$(( t1 - t0 )) ms"
  timeout 60 rocm-smi --showuse --showmemuse 2>&1 | grep -E "GPU use|VRAM" | tee -a $LOG
}
step() {   # name, command...
  local name=$1; shift
  local t0=$(date +%s%3N)
  "$@" > $O/res_$name.log 2>&1
  local rc=$?
  local t1=$(date +%s%3N)
  say "step [$name]: rc=$rc wall=$(( t1 - t0 )) ms   | $(grep -v amdgpu.ids $O/res_$name.log | tail -1 | cut -c1-200)"
  probe $name
}
say "round-end sequence, mode=$MODE repeat=$REPEAT, $(date -u +%FT%TZ)"
probe start
for it in $(seq 1 $REPEAT); do
  say "== pass $it"
  if [ "$MODE" = perfile ]; then
    for f in tests/test_hip_kernels.py tests/test_hip_loop.py tests/test_hip_train.py tests/test_multirank.py; do
      step pytest_$(basename $f .py)_$it python -m pytest $f -x -q -m gpu --durations=8
    done
  else
    step pytest_$it python -m pytest tests/ -x -q -m gpu --durations=15
  fi
  step smoke_$it python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')"
  step bench_$it python3 bench.py --gpus 1 --steps 20 --warmup 5
done
say "sequence done $(date -u +%FT%TZ)"
