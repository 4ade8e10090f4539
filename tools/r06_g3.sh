#!/bin/bash
# scratch: two against three eval lanes with this round's kernels, same box, interleaved
mkdir -p gpurun_out/r06
for r in 1 2; do for l in 2 3; do
  echo "lanes=$l" ; SUBREG_EVAL_LANES=$l python3 bench.py --no-cpu-baseline --sweep-seeds 0 --no-extra-legs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['episodes_per_s_balanced'], d['roofline']['frac'])"
done; done | tee gpurun_out/r06/lanes_2_vs_3.txt
