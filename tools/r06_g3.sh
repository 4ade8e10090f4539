#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_train.py -x -q -k "bn_train or raw_stats or train_step_against or backbone_train or finalize" 2>&1 | grep -v amdgpu.ids | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o t -- python3 $R/tools/bench_train.py --steps 15 > $O/train_bnfin.txt 2>/dev/null
f=$(find /tmp/tr -name "*kernel_stats.csv" | head -1)
grep -i "bn_finalize\|bn_bwd_finalize" $f | cut -c1-200
tail -3 $O/train_bnfin.txt
