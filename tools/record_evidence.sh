#!/bin/bash
# Re-record the round's measurements on the GPU box (run through gpurun from the repo root):
#   bash tools/record_evidence.sh           -> everything under gpurun_out/evidence/
# Then copy what should be judged into profiles/ (see profiles/README.md).
# Needs subspace-reg_amd/subreg_hip/libsubreg_diag.so (a -DR64_DIAG=1 build of conv64_resident.hip) for the stamp table; skipped if absent.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/evidence
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ev_bench -o b -- python3 $R/bench.py --no-cpu-baseline --sweep-seeds 0 > $O/bench_profiled.json 2> /dev/null
f=$(find /tmp/ev_bench -name "*kernel_stats.csv" | head -1); cut -c1-140 "$f" > $O/kernel_stats.csv
f=$(find /tmp/ev_bench -name "*kernel_trace.csv" | head -1); python3 $R/tools/prof_summary.py "$f" --top 25 --conv > $O/kernel_summary.txt 2>&1
for b in 256 700 1125; do python3 $R/tools/bench_conv.py --batch $b 2>&1 | grep -v amdgpu.ids > $O/conv_layers_b$b.txt; done
python3 $R/tools/bench_conv.py --batch 256 --dtype f32 2>&1 | grep -v amdgpu.ids > $O/conv_layers_f32.txt
SUBREG_NO_RESIDENT64=1 python3 $R/tools/bench_conv.py --batch 700 --only L1.conv 2>&1 | grep -v amdgpu.ids > $O/conv_l1_general_kernel_b700.txt
python3 $R/tools/bench_forward.py --lanes 1,2,3 2>&1 | grep -v amdgpu.ids > $O/forward_lanes.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/ev_fetch -o f -- python3 $R/tools/bench_conv.py --batch 256 --iters 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/ev_write -o w -- python3 $R/tools/bench_conv.py --batch 256 --iters 3 > /dev/null 2>&1
ff=$(find /tmp/ev_fetch -name "*counter_collection.csv" | head -1); fw=$(find /tmp/ev_write -name "*counter_collection.csv" | head -1)
(cd $R && python3 tools/traffic_summary.py "$ff" "$fw" 256 bf16 > $O/traffic_layers.txt 2>&1; cp profiles/traffic.json $O/traffic.json)
python3 $R/tools/bench_train.py --steps 20 2>&1 | grep -v amdgpu.ids > $O/train_step.txt
python3 $R/tools/bench_train.py --steps 20 --batch 128 2>&1 | grep -v amdgpu.ids >> $O/train_step.txt
rocprofv3 --kernel-trace --output-format csv -d /tmp/ev_train -o t -- python3 $R/tools/bench_train.py --steps 10 > /dev/null 2>&1
f=$(find /tmp/ev_train -name "*kernel_trace.csv" | head -1); python3 $R/tools/prof_summary.py "$f" --top 25 > $O/train_kernel_summary.txt 2>&1
python3 $R/tools/dp_pretrain_check.py 2>&1 | grep -v "amdgpu.ids\|Gloo\|socket.cpp" > $O/dp_pretrain_check.txt
python3 $R/tools/dp_check.py 2>&1 | grep -v "amdgpu.ids\|Gloo\|socket.cpp" > $O/dp_check.txt
if [ -f $R/subspace-reg_amd/subreg_hip/libsubreg_diag.so ]; then
  SUBREG_LIB=$R/subspace-reg_amd/subreg_hip/libsubreg_diag.so python3 $R/tools/diag_r64.py 256 2>&1 | grep -v amdgpu.ids > $O/conv64_stamps.txt
fi
cd $R && python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
echo evidence done
