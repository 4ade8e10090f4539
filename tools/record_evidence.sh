#!/bin/bash
# Re-record the round's measurements on the GPU box (run through gpurun from the repo root):
#   bash tools/record_evidence.sh [tag]     -> everything under gpurun_out/evidence/
# Then copy what should be judged into profiles/ as <tag>_* (see profiles/README.md).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
FINAL=$R/gpurun_out/evidence
O=$R/gpurun_out/evidence.tmp.$$          # written here, moved into place only when the run got to its end: a failed run keeps the old evidence
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
G="grep -v amdgpu.ids"
# --- the headline line (default flags) and the same command under the kernel trace.  The traced run uses ONE eval lane
#     (SUBREG_EVAL_LANES=1): with two lanes kernels of both lanes overlap and per-kernel durations are inflated; with one lane
#     sum(kernel time of the forward kernels) / images reproduces roofline.frac of ITS OWN bench line (bench_lanes1.json)
python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
SUBREG_EVAL_LANES=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ev_bench -o b -- python3 $R/bench.py --no-cpu-baseline --sweep-seeds 0 --no-extra-legs > $O/bench_lanes1.json 2> /dev/null
f=$(find /tmp/ev_bench -name "*kernel_stats.csv" | head -1); cut -c1-140 "$f" > $O/kernel_stats.csv
# (--images: the 559 000 images of the 8 timed steps + the 27 000 of the warm-up episode, which the trace contains as well)
f=$(find /tmp/ev_bench -name "*kernel_trace.csv" | head -1); python3 $R/tools/prof_summary.py "$f" --top 25 --conv --images 586000 > $O/kernel_summary.txt 2>&1
# --- per-layer conv table (HIP events, random data, 20 back-to-back launches per layer)
for b in 256 700 1125; do python3 $R/tools/bench_conv.py --batch $b 2>&1 | $G > $O/conv_layers_b$b.txt; done
python3 $R/tools/bench_conv.py --batch 700 --unfused --only L1 2>&1 | $G > $O/conv_l1_unfused_b700.txt
python3 $R/tools/bench_conv.py --batch 700 --im2col --only L1 2>&1 | $G > $O/conv_l1_im2col_b700.txt
python3 $R/tools/bench_conv.py --batch 256 --dtype f32 2>&1 | $G > $O/conv_layers_f32.txt
# --- round 5: the one-wave-per-SIMD kernel (conv_wide.hip) forced on every wide layer, its in-kernel stamps (library variant built with
#     -DSUBREG_WIDE_DIAG=3: `make variant NAME=wd3 EXTRA=-DSUBREG_WIDE_DIAG=3` in subspace-reg_amd/), and the probes behind DESIGN 4.3
for mi in 2 3; do
  echo "== SUBREG_WIDE_MI=$mi (2: 256 x 160 tiles, two workgroups per CU; 3: 384 x 160 tiles, one)" >> $O/conv_layers_b700_wide.txt
  SUBREG_WIDE_MI=$mi python3 $R/tools/bench_conv.py --batch 700 --kernel wide 2>&1 | $G >> $O/conv_layers_b700_wide.txt
  if [ -f $R/subspace-reg_amd/subreg_hip/libsubreg_wd3.so ]; then
    echo "== SUBREG_WIDE_MI=$mi" >> $O/wide_stamps_b700.txt
    SUBREG_WIDE_MI=$mi SUBREG_LIB=$R/subspace-reg_amd/subreg_hip/libsubreg_wd3.so python3 $R/tools/diag_conv.py --batch 700 --kernel wide 2>&1 | $G >> $O/wide_stamps_b700.txt
  fi
done
for v in SUBREG_WIDE=0 SUBREG_WIDE=-1 SUBREG_WIDE=0 SUBREG_WIDE=-1; do
  echo "== $v  (0: conv_fwd.hip everywhere; -1: the dispatcher's rule)" >> $O/forward_ab_wide_rule.txt
  env $v python3 $R/tools/bench_forward.py --lanes 2 --batches 250,500,750,1125 2>&1 | $G >> $O/forward_ab_wide_rule.txt
done
for p in dma_issue dma_slot mfma_shape mfma_shape_bare mfma_energy mfma_operand; do
  [ -x $R/tools/probes/$p ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm $R/tools/probes/$p.hip -o $R/tools/probes/$p
  $R/tools/probes/$p > $O/probe_$p.txt 2>&1
done
# --- context: the vendor's libraries on this box, operand values, the eval K split, layer 1's two fused kernels
python3 $R/tools/probes/vendor_ceiling.py 2>&1 | $G > $O/vendor_ceiling.txt
for d in normal zeros narrow half; do echo "== data $d" >> $O/conv_operand_values_b700.txt; python3 $R/tools/bench_conv.py --batch 700 --kernel general --data $d 2>&1 | $G >> $O/conv_operand_values_b700.txt; done
python3 $R/tools/bench_splitk.py 63 125 250 2>&1 | $G > $O/eval_splitk.txt
for k in auto wide auto wide; do python3 $R/tools/bench_conv.py --batch 700 --only L1.conv1 --kernel $k 2>&1 | $G | sed "s/^/kernel=$k  /" >> $O/l1_wide_fused.txt; done
# (tools/torch_rocm_baseline.py - the reference's route through MIOpen - is NOT part of this script: its find mode takes ~10 minutes per table)
# --- whole forward, A/B of this round's layer-1 changes within one box
for v in "" SUBREG_NO_FUSED12=1 SUBREG_IM2COL_FIRST=1; do
  echo "== ${v:-production}" >> $O/forward_ab_layer1.txt
  env $v python3 $R/tools/bench_forward.py --lanes 2 --batches 250,500,750,1125 2>&1 | $G >> $O/forward_ab_layer1.txt
done
python3 $R/tools/bench_forward.py --lanes 1,2,3 2>&1 | $G > $O/forward_lanes.txt
# --- HBM traffic of the conv stack (separate --pmc passes, counters only with --kernel-trace)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/ev_fetch -o f -- python3 $R/tools/bench_conv.py --batch 256 --iters 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/ev_write -o w -- python3 $R/tools/bench_conv.py --batch 256 --iters 3 > /dev/null 2>&1
ff=$(find /tmp/ev_fetch -name "*counter_collection.csv" | head -1); fw=$(find /tmp/ev_write -name "*counter_collection.csv" | head -1)
(cd $R && python3 tools/traffic_summary.py "$ff" "$fw" 256 bf16 > $O/traffic_layers.txt 2>&1; cp profiles/traffic.json $O/traffic.json)
# --- HBM-bound kernels: streaming rates of the box, then the element-wise / BN kernels against them
[ -x $R/tools/probes/stream_bw ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 $R/tools/probes/stream_bw.hip -o $R/tools/probes/stream_bw
$R/tools/probes/stream_bw 1024 > $O/stream_bw.txt 2>&1
python3 $R/tools/bench_elementwise.py 2>&1 | $G > $O/hbm_kernels.txt
# --- pretraining step
python3 $R/tools/bench_train.py --steps 60 --host-time 2>&1 | $G > $O/train_step.txt
python3 $R/tools/bench_train.py --steps 40 --batch 128 2>&1 | $G >> $O/train_step.txt
python3 $R/tools/bench_train.py --steps 40 --batch 8 2>&1 | $G >> $O/train_step.txt
python3 $R/tools/bench_train.py --steps 20 --batch 512 2>&1 | $G >> $O/train_step.txt
for v in SUBREG_TRAIN_ONE_STREAM=1 SUBREG_NO_SPLITK=1 "SUBREG_TRAIN_ONE_STREAM=1 SUBREG_NO_SPLITK=1"; do
  echo "== $v" >> $O/train_step.txt
  env $v python3 $R/tools/bench_train.py --steps 60 2>&1 | $G >> $O/train_step.txt
done
echo "== all on (again)" >> $O/train_step.txt
python3 $R/tools/bench_train.py --steps 60 2>&1 | $G >> $O/train_step.txt
# kernel summary of the step: ONE stream (with two streams kernels overlap and per-kernel durations stretch), then the two-stream
# step's span / busy time / idle gaps from the trace
SUBREG_TRAIN_ONE_STREAM=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/ev_train -o t -- python3 $R/tools/bench_train.py --steps 10 > /dev/null 2>&1
f=$(find /tmp/ev_train -name "*kernel_trace.csv" | head -1); python3 $R/tools/prof_summary.py "$f" --top 34 > $O/train_kernel_summary.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d /tmp/ev_train2 -o t -- python3 $R/tools/bench_train.py --steps 6 > /dev/null 2>&1
f=$(find /tmp/ev_train2 -name "*kernel_trace.csv" | head -1); python3 $R/tools/prof_summary.py "$f" --timeline 300 --step sgd_pack_train > $O/train_timeline_two_streams.txt 2>&1
# --- SQ / GRBM counters of four layers
bash $R/tools/pmc_layers.sh > $O/pmc_raw.txt 2>&1
# --- multi-rank paths on the one GPU
python3 $R/tools/dp_pretrain_check.py 2>&1 | grep -v "amdgpu.ids\|Gloo\|socket.cpp" > $O/dp_pretrain_check.txt
python3 $R/tools/dp_check.py 2>&1 | grep -v "amdgpu.ids\|Gloo\|socket.cpp" > $O/dp_check.txt
python3 $R/tools/rccl_smoke.py 2>&1 | grep -v "amdgpu.ids" > $O/rccl_single_rank_smoke.txt
# --- a step that died must not pass as evidence: a file with a Python traceback (or an empty one) is replaced by the previous run's
#     file, the run is reported as FAILED and exits non-zero
bad=0
for f in $O/*; do
  if [ ! -s "$f" ] || grep -q "Traceback (most recent call last)" "$f"; then
    bad=$((bad+1)); echo "record_evidence: FAILED STEP -> $(basename $f)"; head -c 600 "$f"
    if [ -f "$FINAL/$(basename $f)" ]; then cp "$FINAL/$(basename $f)" "$f.previous"; fi
    mv "$f" "$f.FAILED"
  fi
done
rm -rf $FINAL && mv $O $FINAL
if [ $bad -gt 0 ]; then echo "evidence INCOMPLETE: $bad step(s) failed"; exit 1; fi
echo evidence done
