#!/bin/bash
# Re-record the round's measurements on the GPU box (run through gpurun from the repo root):
#   gpurun -- "SUBREG_EVIDENCE_HEAD=$(git rev-parse --short HEAD) bash tools/record_evidence.sh [tag]"     -> everything under gpurun_out/evidence/
# (the box has no .git: SUBREG_EVIDENCE_HEAD is what profiles/traffic.json records as the commit the HBM traffic was measured on)
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
# the SHIPPED configuration (two eval lanes) under the same trace: per-kernel durations stretch where the lanes overlap, so what is read
# from this one is the UNION of the busy intervals (prof_summary.py --busy) against the line's own ms_per_step
rocprofv3 --kernel-trace --output-format csv -d /tmp/ev_bench2 -o b -- python3 $R/bench.py --no-cpu-baseline --sweep-seeds 0 --no-extra-legs > $O/bench_lanes2_profiled.json 2> /dev/null
f=$(find /tmp/ev_bench2 -name "*kernel_trace.csv" | head -1); python3 $R/tools/prof_summary.py "$f" --top 12 --busy > $O/kernel_summary_two_lanes.txt 2>&1
# --- per-layer conv table (HIP events, random data, 20 back-to-back launches per layer)
for b in 256 700 1125; do python3 $R/tools/bench_conv.py --batch $b 2>&1 | $G > $O/conv_layers_b$b.txt; done
python3 $R/tools/bench_conv.py --batch 700 --unfused --only L1 2>&1 | $G > $O/conv_l1_unfused_b700.txt
python3 $R/tools/bench_conv.py --batch 700 --im2col --only L1 2>&1 | $G > $O/conv_l1_im2col_b700.txt
python3 $R/tools/bench_conv.py --batch 256 --dtype f32 2>&1 | $G > $O/conv_layers_f32.txt
# --- round 6: conv_wide.hip forced on every wide layer in both MFMA shapes (16x16x32 = conv_wide16_kernel, the default; 32x32x16 =
#     conv_wide_kernel<2>) beside the general kernel, interleaved; loop-end stamps of both (library variant built with
#     -DSUBREG_WIDE_DIAG=8: tools/build_variants.sh)
for rep in 1 2; do
  for cfg in general wide wide_alt; do
    echo "== kernel $cfg round $rep  (general: conv_fwd.hip; wide: conv_wide.hip 16x16x32; wide_alt: conv_wide.hip 32x32x16), batch 700" >> $O/conv_layers_b700_wide.txt
    python3 $R/tools/bench_conv.py --batch 700 --kernel $cfg 2>&1 | $G | grep "^L[234]" >> $O/conv_layers_b700_wide.txt
  done
done
for cfg in wide wide_alt; do
  echo "== kernel $cfg, batch 700, ALL-ZERO operands (no power limit)" >> $O/conv_layers_b700_wide.txt
  python3 $R/tools/bench_conv.py --batch 700 --kernel $cfg --data zeros 2>&1 | $G | grep "^L[234]" >> $O/conv_layers_b700_wide.txt
done
if [ -f $R/subspace-reg_amd/subreg_hip/libsubreg_wd8.so ]; then
  for tr in 16 32; do
    echo "== SUBREG_WIDE_TR=$tr (loop begin / end stamps and the in-kernel clock only: -DSUBREG_WIDE_DIAG=8)" >> $O/wide_stamps_b700.txt
    SUBREG_WIDE_TR=$tr SUBREG_LIB=$R/subspace-reg_amd/subreg_hip/libsubreg_wd8.so python3 $R/tools/diag_conv.py --batch 700 --kernel wide 2>&1 | $G >> $O/wide_stamps_b700.txt
  done
fi
for v in SUBREG_WIDE=0 SUBREG_WIDE=-1 SUBREG_WIDE=0 SUBREG_WIDE=-1; do
  echo "== $v  (0: conv_fwd.hip everywhere; -1: the dispatcher's rule)" >> $O/forward_ab_wide_rule.txt
  env $v python3 $R/tools/bench_forward.py --lanes 2 --batches 250,500,750,1125 2>&1 | $G >> $O/forward_ab_wide_rule.txt
done
for p in mfma_shape_asm mfma_energy; do
  [ -x $R/tools/probes/$p ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm $R/tools/probes/$p.hip -o $R/tools/probes/$p
  $R/tools/probes/$p > $O/probe_$p.txt 2>&1
done
# --- context: the vendor's libraries on this box, operand values, the eval K split, layer 1's two fused kernels
python3 $R/tools/probes/vendor_ceiling.py 2>&1 | $G > $O/vendor_ceiling.txt
for d in normal zeros narrow half; do echo "== data $d" >> $O/conv_operand_values_b700.txt; python3 $R/tools/bench_conv.py --batch 700 --kernel general --data $d 2>&1 | $G >> $O/conv_operand_values_b700.txt; done
python3 $R/tools/bench_splitk.py 63 125 250 2>&1 | $G > $O/eval_splitk.txt
for d in 0 1 2 3; do SUBREG_EVAL_PREFETCH=$d python3 $R/tools/bench_prefetch.py 125 2>&1 | grep "prefetch depth" >> $O/prefetch.txt; done
for d in 0 2 0 2; do echo "SUBREG_EVAL_PREFETCH=$d" >> $O/route_a_prefetch.txt; SUBREG_EVAL_PREFETCH=$d python3 $R/bench.py --route-a-only 2>/dev/null | cut -c1-260 >> $O/route_a_prefetch.txt; done
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
echo "== the step as ONE replayed hipGraph (train.GraphedStep)" >> $O/train_step.txt
python3 $R/tools/bench_train.py --steps 60 --graph 2>&1 | $G >> $O/train_step.txt
python3 $R/tools/bench_train.py --steps 40 --batch 128 --graph 2>&1 | $G >> $O/train_step.txt
python3 $R/tools/bench_train.py --steps 40 --graph --dropblock 2>&1 | $G >> $O/train_step.txt
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
# --- the 128-image stash-fed bf16 backward against the NumPy oracle (9 minutes of host time; not in the default suite)
(cd $R && SUBREG_TEST_B128=1 python3 -m pytest tests/test_hip_train.py -q -k "own_forward_stash and B128" -s 2>&1 | $G | tail -6 > $O/test_b128_stash_fed.txt)
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
