# conv_wide16_kernel (round 6): parity through the C ABI, then per-layer times against the 32x32x16 form and the general kernel
mkdir -p gpurun_out/r06
python -m pytest tests/test_hip_kernels.py -x -q -k "conv" > gpurun_out/r06/t_conv16.log 2>&1; tail -5 gpurun_out/r06/t_conv16.log
O=gpurun_out/r06/wide16_vs_32_vs_general.txt; : > $O
for rep in 1 2; do
  for cfg in "general:SUBREG_WIDE_TR=16" "wide:SUBREG_WIDE_TR=32" "wide:SUBREG_WIDE_TR=16"; do
    k=${cfg%%:*}; e=${cfg##*:}
    echo "== kernel $k $e batch 700 round $rep" >> $O
    env $e python tools/bench_conv.py --batch 700 --kernel $k 2>&1 | grep -v amdgpu.ids | grep "^L[234]" >> $O
  done
done
for cfg in "wide:SUBREG_WIDE_TR=32" "wide:SUBREG_WIDE_TR=16"; do
  k=${cfg%%:*}; e=${cfg##*:}
  echo "== kernel $k $e batch 700 ZEROS" >> $O
  env $e python tools/bench_conv.py --batch 700 --kernel $k --data zeros 2>&1 | grep "^L[234]" >> $O
done
cat $O
