mkdir -p gpurun_out/r06
O=gpurun_out/r06/bench_ab_same_box.txt; : > $O
for rep in 1 2 3; do
  for cfg in "SUBREG_WIDE_TR=32" "SUBREG_WIDE_RULE=1" "SUBREG_WIDE_RULE=3" "SUBREG_WIDE=0"; do
    echo -n "$cfg  " >> $O
    env $cfg python bench.py --no-extra-legs --no-cpu-baseline --sweep-seeds 0 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('episodes/s %.4f  balanced %.4f  ms_per_step %.2f  roofline.frac %.4f' % (j['value'], j['episodes_per_s_balanced'], j['ms_per_step'], j['roofline']['frac']))" >> $O
  done
done
cat $O
python -m pytest tests/test_hip_kernels.py tests/test_hip_loop.py -x -q 2>&1 | tail -3
