mkdir -p gpurun_out/r06
python -m pytest tests -x -q -m gpu > gpurun_out/r06/t_all.log 2>&1; grep -v "amdgpu.ids" gpurun_out/r06/t_all.log | tail -15
L=$PWD/subspace-reg_amd/subreg_hip
O=gpurun_out/r06/wide16_vs_general.txt; : > $O
for B in 250 375 500 700 1000 1125; do
  for rep in 1 2; do
    for k in general wide; do
      echo "== batch $B kernel $k round $rep" >> $O
      python tools/bench_conv.py --batch $B --kernel $k 2>&1 | grep "^L[234]" >> $O
    done
  done
done
echo "== loop ends stamps (wide, 16x16x32)" >> $O
SUBREG_LIB=$L/libsubreg_wd8.so python tools/diag_conv.py --batch 700 --kernel wide 2>&1 | grep -v amdgpu.ids >> $O
tail -14 $O
for d in 0 1 2 3; do SUBREG_EVAL_PREFETCH=$d python tools/bench_prefetch.py 125 2>&1 | grep prefetch; done | tee gpurun_out/r06/prefetch.txt
