mkdir -p gpurun_out/r06
O=gpurun_out/r06/wide16s.txt; : > $O
for B in 125 250 375 500 700 1000; do
  for rep in 1 2; do
    for k in general wide256 wide128; do
      echo "== batch $B kernel $k round $rep" >> $O
      python tools/bench_conv.py --batch $B --kernel $k 2>&1 | grep "^L[234]" | grep -v "does not take" >> $O
    done
  done
done
echo "== loop ends stamps, 128-row tiling forced" >> $O
SUBREG_WIDE_ROWS=128 SUBREG_LIB=$PWD/subspace-reg_amd/subreg_hip/libsubreg_wd8.so python tools/diag_conv.py --batch 700 --kernel wide 2>&1 | grep -v amdgpu.ids >> $O
tail -3 $O
