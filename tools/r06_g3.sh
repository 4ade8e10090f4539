#!/bin/bash
# scratch: n-tile-major XCD mapping of conv_wide16_kernel against the m-major one (variant library), parity first
mkdir -p gpurun_out/r06
L=$(pwd)/subspace-reg_amd/subreg_hip
SUBREG_LIB=$L/libsubreg_nmajor.so timeout 900 python -m pytest tests/test_hip_kernels.py -x -q -k "wide or auto" 2>&1 | tail -2
for b in 350 700 1125; do for r in 1 2; do for lib in hip nmajor; do
  echo "-- batch $b lib $lib"
  SUBREG_LIB=$L/libsubreg_$lib.so python3 tools/bench_conv.py --batch $b --iters 20 --kernel wide256 2>&1 | grep "^L3\|^L4\|^conv stack"
done; done; done | tee gpurun_out/r06/nmajor_layers.txt
