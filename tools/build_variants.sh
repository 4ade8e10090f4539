#!/bin/bash
# Measurement builds of conv_wide.hip linked with the production objects (run in the build container after `make`):
#   libsubreg_wd8.so   -DSUBREG_WIDE_DIAG=8   loop begin / end stamps + in-kernel clock (tools/diag_conv.py --kernel wide)
#   libsubreg_a3.so    -DSUBREG_W16_AHEAD=3   conv_wide16_kernel with its B fragments read three groups ahead
set -e
cd "$(dirname "$0")/../subspace-reg_amd"
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -Wno-inline-asm -save-temps=obj"
OBJS="build/backbone.o build/backbone_train.o build/backward.o build/classifier.o build/conv64_resident.o build/conv_first.o build/conv_fwd.o build/elementwise.o"
mkdir -p build_wd8 build_a3
/opt/rocm/bin/hipcc $F -DSUBREG_WIDE_DIAG=8 -c csrc/conv_wide.hip -o build_wd8/conv_wide.o
python3 ../tools/check_isa.py build_wd8 csrc/conv_wide.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o subreg_hip/libsubreg_wd8.so $OBJS build_wd8/conv_wide.o
/opt/rocm/bin/hipcc $F -DSUBREG_W16_AHEAD=3 -c csrc/conv_wide.hip -o build_a3/conv_wide.o
python3 ../tools/check_isa.py build_a3 csrc/conv_wide.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o subreg_hip/libsubreg_a3.so $OBJS build_a3/conv_wide.o
ls -la subreg_hip/*.so
