#!/bin/bash
# scratch: five-slot weight ring of the 128-row wide kernel
mkdir -p gpurun_out/r06
export SUBREG_WIDE_RING=5
timeout 600 python -m pytest tests/test_hip_kernels.py -x -q -k "wide128" 2>&1 | tail -3
timeout 200 python tools/occupancy_steps.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06/occupancy_steps_ring5.txt
timeout 200 python tools/occupancy_steps.py 10 320 320 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06/occupancy_steps_ring5.txt
export SUBREG_WIDE_RING=3
timeout 200 python tools/occupancy_steps.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06/occupancy_steps.txt
timeout 200 python tools/occupancy_steps.py 10 320 320 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r06/occupancy_steps.txt
