#!/bin/bash
mkdir -p gpurun_out/r06
timeout 900 python3 tools/probes/free_running_lanes.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06/free_running_lanes.txt
