#!/bin/bash
mkdir -p gpurun_out/r06
timeout 2400 python -m pytest tests -x -q -m gpu --durations=25 2>&1 | grep -v amdgpu.ids | tail -45 | tee gpurun_out/r06/gpu_suite_durations.txt
