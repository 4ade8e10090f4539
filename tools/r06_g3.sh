#!/bin/bash
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_hip_loop.py -x -q -s -k "freeze" 2>&1 | grep -v amdgpu.ids | grep "backbone update\|passed\|failed\|Error" | tee gpurun_out/r06/freeze_cos.txt
