#!/bin/bash
mkdir -p gpurun_out/r06
( timeout 300 python3 tools/probes/hybrid_tail.py 10 640 640; timeout 300 python3 tools/probes/hybrid_tail.py 21 320 320; timeout 300 python3 tools/probes/hybrid_tail.py 10 320 640 ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06/hybrid_tail.txt
