#!/bin/bash
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_hip_train.py -x -q -k "graphed or another_batch_shape or stash" 2>&1 | grep -v amdgpu.ids | tail -4
