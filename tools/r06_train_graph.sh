mkdir -p gpurun_out/r06
python -m pytest tests/test_hip_train.py -x -q -k "graphed or own_forward_stash" > gpurun_out/r06/t_graph_full.log 2>&1
grep -v "^  File\|^Extension" gpurun_out/r06/t_graph_full.log | head -120
