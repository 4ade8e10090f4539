mkdir -p gpurun_out/r06
python -m pytest tests -x -q -m gpu > gpurun_out/r06/t_all.log 2>&1; grep -v "amdgpu.ids" gpurun_out/r06/t_all.log | tail -6
for d in 0 2 0 2; do echo "SUBREG_EVAL_PREFETCH=$d"; SUBREG_EVAL_PREFETCH=$d python bench.py --route-a-only 2>/dev/null | cut -c1-200; done | tee gpurun_out/r06/route_a_prefetch.txt
python bench.py --no-cpu-baseline 2> gpurun_out/r06/bench_b.err | tee gpurun_out/r06/bench_b.json | cut -c1-900
