for n in 0 1 2 3 5 7 9 12; do python tools/bench_prefetch.py 125 $n 2>&1 | grep "prefetch depth"; done | tee gpurun_out/r06/prefetch_queues2.txt
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --sweep-seeds 0 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(j.get('route_a')))"
python -m pytest tests/test_hip_kernels.py -x -q -k "graphed_eval" 2>&1 | tail -2
