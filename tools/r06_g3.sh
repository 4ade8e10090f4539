mkdir -p gpurun_out/r06
for i in 1 2 3; do python bench.py --steps 4 --warmup 1 --no-cpu-baseline --sweep-seeds 0 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(j.get('route_a'))[:420])"; done
echo "== TN128 experiment (general kernel, L3.1 / L4.x), two rounds"
for rep in 1 2; do for B in 250 500 700 1000; do for e in 0 1; do echo "-- batch $B SUBREG_TN128=$e"; SUBREG_TN128=$e python tools/bench_conv.py --batch $B --kernel general --only L4 2>&1 | grep "^L4"; done; done; done | tee gpurun_out/r06/tn128.txt
SUBREG_TN128=1 python -m pytest tests/test_hip_kernels.py -x -q -k "conv" 2>&1 | tail -3
