#!/bin/bash
mkdir -p gpurun_out/r06
python3 bench.py 2> gpurun_out/r06/bench_final.err > gpurun_out/r06/bench_final.json
python3 -c "
import json; d=json.loads(open('gpurun_out/r06/bench_final.json').read().strip().splitlines()[-1])
print(d['value'], d['episodes_per_s_balanced'], d['roofline']['frac'], d['route_a'], d['feature_reuse']['episodes_per_s'], {k:(v['ms_per_step'], v.get('ms_per_step_eager')) for k,v in d['pretrain']['batches'].items()}, d['cpu_baseline']['value'])
"
