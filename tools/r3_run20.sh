#!/bin/bash
# final numbers of the round on the last build: the driver's default command, then configs 1, 3 (bf16), 4 (bf16), 5
timeout -k 10 400 python bench.py > gpurun_out/r3x_bench_default.json 2> gpurun_out/r3x_bench_default.err || { tail -5 gpurun_out/r3x_bench_default.err; exit 1; }
cut -c1-330 gpurun_out/r3x_bench_default.json
for spec in "cfg1:--config 1" "cfg3bf16:--config 3 --dtype bf16" "cfg4bf16:--config 4 --dtype bf16" "cfg5:--config 5" "cfg2f32:--config 2 --dtype f32"; do
  tag=$(echo "$spec" | cut -d: -f1); args=$(echo "$spec" | cut -d: -f2)
  timeout -k 10 280 python bench.py $args --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], 'ms', d['value'], 'clips/s', d['roofline']['frac'])"
done
