#!/bin/bash
# plain bench lines of all five BASELINE configurations (+ the 16-bit twins of 3 / 4), the per-family breakdown of config 2
# and the PCIe-inclusive run; usage: tools/job_benches.sh <tag>
tag=$1
for spec in "1 f32" "3 f32" "3 bf16" "4 f32" "4 bf16" "5 f16"; do
  set -- $spec
  timeout -k 10 300 python bench.py --config $1 --dtype $2 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_cfg$1_$2.json 2> gpurun_out/${tag}_cfg$1_$2.err || { echo "cfg $1 $2 failed"; tail -3 gpurun_out/${tag}_cfg$1_$2.err; exit 1; }
  python3 -c "import json; d=json.load(open('gpurun_out/${tag}_cfg$1_$2.json')); r=d['roofline']; print('cfg$1 $2', d['ms_per_step'], 'ms', d['value'], 'clips/s |', r['kernel'][:28], r['bound'], r['frac'], '| step', d['step_roofline']['frac'], d['step_roofline']['bound'])"
done
for spec in "1 f32" "3 f32" "3 bf16" "4 f32" "4 bf16" "5 f16" "2 bf16"; do
  set -- $spec
  timeout -k 10 300 python bench.py --config $1 --dtype $2 --steps 6 --warmup 2 --no-cpu-baseline --no-vendor-gemm --breakdown > /dev/null 2> gpurun_out/${tag}_bd_cfg$1_$2.txt || exit 1
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --h2d > gpurun_out/${tag}_h2d.json 2> gpurun_out/${tag}_h2d.err || exit 1
python3 -c "import json; d=json.load(open('gpurun_out/${tag}_h2d.json')); print('h2d', d['ms_per_step'], d['value'], d['config']['input'][:40])"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_default.json 2> gpurun_out/${tag}_default.err || exit 1
python3 -c "import json; d=json.load(open('gpurun_out/${tag}_default.json')); print('default', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline'].get('vendor_gemm_bf16_tflops'), d['step_roofline']['frac'])"
