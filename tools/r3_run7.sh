timeout -k 10 200 python tools/gcn_exp.py bf16 bwd ISTGCN_GCN_RC 0,1 2>&1 | grep -v amdgpu.ids | head -3
timeout -k 10 840 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r3g_tests.log 2>&1; rc=$?
tail -4 gpurun_out/r3g_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 240 python bench.py --steps 10 --warmup 3 --breakdown --no-cpu-baseline > gpurun_out/r3g_bench_bf16.json 2> gpurun_out/r3g_bench_bf16.err || exit 1
python - <<PY
import json
d=json.loads(open('gpurun_out/r3g_bench_bf16.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'])
PY
grep -B14 "kernels .* ms/step" gpurun_out/r3g_bench_bf16.err
timeout -k 10 240 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
