mkdir -p gpurun_out
for cfg in "3" "4" "4 --dtype bf16" "5" "3 --dtype bf16"; do
  tag=$(echo "$cfg" | tr -d ' -' )
  timeout -k 10 280 python bench.py --config $cfg --steps 5 --warmup 2 --breakdown --no-cpu-baseline > gpurun_out/r3d_cfg$tag.json 2> gpurun_out/r3d_cfg$tag.err || { tail -5 gpurun_out/r3d_cfg$tag.err; exit 1; }
  echo "== config $cfg"; python - <<PY
import json
d=json.loads(open('gpurun_out/r3d_cfg$tag.json').read().strip().splitlines()[-1])
print(d['value'], d['unit'], d['ms_per_step'], 'ms', d['dtype'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['bound'])
PY
  grep -B30 "kernels .* ms/step" gpurun_out/r3d_cfg$tag.err | tail -14
done
