mkdir -p gpurun_out
for cfg in "2" "2 --dtype f32" "3" "3 --dtype bf16" "4" "4 --dtype bf16" "5" "1"; do
  tag=$(echo "$cfg" | tr -d ' -' )
  timeout -k 10 280 python bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3i_cfg$tag.json 2> gpurun_out/r3i_cfg$tag.err || { tail -5 gpurun_out/r3i_cfg$tag.err; exit 1; }
  timeout -k 10 280 python bench.py --config $cfg --steps 5 --warmup 2 --breakdown --no-cpu-baseline > gpurun_out/r3i_cfg${tag}_bd.json 2> gpurun_out/r3i_cfg${tag}_bd.err || { tail -5 gpurun_out/r3i_cfg${tag}_bd.err; exit 1; }
  echo "== config $cfg"; python - <<PY
import json
d=json.loads(open('gpurun_out/r3i_cfg$tag.json').read().strip().splitlines()[-1])
print(d['value'], d['unit'], d['ms_per_step'], 'ms', d['dtype'], d['roofline']['kernel'][:30], d['roofline']['frac'], d['roofline']['bound'])
PY
  grep -B12 "kernels .* ms/step" gpurun_out/r3i_cfg${tag}_bd.err | grep -v amdgpu | cut -c1-100
done
timeout -k 10 300 python tools/gcn_exp.py f32 bwd ISTGCN_GCN_RC 0 2>&1 | grep -v amdgpu.ids
