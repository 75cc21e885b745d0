# all five BASELINE configurations in the reference's arithmetic + the 16-bit variants, plain bench and breakdown
mkdir -p gpurun_out
for spec in "cfg1:--config 1" "cfg2:--config 2" "cfg2f32:--config 2 --dtype f32" "cfg3:--config 3" "cfg3bf16:--config 3 --dtype bf16" "cfg4:--config 4" "cfg4bf16:--config 4 --dtype bf16" "cfg5:--config 5"; do
  tag=$(echo "$spec" | cut -d: -f1); args=$(echo "$spec" | cut -d: -f2)
  timeout -k 10 280 python bench.py $args --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r3z_${tag}.json 2> gpurun_out/r3z_${tag}.err || { tail -5 gpurun_out/r3z_${tag}.err; exit 1; }
  timeout -k 10 280 python bench.py $args --steps 5 --warmup 2 --no-cpu-baseline --breakdown > gpurun_out/r3z_${tag}_bd.json 2> gpurun_out/r3z_${tag}_bd.err || { tail -5 gpurun_out/r3z_${tag}_bd.err; exit 1; }
  python - <<PY
import json
d=json.loads(open('gpurun_out/r3z_${tag}.json').read().strip().splitlines()[-1])
print('${tag}', d['ms_per_step'], 'ms', d['value'], 'clips/s', d['roofline']['kernel'][:30], d['roofline']['frac'])
PY
done
