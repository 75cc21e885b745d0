#!/bin/bash
# whole-step A/B of experiment builds (tools/build_variant.sh) on ONE box: config 2 (default), per variant the step time and the
# glue families of the breakdown; usage: tools/ab_step.sh <tag> <reps> <variant>...   ("base" = the in-tree library)
tag=$1; reps=$2; shift 2
for rep in $(seq 1 $reps); do
  for v in "$@"; do
    lib=""; [ "$v" != base ] && lib=tools/bin/lib_$v.so
    ISTGCN_LIB_PATH=$lib timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-vendor-gemm --breakdown \
      > gpurun_out/${tag}_${v}_$rep.json 2> gpurun_out/${tag}_${v}_$rep.txt || { echo "$v failed"; tail -5 gpurun_out/${tag}_${v}_$rep.txt; exit 1; }
    python3 - <<PY
import json, re
d = json.load(open('gpurun_out/${tag}_${v}_$rep.json'))
t = open('gpurun_out/${tag}_${v}_$rep.txt').read()
fam = {}
for line in t.splitlines():
    m = re.match(r'\s*(istgcn_\w+)\s+n=\s*(\d+)\s+([\d.]+) ms/step', line)
    if m: fam[m.group(1)] = float(m.group(3))
keys = ['istgcn_affine2', 'istgcn_block_out_bwd', 'istgcn_block_out_fwd', 'istgcn_tconv', 'istgcn_tconv_wgrad', 'istgcn_gcn_wgrad', 'istgcn_gcn_bwd_data', 'istgcn_gcn_fwd']
print('%-10s rep $rep  %.3f ms/step  ' % ('$v', d['ms_per_step']) + '  '.join('%s %.3f' % (k[7:], fam.get(k, -1)) for k in keys), flush=True)
PY
  done
done
