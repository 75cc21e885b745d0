#!/bin/bash
# A/B on one box: the block backward without the dres tensor (default since round 5) vs with it (ISTGCN_DRES_FREE=0)
for rep in 1 2; do
  for v in 1 0; do
    ISTGCN_DRES_FREE=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-vendor-gemm > gpurun_out/d_free$v.json 2>/dev/null || exit 1
    echo "cfg2 dres_free=$v rep $rep: $(python3 -c "import json; d=json.load(open('gpurun_out/d_free$v.json')); print(d['ms_per_step'], 'ms', d['value'], 'clips/s')")"
  done
done
for v in 1 0; do
  ISTGCN_DRES_FREE=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-vendor-gemm --breakdown > /dev/null 2> gpurun_out/d_bd$v.txt || exit 1
  echo "== breakdown dres_free=$v"; grep "block_out_bwd\|affine2\|gcn_bwd\|wall" gpurun_out/d_bd$v.txt
done
for c in "4 bf16" "5 f16"; do set -- $c; for v in 1 0; do
  ISTGCN_DRES_FREE=$v python bench.py --config $1 --dtype $2 --steps 6 --warmup 2 --no-cpu-baseline --no-vendor-gemm > gpurun_out/d_c$1_$v.json 2>/dev/null || exit 1
  echo "cfg$1 $2 dres_free=$v: $(python3 -c "import json; d=json.load(open('gpurun_out/d_c$1_$v.json')); print(d['ms_per_step'], 'ms')")"
done; done
