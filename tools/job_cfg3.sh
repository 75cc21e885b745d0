#!/bin/bash
# config 3 in fp32 (V = 18): the register-chained fp32 weight gradient dispatched from 18 joints (experiment build lib_wg18) vs from 20
for rep in 1 2; do
for v in intree wg18; do
  if [ $v = intree ]; then unset ISTGCN_LIB_PATH; else export ISTGCN_LIB_PATH=tools/bin/lib_$v.so; fi
  python bench.py --config 3 --steps 4 --warmup 2 --no-cpu-baseline --no-vendor-gemm --breakdown > gpurun_out/c3_$v.json 2> gpurun_out/c3_$v.txt || exit 1
  echo "== $v rep $rep: $(python3 -c "import json; d=json.load(open('gpurun_out/c3_$v.json')); print(d['ms_per_step'])") ms/step (with events)"; grep "gcn_wgrad\|gcn_fwd\|gcn_bwd" gpurun_out/c3_$v.txt
done; done
