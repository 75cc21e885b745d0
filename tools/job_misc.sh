#!/bin/bash
# small A/Bs on one box: "last workgroup finalises" BatchNorm tails (ISTGCN_BN_TAILS=1) on the round-5 step, hipGraph replay for the
# launch-bound config 1, the PCIe-inclusive run through harness.DeviceStager
for rep in 1 2; do
  for t in 0 1; do
    ISTGCN_BN_TAILS=$t python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-vendor-gemm > gpurun_out/m_tails$t.json 2>/dev/null || exit 1
    echo "cfg2 bn_tails=$t rep $rep: $(python3 -c "import json; d=json.load(open('gpurun_out/m_tails$t.json')); print(d['ms_per_step'], 'ms', d['value'], 'clips/s')")"
  done
done
for g in "" "--graph"; do
  python bench.py --config 1 --steps 20 --warmup 5 --no-cpu-baseline $g > gpurun_out/m_cfg1$g.json 2>/dev/null || exit 1
  echo "cfg1 $g: $(python3 -c "import json; d=json.load(open('gpurun_out/m_cfg1$g.json')); print(d['ms_per_step'], 'ms', d['value'], 'clips/s')")"
done
for h in "" "--h2d"; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-vendor-gemm $h > gpurun_out/m_h2d$h.json 2>/dev/null || exit 1
  echo "cfg2 $h: $(python3 -c "import json; d=json.load(open('gpurun_out/m_h2d$h.json')); print(d['ms_per_step'], 'ms', d['value'], 'clips/s')")"
done
