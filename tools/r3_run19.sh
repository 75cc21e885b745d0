#!/bin/bash
# static-k compute loop (SK) for the data gradients and the 64-channel forward: whole GPU suite, kernel A/B, step A/B
set -o pipefail
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 || exit 1
ISTGCN_TCONV_SK=0 timeout -k 10 200 python tools/tconv_var_exp.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 200 python tools/tconv_var_exp.py 2>&1 | grep -v amdgpu.ids
for r in 1 2 3; do
  for v in 0 1; do
    ISTGCN_TCONV_SK=$v timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('SK=$v cfg2', d['ms_per_step'], d['roofline']['frac'])"
  done
done
for v in 0 1 0 1; do
  ISTGCN_TCONV_SK=$v timeout -k 10 200 python bench.py --config 4 --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('SK=$v cfg4bf16', d['ms_per_step'])"
done
for v in 0 1; do
  ISTGCN_TCONV_SK=$v timeout -k 10 200 python bench.py --config 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('SK=$v cfg1', d['ms_per_step'])"
done
