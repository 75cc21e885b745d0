#!/bin/bash
# new tconv memory role (buffer-descriptor loads, branch-free packed commit): parity tests, kernel A/B, step A/B vs HEAD's library
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_tconv.py tests/test_gpu_block.py -x -q -m gpu 2>&1 | tail -4 || exit 1
ISTGCN_LIB_PATH=tools/bin/lib_head.so timeout -k 10 200 python tools/tconv_var_exp.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 200 python tools/tconv_var_exp.py 2>&1 | grep -v amdgpu.ids
ISTGCN_LIB_PATH=tools/bin/lib_head.so timeout -k 10 200 python tools/tconv_var_exp.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 200 python tools/tconv_var_exp.py 2>&1 | grep -v amdgpu.ids
for r in 1 2 3; do
  for v in head new; do
    if [ $v = head ]; then export ISTGCN_LIB_PATH=tools/bin/lib_head.so; else unset ISTGCN_LIB_PATH; fi
    timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v cfg2', d['ms_per_step'], d['roofline']['frac'])"
  done
done
for v in head new head new; do
  if [ $v = head ]; then export ISTGCN_LIB_PATH=tools/bin/lib_head.so; else unset ISTGCN_LIB_PATH; fi
  timeout -k 10 200 python bench.py --config 4 --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v cfg4bf16', d['ms_per_step'])"
done
