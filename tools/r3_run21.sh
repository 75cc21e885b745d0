#!/bin/bash
# same-box A/B of the round's last library against the library of commit 1854e93 (tools/bin/lib_head.so): configs 5, 3 (bf16), 2
for spec in "cfg5:--config 5" "cfg3bf16:--config 3 --dtype bf16" "cfg2:--config 2"; do
  tag=$(echo "$spec" | cut -d: -f1); args=$(echo "$spec" | cut -d: -f2)
  for v in head new head new; do
    if [ $v = head ]; then export ISTGCN_LIB_PATH=tools/bin/lib_head.so; else unset ISTGCN_LIB_PATH; fi
    timeout -k 10 280 python bench.py $args --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v $tag', d['ms_per_step'], 'ms')"
  done
done
