timeout -k 10 300 python -m pytest tests/test_gpu_bneck.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
for rep in 1 2; do
  for e in "" "ISTGCN_BOUT_NW8=1"; do
    for cfg in "5" "3 --dtype bf16"; do echo -n "[$e] cfg $cfg: "; env $e timeout -k 10 200 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c135-160; done
  done
done
