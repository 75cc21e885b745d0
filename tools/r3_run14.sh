timeout -k 10 600 python -m pytest tests/test_gpu_block.py -m gpu -q -x -k "bn_tails" -p no:cacheprovider > gpurun_out/r3q_tests.log 2>&1; rc=$?
tail -12 gpurun_out/r3q_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for t in 1 0; do
  echo "== ISTGCN_BN_TAILS=$t"
  for cfg in "2" "1" "5"; do
    ISTGCN_BN_TAILS=$t timeout -k 10 280 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-160
  done
done
