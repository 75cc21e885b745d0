timeout -k 10 1000 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider > gpurun_out/r3m_tests.log 2>&1; rc=$?
tail -6 gpurun_out/r3m_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "suite killed"; exit $rc; fi
for cfg in "2" "5" "3 --dtype bf16" "4 --dtype bf16"; do
  timeout -k 10 280 python bench.py --config $cfg --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
done
exit $rc
