timeout -k 10 840 python -m pytest tests/test_gpu_block.py tests/test_gpu_f2f4.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r3j_tests.log 2>&1; rc=$?
tail -4 gpurun_out/r3j_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for cfg in "5" "3 --dtype bf16" "3"; do
  timeout -k 10 280 python bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-160
done
