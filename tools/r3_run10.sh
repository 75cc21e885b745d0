timeout -k 10 500 python -m pytest tests/test_gpu_bneck.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r3l_bneck_tests.log 2>&1; rc=$?
tail -15 gpurun_out/r3l_bneck_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for cfg in "5" "3 --dtype bf16"; do
  timeout -k 10 280 python bench.py --config $cfg --steps 5 --warmup 2 --no-cpu-baseline --breakdown 2>&1 >/dev/null | grep -v amdgpu | cut -c1-100
done
