timeout -k 10 600 python -m pytest tests/test_gpu_tconv.py tests/test_gpu_block.py -m gpu -q -x -k "tconv or golden" -p no:cacheprovider > gpurun_out/r3t_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r3t_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for i in 1 2; do timeout -k 10 280 python bench.py --config 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-150; done
timeout -k 10 280 python bench.py --config 1 --steps 10 --warmup 3 --no-cpu-baseline --breakdown 2>&1 >/dev/null | grep -v amdgpu | head -6
timeout -k 10 280 python bench.py --config 1 --steps 20 --warmup 5 --no-cpu-baseline --graph 2>/dev/null | tail -1 | cut -c1-150
