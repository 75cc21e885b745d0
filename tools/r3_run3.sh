mkdir -p gpurun_out
timeout -k 10 840 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r3c_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r3c_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 240 python bench.py --steps 10 --warmup 3 --breakdown --no-cpu-baseline > gpurun_out/r3c_bench_bf16.json 2> gpurun_out/r3c_bench_bf16.err || exit 1
tail -c 1500 gpurun_out/r3c_bench_bf16.json; grep -A30 "ms/step" gpurun_out/r3c_bench_bf16.err | head -40
