ISTGCN_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 5 --warmup 2 --batch 32 --no-cpu-baseline > gpurun_out/r3p_bench_2rank_gloo.json 2> gpurun_out/r3p_bench_2rank_gloo.err || { tail -20 gpurun_out/r3p_bench_2rank_gloo.err; exit 1; }
tail -c 1500 gpurun_out/r3p_bench_2rank_gloo.json
