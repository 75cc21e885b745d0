#!/bin/bash
# Last GPU call of a round, on the build whose profiles/ summaries were just committed: smoke(), the default bench line
# (its roofline.traffic now comes from those summaries), instruction mix + LDS conflicts of config 2, the weight-gradient
# evidence (per-call times, stamps of the experiment build tools/bin/lib_twgstamp.so when present).
# usage: tools/gpu_final.sh <tag>
tag=$1
mkdir -p gpurun_out
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${tag}_smoke.log 2>&1 || { tail -5 gpurun_out/${tag}_smoke.log; exit 1; }
tail -1 gpurun_out/${tag}_smoke.log
timeout -k 10 300 python bench.py > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err || { tail -5 gpurun_out/${tag}_bench_default.err; exit 1; }
cut -c1-260 gpurun_out/${tag}_bench_default.json
tools/pmc_bench_mix.sh r04_cfg2 || exit 1
cp gpurun_out/pmcmix_r04_cfg2.txt gpurun_out/r04_st_gcn_msgcn_bf16_b64_instruction_mix_lds_conflicts.txt
rm -rf gpurun_out/pmcmix_r04_cfg2
{ timeout -k 10 200 python tools/twg_bench.py bf16 9; timeout -k 10 200 python tools/twg_bench.py bf16 15; } 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_twg_bench.txt
cat gpurun_out/${tag}_twg_bench.txt
if [ -f tools/bin/lib_twgstamp.so ]; then
  ISTGCN_LIB_PATH=tools/bin/lib_twgstamp.so timeout -k 10 200 python tools/twg_stamp_exp.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_twg_stamps.txt
  cat gpurun_out/${tag}_twg_stamps.txt
fi
# two ranks sharing the one card over gloo: the data-parallel path end to end (layout agreement, early bucket, per-rank report)
ISTGCN_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --batch 32 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_bench_2rank_gloo.json 2> gpurun_out/${tag}_bench_2rank_gloo.err || { tail -5 gpurun_out/${tag}_bench_2rank_gloo.err; exit 1; }
cut -c1-200 gpurun_out/${tag}_bench_2rank_gloo.json
