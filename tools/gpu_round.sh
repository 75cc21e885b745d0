#!/bin/bash
# One gpurun call: GPU parity suite, then (unless the suite was killed by its timeout) the bench lines.
# usage: tools/gpu_round.sh <tag> [extra]   (extra = "full": also f16 config-5 bench, 2-rank gloo rehearsal, inference bench)
tag=$1; extra=$2
mkdir -p gpurun_out
timeout -k 10 840 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > gpurun_out/${tag}_tests.log 2>&1
rc=$?
echo "pytest rc=$rc" | tee -a gpurun_out/${tag}_tests.log
tail -5 gpurun_out/${tag}_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "suite killed: no further GPU step"; exit $rc; fi
timeout -k 10 240 python bench.py --steps 10 --warmup 3 --breakdown > gpurun_out/${tag}_bench_bf16.json 2> gpurun_out/${tag}_bench_bf16.err || exit 1
tail -c 1500 gpurun_out/${tag}_bench_bf16.json; grep "ms/step" gpurun_out/${tag}_bench_bf16.err
if [ "$extra" = "full" ]; then
  timeout -k 10 300 python bench.py --model st_gcn_mstcn_1x1_deep --dtype f16 --batch 128 --steps 5 --warmup 2 --breakdown --no-cpu-baseline > gpurun_out/${tag}_bench_cfg5_f16.json 2> gpurun_out/${tag}_bench_cfg5_f16.err || exit 1
  tail -c 1200 gpurun_out/${tag}_bench_cfg5_f16.json; grep "ms/step" gpurun_out/${tag}_bench_cfg5_f16.err
  ISTGCN_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 5 --warmup 2 --batch 32 --no-cpu-baseline > gpurun_out/${tag}_bench_2rank_gloo.json 2> gpurun_out/${tag}_bench_2rank_gloo.err || { tail -20 gpurun_out/${tag}_bench_2rank_gloo.err; exit 1; }
  tail -c 800 gpurun_out/${tag}_bench_2rank_gloo.json
  timeout -k 10 200 python tools/infer_bench.py st_gcn_msgcn 64 bf16 > gpurun_out/${tag}_infer.json 2> gpurun_out/${tag}_infer.err || { tail -20 gpurun_out/${tag}_infer.err; exit 1; }
  timeout -k 10 200 python tools/infer_bench.py st_gcnold 64 f32 >> gpurun_out/${tag}_infer.json 2>> gpurun_out/${tag}_infer.err
  cat gpurun_out/${tag}_infer.json
fi
exit $rc
