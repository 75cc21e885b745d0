#!/bin/bash
# The kernels every BASELINE configuration reaches, in every storage type it is run in: short bench.py runs under the
# library's dispatch trace -> gpurun_out/kernel_coverage_bench.tsv (read by tools/kernel_coverage.py)
out=gpurun_out/kernel_coverage_bench.tsv
rm -f $out
for spec in "1 f32" "1 bf16" "2 bf16" "2 f32" "2 f16" "3 f32" "3 bf16" "4 f32" "4 bf16" "5 f16" "5 f32"; do
  set -- $spec
  b=""; if [ "$1" = "5" ] && [ "$2" = "f32" ]; then b="--batch 32"; fi
  ISTGCN_TRACE_KERNELS=$out timeout -k 10 300 python bench.py --config $1 --dtype $2 $b --steps 2 --warmup 1 --no-cpu-baseline --no-vendor-gemm > /dev/null 2> gpurun_out/cov_$1_$2.err || { echo "config $1 $2 failed"; tail -3 gpurun_out/cov_$1_$2.err; exit 1; }
  echo "config $1 $2 ok"
done
