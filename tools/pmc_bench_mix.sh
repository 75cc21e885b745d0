#!/bin/bash
# Instruction mix + LDS conflict counters over one bench command.  usage: tools/pmc_bench_mix.sh <tag> [bench.py arguments]
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/pmcmix_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-vendor-gemm $*"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d $out/a -- $B > $out/run_a.log 2>&1 || { tail -5 $out/run_a.log; exit 1; }
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS \
  --kernel-trace --output-format csv -d $out/b -- $B > $out/run_b.log 2>&1 || { tail -5 $out/run_b.log; exit 1; }
cd $R
python3 tools/pmc_parse.py $out > gpurun_out/pmcmix_$tag.txt 2>&1
grep -c . gpurun_out/pmcmix_$tag.txt
