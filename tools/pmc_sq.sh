#!/bin/bash
# SQ counter pass (MFMA utilisation, wait breakdown) over a kernel micro-benchmark.  usage: tools/pmc_sq.sh <tag> <kbench args...>
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d $out -- python3 $R/tools/kbench.py "$@" > $out/run.log 2>&1
cd $R
python3 tools/pmc_parse.py $out > gpurun_out/pmc_$tag.txt 2>&1
cat gpurun_out/pmc_$tag.txt
