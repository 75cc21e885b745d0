#!/bin/bash
# SQ / LDS counter passes over tools/gcn_exp.py (one dtype, one op).  usage: tools/pmc_gcn.sh <tag> <dtype> <op>
tag=$1; dt=$2; op=$3
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d $out/a -- python3 $R/tools/gcn_exp.py $dt $op X 0 > $out/run_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS \
  --kernel-trace --output-format csv -d $out/b -- python3 $R/tools/gcn_exp.py $dt $op X 0 > $out/run_b.log 2>&1
cd $R
python3 tools/pmc_parse.py $out > gpurun_out/pmc_$tag.txt 2>&1
cat gpurun_out/pmc_$tag.txt
