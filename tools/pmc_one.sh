#!/bin/bash
# SQ counter passes over tools/one_kernel.py.  usage: tools/pmc_one.sh <tag> <one_kernel.py arguments>
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/pmcone_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $R/tools/one_kernel.py $*"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d $out/a -- $B > $out/run_a.log 2>&1 || { tail -5 $out/run_a.log; exit 1; }
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS \
  --kernel-trace --output-format csv -d $out/b -- $B > $out/run_b.log 2>&1 || { tail -5 $out/run_b.log; exit 1; }
rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS \
  --kernel-trace --output-format csv -d $out/c -- $B > $out/run_c.log 2>&1 || { tail -5 $out/run_c.log; }
cd $R
python3 tools/pmc_parse.py $out > gpurun_out/pmcone_$tag.txt 2>&1
grep -v "^    raw" gpurun_out/pmcone_$tag.txt | grep -A3 "rc_wgrad\|twg_lean\|gcn_rc" | cut -c1-250
grep "raw" gpurun_out/pmcone_$tag.txt | head -3 | cut -c1-900
rm -rf $out
