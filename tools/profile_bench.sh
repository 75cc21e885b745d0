#!/bin/bash
# rocprofv3 passes over the default bench command (bf16, configs[1]): kernel trace + stats, HBM traffic (FETCH_SIZE and
# WRITE_SIZE in separate passes: they do not fit one pass), SQ counters (MFMA busy, wait breakdown).  usage: tools/profile_bench.sh <tag>
tag=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$tag
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $B > $O/fetch.log 2>&1 || { tail -5 $O/fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $B > $O/write.log 2>&1 || { tail -5 $O/write.log; exit 1; }
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d $O/sq -- $B > $O/sq.log 2>&1 || { tail -5 $O/sq.log; exit 1; }
cd $R
python3 tools/profile_summarise.py $O $tag
