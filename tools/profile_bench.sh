#!/bin/bash
# rocprofv3 passes over one bench command: kernel trace + stats, HBM traffic (FETCH_SIZE and WRITE_SIZE in separate
# passes: they do not fit one pass), SQ counters (MFMA busy, wait breakdown).  Counter passes carry --kernel-trace only.
# usage: tools/profile_bench.sh <tag> [bench.py arguments, e.g. --config 3]
#   the summaries land in gpurun_out/<tag>_{kernel_stats.csv,pmc.json}; name the tag r<round>_<model>_<dtype>_b<batch>
#   and copy both into profiles/: bench.py looks its counters up under exactly that name
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$tag
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-vendor-gemm $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B > $O/trace.log 2>&1 || { tail -5 $O/trace.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- $B > $O/fetch.log 2>&1 || { tail -5 $O/fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- $B > $O/write.log 2>&1 || { tail -5 $O/write.log; exit 1; }
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d $O/sq -- $B > $O/sq.log 2>&1 || { tail -5 $O/sq.log; exit 1; }
cd $R
python3 tools/profile_summarise.py $O $tag $*
rm -rf $O/trace/*/*_agent_info.csv 2>/dev/null; du -sh $O | tail -1
