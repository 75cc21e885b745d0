#!/bin/bash
# rocprofv3 summaries (kernel stats + PMC) of the given configurations on the build in the tree -> gpurun_out/r05_*; copy to profiles/
# usage: tools/job_profiles.sh "2 4 5"
for c in $1; do
  case $c in
    1) n=st_gcnold_f32_b2; a="--config 1";;
    2) n=st_gcn_msgcn_bf16_b64; a="";;
    3) n=st_gcn_mstcn_1x1_f32_b256; a="--config 3";;
    4) n=st_gcn_multi3_fix_3A_mstcn_bf16_b64; a="--config 4 --dtype bf16";;
    4f) n=st_gcn_multi3_fix_3A_mstcn_f32_b64; a="--config 4";;
    5) n=st_gcn_mstcn_1x1_deep_f16_b128; a="--config 5";;
  esac
  tools/profile_bench.sh r05_$n $a > gpurun_out/prof_$c.log 2>&1 || { echo "profile $c failed"; tail -5 gpurun_out/prof_$c.log; exit 1; }
  echo "profile $c ok: $(tail -1 gpurun_out/prof_$c.log)"
  rm -rf gpurun_out/prof_r05_$n
done
