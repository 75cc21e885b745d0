# refresh of the 16-bit profiles on the round's last build (tools/profile_bench.sh): configs 4 (bf16), 5, 3 (bf16)
for spec in "r03_st_gcn_multi3_fix_3A_mstcn_bf16_b64:--config 4 --dtype bf16" "r03_st_gcn_mstcn_1x1_deep_f16_b128:--config 5" "r03_st_gcn_mstcn_1x1_bf16_b256:--config 3 --dtype bf16"; do
  tag=$(echo "$spec" | cut -d: -f1); args=$(echo "$spec" | cut -d: -f2)
  echo "=== $tag ($args)"
  bash tools/profile_bench.sh $tag $args > gpurun_out/${tag}_profile.log 2>&1 || { tail -20 gpurun_out/${tag}_profile.log; exit 1; }
  grep "total kernel time" gpurun_out/${tag}_profile.log
  rm -rf gpurun_out/prof_$tag
done
