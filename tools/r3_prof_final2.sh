for spec in "r03_st_gcn_msgcn_bf16_b64:--config 2" "r03_st_gcn_mstcn_1x1_deep_f16_b128:--config 5" "r03_st_gcn_mstcn_1x1_bf16_b256:--config 3 --dtype bf16" "r03_st_gcn_multi3_fix_3A_mstcn_bf16_b64:--config 4 --dtype bf16" "r03_st_gcnold_f32_b2:--config 1"; do
  tag=$(echo "$spec" | cut -d: -f1); args=$(echo "$spec" | cut -d: -f2)
  echo "=== $tag ($args)"
  bash tools/profile_bench.sh $tag $args > gpurun_out/${tag}_profile.log 2>&1 || { tail -20 gpurun_out/${tag}_profile.log; exit 1; }
  grep "total kernel time" gpurun_out/${tag}_profile.log
  rm -rf gpurun_out/prof_$tag
done
timeout -k 10 1000 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider > gpurun_out/r3s_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r3s_tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
exit $rc
