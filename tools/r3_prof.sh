# usage: tools/r3_prof.sh <tag> [bench args]   -> gpurun_out/<tag>_{kernel_stats.csv,pmc.json} + pmcmix_<tag>.txt
tag=$1; shift
bash tools/profile_bench.sh $tag $* > gpurun_out/${tag}_profile.log 2>&1 || { tail -20 gpurun_out/${tag}_profile.log; exit 1; }
tail -30 gpurun_out/${tag}_profile.log
bash tools/pmc_bench_mix.sh $tag $* || exit 1
rm -rf gpurun_out/prof_$tag gpurun_out/pmcmix_$tag
