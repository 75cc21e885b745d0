#!/bin/bash
# One gpurun call at a milestone: GPU parity suite, then (unless the suite was killed by its timeout) the bench lines of all
# five BASELINE configurations and the rocprofv3 summaries named so that bench.py finds them.
# usage: tools/gpu_round.sh <tag> [profile configs, e.g. "2 4 5"]
tag=$1; prof=${2:-}
mkdir -p gpurun_out
rm -f gpurun_out/err16_measured.txt
timeout -k 10 900 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider > gpurun_out/${tag}_tests.log 2>&1
rc=$?
echo "pytest rc=$rc" | tee -a gpurun_out/${tag}_tests.log
tail -4 gpurun_out/${tag}_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "suite killed: no further GPU step"; exit $rc; fi
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err || { tail -5 gpurun_out/${tag}_bench_default.err; exit 1; }
cut -c1-330 gpurun_out/${tag}_bench_default.json
for c in 1 3 4 5; do
  timeout -k 10 300 python bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --breakdown > gpurun_out/${tag}_bench_cfg$c.json 2> gpurun_out/${tag}_bench_cfg$c.err || { tail -5 gpurun_out/${tag}_bench_cfg$c.err; exit 1; }
  python3 -c "import json,sys; d=json.load(open('gpurun_out/${tag}_bench_cfg$c.json')); print('cfg$c', d['ms_per_step'], d['value'], d['roofline']['kernel'][:30], d['roofline']['frac'])"
done
for dt in bf16; do for c in 3 4; do
  timeout -k 10 300 python bench.py --config $c --dtype $dt --steps 5 --warmup 2 --no-cpu-baseline --breakdown > gpurun_out/${tag}_bench_cfg${c}_$dt.json 2> gpurun_out/${tag}_bench_cfg${c}_$dt.err || exit 1
  python3 -c "import json,sys; d=json.load(open('gpurun_out/${tag}_bench_cfg${c}_$dt.json')); print('cfg$c $dt', d['ms_per_step'], d['value'], d['roofline']['kernel'][:30], d['roofline']['frac'])"
done; done
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --breakdown > /dev/null 2> gpurun_out/${tag}_breakdown_cfg2.txt; grep "ms/step" gpurun_out/${tag}_breakdown_cfg2.txt
for c in $prof; do
  case $c in
    1) n=st_gcnold_f32_b2; a="--config 1";;
    2) n=st_gcn_msgcn_bf16_b64; a="";;
    3) n=st_gcn_mstcn_1x1_f32_b256; a="--config 3";;
    4) n=st_gcn_multi3_fix_3A_mstcn_bf16_b64; a="--config 4 --dtype bf16";;
    4f) n=st_gcn_multi3_fix_3A_mstcn_f32_b64; a="--config 4";;
    5) n=st_gcn_mstcn_1x1_deep_f16_b128; a="--config 5";;
  esac
  tools/profile_bench.sh r05_$n $a > gpurun_out/${tag}_prof_$c.log 2>&1 || { tail -5 gpurun_out/${tag}_prof_$c.log; exit 1; }
  tail -3 gpurun_out/${tag}_prof_$c.log
  rm -rf gpurun_out/prof_r05_$n
done
exit $rc
