mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_gcn.py tests/test_gpu_block.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r3a_gcn_tests.log 2>&1; rc=$?
tail -6 gpurun_out/r3a_gcn_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for w in ${1:-fwd}; do
timeout -k 10 200 python tools/gcn_exp.py bf16 $w ISTGCN_GCN_RC 0,1 > gpurun_out/r3a_gcn_exp_$w.log 2>&1
cat gpurun_out/r3a_gcn_exp_$w.log
done
