mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_gcn.py tests/test_gpu_fullsize.py tests/test_gpu_block.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r3b_tests.log 2>&1; rc=$?
tail -8 gpurun_out/r3b_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 200 python tools/gcn_exp.py bf16 wgrad ISTGCN_GCN_RC 0,1 > gpurun_out/r3b_gcn_exp_wgrad.log 2>&1
cat gpurun_out/r3b_gcn_exp_wgrad.log
timeout -k 10 200 python tools/gcn_exp.py bf16 bwd ISTGCN_GCN_RC 0,1 > gpurun_out/r3b_gcn_exp_bwd.log 2>&1
cat gpurun_out/r3b_gcn_exp_bwd.log
ISTGCN_RC_NCT=2 timeout -k 10 200 python tools/gcn_exp.py bf16 bwd ISTGCN_GCN_RC 0,1 > gpurun_out/r3b_gcn_exp_bwd_nct2.log 2>&1
cat gpurun_out/r3b_gcn_exp_bwd_nct2.log
