timeout -k 10 600 python -m pytest tests/test_gpu_gcn.py tests/test_gpu_fullsize.py tests/test_gpu_block.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r3h_tests.log 2>&1; rc=$?
tail -6 gpurun_out/r3h_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python tools/gcn_exp.py f32 ${1:-fwd} ISTGCN_GCN_RC 0,1 2>&1 | grep -v amdgpu.ids
