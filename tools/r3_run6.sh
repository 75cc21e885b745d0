mkdir -p gpurun_out
timeout -k 10 200 python tools/twg_exp.py bf16 9 > gpurun_out/r3f_twg9.log 2>&1; cat gpurun_out/r3f_twg9.log
timeout -k 10 200 python tools/twg_exp.py bf16 15 > gpurun_out/r3f_twg15.log 2>&1; cat gpurun_out/r3f_twg15.log
ISTGCN_TWG_RC=2 timeout -k 10 600 python -m pytest tests/test_gpu_tconv.py tests/test_gpu_block.py tests/test_gpu_fullsize.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r3f_tests.log 2>&1; tail -5 gpurun_out/r3f_tests.log
