timeout -k 10 600 python -m pytest tests/test_gpu_gcn.py tests/test_gpu_fullsize.py tests/test_gpu_block.py -m gpu -q -x -k "wgrad or golden or full_batch" -p no:cacheprovider > gpurun_out/r3n_tests.log 2>&1; rc=$?
tail -4 gpurun_out/r3n_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for ot in 1 2; do echo "== ISTGCN_GWG_OT=$ot"; ISTGCN_GWG_OT=$ot timeout -k 10 120 python tools/kbench.py --only gcn_wgrad 2>&1 | grep -v amdgpu.ids; done
for ot in 1 2; do echo "== ISTGCN_GWG_OT=$ot"; ISTGCN_GWG_OT=$ot timeout -k 10 120 python tools/kbench.py --only gcn_wgrad 2>&1 | grep -v amdgpu.ids; done
