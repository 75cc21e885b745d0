timeout -k 10 1000 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider > gpurun_out/r3s_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r3s_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "suite killed"; exit $rc; fi
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
exit $rc
