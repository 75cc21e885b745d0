mkdir -p gpurun_out; rm -f gpurun_out/err16_measured.txt
timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=20 > gpurun_out/r3e_tests.log 2>&1; rc=$?
tail -25 gpurun_out/r3e_tests.log
exit $rc
