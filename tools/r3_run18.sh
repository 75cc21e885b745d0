#!/bin/bash
# final verification of the round's last code change: whole GPU suite, smoke, default bench (the driver's command)
set -o pipefail
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 || exit 1
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 || exit 1
timeout -k 10 400 python bench.py > gpurun_out/r3w_bench_default.json 2> gpurun_out/r3w_bench_default.err || { tail -5 gpurun_out/r3w_bench_default.err; exit 1; }
cut -c1-900 gpurun_out/r3w_bench_default.json
