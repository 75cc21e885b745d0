#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 200 python tools/gwg_bits.py > gpurun_out/gwg_bits_new.txt 2>&1 || { tail -5 gpurun_out/gwg_bits_new.txt; exit 1; }
ISTGCN_LIB_PATH=tools/bin/lib_sdot.so timeout -k 10 200 python tools/gwg_bits.py > gpurun_out/gwg_bits_sdot.txt 2>&1 || { tail -5 gpurun_out/gwg_bits_sdot.txt; exit 1; }
grep -c rel-err gpurun_out/gwg_bits_new.txt gpurun_out/gwg_bits_sdot.txt
python3 - <<'PY'
import re
def load(f):
    return [l.split() for l in open(f) if 'rel-err' in l]
a, b = load('gpurun_out/gwg_bits_new.txt'), load('gpurun_out/gwg_bits_sdot.txt')
same_s = sum(1 for x, y in zip(a, b) if x[10] == y[10])
print('S hash equal (new vs sdot) in %d of %d cases; worst rel-err new %s sdot %s' % (same_s, len(a), max(float(x[-1]) for x in a), max(float(x[-1]) for x in b)))
PY
GWG_LAYERS=128x128x150,256x256x75 timeout -k 10 400 tools/ab_gwg.sh -r 2 sdot a8 a16 a24 a31 2>&1 | grep -v amdgpu.ids
