#!/bin/bash
mkdir -p gpurun_out
ISTGCN_LIB_PATH=tools/bin/lib_r5base.so timeout -k 10 200 python tools/gwg_bits.py > gpurun_out/gwg_bits_base.txt 2>&1 || { tail -5 gpurun_out/gwg_bits_base.txt; exit 1; }
timeout -k 10 200 python tools/gwg_bits.py > gpurun_out/gwg_bits_new.txt 2>&1 || { tail -5 gpurun_out/gwg_bits_new.txt; exit 1; }
python3 - <<'PY'
def load(f):
    return [l.split() for l in open(f) if 'rel-err' in l]
a, b = load('gpurun_out/gwg_bits_base.txt'), load('gpurun_out/gwg_bits_new.txt')
# columns: 0 dtype 1 NM 2 T 3 V 4 cin->cout 5 K 6 'dW' 7 hash 8 'S' 9 hash 10 'rerun-identical' 11 bool 12 'rel-err-vs-torch' 13 err 14 'S-err' 15 err
print('%d cases; S hash identical base/new: %d; dW hash identical: %d; worst dW err base %s new %s; worst S err base %s new %s' % (
    len(a), sum(x[9] == y[9] for x, y in zip(a, b)), sum(x[7] == y[7] for x, y in zip(a, b)),
    max(float(x[13]) for x in a), max(float(x[13]) for x in b), max(float(x[15]) for x in a), max(float(x[15]) for x in b)))
for x, y in zip(a, b):
    if float(y[15]) > 1e-3: print('BAD', ' '.join(y))
PY
timeout -k 10 500 python -m pytest tests/test_gpu_gcn.py -q -p no:cacheprovider -x 2>&1 | tail -3
