"""Distribution of a 16-bit-storage gate over repeated runs in ONE process (gpu_util.gate16 appends every measured value to
gpurun_out/err16_measured.txt): usage python tools/repeat_gate.py <pytest node id> [n]"""
import os, sys
import pytest
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
node, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 10
out = os.path.join(ROOT, 'gpurun_out', 'err16_measured.txt')
if os.path.exists(out):
    os.remove(out)
bad = 0
for i in range(n):
    rc = pytest.main([node, '-x', '-q', '-m', 'gpu', '-p', 'no:cacheprovider'])
    bad += int(rc != 0)
    print('run %d rc %s' % (i, rc), flush=True)
print('failures: %d of %d' % (bad, n))
