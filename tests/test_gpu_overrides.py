"""GPU: the three dispatch overrides the shipped library reads each select the PREVIOUS generation of one
kernel family -- ISTGCN_GCN_RC=0 the round-1/2 graph-conv kernels, ISTGCN_TCONV_LEAN=0 the round-3 temporal conv,
ISTGCN_TWG_LEAN=0 the round-1 temporal-conv weight gradient -- which also serve the shapes / storage types the newer kernels
decline.  (`ISTGCN_DRES_FREE=0`, read by the Python side, is the fourth: the block's backward with the dres tensor written as
before round 5.)  An override is read once per process, so each setting runs the reference-pinned wide-block test
(test_gpu_block.py::test_blocks_wide_golden, all five block kinds, bfloat16 and float16) in a child process; that test asserts
from the library's dispatch trace that the older kernels really ran, against the same reference fixtures and gates."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('var', ['ISTGCN_GCN_RC', 'ISTGCN_TCONV_LEAN', 'ISTGCN_TWG_LEAN', 'ISTGCN_DRES_FREE'])
def test_wide_blocks_golden_under_override(var):
    env = dict(os.environ)
    env[var] = '0'
    env['ISTGCN_COVERAGE_TAG'] = ' {%s=0}' % var
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_gpu_block.py') + '::test_blocks_wide_golden',
                        '-q', '-x', '-p', 'no:cacheprovider', '-k', 'dt1 or dt2'], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode(errors='replace')
    assert r.returncode == 0, out[-3000:]
    assert '10 passed' in out, out[-500:]
