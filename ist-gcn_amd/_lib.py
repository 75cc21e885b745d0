"""Build + load the C-ABI shared library (`libistgcn_hip.so`, hand-written HIP for gfx950).

The product path has NO fallback: if the library cannot be loaded every compute entry point
raises `RuntimeError` (there is no CPU / eager path -- the CPU restatement lives in `oracle/`
and is test infrastructure only).
"""
import ctypes
import glob
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB_PATH = os.path.join(HERE, 'libistgcn_hip.so')
HEADER = os.path.join(os.path.dirname(HERE), 'include', 'istgcn.h')
ARCH = 'gfx950'

_lib = None
_err = None


def _hipcc():
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    return None


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, '*.hpp'))
    return any(os.path.getmtime(s) > t for s in deps)


def build(force=False, verbose=False, jobs=None):
    """hipcc --offload-arch=gfx950 every csrc/*.hip -> one in-tree .so (cross-compiles without a GPU)."""
    if not force and not needs_build():
        return LIB_PATH
    hipcc = _hipcc()
    if hipcc is None:
        raise RuntimeError('hipcc not found: cannot build %s' % LIB_PATH)
    objdir = os.path.join(HERE, 'build')
    os.makedirs(objdir, exist_ok=True)
    flags = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-ffp-contract=fast', '-I', CSRC]
    procs, objs = [], []
    jobs = jobs or min(8, os.cpu_count() or 1)
    pending = list(sources())
    running = []
    failed = []

    def reap(block):
        for item in list(running):
            p, src = item
            if block:
                p.wait()
            if p.poll() is not None:
                out = p.stdout.read().decode(errors='replace')
                if p.returncode != 0:
                    failed.append((src, out))
                elif verbose and out.strip():
                    print(out)
                running.remove(item)

    for src in pending:
        obj = os.path.join(objdir, os.path.basename(src) + '.o')
        objs.append(obj)
        if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(src)
                and all(os.path.getmtime(obj) > os.path.getmtime(h) for h in glob.glob(os.path.join(CSRC, '*.hpp')))):
            continue
        while len(running) >= jobs:
            reap(True)
        p = subprocess.Popen([hipcc] + flags + ['-c', src, '-o', obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        running.append((p, src))
    while running:
        reap(True)
    if failed:
        raise RuntimeError('hipcc failed:\n' + '\n'.join('%s\n%s' % f for f in failed))
    r = subprocess.run([hipcc, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB_PATH] + objs,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError('link failed:\n' + r.stdout.decode(errors='replace'))
    return LIB_PATH


def load():
    """ctypes handle of the C-ABI library; raises RuntimeError when it is missing (no fallback)."""
    global _lib, _err
    if _lib is not None:
        return _lib
    # PyTorch bundles its own HIP runtime (SONAME libamdhip64.so.7).  It must be in the process BEFORE this library
    # is dlopen'ed so that both share ONE runtime (device context, streams, allocations); loaded the other way round
    # the system libamdhip64 would come in as a second, device-less runtime (hipErrorNoDevice at the first launch).
    import torch  # noqa: F401
    path = os.environ.get('ISTGCN_LIB_PATH') or LIB_PATH      # experiment builds (tools/build_variant.sh); default: the in-tree library
    if not os.path.exists(path):
        _err = ('%s is missing: run `python -c "import __graft_entry__ as g; g.build()"` (hipcc, gfx950). '
                'There is no CPU fallback for the IST-GCN hot path.' % path)
        raise RuntimeError(_err)
    try:
        _lib = ctypes.CDLL(path)
    except OSError as e:
        raise RuntimeError('cannot load %s: %s' % (path, e))
    for name in ('istgcn_pack_gcn_elems', 'istgcn_pack_tconv_elems', 'istgcn_pack_gcn_bwd_elems', 'istgcn_gcn_rc_offset', 'istgcn_gcn_bwd_rc_offset'):
        getattr(_lib, name).restype = ctypes.c_longlong          # element counts; every other entry returns int
    return _lib


def declared_symbols():
    """Names of every `int istgcn_*(...)` entry point declared in include/istgcn.h."""
    import re
    txt = open(HEADER).read()
    return sorted(set(re.findall(r'\b(?:int|long long)\s+(istgcn_\w+)\s*\(', txt)))
