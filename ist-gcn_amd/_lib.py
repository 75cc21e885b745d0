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


def _deps(src, seen=None):
    """`src` and the csrc headers it includes (transitively): a header edit rebuilds only the sources that see it --
    tconv.hip alone takes five minutes."""
    import re
    seen = set() if seen is None else seen
    if src in seen or not os.path.exists(src):
        return seen
    seen.add(src)
    for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(src).read(), re.M):
        _deps(os.path.join(CSRC, inc), seen)
    return seen


def source_flags(src):
    """Extra hipcc flags a source asks for in a `// hipcc-flags: ...` line (e.g. tconv_lean.hip: -fno-slp-vectorize)."""
    import re
    m = re.search(r'^//\s*hipcc-flags:\s*(.+)$', open(src).read(), re.M)
    return m.group(1).split() if m else []


def csrc_hash():
    """First 16 hex digits of a SHA-256 over the kernel sources (names + contents, build_id.hip excluded): what
    `istgcn_build_id()` of a library built from this tree returns."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(CSRC, '*.hip')) + glob.glob(os.path.join(CSRC, '*.hpp'))):
        if os.path.basename(f) == 'build_id.hip':
            continue
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def build_id():
    """`istgcn_build_id()` of the LOADED library (the tree hash it was built from), or None for a library without it."""
    lib = load()
    if not hasattr(lib, 'istgcn_build_id'):
        return None
    lib.istgcn_build_id.restype = ctypes.c_char_p
    return lib.istgcn_build_id().decode()


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    idf = LIB_PATH + '.build_id'         # next to the library: it travels to the GPU box with it (the object directory does not)
    if not os.path.exists(idf) or open(idf).read().strip() != csrc_hash():
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
    tree = csrc_hash()
    idf = LIB_PATH + '.build_id'
    stale_id = not os.path.exists(idf) or open(idf).read().strip() != tree
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
        is_id = os.path.basename(src) == 'build_id.hip'
        if (not force and os.path.exists(obj) and all(os.path.getmtime(obj) > os.path.getmtime(d) for d in _deps(src))
                and not (is_id and stale_id)):
            continue
        while len(running) >= jobs:
            reap(True)
        extra = (['-DISTGCN_BUILD_ID="%s"' % tree] if is_id else []) + source_flags(src)
        p = subprocess.Popen([hipcc] + flags + extra + ['-c', src, '-o', obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        running.append((p, src))
    while running:
        reap(True)
    if failed:
        raise RuntimeError('hipcc failed:\n' + '\n'.join('%s\n%s' % f for f in failed))
    r = subprocess.run([hipcc, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB_PATH] + objs,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError('link failed:\n' + r.stdout.decode(errors='replace'))
    with open(idf, 'w') as f:
        f.write(tree + '\n')
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
