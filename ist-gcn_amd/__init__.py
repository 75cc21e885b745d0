"""istgcn_amd -- MI355X-native engine for the IST-GCN st_gcn hot path.

Import name `istgcn_amd` (the directory is `ist-gcn_amd/`; `istgcn_amd.py` at the repo root aliases
it because a hyphen is not importable).  Layout:
  csrc/          hand-written HIP (gfx950) kernels + the extern "C" ABI of include/istgcn.h
  _lib.py        hipcc build + ctypes loader (no fallback: missing library => RuntimeError)
  ops.py         tensor-level wrappers over the C ABI
  net/           drop-in mirrors of the reference's `net.*` Model classes (same ctor, forward,
                 extract_feature, state_dict keys)
"""
from . import _lib  # noqa: F401

__all__ = ['_lib']
