"""Drop-in for net/utils/tgcn_multi3_fix_3A.py (class name kept)."""
from .tgcn import ConvTemporalGraphical3A as ConvTemporalGraphical  # noqa: F401
