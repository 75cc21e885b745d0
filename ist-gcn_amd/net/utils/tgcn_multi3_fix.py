"""Drop-in for net/utils/tgcn_multi3_fix.py (class name kept)."""
from .tgcn import ConvTemporalGraphicalMulti3Fix as ConvTemporalGraphical  # noqa: F401
