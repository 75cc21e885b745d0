"""Drop-in for net/utils/tgcn_multi3.py (class name kept)."""
from .tgcn import ConvTemporalGraphicalMulti3 as ConvTemporalGraphical  # noqa: F401
