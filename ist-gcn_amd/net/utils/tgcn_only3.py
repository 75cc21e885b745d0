"""Drop-in for net/utils/tgcn_only3.py (class name kept)."""
from .tgcn import ConvTemporalGraphicalOnly3 as ConvTemporalGraphical  # noqa: F401
