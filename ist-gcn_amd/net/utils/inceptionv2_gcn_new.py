"""Drop-in for net/utils/inceptionv2_gcn_new.py (identical arithmetic to inceptionv2_gcn.py)."""
from .inceptionv2_gcn import BasicConv2d, Inception2  # noqa: F401
