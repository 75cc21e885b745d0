"""ORACLE package: CPU restatement of the reference's hot path (test infrastructure only).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
