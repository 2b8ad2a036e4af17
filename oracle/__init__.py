"""Oracle package: CPU restatements of the reference hot path. TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never from recnext_amd/.
"""
