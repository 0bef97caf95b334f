"""CPU oracle for the pynndescent build path -- TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  See nnd_oracle.c for the restatement and its reference
citations.
"""
