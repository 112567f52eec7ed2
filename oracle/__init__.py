"""CPU oracle for Neumann's vector_engine SIMILAR TOP-K path.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product (neumann_amd/, libneumann_gpu.so) never imports this package.
See the header of oracle/nmn_oracle.c for what is restated and how parity is pinned.
"""
