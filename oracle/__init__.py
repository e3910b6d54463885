"""CPU oracle for the learning3d hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  Nothing under learning3d_amd/ does (tests/test_no_oracle_leak.py
enforces it).  See oracle.c for the restated algorithms and their reference
citations, and tests/golden/make_golden.py for how the oracle is pinned against
the real reference.
"""
from .oracle import *  # noqa: F401,F403
