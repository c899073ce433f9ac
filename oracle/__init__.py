"""CPU oracle (test infrastructure only -- see oracle/pcgol_oracle.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package; the product (pcgol_amd) never does.
"""
from .oracle import *  # noqa: F401,F403
