#!/usr/bin/env python3
"""Per-rank SpMM time of the column-sharded strategy at config 5, measured on ONE GPU: with N ranks
each rank runs the full CSR on F/N feature columns, so the single-GPU time at width F/N is the
per-rank compute time of an N-GPU run (communication not included)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.spmm_sweep import sweep
sweep(10_000_000, 200_000_000, [32, 64, 128, 256], seed=43)
