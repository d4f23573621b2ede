#!/usr/bin/env python3
"""Config-4 training epoch alone (for rocprofv3 --kernel-trace --stats)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
print(bench.epoch_time(0, 1, dev, False))
