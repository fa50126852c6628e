#!/usr/bin/env python
"""The events leg of bench.py with the per-phase wall-clock laps of every step printed (HEYOKA_AMD_EVENTS_TIMING=1): which
phase of which step is slow in the terminal variant?"""
import os, sys
os.environ["HEYOKA_AMD_EVENTS_TIMING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
import heyoka_amd as hy
from heyoka_amd import configs
ctx = dict(torch=torch, hy=hy, configs=configs, dev=torch.device("cuda:0"), dev_index=0)
r = bench.events_leg(ctx, int(sys.argv[1]) if len(sys.argv) > 1 else 1048576, n_steps=int(sys.argv[2]) if len(sys.argv) > 2 else 6)
print({k: v for k, v in r.items() if k != "config"})
