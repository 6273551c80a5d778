"""Ablation timing of k_transient_bins (make abl ABL=1..3 in csrc): which part of a tile costs what."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import nrc_amd
from nrc_amd import rc_ext
lib = sys.argv[1] if len(sys.argv) > 1 else ""
if lib:
    rc_ext.library_path = lambda: os.path.join(R, "tools", "diag", lib, "librc_hip.so")
import bench
r = bench.transient_line(0, torch.device("cuda:0"))
print(lib or "product", f"{r['ms_per_step']:.3f} ms/step")
