import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import bench
print(bench.transient_line(0, torch.device("cuda:0")))
