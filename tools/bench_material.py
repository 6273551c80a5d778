import sys
sys.path.insert(0, ".")
import torch, bench
print(bench.material_line(0, torch.device("cuda:0")))
