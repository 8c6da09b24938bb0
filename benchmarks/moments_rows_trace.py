import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cheetah_amd import _ops
for B, N in [(4096, 1000), (65535, 100), (256, 16000)]:
    x = torch.randn(B, N, 7, device="cuda"); w = torch.rand(B, N, device="cuda")
    for _ in range(5):
        _ops._moments_raw(x, w, B, N)
    torch.cuda.synchronize()
