"""Profiling driver (not product): run the W4A16 GEMM `iters` times at one shape so that a rocprofv3 pass sees a
clean population of dispatches.   python tools/prof_gemm.py --M 512 [--kernel 0] [--iters 30]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ap = argparse.ArgumentParser()
ap.add_argument("--M", type=int, nargs="+", default=[512])
ap.add_argument("--K", type=int, default=4096)
ap.add_argument("--N", type=int, default=4096)
ap.add_argument("--G", type=int, default=128)
ap.add_argument("--kernel", type=int, default=0)
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--sets", type=int, default=38)
a = ap.parse_args()
import quick_amd
from quick_amd import packing
dev = torch.device("cuda:0")
K, N, G = a.K, a.N, a.G
gen = torch.Generator(device=dev).manual_seed(1)
sets = []
for _ in range(a.sets):
    qw, sc, qz = packing.random_mi355x(K, N, G, dev, gen)
    sets.append((qw, sc, qz))
for M in a.M:
    x = torch.randn(M, K, device=dev, generator=gen).half()
    torch.cuda.synchronize()
    for i in range(a.iters):
        y = quick_amd.gemm_forward(x, *sets[i % a.sets], kernel_id=a.kernel)
    torch.cuda.synchronize()
print("done")
