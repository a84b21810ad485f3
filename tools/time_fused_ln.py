"""A/B: gemm with the RMSNorm prologue against rmsnorm + gemm, replayed from one hipGraph each.  argv: M K N ..."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quick_amd import kernels as K, packing

dev = torch.device("cuda:0")
args = [int(v) for v in sys.argv[1:]] or [64, 4096, 22016]
for M, Kd, N in zip(args[0::3], args[1::3], args[2::3]):
    G = 128
    sets = [packing.random_mi355x(Kd, N, G, dev) for _ in range(6)]
    x = torch.randn(M, Kd, device=dev).half()
    lnw = (torch.rand(Kd, device=dev) + 0.5).half()
    h = torch.empty_like(x)
    y = torch.empty(M, N // 2, dtype=torch.float16, device=dev)
    print(M, Kd, N, "fusable" if K.can_fuse_rmsnorm(M, Kd, N, G) else "NOT fusable", K.plan_describe(M, Kd, N, G))

    def fused(i):
        qw, sc, qz = sets[i % len(sets)]
        K.gemm_forward(x, qw, sc, qz, out=y, rmsnorm_weight=lnw, silu_mul=True)

    def two(i):
        qw, sc, qz = sets[i % len(sets)]
        K.rmsnorm(x, lnw, out=h)
        K.gemm_forward(h, qw, sc, qz, out=y, silu_mul=True)

    for name, fn in (("two launches", two), ("fused", fused), ("two launches", two), ("fused", fused)):
        if name == "fused" and not K.can_fuse_rmsnorm(M, Kd, N, G):
            continue
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for i in range(3):
                fn(i)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                for i in range(48):
                    fn(i)
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st); g.replay(); g.replay(); e1.record(st); torch.cuda.synchronize()
        print(f"   {name:14s} {e0.elapsed_time(e1) * 1e3 / 96:8.2f} us per layer-half")
