"""Exploration tool (not product): the four GEMMs of a decoder layer after attention (o, gate_up, down, next qkv) at M tokens --
one chained launch against four single launches, both replayed from a hipGraph holding `layers` such groups (different
weights per group, so nothing is cache-resident), same session."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quick_amd import kernels  # noqa: E402
from quick_amd.decoder import random_wqlinear  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hidden", type=int, default=4096)
    ap.add_argument("--inter", type=int, default=11008)
    ap.add_argument("--qkv", type=int, default=12288)
    ap.add_argument("--M", type=int, nargs="+", default=[1])
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--reps", type=int, default=30)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(0)
    H, I, Q = args.hidden, args.inter, args.qkv
    sets = [[random_wqlinear(H, H, 128, dev, gen), random_wqlinear(H, 2 * I, 128, dev, gen), random_wqlinear(I, H, 128, dev, gen),
             random_wqlinear(H, Q, 128, dev, gen)] for _ in range(args.layers)]
    ln = torch.ones(H, device=dev, dtype=torch.float16)
    for M in args.M:
        att = torch.randn(M, H, device=dev).half()
        x = torch.randn(M, H, device=dev).half()
        act = torch.empty(M, I, device=dev, dtype=torch.float16)
        qkv = torch.empty(M, Q, device=dev, dtype=torch.float16)
        T = lambda m, xin, out, **kw: dict(in_feats=xin, kernel=m.qweight, scaling_factors=m.scales, zeros=m.qzeros, out=out, **kw)

        def group(s, chained):
            o, gu, dn, qk = s
            tasks = [T(o, att, x, residual=x), T(gu, x, act, rmsnorm_weight=ln, silu_mul=True), T(dn, act, x, residual=x),
                     T(qk, x, qkv, rmsnorm_weight=ln)]
            if chained:
                kernels.gemm_chain(tasks)
            else:
                for t in tasks:
                    t = dict(t)
                    kernels.gemm_forward(t.pop("in_feats"), t.pop("kernel"), t.pop("scaling_factors"), t.pop("zeros"), **t)

        res = {}
        for chained in (False, True):
            x.copy_(torch.randn(M, H, device=dev).half())
            for s in sets:
                group(s, chained)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for s in sets:
                    group(s, chained)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            times = []
            for _ in range(args.reps):
                ev[0].record()
                g.replay()
                ev[1].record()
                torch.cuda.synchronize()
                times.append(ev[0].elapsed_time(ev[1]) * 1e3 / args.layers)
            times.sort()
            res[chained] = times[len(times) // 2]
        print(f"M={M}: four launches {res[False]:.2f} us per layer group, one chained launch {res[True]:.2f} us  ({res[False] / res[True]:.3f}x)", flush=True)


if __name__ == "__main__":
    main()
