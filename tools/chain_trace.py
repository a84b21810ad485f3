"""Exploration tool (not product): where the time of a chained launch goes.  Runs the four GEMMs of a Llama-2-7B decoder layer
(o, gate_up, down, next qkv) at M = 1 as one chain from the stamping build and prints, per task, when the workgroups passed
each point (us after the first workgroup's first stamp: min / median / max over workgroups)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quick_amd import kernels  # noqa: E402
from quick_amd.decoder import random_wqlinear  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hidden", type=int, default=4096)
    ap.add_argument("--inter", type=int, default=11008)
    ap.add_argument("--qkv", type=int, default=12288)
    ap.add_argument("--M", type=int, default=1)
    ap.add_argument("--sets", type=int, default=6, help="weight sets cycled through so that launches miss the Infinity Cache")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(0)
    H, I, Q, M = args.hidden, args.inter, args.qkv, args.M
    sets = [[random_wqlinear(H, H, 128, dev, gen), random_wqlinear(H, 2 * I, 128, dev, gen), random_wqlinear(I, H, 128, dev, gen),
             random_wqlinear(H, Q, 128, dev, gen)] for _ in range(args.sets)]
    att = torch.randn(M, H, device=dev).half()
    x = torch.randn(M, H, device=dev).half()
    act = torch.empty(M, I, device=dev, dtype=torch.float16)
    qkv = torch.empty(M, Q, device=dev, dtype=torch.float16)
    ln = torch.ones(H, device=dev, dtype=torch.float16)
    T = lambda m, xin, out, **kw: dict(in_feats=xin, kernel=m.qweight, scaling_factors=m.scales, zeros=m.qzeros, out=out, **kw)
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    names = ["o", "gate_up", "down", "qkv"]
    points = ["begin", "cells arrived", "x staged", "blocks done", "residuals in", "next ctx", "prefetch out", "last mfma"]
    order = [0, 1, 2, 4, 5, 6, 7, 3]
    acc = []
    for it in range(3 * args.sets):
        o, gu, dn, qk = sets[it % args.sets]
        tasks = [T(o, att, x, residual=x), T(gu, x, act, rmsnorm_weight=ln, silu_mul=True), T(dn, act, x, residual=x),
                 T(qk, x, qkv, rmsnorm_weight=ln)]
        trace = torch.zeros(cus, 6, 8, dtype=torch.int64, device=dev)
        kernels.gemm_chain(tasks, trace=trace)
        torch.cuda.synchronize()
        if it >= args.sets:
            t = trace.cpu().numpy().astype(np.float64)[:, :4, :8]
            t = (t - t[:, 0, 0].min()) / 100.0
            acc.append(t)
    t = np.mean(acc, axis=0)        # [wg, task, point]
    print(f"chain of {names} at M={M} (H={H} I={I} qkv={Q}), mean of {len(acc)} launches; us since the first workgroup began")
    for k, n in enumerate(names):
        for p in order:
            pn = points[p]
            v = t[:, k, p]
            print(f"  {n:8s} {pn:13s} min {v.min():7.2f}  median {np.median(v):7.2f}  max {v.max():7.2f}")
    print(f"  whole launch (last block of the last task): {t[:, 3, 3].max():.2f} us")


if __name__ == "__main__":
    main()
