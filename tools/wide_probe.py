#!/usr/bin/env python3
"""Timing + parity probe of kernel variants on one GPU (builder tool, not a test).

    python tools/wide_probe.py [--shapes MxKxN,...] [--variants name=kernel_id[:split],...] [--iters 40]

For every shape: the result of each variant against the planner's default kernel (max relative difference), and the
kernel's own duration (event pair per dispatch, weight sets cycled past the Infinity Cache), best-of-3 medians.
"""
import argparse
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quick_amd import _lib, packing, gemm_forward  # noqa: E402

WIDE = 3


def wide(mb, pairs):
    return WIDE | (mb << 4) | (pairs << 8)


NORING = 1 << 12


def ring(n):
    return n << 22


DEFAULT_VARIANTS = {
    "auto": (0, 0), "tiled": (2, 0), "tiled_big": (2 | (1 << 27), 0),
    "w2x1": (wide(2, 1), 0), "w2x1n4": (wide(2, 1) | ring(4), 0), "w2x1n3": (wide(2, 1) | ring(3), 0), "w2x1nr": (wide(2, 1) | NORING, 0),
    "w2x2": (wide(2, 2), 0), "w2x2nr": (wide(2, 2) | NORING, 0),
    "w4x1": (wide(4, 1), 0), "w4x1nr": (wide(4, 1) | NORING, 0), "w4x2": (wide(4, 2), 0), "w4x2nr": (wide(4, 2) | NORING, 0),
    "w8x1": (wide(8, 1), 0), "w8x2": (wide(8, 2), 0),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="512x4096x4096,1024x4096x4096,2048x4096x4096,4096x4096x4096,8192x4096x4096,4096x8192x8192")
    ap.add_argument("--variants", default="")
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--G", type=int, default=128)
    ap.add_argument("--out", default="")
    ap.add_argument("--xfill", default="randn", help="randn | zeros | ones: what the activations hold (DVFS experiments)")
    args = ap.parse_args()
    variants = dict(DEFAULT_VARIANTS)
    if args.variants:
        variants = {}
        for tok in args.variants.split(","):
            name, spec = tok.split("=")
            kid, _, sp = spec.partition(":")
            variants[name] = (int(kid, 0), int(sp or 0))
    lib = _lib.load()
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(7)
    stream = torch.cuda.current_stream()
    G = args.G
    rows = []
    for spec in args.shapes.split(","):
        M, K, N = (int(v) for v in spec.lower().split("x"))
        sb = K * N // 2 + (K // G) * 2 * N * 2 + (K // G) * (N // 4) * 4
        ns = max(2, -(-(320 << 20) // sb))
        sets = [packing.random_mi355x(K, N, G, dev, gen) for _ in range(ns)]
        arr = lambda i: (ctypes.c_void_p * ns)(*[st[i].data_ptr() for st in sets])
        qa_, sa_, za_ = arr(0), arr(1), arr(2)
        x = (torch.randn((M, K), device=dev, generator=gen) * 0.5).half()
        if args.xfill == "zeros":
            x.zero_()
        elif args.xfill == "ones":
            x.fill_(1.0)
        y = torch.empty((M, N), dtype=torch.float16, device=dev)
        ref = gemm_forward(x, *sets[0]).float()
        scale = float(ref.abs().max())
        flops = 2.0 * M * N * K
        for name, (kid, split) in variants.items():
            try:
                got = gemm_forward(x, *sets[0], kernel_id=kid, grid_split_k=split).float()
            except Exception as e:  # unsupported combination
                print(f"{spec:>18} {name:>10}: {type(e).__name__}: {e}", flush=True)
                continue
            torch.cuda.synchronize()
            diff = float((got - ref).abs().max()) / max(scale, 1e-30)
            wsb = lib.quick_w4a16_workspace_bytes_ex(M, K, N, G, kid, split)
            ws = torch.zeros(max(wsb, 1), dtype=torch.uint8, device=dev)
            meds = []
            for _ in range(3):
                kus = (ctypes.c_float * args.iters)()
                rc = lib.quick_w4a16_gemm_profile(x.data_ptr(), qa_, sa_, za_, ns, y.data_ptr(), ws.data_ptr(), wsb, M, K, N, G,
                                                  kid, split, args.iters, kus, stream.cuda_stream)
                if rc != 0:
                    raise RuntimeError(_lib.last_error())
                meds.append(float(np.median(np.asarray(kus[:])[3:])))
            us = min(meds)
            buf = ctypes.create_string_buffer(256)
            lib.quick_w4a16_plan_describe(M, K, N, G, kid, split, buf, 256)
            row = {"shape": spec, "variant": name, "kernel_us": us, "tflops": flops / us / 1e6, "frac_mfma_peak": flops / us / 1e6 / 2500.0,
                   "rel_diff_vs_auto": diff, "plan": buf.value.decode()}
            rows.append(row)
            print(f"{spec:>18} {name:>10}: {us:9.2f} us  {row['tflops']:8.1f} TF  ({100 * row['frac_mfma_peak']:5.1f}%)  diff {diff:.1e}  {row['plan']}", flush=True)
        del sets
    if args.out:
        with open(args.out, "w") as f:
            for r in rows:
                f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
