"""lm_head of a decode step: quick_lm_head_argmax_f16 (final RMSNorm + fp16 GEMV + arg-max, two launches) against torch
(quick_rmsnorm_f16 + matmul + argmax), both as hipGraph chains.  Usage: python tools/time_lm_head.py [--vocab 32000 --hidden 4096]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from quick_amd import kernels  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--vocab", type=int, default=32000)
    ap.add_argument("--hidden", type=int, default=4096)
    ap.add_argument("--batches", default="1,2,4")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    V, H = a.vocab, a.hidden
    n_sets = max(2, -(-(320 << 20) // (V * H * 2)))           # distinct weight copies: every launch streams from HBM
    ws = [(torch.randn(V, H, device=dev) * 0.02).half() for _ in range(n_sets)]
    nw = torch.ones(H, device=dev).half()
    for B in [int(b) for b in a.batches.split(",")]:
        x = torch.randn(B, H, device=dev).half()
        res = {}
        for name in ("torch", "quick"):
            def step(i):
                w = ws[i % n_sets]
                if name == "torch":
                    return (kernels.rmsnorm(x, nw) @ w.t()).argmax(-1)
                return kernels.lm_head_argmax(x, w, nw, want_hidden=True)[0]
            for i in range(3):
                step(i)
            torch.cuda.synchronize()
            L = n_sets * 4
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for i in range(L):
                    step(i)
            g.replay()
            torch.cuda.synchronize()
            ts = []
            for _ in range(10):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                g.replay()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3 / L)
            ts.sort()
            res[name] = ts[len(ts) // 2]
        gb = V * H * 2 / 1e9
        print(f"B={B} V={V} H={H}: torch {res['torch']:.1f} us ({gb / res['torch'] * 1e3:.2f} TB/s)   quick {res['quick']:.1f} us ({gb / res['quick'] * 1e3:.2f} TB/s)", flush=True)


if __name__ == "__main__":
    main()
