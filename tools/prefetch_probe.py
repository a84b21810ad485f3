"""Does pulling the NEXT launch's weights into the memory-side cache on a second stream shorten a chain of small-M GEMMs?
A chain of L launches over distinct weight sets (> 320 MiB in total, so every set comes from HBM) is captured in a hipGraph
three ways: plain; with quick_prefetch of set i + 1 on a side stream forked when launch i is enqueued; one set only (the
cache-resident bound).  Usage: python tools/prefetch_probe.py --shapes 1x4096x4096,1x4096x22016 [--workgroups 64] [--ahead 1]"""
import argparse
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from quick_amd import kernels, packing  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="1x4096x4096,1x4096x12288,1x4096x22016,1x11008x4096")
    ap.add_argument("--workgroups", default="64")
    ap.add_argument("--ahead", type=int, default=1)
    ap.add_argument("--frac", type=float, default=1.0, help="fraction of the next weight matrix to pull")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--kernel-id", type=int, default=0, help="kernel id of the 'hint' and 'plainid' chains (e.g. 1 | 1 << 22: one persistent slot per CU)")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(7)
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    for shape in a.shapes.split(","):
        M, K, N = map(int, shape.split("x"))
        G = 128
        set_bytes = K * N // 2
        n_sets = max(4, -(-(320 << 20) // set_bytes))
        sets = [packing.random_mi355x(K, N, G, dev, gen) for _ in range(n_sets)]
        x = torch.randn(M, K, dtype=torch.float16, device=dev) * 0.5
        L = n_sets * 2
        side = torch.cuda.Stream()

        def chain(mode, wg):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for i in range(L):
                    qw, sc, qz = sets[0] if mode == "hot" else sets[i % n_sets]
                    if mode in ("pair", "pfonly"):   # same stream: pull set i, then run on it
                        kernels.prefetch(qw, wg)
                    if mode == "hint":      # the launch itself pulls the next set
                        kernels.gemm_forward(x, qw, sc, qz, prefetch=sets[(i + 1) % n_sets][0], kernel_id=a.kernel_id)
                    elif mode == "hint_tiny":   # the extra wave with (almost) nothing to touch: what the ninth wave itself costs
                        kernels.gemm_forward(x, qw, sc, qz, prefetch=sets[(i + 1) % n_sets][0].view(-1)[:64], kernel_id=a.kernel_id)
                    elif mode == "hint_hot":    # own stream cache-resident (set 0), cold touches
                        kernels.gemm_forward(x, *sets[0], prefetch=sets[(i + 1) % n_sets][0], kernel_id=a.kernel_id)
                    elif mode == "hint_same":   # touch what was touched before: the touches hit
                        kernels.gemm_forward(x, qw, sc, qz, prefetch=sets[0][0], kernel_id=a.kernel_id)
                    elif mode == "plainid":
                        kernels.gemm_forward(x, qw, sc, qz, kernel_id=a.kernel_id)
                    elif mode != "pfonly":
                        kernels.gemm_forward(x, qw, sc, qz)
            return g

        line = [f"{shape:>16}"]
        wx = torch.randn(2048, 2048, dtype=torch.float16, device=dev)
        wx @ wx
        torch.cuda.synchronize()
        warm = torch.cuda.CUDAGraph()
        with torch.cuda.graph(warm):
            for _ in range(4):
                wx @ wx
        modes = [("plain", 0), ("hot", 0), ("hint", 0), ("hint_tiny", 0), ("hint_hot", 0), ("hint_same", 0)]
        for w in [w for w in a.workgroups.split(",") if w]:
            for dw in (1, 2, 4, 32):
                modes += [("pfonly", int(w) | (dw << 16)), ("pair", int(w) | (dw << 16))]
        for mode, wg in modes:
            g = chain(mode, wg)
            g.replay()
            torch.cuda.synchronize()
            best = []
            for _ in range(a.reps):
                flush.fill_(1)
                for _ in range(3):
                    warm.replay()      # clocks back up on something else
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                g.replay()
                e1.record()
                torch.cuda.synchronize()
                best.append(e0.elapsed_time(e1) * 1e3 / L)
            best.sort()
            line.append(f"{mode}{'' if wg == 0 else '/wg%d/dw%d' % (wg & 0xffff, wg >> 16)}: {best[len(best) // 2]:6.2f}")
        print("  ".join(line), flush=True)


if __name__ == "__main__":
    main()
