"""Does pulling the NEXT launch's weights into the memory-side cache shorten a chain of small-M GEMMs?  (It does not: DESIGN.md 8.)
A chain of L launches over distinct weight sets (> 320 MiB in total, so every set comes from HBM) is captured in a hipGraph: plain;
one set only (hot: the L2-resident bound); quick_prefetch of set i + 1 on a side stream forked when launch i is enqueued (fork);
quick_prefetch of set i and then the launch on it, same stream (pair; pfonly = the touch kernels alone).  The in-kernel variant -- one
extra wave per workgroup touching the next set -- is in the history (commit "Experiment: pulling the next launch's weights ..."),
its numbers in profiles/r03_prefetch_probe.txt.  Usage: python tools/prefetch_probe.py --shapes 1x4096x4096,1x4096x22016 --workgroups 64,256"""
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
    ap.add_argument("--kernel-id", type=int, default=0, help="kernel id of the GEMM launches")
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
                main = torch.cuda.current_stream()
                for i in range(L):
                    qw, sc, qz = sets[0] if mode == "hot" else sets[i % n_sets]
                    if mode == "fork":      # second stream, forked when launch i is enqueued: pull set i + 1
                        ev = torch.cuda.Event()
                        ev.record(main)
                        side.wait_event(ev)
                        kernels.prefetch(sets[(i + 1) % n_sets][0], wg, side)
                    if mode in ("pair", "pfonly"):   # same stream: pull set i, then run on it
                        kernels.prefetch(qw, wg)
                    if mode != "pfonly":
                        kernels.gemm_forward(x, qw, sc, qz, kernel_id=a.kernel_id)
                if mode == "fork":
                    ev = torch.cuda.Event()
                    ev.record(side)
                    main.wait_event(ev)
            return g

        line = [f"{shape:>16}"]
        wx = torch.randn(2048, 2048, dtype=torch.float16, device=dev)
        wx @ wx
        torch.cuda.synchronize()
        warm = torch.cuda.CUDAGraph()
        with torch.cuda.graph(warm):
            for _ in range(4):
                wx @ wx
        modes = [("plain", 0), ("hot", 0)]
        for w in [int(w) for w in a.workgroups.split(",") if w]:
            modes.append(("fork", w))
            for dw in (1, 2, 4, 32):
                modes += [("pfonly", w | (dw << 16)), ("pair", w | (dw << 16))]
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
