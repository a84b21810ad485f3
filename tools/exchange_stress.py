"""Stress of the exchange launches' give-up paths (not part of the test suite: runs for minutes).

Three streams at once: two launch exchange GEMMs (four-wave and exchange-K tiles, 2 / 4 / 8 slices, random shapes out of a fixed set) with a
random poll limit per launch (20 ns .. 5 us, so that owners give up while their partners are on their way, partners claim, owners win
their blocks back ...), the third keeps most of the chip busy with dense matmuls of changing size.  Every result must equal, bit for bit,
the one the same launch gave undisturbed, and the workspace must be all-zero again at the end of every round.
    python tools/exchange_stress.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from quick_amd import kernels, packing

XK, XW = 4, 5
def xw(mb, pairs, s): return XW | ((mb << 4) if mb != 4 else 0) | ((1 << 12) if pairs == 1 else 0) | (s << 8)
def xk(mb, s): return XK | (mb << 4) | (s << 8)

dev = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
ONLY = [int(v) for v in os.environ.get("STRESS_CASES", "").split(",") if v]          # (debugging aids: a subset of the cases, one stream, no noise, fixed limits)
NSTREAMS = int(os.environ.get("STRESS_STREAMS", "2"))
NOISE = int(os.environ.get("STRESS_NOISE", "1"))
POLLS = [int(v) for v in os.environ.get("STRESS_POLLS", "1,2,4,6,7,8,9,12").split(",")]
gen = torch.Generator(device=dev).manual_seed(7)
cases = []
for (M, K, N), kid in (((512, 4096, 4096), xw(4, 2, 4)), ((512, 4096, 4096), xw(4, 1, 2)), ((256, 4096, 4096), xw(2, 1, 2)), ((1024, 4096, 4096), xw(4, 2, 2)),
                       ((128, 8192, 2048), xw(4, 1, 4)), ((300, 2048, 4096), xw(4, 1, 2)), ((128, 4096, 4096), xk(2, 4)), ((64, 11008, 4096), xk(2, 8)),
                       ((512, 4096, 4096), xk(4, 2)), ((200, 4096, 2048), xk(4, 4)), ((64, 4096, 6144), xk(2, 4)), ((96, 8192, 1024), xk(2, 8))):
    qw, sc, qz = packing.random_mi355x(K, N, 128, dev, gen)
    x = (torch.randn(M, K, device=dev, generator=gen) * 0.5).half()
    plan = kernels.plan_describe(M, K, N, 128, kid)
    assert "slices=1 " not in plan, plan
    y0 = kernels.gemm_forward(x, qw, sc, qz, kernel_id=kid)
    ref = x.float() @ kernels.dequantize_mi355x(qw, sc, qz).float()
    assert ((y0.float() - ref).abs().max() / ref.abs().max()).item() < 2e-3, plan
    cases.append((x, (qw, sc, qz), kid, y0, plan))
torch.cuda.synchronize()
if ONLY:
    cases = [cases[i] for i in ONLY]
streams = [torch.cuda.Stream() for _ in range(NSTREAMS)]
noise = torch.cuda.Stream()
bigs = [torch.randn(n, n, device=dev).half() for n in (2048, 4096, 8192)]
rng = np.random.default_rng(1)
t0, rounds, launches, bad = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    with torch.cuda.stream(noise):
        for _ in range(int(rng.integers(0, 3)) if NOISE else 0):
            b = bigs[int(rng.integers(0, 3))]
            torch.matmul(b, b)
    outs = []
    for si, st in enumerate(streams):
        with torch.cuda.stream(st):
            for _ in range(int(rng.integers(2, 7))):
                c = cases[int(rng.integers(0, len(cases)))]
                os.environ["QUICK_AMD_EXCHANGE_POLL_LOG2"] = str(int(rng.choice(POLLS)))
                out = torch.full_like(c[3], float("nan"))   # (poisoned: a block nobody finished shows, whatever the allocator hands back)
                outs.append((kernels.gemm_forward(c[0], *c[1], kernel_id=c[2], out=out), c))
                launches += 1
    os.environ.pop("QUICK_AMD_EXCHANGE_POLL_LOG2", None)
    torch.cuda.synchronize()
    for y, c in outs:
        if not torch.equal(y, c[3]):
            bad += 1
            print("MISMATCH", c[4], (y.float() - c[3].float()).abs().max().item(), flush=True)
    for key, ws in kernels._WORKSPACES.items():
        head = ws[: (64 << 10) + (16 << 20)]
        if int(head.count_nonzero()) != 0:
            bad += 1
            w = head.view(torch.int32)
            nz = w.nonzero().flatten()
            print(f"round {rounds}: workspace {key} not handed back zeroed: {len(nz)} words, first at word {int(nz[0])} (= {hex(int(w[nz[0]]) & 0xffffffff)}), "
                  f"{int((nz < 16384).sum())} of them in the counter region; launches of the round: {[c[4][:60] for _, c in outs]}", flush=True)
            head.zero_()
    rounds += 1
    if bad >= 12:
        break
print(f"{rounds} rounds, {launches} exchange launches under contention with poll limits of 20 ns .. 41 us: {bad} failures")
sys.exit(1 if bad else 0)
