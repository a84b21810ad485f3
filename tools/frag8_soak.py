"""Soak of the straight-line eight-tile fragment kernel (w4a16_frag8_kernel, forced by kernel id) on random shapes: 9..16 tokens, every T it is built
for, 1 / 2 / 4 K slices, random epilogue, each launch twice (bit-equal) against fp32 matmul over the GPU-dequantised weights, into NaN-poisoned
outputs, with a dense matmul on a second stream as noise.     python tools/frag8_soak.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from quick_amd import kernels, packing
dev = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(5)
gen = torch.Generator(device=dev)
noise, big = torch.cuda.Stream(), torch.randn(4096, 4096, device=dev).half()
kid = kernels.KERNEL_SKINNY | (8 << 4)
t0, n, worst, seen = time.time(), 0, 0.0, set()
while time.time() - t0 < budget:
    ks = int(rng.choice([1, 2, 4]))
    T = int(rng.choice([2, 4, 7, 8]))
    K = 128 * 8 * ks * T
    N = int(rng.integers(1, 49)) * 128
    M = int(rng.integers(9, 17))
    plan = kernels.plan_describe(M, K, N, 128, kid, ks)
    assert plan.startswith("skinny ntw=8 ") and f"ksplit={ks}" in plan, plan
    gen.manual_seed(n)
    qw, sc, qz = packing.random_mi355x(K, N, 128, dev, generator=gen)
    x = (torch.randn(M, K, device=dev, generator=gen) * 0.5).half()
    ref = x.float() @ kernels.dequantize_mi355x(qw, sc, qz).float()
    mode = int(rng.integers(0, 3))
    bias = torch.randn(N, device=dev, generator=gen).half() if mode == 1 else None
    res = torch.randn(M, N, device=dev, generator=gen).half() if mode == 1 else None
    if mode == 1:
        ref = ref + bias.float() + res.float()
    if rng.integers(0, 2):
        with torch.cuda.stream(noise):
            torch.matmul(big, big)
    outs = []
    for _ in range(2):
        out = torch.full((M, N // 2 if mode == 2 else N), float("nan"), dtype=torch.float16, device=dev)
        outs.append(kernels.gemm_forward(x, qw, sc, qz, bias=bias, residual=res, silu_mul=(mode == 2), kernel_id=kid, grid_split_k=ks, out=out))
    assert torch.equal(outs[0], outs[1]), (n, M, K, N, ks, mode)
    if mode == 2:
        r = ref.view(M, N // 16, 2, 8)
        ref = (torch.nn.functional.silu(r[:, :, 0].half().float()).half().float() * r[:, :, 1].half().float()).reshape(M, N // 2)
    err = ((outs[0].float() - ref).abs().max() / ref.abs().max()).item()
    assert bool(torch.isfinite(outs[0]).all()) and err <= (4e-3 if mode == 2 else 2e-3), (n, M, K, N, ks, mode, err, plan)
    worst = max(worst, err); seen.add((T, ks)); n += 1
torch.cuda.synchronize()
print(f"{n} random eight-tile launches x 2 (T x slices combinations seen: {len(seen)}), all finite, bit-equal in pairs, worst relative error {worst:.2e}")
