#!/bin/bash
# the seven-tile fragment launch: parity tests, then A/B against eight tiles (QUICK_AMD_FRAG7=0) on the Llama-2-70B gate_up shape and the 70B decode row
mkdir -p gpurun_out/r06
out=gpurun_out/r06/frag7.txt; : > $out
timeout 900 python -m pytest tests/test_gemm_gpu.py -q -m gpu -k "seven_tile or runs_seven_tiles or eight_tile or (e2e_layer_shapes and 57344)" 2>&1 | tail -3 >> $out
for r in 1 2 3; do for f in 1 0; do
  echo "== QUICK_AMD_FRAG7=$f (round $r)" >> $out
  QUICK_AMD_FRAG7=$f timeout 300 python - >> $out 2>&1 <<'P'
import ctypes, numpy as np, torch
from quick_amd import _lib, packing, kernels
lib = _lib.load(); dev = torch.device("cuda:0"); G = 128
def arr(ts): return (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
for (M, K, N) in [(16, 8192, 57344), (12, 8192, 57344), (16, 8192, 28672)]:
    nsets = 3
    sets = [packing.random_mi355x(K, N, G, dev) for _ in range(nsets)]
    x = (torch.randn(M, K, device=dev) * 0.5).half(); y = torch.empty(M, N, dtype=torch.float16, device=dev); ws = torch.zeros(48 << 20, dtype=torch.uint8, device=dev)
    qa, sa, za = arr([s[0] for s in sets]), arr([s[1] for s in sets]), arr([s[2] for s in sets])
    it = 60; us = (ctypes.c_float * it)()
    rc = lib.quick_w4a16_gemm_profile(x.data_ptr(), qa, sa, za, nsets, y.data_ptr(), ws.data_ptr(), ws.numel(), M, K, N, G, 0, 0, it, us, None)
    sp = (ctypes.c_float * 48)()
    rc2 = lib.quick_w4a16_gemm_span(x.data_ptr(), qa, sa, za, nsets, y.data_ptr(), ws.data_ptr(), ws.numel(), M, K, N, G, 0, 0, 48, sp, None)
    algo = K * N / 2 + (K // G) * N * 2.5 + 2 * M * K + 2 * M * N
    d = float(np.median(np.asarray(us[:])[10:]))
    print(f"   {M} x {K} x {N}: dispatch {d:7.2f} us ({algo / d / 8e6 * 100:5.1f} % of 8 TB/s)  span {float(np.median(np.asarray(sp[:])[8:])) if rc2 == 0 else float('nan'):7.2f} us  [{kernels.plan_describe(M, K, N, G)}]")
    del sets
P
done; done
for f in 1 0; do echo "== decode Llama-2-70B bs=16, QUICK_AMD_FRAG7=$f" >> $out; QUICK_AMD_FRAG7=$f timeout 600 python bench_decode.py --model llama2-70b --bs 16 2>&1 | tail -3 >> $out; done
cat $out
