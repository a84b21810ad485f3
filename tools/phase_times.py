"""Diagnostic: per-phase cycle totals of the tiled kernel (ablation bit 16 = s_memtime stamps), M=512 K=N=4096."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from quick_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
M, K, N, G = 512, 4096, 4096, 128
x = torch.randn(M, K, device=dev).half()
from quick_amd import packing
qw, sc, qz = packing.random_mi355x(K, N, G, dev)
y = torch.empty(M, N, dtype=torch.float16, device=dev)
ws = torch.zeros(4096 * 8 * 64 // 8, dtype=torch.int64, device=dev)
kid = 2 + ((16 + (int(sys.argv[1]) if len(sys.argv) > 1 else 0)) << 16)
for _ in range(3):
    rc = lib.quick_w4a16_gemm_f16_ex(x.data_ptr(), qw.data_ptr(), sc.data_ptr(), qz.data_ptr(), None, y.data_ptr(), ws.data_ptr(),
                                     ws.numel() * 8, M, K, N, G, kid, 0, None)
    assert rc == 0, _lib.last_error()
torch.cuda.synchronize()
d = ws.cpu().numpy().reshape(-1, 8)[: 256 * 8].reshape(256, 8, 8).astype(np.float64)
raw = ws.cpu().numpy().reshape(-1, 8)[: 256 * 8]
nst = float(int(raw[0, 5]) & 255)
names = ["x regs->LDS (vmcnt wait + ds_write)", "issue x loads", "compute (ds_read + dequant + mfma)", "issue w loads", "barrier"]
tot = d[:, :, :5].sum(axis=2).mean()
print(f"stages {nst:.0f}; mean cycles per wave in the K loop {tot:.0f} (s_memtime ticks = shader cycles)")
for i, n in enumerate(names):
    v = d[:, :, i]
    print(f"  {n:40s} {v.mean() / nst:8.0f} cyc/stage  ({100 * v.mean() / tot:4.1f} %)   wave spread {v.min() / nst:.0f}..{v.max() / nst:.0f}")
for wk in (0, 1):
    v = d[:, 4 * wk:4 * wk + 4, :5].mean(axis=(0, 1)) / nst
    print(f"  waves wk={wk}: " + "  ".join(f"{x:.0f}" for x in v))
# words 5-7 of a wave's record: stage count | start (100 MHz ticks) << 8;  prologue cycles;  cycles << 24 | 100 MHz ticks
# from kernel entry to the end of the K loop (the ratio is the shader clock the kernel actually ran at)
pro = d[:, :, 6].mean()
cyc = np.mean([int(v) >> 24 for v in raw[:, 7]])
rt = np.mean([int(v) & 0xffffff for v in raw[:, 7]])
print(f"prologue {pro:.0f} cycles; entry -> end of K loop {cyc:.0f} shader cycles = {rt:.1f} ticks of 100 MHz -> {cyc / (rt * 10):.2f} GHz")
st = np.array([int(v) >> 8 for v in raw[:, 5]], dtype=np.float64)
en = st + np.array([int(v) & 0xffffff for v in raw[:, 7]], dtype=np.float64)
print(f"wave start spread {(st.max() - st.min()) / 100:.2f} us; first start -> last K-loop end {(en.max() - st.min()) / 100:.2f} us; "
      f"per-wave entry -> loop end min/mean/max {(en - st).min() / 100:.2f}/{(en - st).mean() / 100:.2f}/{(en - st).max() / 100:.2f} us")
