"""Diagnostic: time the individual launches of one decode layer (bs given), each as a 64-deep hipGraph chain."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quick_amd import kernels as K
from quick_amd.decoder import random_wqlinear
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
H, I, nh, nkv, D, L, G = 4096, 11008, 32, 32, 128, 256, 128
g = torch.Generator(device=dev).manual_seed(0)
sets = 12   # rotate weights so they come from HBM
qkv = [random_wqlinear(H, 3 * H, G, dev, g) for _ in range(sets)]
o = [random_wqlinear(H, H, G, dev, g) for _ in range(sets)]
gu = [random_wqlinear(H, 2 * I, G, dev, g) for _ in range(sets)]
dn = [random_wqlinear(I, H, G, dev, g) for _ in range(sets)]
x = torch.randn(B, H, device=dev).half(); lnw = torch.ones(H, device=dev).half()
h = torch.empty_like(x); qkv_o = torch.empty(B, 3 * H, device=dev).half(); att = torch.empty(B, H, device=dev).half()
gu_o = torch.empty(B, 2 * I, device=dev).half(); act = torch.empty(B, I, device=dev).half(); q_o = torch.empty(B, nh, D, device=dev).half()
kc = torch.randn(B, nkv, L, D, device=dev).half(); vc = torch.randn(B, nkv, L, D, device=dev).half()
pos = torch.full((1,), 200, dtype=torch.int64, device=dev)
ang = torch.outer(torch.arange(L, device=dev).float(), 1.0 / (10000 ** (torch.arange(0, D, 2, device=dev).float() / D)))
cos, sin = torch.cat((ang.cos(), ang.cos()), -1).half(), torch.cat((ang.sin(), ang.sin()), -1).half()
gm = lambda m, xin, out, **kw: K.gemm_forward(xin, m.qweight, m.scales, m.qzeros, out=out, **kw)
ops = {
  "rmsnorm": lambda i: K.rmsnorm(x, lnw, out=h),
  "qkv gemm": lambda i: gm(qkv[i % sets], h, qkv_o),
  "rope_kv_append": lambda i: K.rope_kv_append(qkv_o, cos, sin, pos, q_o, kc, vc, nh, nkv, D),
  "decode_attention": lambda i: K.decode_attention(q_o, kc, vc, pos, att, nh, nkv, D),
  "rope_attention (fused)": lambda i: K.rope_attention(qkv_o, cos, sin, pos, kc, vc, att, nh, nkv, D),
  "o gemm + residual": lambda i: gm(o[i % sets], att, x, residual=x),
  "gate_up gemm": lambda i: gm(gu[i % sets], h, gu_o),
  "gate_up gemm + silu_mul": lambda i: gm(gu[i % sets], h, act, silu_mul=True),
  "silu_mul": lambda i: K.silu_mul(gu_o, out=act),
  "down gemm + residual": lambda i: gm(dn[i % sets], act, x, residual=x),
}
if K.can_fuse_rmsnorm(B, H, 3 * H, G):
    ops["qkv gemm + rmsnorm"] = lambda i: gm(qkv[i % sets], x, qkv_o, rmsnorm_weight=lnw)
    ops["gate_up gemm + rmsnorm + silu_mul"] = lambda i: gm(gu[i % sets], x, act, rmsnorm_weight=lnw, silu_mul=True)
for name, fn in ops.items():
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for i in range(64): fn(i)
    gr.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); gr.replay(); e.record(); torch.cuda.synchronize()
    print(f"B={B} {name:36s} {s.elapsed_time(e) / 64 * 1000:7.2f} us per launch")
