"""Times quick_decode_rope_attention_f16 alone: us per launch vs batch, KV heads and context length (hipGraph replay)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quick_amd import kernels as K_
dev = torch.device("cuda:0")
D, L = 128, 1024
ang = torch.outer(torch.arange(L, device=dev).float(), 1.0 / (10000 ** (torch.arange(0, D, 2, device=dev).float() / D)))
cos, sin = torch.cat((ang.cos(), ang.cos()), -1).half(), torch.cat((ang.sin(), ang.sin()), -1).half()
for B, nh, nkv in [tuple(int(v) for v in a.split('x')) for a in (sys.argv[1:] or ['1x32x32', '64x32x32', '64x32x8', '16x32x8', '16x64x8', '64x64x8'])]:
    layers = 4  # rotate caches so that they are HBM-cold like in a model
    kc = [torch.randn(B, nkv, L, D, device=dev).half() for _ in range(layers)]
    vc = [torch.randn(B, nkv, L, D, device=dev).half() for _ in range(layers)]
    qkv = torch.randn(B, (nh + 2 * nkv) * D, device=dev).half()
    out = torch.empty(B, nh * D, dtype=torch.float16, device=dev)
    for ctx in (128, 192, 512):
        pos = torch.full((1,), ctx, dtype=torch.int64, device=dev)
        def run():
            for i in range(layers):
                K_.rope_attention(qkv, cos, sin, pos, kc[i], vc[i], out, nh, nkv, D)
        run(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(5): run()
        g.replay(); torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record(); g.replay(); g.replay(); t1.record(); torch.cuda.synchronize()
        us = t0.elapsed_time(t1) * 1e3 / (2 * 5 * layers)
        mb = B * nkv * ctx * D * 2 * 2 / 1e6
        print(f"B={B:3d} nh={nh} nkv={nkv:2d} ctx={ctx:4d}  {us:7.2f} us   KV {mb:7.1f} MB  {mb / us * 1e-3 * 1e3:6.2f} GB/ms" )
