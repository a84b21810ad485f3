#!/usr/bin/env python3
"""End-to-end decode throughput on synthetic AWQ-QUICK (w4, g128) decoder stacks -- the second half of the BASELINE
metric ("decode tok/s Llama-2-7B bs=1,64").  Methodology of the reference's examples/benchmark.py:38-67,127-129:
prefill tok/s = ctx * bs / prefill time; decode tok/s = bs / median(step time); prefill/decode = 128/128.

    python bench_decode.py --model llama2-7b --bs 1 64            # one JSON line per (model, bs)

Weights are random packed tensors (no checkpoints / network here); every projection runs through WQLinear_QUICK ->
libquick_amd.so; attention/norm/RoPE are torch ops.  One decode step is captured in a hipGraph and replayed.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", nargs="+", default=["llama2-7b"], help="llama2-7b mistral-7b llama2-70b tiny")
    ap.add_argument("--bs", type=int, nargs="+", default=[1, 64])
    ap.add_argument("--ctx", type=int, default=128)
    ap.add_argument("--gen", type=int, default=128)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--torch-glue", action="store_true", help="eager torch ops around the GEMMs instead of the HIP glue kernels")
    args = ap.parse_args()
    import torch
    from quick_amd.decoder import CONFIGS, SyntheticDecoder, run_generation
    dev = torch.device("cuda:0")
    for name in args.model:
        cfg = CONFIGS[name]
        for bs in args.bs:
            model = SyntheticDecoder(cfg, bs, args.ctx + args.gen, dev)
            torch.cuda.synchronize()
            fused = not args.torch_glue and cfg.head_dim == 128
            run_generation(model, args.ctx, min(args.gen, 8), use_graph=False, fused=fused)   # warm-up (allocator, lazy init)
            prefill, steps = run_generation(model, args.ctx, args.gen, use_graph=not args.no_graph, fused=fused)
            med = float(np.median(steps))
            out = {"metric": "decode_tok_s", "model": cfg.name, "batch": bs, "prefill_len": args.ctx, "decode_len": args.gen,
                   "prefill_tok_s": args.ctx * bs / prefill, "decode_tok_s": bs / med, "decode_ms_per_step": med * 1e3,
                   "weights_GB": model.weight_bytes() / 1e9, "weight_stream_GBs_at_decode": model.weight_bytes() / med / 1e9,
                   "launch": "eager" if args.no_graph else "hipgraph", "glue": "hip kernels" if fused else "torch ops",
                   "data": "synthetic random weights",
                   "vram_GB": torch.cuda.max_memory_allocated(dev) / 1e9}
            print(json.dumps(out), flush=True)
            del model
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
