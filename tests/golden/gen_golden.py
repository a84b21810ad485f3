#!/usr/bin/env python3
"""Generate golden fixtures for the W4A16 path FROM THE REFERENCE'S OWN PYTHON.

Runs only in the build container, where /root/reference exists (it does not exist on the GPU
box).  Nothing from the reference is copied: its modules are imported by file path with the
harness below and only *data* -- inputs and the outputs the reference
computed -- is written to tests/golden/*.npz.

    python tests/golden/gen_golden.py            # regenerates every fixture

What is pinned:
  pack     WQLinear_QUICK.from_linear (quick/awq/modules/linear/quick.py:60-156), executed on CPU by
           replacing the three hard-coded 'cuda' literals (lines 95, 101, 147) in memory;
  dequant  dequantize_gemm (quick/awq/utils/packing_utils.py:82-97) on the GEMM-format pack of the
           same layer (WQLinear_GEMM.from_linear, quick/awq/modules/linear/gemm.py:64-150);
  gemm     WQLinear_GEMM.forward without awq_ext = dequantize_gemm + torch.matmul (gemm.py:173-181);
  cat      QUICK_cat (quick/awq/utils/fused_utils.py:119-159);
  quant    AwqQuantizer.pseudo_quantize_tensor (quick/awq/quantize/quantizer.py:46-72) feeding
           from_linear, for the non-exact rounding case;
  pin      the BASELINE size K = N = 4096, g = 128: seed + SHA-256 of the reference packer's three buffers + sampled
           dequantised weights and sampled outputs of the reference CPU path (`python gen_golden.py pin` makes only this).
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _load_reference():
    stub = types.ModuleType("quick_kernels")
    stub.gemm_forward_cuda_quick = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("stub"))
    sys.modules["quick_kernels"] = stub
    for name in ["quick", "quick.awq", "quick.awq.utils", "quick.awq.modules", "quick.awq.modules.linear"]:
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m

    def load(dotted, rel):
        spec = importlib.util.spec_from_file_location(dotted, os.path.join(REF, rel))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[dotted] = mod
        spec.loader.exec_module(mod)
        return mod

    load("quick.awq.utils.utils", "quick/awq/utils/utils.py")
    packing = load("quick.awq.utils.packing_utils", "quick/awq/utils/packing_utils.py")
    gemm = load("quick.awq.modules.linear.gemm", "quick/awq/modules/linear/gemm.py")

    # WQLinear_QUICK: same source, the three device literals pointed at the CPU, compiled in memory.
    src = open(os.path.join(REF, "quick/awq/modules/linear/quick.py")).read()
    assert src.count("'cuda'") == 3
    qmod = types.ModuleType("ref_quick_cpu")
    exec(compile(src.replace("'cuda'", "'cpu'"), "ref_quick_cpu", "exec"), qmod.__dict__)

    # QUICK_cat: only that function's text (the module imports the whole package).
    fsrc = open(os.path.join(REF, "quick/awq/utils/fused_utils.py")).read()
    a, b = fsrc.index("def QUICK_cat"), fsrc.index("def get_attention_shapes")
    cmod = types.ModuleType("ref_quick_cat")
    exec(compile("import torch\nfrom typing import Optional, Tuple\n" + fsrc[a:b], "ref_quick_cat", "exec"), cmod.__dict__)

    # pseudo_quantize_tensor: the method's text only (the module imports datasets/transformers glue).
    zsrc = open(os.path.join(REF, "quick/awq/quantize/quantizer.py")).read()
    a, b = zsrc.index("    def pseudo_quantize_tensor"), zsrc.index("    def pseudo_dequantize_tensor")
    import textwrap
    pmod = types.ModuleType("ref_pseudo_quant")
    exec(compile("import torch\n" + textwrap.dedent(zsrc[a:b]), "ref_pseudo_quant", "exec"), pmod.__dict__)
    return qmod.WQLinear_QUICK, gemm.WQLinear_GEMM, packing.dequantize_gemm, cmod.QUICK_cat, pmod.pseudo_quantize_tensor


def _linear(w):
    lin = torch.nn.Linear(w.shape[1], w.shape[0], bias=False)
    lin.weight.data = w.clone()
    return lin.half()


def _exact_layer(K, N, G, seed):
    """Layer whose quantisation is exact: W = fp16((iw - z) * s), so packing is the only thing under test."""
    g = torch.Generator().manual_seed(seed)
    iw = torch.randint(0, 16, (N, K), generator=g)
    z = torch.randint(0, 16, (N, K // G), generator=g)
    s = (torch.rand(N, K // G, generator=g) * 0.02 + 0.005).half()
    w = ((iw - z.repeat_interleave(G, 1)).half() * s.repeat_interleave(G, 1)).half()
    return w, s, z.half(), iw


def make_pin(QUICK, GEMM, dequantize_gemm):
    # BASELINE size (K = N = 4096, g = 128): too big to store, so the fixture holds the seed the layer is regenerated from,
    # the SHA-256 of the three buffers the REFERENCE packer made of it, sampled dequantised weights and sampled outputs of
    # the REFERENCE CPU path ("for 4096^2 only seeds + SHA-256 of packed buffers + sampled outputs")
    import hashlib
    K, N, G, M, seed = 4096, 4096, 128, 16, 900
    w, s, z, iw = _exact_layer(K, N, G, seed=seed)
    lin = _linear(w)
    ql = QUICK.from_linear(lin, 4, G, False, s.clone(), z.clone())
    gl = GEMM.from_linear(lin, 4, G, False, s.t().contiguous(), z.t().contiguous())
    x = torch.randn(M, K, generator=torch.Generator().manual_seed(seed + 1)).half()
    with torch.no_grad():
        y = gl(x)
        wdeq = dequantize_gemm(gl.qweight, gl.qzeros, gl.scales, 4, G)
    rng = np.random.default_rng(seed + 2)
    ys_m, ys_n = rng.integers(0, M, 2048), rng.integers(0, N, 2048)
    ws_k, ws_n = rng.integers(0, K, 4096), rng.integers(0, N, 4096)
    sha = lambda t: hashlib.sha256(np.ascontiguousarray(t.numpy()).tobytes()).hexdigest()
    np.savez_compressed(
        os.path.join(OUT, "pin_k4096n4096g128.npz"), K=K, N=N, G=G, M=M, seed=seed, x=x.numpy(),
        sha_qweight=sha(ql.qweight), sha_qscales=sha(ql.scales), sha_qzeros=sha(ql.qzeros),
        y_rows=ys_m, y_cols=ys_n, y_ref=y.numpy()[ys_m, ys_n],
        w_k=ws_k, w_n=ws_n, w_ref=wdeq.numpy()[ws_k, ws_n],
        col_abs_sum=y.float().abs().sum(0).numpy(),      # one number per output channel: catches a wrong column anywhere
    )
    print("wrote 4096^2 pin:", sha(ql.qweight)[:16], sha(ql.scales)[:16], sha(ql.qzeros)[:16])



def main():
    torch.manual_seed(0)
    QUICK, GEMM, dequantize_gemm, QUICK_cat, pseudo_quantize_tensor = _load_reference()

    if len(sys.argv) > 1 and sys.argv[1] == "pin":     # only the BASELINE-size pin (the other fixtures are untouched)
        make_pin(QUICK, GEMM, dequantize_gemm)
        return
    cases = [("k64n128g64", 64, 128, 64, 3), ("k128n128g128", 128, 128, 128, 5), ("k128n256g32", 128, 256, 32, 7),
             ("k256n512g64", 256, 512, 64, 16), ("k384n256g128", 384, 256, 128, 33), ("k512n768g128", 512, 768, 128, 1)]
    for i, (name, K, N, G, M) in enumerate(cases):
        w, s, z, iw = _exact_layer(K, N, G, seed=100 + i)
        lin = _linear(w)
        ql = QUICK.from_linear(lin, 4, G, False, s.clone(), z.clone())
        gl = GEMM.from_linear(lin, 4, G, False, s.t().contiguous(), z.t().contiguous())
        x = torch.randn(M, K, generator=torch.Generator().manual_seed(200 + i)).half()
        with torch.no_grad():
            y = gl(x)
            wdeq = dequantize_gemm(gl.qweight, gl.qzeros, gl.scales, 4, G)
        np.savez_compressed(
            os.path.join(OUT, f"exact_{name}.npz"),
            K=K, N=N, G=G, M=M,
            weight=w.numpy(), scales_nk=s.numpy(), zeros_nk=z.numpy(), intweight_nk=iw.numpy().astype(np.uint8),
            x=x.numpy(),
            ref_qweight=ql.qweight.numpy(), ref_qscales=ql.scales.numpy(), ref_qzeros=ql.qzeros.numpy(),
            ref_wdeq=wdeq.numpy(), ref_y=y.numpy(),
            ref_gemm_qweight=gl.qweight.numpy(), ref_gemm_qzeros=gl.qzeros.numpy(), ref_gemm_scales=gl.scales.numpy(),
        )
        print("wrote", name, tuple(ql.qweight.shape), tuple(ql.scales.shape), tuple(ql.qzeros.shape))

    # QUICK_cat on three equal-shape layers (q/k/v of an MHA block)
    K, N, G = 128, 256, 64
    packs, logical = [], []
    for j in range(3):
        w, s, z, iw = _exact_layer(K, N, G, seed=300 + j)
        ql = QUICK.from_linear(_linear(w), 4, G, False, s.clone(), z.clone())
        packs.append(ql)
        logical.append((w, s, z))
    wcat = torch.cat([l[0] for l in logical], 0)
    scat = torch.cat([l[1] for l in logical], 0)
    zcat = torch.cat([l[2] for l in logical], 0)
    qcat = QUICK.from_linear(_linear(wcat), 4, G, False, scat.clone(), zcat.clone())
    out = {}
    for opt, attr in (("qweight", "qweight"), ("qzeros", "qzeros"), ("scales", "scales")):
        out["cat_" + opt] = QUICK_cat(*[getattr(p, attr) for p in packs], options=opt).numpy()
        out["full_" + opt] = getattr(qcat, attr).numpy()
        for j, p in enumerate(packs):
            out[f"in{j}_{opt}"] = getattr(p, attr).numpy()
    np.savez_compressed(os.path.join(OUT, "quick_cat_k128n256g64.npz"), K=K, N=N, G=G, **out)
    print("wrote quick_cat; QUICK_cat == pack(concat):",
          all(np.array_equal(out["cat_" + o], out["full_" + o]) for o in ("qweight", "qzeros", "scales")))

    # non-exact quantisation: real-valued weights through pseudo_quantize_tensor (AwqQuantizer._apply_quant)
    K, N, G = 256, 256, 128
    w = (torch.randn(N, K, generator=torch.Generator().manual_seed(400)) * 0.03).half()
    ns = types.SimpleNamespace(group_size=G, w_bit=4)
    wq, s, z = pseudo_quantize_tensor(ns, w.clone(), get_scale_zp=True)
    ql = QUICK.from_linear(_linear(wq), 4, G, False, s.clone(), z.clone())
    gl = GEMM.from_linear(_linear(wq), 4, G, False, s.t().contiguous(), z.t().contiguous())
    x = torch.randn(9, K, generator=torch.Generator().manual_seed(401)).half()
    with torch.no_grad():
        y = gl(x)
        wdeq = dequantize_gemm(gl.qweight, gl.qzeros, gl.scales, 4, G)
    np.savez_compressed(
        os.path.join(OUT, "quant_k256n256g128.npz"), K=K, N=N, G=G, M=9,
        weight=wq.numpy(), scales_nk=s.numpy(), zeros_nk=z.numpy(), x=x.numpy(),
        ref_qweight=ql.qweight.numpy(), ref_qscales=ql.scales.numpy(), ref_qzeros=ql.qzeros.numpy(),
        ref_wdeq=wdeq.numpy(), ref_y=y.numpy(),
    )
    print("wrote quant case")
    make_pin(QUICK, GEMM, dequantize_gemm)


if __name__ == "__main__":
    main()
