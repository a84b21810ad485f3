"""Synthetic Llama-like decoder stack that calls the W4A16 operator at the real layer shapes.

This is the caller side of the hot path: the counterpart, for measurement only, of the
reference's fused runtime (quick/awq/modules/fused/{model,block,attn}.py) and of examples/benchmark.py.  There are no
checkpoints or network here, so weights are random packed tensors of the right shapes; attention, RMSNorm, RoPE and the
KV cache are plain torch-ROCm ops (the reference uses out-of-tree awq_ext / awq_ft_ext kernels for those); every
quantised projection goes through ``WQLinear_QUICK`` -> libquick_amd.so.

Differences from the reference's layer wiring, both on the GEMM side of the boundary:
  * q/k/v are one fused GEMM also for GQA models (the reference's QUICK_cat rejects unequal widths);
  * gate_proj and up_proj are one GEMM of width 2*intermediate (the intent of the reference's unused QuantFusedMLP,
    quick/awq/modules/fused/mlp.py:52-71) whose output channels interleave gate and up in blocks of 8, so that the
    SiLU*mul can run in the GEMM epilogue.
"""
from dataclasses import dataclass

import os

import torch
import torch.nn.functional as F

from . import kernels
from . import packing
from .linear import WQLinear_QUICK


@dataclass
class DecoderConfig:
    name: str
    hidden: int
    layers: int
    heads: int
    kv_heads: int
    intermediate: int
    vocab: int = 32000
    group_size: int = 128
    rope_theta: float = 10000.0

    @property
    def head_dim(self):
        return self.hidden // self.heads


CONFIGS = {
    "llama2-7b": DecoderConfig("Llama-2-7B", 4096, 32, 32, 32, 11008),
    "mistral-7b": DecoderConfig("Mistral-7B", 4096, 32, 32, 8, 14336),
    "llama2-70b": DecoderConfig("Llama-2-70B", 8192, 80, 64, 8, 28672),
    "tiny": DecoderConfig("tiny-test", 512, 2, 4, 2, 1024, vocab=512),
}


def random_wqlinear(K, N, G, device, gen):
    """A WQLinear_QUICK whose buffers are random bits already in MI355X order (timing only)."""
    m = WQLinear_QUICK(4, G, K, N, False, device)
    qw, sc, qz = packing.random_mi355x(K, N, G, device, gen, zero_point=8, scale_lo=0.001, scale_span=0.004)  # weights centred on 0
    m._set_packed(qw, sc, qz, prepared=True)
    return m


def _rms_norm(x, w, eps=1e-5):
    v = x.float()
    return (v * torch.rsqrt(v.pow(2).mean(-1, keepdim=True) + eps)).to(x.dtype) * w


def _rope(x, cos, sin):
    # x [B, H, T, D]; cos/sin [T, D] (HF rotate-half convention)
    d = x.shape[-1] // 2
    rot = torch.cat((-x[..., d:], x[..., :d]), dim=-1)
    return x * cos + rot * sin


class SyntheticDecoder:
    def __init__(self, cfg: DecoderConfig, batch, max_len, device, seed=0):
        self.cfg, self.B, self.max_len, self.dev = cfg, batch, max_len, device
        g = torch.Generator(device=device).manual_seed(seed)
        H, KV, D, I, G = cfg.hidden, cfg.kv_heads * cfg.head_dim, cfg.head_dim, cfg.intermediate, cfg.group_size
        self.layers = []
        for _ in range(cfg.layers):
            self.layers.append(dict(
                qkv=random_wqlinear(H, H + 2 * KV, G, device, g), o=random_wqlinear(H, H, G, device, g),
                gate_up=random_wqlinear(H, 2 * I, G, device, g), down=random_wqlinear(I, H, G, device, g),
                ln1=torch.ones(H, dtype=torch.float16, device=device), ln2=torch.ones(H, dtype=torch.float16, device=device),
                k=torch.zeros(batch, cfg.kv_heads, max_len, D, dtype=torch.float16, device=device),
                v=torch.zeros(batch, cfg.kv_heads, max_len, D, dtype=torch.float16, device=device)))
        self.embed = (torch.randn(cfg.vocab, H, device=device, generator=g) * 0.02).half()
        self.lm_head = (torch.randn(cfg.vocab, H, device=device, generator=g) * 0.02).half()
        self.norm = torch.ones(H, dtype=torch.float16, device=device)
        inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, D, 2, device=device).float() / D))
        ang = torch.outer(torch.arange(max_len, device=device).float(), inv)
        self.cos = torch.cat((ang.cos(), ang.cos()), -1).half()
        self.sin = torch.cat((ang.sin(), ang.sin()), -1).half()

        # static activations of the fused decode step (one new token per sequence)
        nh, nkv = cfg.heads, cfg.kv_heads
        f16 = dict(dtype=torch.float16, device=device)
        self._h = torch.empty(batch, H, **f16)
        self._qkv = torch.empty(batch, H + 2 * KV, **f16)
        self._q = torch.empty(batch, nh, D, **f16)
        self._att = torch.empty(batch, H, **f16)
        self._gu = torch.empty(batch, 2 * I, **f16)
        self._act = torch.empty(batch, I, **f16)

    def weight_bytes(self):
        return sum(l[k].qweight.numel() * 4 for l in self.layers for k in ("qkv", "o", "gate_up", "down"))

    @torch.no_grad()
    def forward(self, tokens, pos, mask, hip_glue=True):
        """tokens [B, T] int64; pos int64 [T] (positions being written); mask additive [1, 1, T, max_len] or None for
        causal prefill from position 0.  Returns the next-token ids [B] (argmax of the last position's logits).
        ``hip_glue``: RMSNorm through quick_rmsnorm_f16 and the residual adds / SiLU*mul in the GEMM epilogues (what a
        prefill wants); False = every op around the GEMMs in eager torch (the arithmetic the tests compare against)."""
        cfg, B, T = self.cfg, tokens.shape[0], tokens.shape[1]
        H, nh, nkv, D = cfg.hidden, cfg.heads, cfg.kv_heads, cfg.head_dim
        x = self.embed[tokens]                                        # [B, T, H]
        cos, sin = self.cos.index_select(0, pos), self.sin.index_select(0, pos)
        norm = (lambda v, w: kernels.rmsnorm(v.reshape(-1, H), w).view(v.shape)) if hip_glue else _rms_norm
        p0 = int(pos[0]) if (hip_glue and mask is None) else 0   # one host read per prefill, not per layer
        for l in self.layers:
            h = norm(x, l["ln1"])
            qkv = l["qkv"](h)                                          # W4A16 GEMM, M = B*T
            if hip_glue and mask is None and B * T <= 65535:           # prefill: RoPE + cache write in one launch
                q = torch.empty((B, nh, T, D), dtype=torch.float16, device=x.device)
                kernels.rope_kv_write(qkv.reshape(B * T, -1), self.cos, self.sin, pos[:1], q, l["k"], l["v"], T, nh, nkv, D)
                att = F.scaled_dot_product_attention(q, l["k"][:, :, p0:p0 + T], l["v"][:, :, p0:p0 + T], is_causal=True,
                                                     enable_gqa=nkv != nh)
                att = att.transpose(1, 2).reshape(B * T, H)
                o, gu, dn = l["o"], l["gate_up"], l["down"]
                x2 = x.reshape(B * T, H).contiguous()
                kernels.gemm_forward(att, o.qweight, o.scales, o.qzeros, residual=x2, out=x2)
                act = kernels.gemm_forward(kernels.rmsnorm(x2, l["ln2"]), gu.qweight, gu.scales, gu.qzeros, silu_mul=True)
                kernels.gemm_forward(act, dn.qweight, dn.scales, dn.qzeros, residual=x2, out=x2)
                x = x2.view(B, T, H)
                continue
            q, k, v = qkv.split((H, nkv * D, nkv * D), dim=-1)
            q = _rope(q.view(B, T, nh, D).transpose(1, 2), cos, sin)
            k = _rope(k.view(B, T, nkv, D).transpose(1, 2), cos, sin)
            v = v.view(B, T, nkv, D).transpose(1, 2)
            l["k"].index_copy_(2, pos, k)
            l["v"].index_copy_(2, pos, v)
            if mask is None:                                           # prefill from an empty cache: causal over T
                att = F.scaled_dot_product_attention(q, k, v, is_causal=True, enable_gqa=nkv != nh)
            else:
                att = F.scaled_dot_product_attention(q, l["k"], l["v"], attn_mask=mask, enable_gqa=nkv != nh)
            att = att.transpose(1, 2).reshape(B * T, H)
            if hip_glue:
                o, gu, dn = l["o"], l["gate_up"], l["down"]
                x2 = x.reshape(B * T, H).contiguous()
                kernels.gemm_forward(att, o.qweight, o.scales, o.qzeros, residual=x2, out=x2)        # x += o(att)
                act = kernels.gemm_forward(kernels.rmsnorm(x2, l["ln2"]), gu.qweight, gu.scales, gu.qzeros, silu_mul=True)
                kernels.gemm_forward(act, dn.qweight, dn.scales, dn.qzeros, residual=x2, out=x2)     # x += down(silu(gate) * up)
                x = x2.view(B, T, H)
                continue
            x = x + l["o"](att.view(B, T, H))                          # W4A16 GEMM
            h = norm(x, l["ln2"])
            gu = l["gate_up"](h).view(B, T, cfg.intermediate // 8, 2, 8)   # W4A16 GEMM, N = 2*intermediate,
            gate = gu[..., 0, :].reshape(B, T, cfg.intermediate)           # gate/up channels interleaved in blocks of 8
            up = gu[..., 1, :].reshape(B, T, cfg.intermediate)
            x = x + l["down"](F.silu(gate) * up)                       # W4A16 GEMM
        logits = norm(x[:, -1], self.norm) @ self.lm_head.t()
        return logits.argmax(-1)


def _gemm(m: WQLinear_QUICK, x, out, residual=None):
    return kernels.gemm_forward(x, m.qweight, m.scales, m.qzeros, residual=residual, out=out)


@torch.no_grad()
def decode_step_fused(model: SyntheticDecoder, tok, pos):
    """One decode step (T = 1) with the glue around the GEMMs fused.  Per layer
         qkv GEMM (RMSNorm prologue) | RoPE + KV append + single-query attention | o GEMM (+ residual) |
         gate_up GEMM (RMSNorm prologue, SiLU*mul epilogue) | down GEMM (+ residual)
    -- 5 launches, 7 where the planner's kernel for the shape cannot take the RMSNorm prologue.  (Running the four GEMMs
    between two attention kernels as ONE persistent launch was built and measured in r02 and is slower than the four
    launches: profiles/r02_chain_experiment.txt.)  Same arithmetic as ``SyntheticDecoder.forward`` up to fp16 rounding
    order.  Returns (next tokens [B], hidden [B, H])."""
    cfg = model.cfg
    nh, nkv, D, H, I, G = cfg.heads, cfg.kv_heads, cfg.head_dim, cfg.hidden, cfg.intermediate, cfg.group_size
    B = tok.shape[0]
    x = model.embed.index_select(0, tok.view(-1))                  # [B, H], a fresh buffer: the residual stream
    fuse_qkv = kernels.can_fuse_rmsnorm(B, H, H + 2 * nkv * D, G)
    fuse_gu = kernels.can_fuse_rmsnorm(B, H, 2 * I, G)
    for l in model.layers:
        qkv, gu = l["qkv"], l["gate_up"]
        if fuse_qkv:
            kernels.gemm_forward(x, qkv.qweight, qkv.scales, qkv.qzeros, out=model._qkv, rmsnorm_weight=l["ln1"])
        else:
            kernels.rmsnorm(x, l["ln1"], out=model._h)
            _gemm(qkv, model._h, model._qkv)
        kernels.rope_attention(model._qkv, model.cos, model.sin, pos, l["k"], l["v"], model._att, nh, nkv, D)
        _gemm(l["o"], model._att, x, residual=x)                    # x += o_proj(att), added in the GEMM epilogue
        if fuse_gu:
            kernels.gemm_forward(x, gu.qweight, gu.scales, gu.qzeros, out=model._act, rmsnorm_weight=l["ln2"], silu_mul=True)
        else:
            kernels.rmsnorm(x, l["ln2"], out=model._h)
            kernels.gemm_forward(model._h, gu.qweight, gu.scales, gu.qzeros, out=model._act, silu_mul=True)
        _gemm(l["down"], model._act, x, residual=x)
    if B <= 4 and H % 512 == 0 and B * H <= 36864 and os.environ.get("QUICK_AMD_TORCH_LM_HEAD") != "1":  # (the variable: A/B runs)   # final RMSNorm + fp16 lm_head + greedy arg-max: one weight stream, two launches
        tok, hidden, _ = kernels.lm_head_argmax(x, model.lm_head, model.norm, want_hidden=True)
        return tok, hidden
    hidden = kernels.rmsnorm(x, model.norm)
    return (hidden @ model.lm_head.t()).argmax(-1), hidden


@torch.no_grad()
def run_generation(model: SyntheticDecoder, ctx, n_generate, use_graph=True, fused=True):
    """examples/benchmark.py:38-67 methodology: events around every forward; prefill = iteration 0, decode = the rest.
    Returns (prefill_seconds, [decode step seconds])."""
    B, dev, L = model.B, model.dev, model.max_len
    assert ctx + n_generate <= L
    tokens = torch.randint(0, model.cfg.vocab, (B, ctx), device=dev)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    s, e = ev(), ev()
    s.record()
    nxt = model.forward(tokens, torch.arange(ctx, device=dev), None)
    e.record()
    torch.cuda.synchronize()
    prefill = s.elapsed_time(e) * 1e-3

    # decode: static tensors so that one step can be captured in a hipGraph and replayed
    tok = nxt.view(B, 1).clone()
    pos = torch.full((1,), ctx, dtype=torch.int64, device=dev)
    mask = torch.full((1, 1, 1, L), float("-inf"), dtype=torch.float16, device=dev)
    mask[..., :ctx] = 0

    def step():
        if fused:
            out, _ = decode_step_fused(model, tok, pos)
        else:
            mask.index_fill_(3, pos, 0.0)
            out = model.forward(tok, pos, mask)
        tok.copy_(out.view(B, 1))
        pos.add_(1)

    graph = None
    if use_graph:
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):      # warm-up on a side stream, as torch requires before capture
            step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
        n_generate -= 1                     # one step was executed by the warm-up (capture executes nothing)
    times = []
    for _ in range(n_generate):
        s.record()
        graph.replay() if graph is not None else step()
        e.record()
        torch.cuda.synchronize()
        times.append(s.elapsed_time(e) * 1e-3)
    return prefill, times
