"""Python face of libquick_amd.so: the functions the reference's ``quick_kernels`` extension exports
(csrc/pybind.cpp:5-8), plus the format bridge.  Device pointers and the current HIP stream come from
torch; the arithmetic is entirely in the HIP library -- there is no CPU / eager fallback.
"""
import ctypes
import weakref

import torch

from . import _lib

KERNEL_AUTO, KERNEL_SKINNY, KERNEL_TILED, KERNEL_WIDE, KERNEL_XK, KERNEL_XW, KERNEL_LEAN, KERNEL_XM = 0, 1, 2, 3, 4, 5, 6, 7
_OK, _INVALID, _WORKSPACE, _LAUNCH, _UNSUPPORTED = 0, 1, 2, 3, 4


def _raise(rc):
    msg = _lib.last_error() or f"libquick_amd error {rc}"
    if rc == _INVALID:
        raise ValueError(msg)          # std::invalid_argument in the reference (gemm_cuda_quick.cu:1479-1484)
    if rc == _UNSUPPORTED:
        raise NotImplementedError(msg)
    raise RuntimeError(msg)


def _expect(t, dtype, name):
    if t.dtype != dtype:               # data_ptr<T>() throws c10::Error -> RuntimeError in the reference
        raise RuntimeError(f"expected scalar type {dtype} for {name} but found {t.dtype}")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a GPU tensor: the W4A16 GEMM has no CPU implementation")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")


def _stream():
    return torch.cuda.current_stream().cuda_stream


_WORKSPACES = {}          # (device index, stream handle) -> zero-filled uint8 tensor; insertion order = recency
_WORKSPACES_MAX = 16      # stream handles are recycled by the runtime: keep the few most recently used, drop the rest
_CAPTURED = set()         # keys whose buffer a hipGraph capture has baked into kernel arguments: never evicted
_KEEPALIVE = []           # buffers a capture used and that were outgrown since: a replayed graph still writes into them


def _workspace(device, nbytes):
    """Zero-filled scratch for the in-kernel split-K reduction / K-slice exchange, one per (device, stream), grown on demand.
    The library hands it back zeroed after every launch, so it is allocated and cleared only when it has to grow.  A buffer
    that was used while its stream was being captured is pinned: the graph holds its address (counters and mailboxes that must
    be zero on entry), so it is neither evicted by the LRU cap nor freed when a later, larger request replaces it."""
    key = (device.index, _stream())
    capturing = torch.cuda.is_current_stream_capturing()
    ws = _WORKSPACES.pop(key, None)
    if ws is None or ws.numel() < nbytes:
        if ws is not None and key in _CAPTURED:
            _KEEPALIVE.append(ws)
        ws = torch.zeros(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
    _WORKSPACES[key] = ws                                  # most recently used last
    if capturing:
        _CAPTURED.add(key)
    evictable = [k for k in _WORKSPACES if k not in _CAPTURED]
    while len(evictable) > _WORKSPACES_MAX:
        _WORKSPACES.pop(evictable.pop(0))                  # (in-flight launches keep their buffer alive through the allocator's stream ordering)
    return ws


def can_fuse_rmsnorm(M, K, N, G):
    """True when gemm_forward(..., rmsnorm_weight=...) is available for this shape (small M, x held whole in LDS)."""
    return bool(_lib.load().quick_w4a16_can_fuse_rmsnorm(M, K, N, G))


def plan_describe(M, K, N, G, kernel_id=KERNEL_AUTO, grid_split_k=0):
    """One line saying what a launch of this shape runs (quick_w4a16_plan_describe; host-only, no GPU needed)."""
    import ctypes
    buf = ctypes.create_string_buffer(256)
    rc = _lib.load().quick_w4a16_plan_describe(M, K, N, G, kernel_id, grid_split_k, buf, len(buf))
    if rc != _OK:
        _raise(rc)
    return buf.value.decode()


def _check_gemm(in_feats, kernel, scaling_factors, zeros, bias, residual, out, rmsnorm_weight, silu_mul):
    """The library takes raw pointers: everything it will dereference is checked here (dtype, device, contiguity, shape).
    Returns (M, K, N, G, out) with ``out`` allocated if it was None."""
    _expect(in_feats, torch.float16, "in_feats")
    _expect(kernel, torch.int32, "kernel")
    _expect(scaling_factors, torch.float16, "scaling_factors")
    _expect(zeros, torch.int32, "zeros")
    if in_feats.dim() != 2:
        raise RuntimeError("in_feats must be 2-D [M, K]")
    M, K = in_feats.shape
    N = kernel.shape[1] // 4 * 8                      # gemm_cuda_quick.cu:1468
    if kernel.shape[0] * 4 != K:
        raise ValueError(f"kernel has {kernel.shape[0] * 4} input channels, in_feats has {K}")
    G = K // scaling_factors.shape[0]                 # gemm_cuda_quick.cu:1477
    if tuple(scaling_factors.shape) != (K // G, 2 * N) or tuple(zeros.shape) != (K // G, N // 4):
        raise ValueError(f"scaling_factors / zeros must be [{K // G}, {2 * N}] / [{K // G}, {N // 4}] for K={K} N={N} G={G}, "
                         f"got {tuple(scaling_factors.shape)} / {tuple(zeros.shape)}")
    n_out = N // 2 if silu_mul else N
    for t, shape, name in ((out, (M, n_out), "out"), (bias, (N,), "bias"), (residual, (M, N), "residual"),
                           (rmsnorm_weight, (K,), "rmsnorm_weight")):
        if t is None:
            continue
        _expect(t, torch.float16, name)
        if tuple(t.shape) != shape or t.device != in_feats.device:
            raise ValueError(f"{name} must be a {list(shape)} fp16 tensor on {in_feats.device}, got {list(t.shape)} on {t.device}")
    for t, name in ((kernel, "kernel"), (scaling_factors, "scaling_factors"), (zeros, "zeros")):
        if t.device != in_feats.device:
            raise RuntimeError(f"{name} is on {t.device}, in_feats on {in_feats.device}")
    if out is None:
        out = torch.empty((M, n_out), dtype=torch.float16, device=in_feats.device)
    return M, K, N, G, out


def gemm_forward(in_feats, kernel, scaling_factors, zeros, bias=None, kernel_id=KERNEL_AUTO, grid_split_k=0, residual=None,
                 out=None, rmsnorm_weight=None, rmsnorm_eps=1e-5, silu_mul=False):
    """y [M, N] fp16 = in_feats [M, K] @ dequant(kernel, scaling_factors, zeros) (+ bias) (+ residual), MI355X-order
    weights.  ``out`` (optional, may be ``residual``) receives the result.  ``rmsnorm_weight`` [K] normalises in_feats on
    the way in (see can_fuse_rmsnorm); ``silu_mul`` treats the output channels as gate/up interleaved in blocks of 8 and
    returns silu(gate) * up, [M, N/2]."""
    M, K, N, G, out = _check_gemm(in_feats, kernel, scaling_factors, zeros, bias, residual, out, rmsnorm_weight, silu_mul)
    lib = _lib.load()
    if M == 0:
        return out
    with torch.cuda.device(in_feats.device):          # OptionalCUDAGuard, gemm_cuda_quick.cu:1465
        ws_bytes = lib.quick_w4a16_workspace_bytes_ex(M, K, N, G, kernel_id, grid_split_k)
        ws = _workspace(in_feats.device, ws_bytes) if ws_bytes else None
        fusion = _lib.GemmFusion(bias.data_ptr() if bias is not None else None,
                                 residual.data_ptr() if residual is not None else None,
                                 rmsnorm_weight.data_ptr() if rmsnorm_weight is not None else None, rmsnorm_eps, int(silu_mul))
        rc = lib.quick_w4a16_gemm_f16_fused(
            in_feats.data_ptr(), kernel.data_ptr(), scaling_factors.data_ptr(), zeros.data_ptr(), ctypes.byref(fusion),
            out.data_ptr(),
            ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, M, K, N, G, kernel_id,
            grid_split_k, _stream())
    if rc != _OK:
        if ws is not None:                                # whatever went wrong, the next launch finds the state it expects: all-zero
            ws.zero_()
        _raise(rc)
    return out


def workspace_check(device=None):
    """Verify that this stream's workspace is all-zero where the library expects it (quick_w4a16_workspace_check: synchronises).
    Raises RuntimeError naming the first dirty byte -- and zeroes the buffer again -- otherwise."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    with torch.cuda.device(device):      # (the stream is THAT device's current stream: key and call both)
        stream = torch.cuda.current_stream(device).cuda_stream
        ws = _WORKSPACES.get((device.index, stream))
        if ws is None:
            return True
        rc = _lib.load().quick_w4a16_workspace_check(ws.data_ptr(), ws.numel(), stream)
    if rc != _OK:
        ws.zero_()
        _raise(rc)
    return True


def _tensor_version(t):
    return 0 if t.is_inference() else t._version     # inference tensors have no version counter (and cannot be rewritten in place outside inference mode)


class _Repacked:
    __slots__ = ("refs", "versions", "packed", "stream", "event", "done", "seen")


_REPACK_CACHE = {}     # (data_ptr of qweight, scales, qzeros) -> _Repacked; entries die with the tensors they mirror


def reference_to_mi355x_cached(kernel, scaling_factors, zeros):
    """MI355X-order copies of reference-order packed tensors, made once per tensor triple by the HIP repack kernel.

    This is what lets the reference's own ``WQLinear_QUICK.forward`` (quick/awq/modules/linear/quick.py:158-166), whose
    buffers hold the checkpoint order of ``from_linear`` (quick.py:88-150), call this library unchanged.  The entry is
    keyed on the identity (weak references) and version counters of the three tensors: an in-place rewrite
    (``load_state_dict``, ``copy_``) or a new tensor object invalidates it, and it is dropped when a tensor dies.  Cost: a
    second copy of the layer's packed weights in HBM -- ``quick_amd.WQLinear_QUICK`` avoids it by permuting its own
    buffers in place and calling :func:`gemm_forward` directly.
    """
    tensors = (kernel, scaling_factors, zeros)
    key = tuple(t.data_ptr() for t in tensors)
    versions = tuple(_tensor_version(t) for t in tensors)
    ent = _REPACK_CACHE.get(key)
    if ent is not None and ent.versions == versions and all(r() is t for r, t in zip(ent.refs, tensors)):
        if kernel.is_cuda:
            cur = torch.cuda.current_stream(kernel.device)
            sid = (cur.stream_id, cur.cuda_stream)     # (torch's id AND the raw handle: the runtime recycles handles of destroyed streams)
            if cur != ent.stream:
                # Made on another stream.  Outside a capture: wait for the repack ONCE on the host -- from then on the copy is visible to
                # every stream and later hits cost two comparisons.  Inside a capture nothing may block: the capturing stream
                # waits for the event, on every hit, until a call outside one has synchronised.  Either way the allocator learns once per
                # stream who else reads the copy (record_stream is capture-safe: a graph captured on a side stream is covered -- ADVICE r04).
                if not ent.done:
                    if torch.cuda.is_current_stream_capturing():
                        cur.wait_event(ent.event)
                    else:
                        ent.event.synchronize()
                        ent.done = True
                if sid not in ent.seen:
                    for t in ent.packed:
                        t.record_stream(cur)
                    ent.seen.add(sid)
        return ent.packed
    ent = _Repacked()
    ent.stream = ent.event = None
    ent.done, ent.seen = False, set()
    if kernel.shape[0] * 4 % 128 == 0:
        ent.packed = repack_cuda_to_mi355x(kernel, scaling_factors, zeros)
    else:
        ent.packed = _padded_mi355x(kernel, scaling_factors, zeros)
    ent.versions = versions
    if kernel.is_cuda:
        ent.stream = torch.cuda.current_stream(kernel.device)
        ent.event = torch.cuda.Event()
        ent.event.record(ent.stream)
    drop = lambda _ref, key=key: _REPACK_CACHE.pop(key, None)
    ent.refs = tuple(weakref.ref(t, drop) for t in tensors)
    _REPACK_CACHE[key] = ent
    return ent.packed


def padded_in_features(K, G):
    """The reference accepts in_features % 32 == 0 (csrc/gemm_cuda_quick.cu:1479-1484); the MI355X weight order is made of
    128-k tiles.  A layer in between (only possible with group sizes that are not multiples of 128) runs on a copy padded
    along K to the next multiple of lcm(128, G): weights 0, zero points 0, scales 0 in the added groups -- they contribute
    exactly 0 -- and the activations padded with zeros per call."""
    import math
    unit = 128 * G // math.gcd(128, G)
    return (K + unit - 1) // unit * unit


def _padded_mi355x(kernel, scaling_factors, zeros):
    """MI355X-order tensors of a reference-order layer whose K is not a multiple of 128, once per layer: the HIP repack
    kernel's padded flavour on the GPU, torch ops on CPU tensors (tests)."""
    from . import packing
    if kernel.is_cuda:
        K, N = kernel.shape[0] * 4, kernel.shape[1] * 2
        G = K // scaling_factors.shape[0]
        lib = _lib.load()
        Kp = lib.quick_padded_in_features(K, G)
        if Kp == 0:
            raise ValueError(f"quick_repack_cuda_to_mi355x_padded: invalid shape K={K} N={N} G={G}")
        outs = [torch.empty((Kp // 4, N // 2), dtype=torch.int32, device=kernel.device),
                torch.empty((Kp // G, 2 * N), dtype=torch.float16, device=kernel.device),
                torch.empty((Kp // G, N // 4), dtype=torch.int32, device=kernel.device)]
        with torch.cuda.device(kernel.device):
            rc = lib.quick_repack_cuda_to_mi355x_padded(kernel.data_ptr(), scaling_factors.data_ptr(), zeros.data_ptr(),
                                                        outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), K, N, G, _stream())
        if rc != _OK:
            _raise(rc)
        return tuple(outs)
    iw, s, z = packing.unpack_cuda_order(kernel, scaling_factors, zeros)
    K, G = iw.shape[0], iw.shape[0] // s.shape[0]
    Kp = padded_in_features(K, G)
    pad = lambda t, rows: torch.cat([t, torch.zeros((rows - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)])
    return packing.pack_mi355x(pad(iw, Kp), pad(s, Kp // G), pad(z, Kp // G))


def gemm_forward_cuda_quick(in_feats, kernel, scaling_factors, zeros, split_k_iters):
    """Drop-in for ``quick_kernels.gemm_forward_cuda_quick`` (csrc/gemm_cuda_quick.h:3-8, csrc/pybind.cpp:5-8).

    Same arguments, dtypes, error behaviour and -- like the reference -- the REFERENCE's packed order: ``kernel`` /
    ``scaling_factors`` / ``zeros`` are what ``WQLinear_QUICK.from_linear`` of the reference writes and what its
    checkpoints hold (quick.py:88-150), so the reference's unchanged module can call it.  The MI355X-order copy the HIP
    kernels consume is made on first use and cached (:func:`reference_to_mi355x_cached`).  Return shape follows the
    reference: ``[M, N]`` when ``split_k_iters > 1`` (its ``.sum(0)``), ``[1, M, N]`` otherwise
    (gemm_cuda_quick.cu:1515-1516).  ``split_k_iters`` is a tuning hint for NVIDIA parts; this library chooses its own K
    partitioning and always returns the fully reduced result.
    """
    if split_k_iters < 1:
        raise ValueError("split_k_iters must be >= 1")
    _expect(in_feats, torch.float16, "in_feats")
    _expect(kernel, torch.int32, "kernel")
    _expect(scaling_factors, torch.float16, "scaling_factors")
    _expect(zeros, torch.int32, "zeros")
    if in_feats.dim() != 2:
        raise RuntimeError("in_feats must be 2-D [M, K]")
    K, N = kernel.shape[0] * 4, kernel.shape[1] // 4 * 8       # gemm_cuda_quick.cu:1468
    if in_feats.shape[1] != K:
        raise ValueError(f"kernel has {K} input channels, in_feats has {in_feats.shape[1]}")
    G = K // scaling_factors.shape[0]
    Kp = K if K % 128 == 0 or K % 32 != 0 or G % 32 != 0 or K % G != 0 else padded_in_features(K, G)
    plan_describe(max(int(in_feats.shape[0]), 1), Kp, N, G)   # the reference's shape errors, before any repack
    if Kp != K:                                              # in_features % 128 != 0: the padded copy, zero-padded activations
        in_feats = torch.nn.functional.pad(in_feats, (0, Kp - K))
    out = gemm_forward(in_feats, *reference_to_mi355x_cached(kernel, scaling_factors, zeros))
    return out if split_k_iters > 1 else out.unsqueeze(0)


def _repack(fn_name, qweight, scales, qzeros):
    for t, d, nm in ((qweight, torch.int32, "qweight"), (scales, torch.float16, "scales"), (qzeros, torch.int32, "qzeros")):
        _expect(t, d, nm)
    lib = _lib.load()
    K, N = qweight.shape[0] * 4, qweight.shape[1] * 2
    G = K // scales.shape[0]
    outs = [torch.empty_like(qweight), torch.empty_like(scales), torch.empty_like(qzeros)]
    with torch.cuda.device(qweight.device):
        rc = getattr(lib, fn_name)(qweight.data_ptr(), scales.data_ptr(), qzeros.data_ptr(),
                                   outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), K, N, G, _stream())
    if rc != _OK:
        if rc == _INVALID:
            raise ValueError(f"{fn_name}: invalid shape K={K} N={N} G={G}")
        if rc == _UNSUPPORTED:
            raise NotImplementedError(f"{fn_name}: in_features ({K}) must be a multiple of 128 on MI355X")
        raise RuntimeError(f"{fn_name} failed ({rc})")
    return tuple(outs)


def repack_cuda_to_mi355x(qweight, scales, qzeros):
    """Reference-order packed tensors (a QUICK checkpoint) -> MI355X order, on the GPU."""
    return _repack("quick_repack_cuda_to_mi355x", qweight, scales, qzeros)


def repack_mi355x_to_cuda(qweight, scales, qzeros):
    return _repack("quick_repack_mi355x_to_cuda", qweight, scales, qzeros)


def dequantize_mi355x(qweight, scales, qzeros):
    """Dense fp16 [K, N] = fp16((w - z) * s) from MI355X-order tensors (parity aid)."""
    lib = _lib.load()
    K, N = qweight.shape[0] * 4, qweight.shape[1] * 2
    G = K // scales.shape[0]
    out = torch.empty((K, N), dtype=torch.float16, device=qweight.device)
    with torch.cuda.device(qweight.device):
        rc = lib.quick_dequantize_mi355x_f16(qweight.data_ptr(), scales.data_ptr(), qzeros.data_ptr(), out.data_ptr(),
                                             K, N, G, _stream())
    if rc != _OK:
        _raise(rc)
    return out


# ---------------------------------------------------------------------------------------------- decode-step glue
def rmsnorm(x, weight, eps=1e-5, out=None):
    """RMSNorm over the last dimension (fp16, H % 8 == 0)."""
    lib = _lib.load()
    H = x.shape[-1]
    out = torch.empty_like(x) if out is None else out
    rc = lib.quick_rmsnorm_f16(x.data_ptr(), weight.data_ptr(), out.data_ptr(), x.numel() // H, H, eps, _stream())
    if rc != _OK:
        raise RuntimeError(f"quick_rmsnorm_f16 failed ({rc})")
    return out


def rope_kv_append(qkv, cos_table, sin_table, pos, q_out, k_cache, v_cache, n_heads, n_kv_heads, head_dim):
    """Decode step: rotate q/k of the fused qkv GEMM output [B, (nh + 2 nkv) D], append k/v to the caches at *pos."""
    lib = _lib.load()
    rc = lib.quick_rope_kv_append_f16(qkv.data_ptr(), cos_table.data_ptr(), sin_table.data_ptr(), pos.data_ptr(), q_out.data_ptr(),
                                      k_cache.data_ptr(), v_cache.data_ptr(), qkv.shape[0], n_heads, n_kv_heads, head_dim,
                                      k_cache.shape[2], _stream())
    if rc != _OK:
        raise RuntimeError(f"quick_rope_kv_append_f16 failed ({rc})")
    return q_out


def rope_kv_write(qkv, cos_table, sin_table, pos0, q_out, k_cache, v_cache, tokens, n_heads, n_kv_heads, head_dim):
    """Prefill: qkv [B * T, (nh + 2 nkv) D] -> rotated q in q_out [B, nh, T, D], rotated k and v into the caches at
    positions *pos0 .. *pos0 + T - 1."""
    lib = _lib.load()
    rc = lib.quick_rope_kv_write_f16(qkv.data_ptr(), cos_table.data_ptr(), sin_table.data_ptr(), pos0.data_ptr(), q_out.data_ptr(),
                                     k_cache.data_ptr(), v_cache.data_ptr(), qkv.shape[0] // tokens, tokens, n_heads, n_kv_heads,
                                     head_dim, k_cache.shape[2], _stream())
    if rc != _OK:
        raise RuntimeError(f"quick_rope_kv_write_f16 failed ({rc})")
    return q_out


def decode_attention(q, k_cache, v_cache, pos, out, n_heads, n_kv_heads, head_dim):
    """Single-query attention over cache positions 0..*pos; q [B, nh, D] -> out [B, nh * D]."""
    lib = _lib.load()
    rc = lib.quick_decode_attention_f16(q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), pos.data_ptr(), out.data_ptr(),
                                        q.shape[0], n_heads, n_kv_heads, head_dim, k_cache.shape[2], head_dim ** -0.5, _stream())
    if rc != _OK:
        raise RuntimeError(f"quick_decode_attention_f16 failed ({rc})")
    return out


def rope_attention(qkv, cos_table, sin_table, pos, k_cache, v_cache, out, n_heads, n_kv_heads, head_dim):
    """rope_kv_append + decode_attention in one launch: qkv [B, (nh + 2 nkv) D] -> out [B, nh * D]."""
    lib = _lib.load()
    rc = lib.quick_decode_rope_attention_f16(qkv.data_ptr(), cos_table.data_ptr(), sin_table.data_ptr(), pos.data_ptr(),
                                             k_cache.data_ptr(), v_cache.data_ptr(), out.data_ptr(), qkv.shape[0], n_heads,
                                             n_kv_heads, head_dim, k_cache.shape[2], head_dim ** -0.5, _stream())
    if rc != _OK:
        raise RuntimeError(f"quick_decode_rope_attention_f16 failed ({rc})")
    return out


_LM_WS = {}


def lm_head_argmax(x, weight, norm_weight=None, eps=1e-5, want_hidden=False, want_logits=False):
    """Greedy next token of a decode step: argmax(RMSNorm(x) @ weight.T) in two launches (quick_lm_head_argmax_f16); x [B, H] fp16,
    weight [V, H] fp16, B <= 4.  Returns (next_token [B] int64, hidden [B, H] or None, logits [B, V] or None).  Raises
    NotImplementedError where the kernel has no build (the caller then runs torch)."""
    B, H = x.shape
    V = weight.shape[0]
    lib = _lib.load()
    key = (x.device.index, _stream(), B)   # per stream: two streams must not share the partial-key buffer
    if key not in _LM_WS:
        _LM_WS[key] = torch.empty(lib.quick_lm_head_workspace_bytes(B), dtype=torch.uint8, device=x.device)
    ws = _LM_WS[key]
    tok = torch.empty(B, dtype=torch.int64, device=x.device)
    hidden = torch.empty_like(x) if want_hidden else None
    logits = torch.empty((B, V), dtype=torch.float16, device=x.device) if want_logits else None
    rc = lib.quick_lm_head_argmax_f16(x.data_ptr(), norm_weight.data_ptr() if norm_weight is not None else None, eps, weight.data_ptr(),
                                      hidden.data_ptr() if hidden is not None else None,
                                      logits.data_ptr() if logits is not None else None, tok.data_ptr(), ws.data_ptr(), ws.numel(),
                                      B, V, H, _stream())
    if rc == 4:
        raise NotImplementedError("quick_lm_head_argmax_f16: batch <= 4, hidden % 512 == 0")
    if rc != _OK:
        raise RuntimeError(f"quick_lm_head_argmax_f16 failed ({rc})")
    return tok, hidden, logits


def prefetch(t, workgroups=0, stream=None):
    """Measurement aid: pull tensor `t` through HBM into the memory-side cache on `stream` (default: the current one); quick_prefetch."""
    st = _stream() if stream is None else stream.cuda_stream
    rc = _lib.load().quick_prefetch(t.data_ptr(), t.numel() * t.element_size(), workgroups, st)
    if rc != _OK:
        raise RuntimeError(f"quick_prefetch failed ({rc})")


def silu_mul(gate_up, out=None):
    """silu(gate) * up for the fused gate_up GEMM output [M, 2 I] (gate/up interleaved in blocks of 8) -> [M, I]."""
    lib = _lib.load()
    M, I = gate_up.shape[0], gate_up.shape[1] // 2
    out = torch.empty((M, I), dtype=torch.float16, device=gate_up.device) if out is None else out
    rc = lib.quick_silu_mul_f16(gate_up.data_ptr(), out.data_ptr(), M, I, _stream())
    if rc != _OK:
        raise RuntimeError(f"quick_silu_mul_f16 failed ({rc})")
    return out
