"""CPU restatement (numpy) of the QUICK W4A16 GEMM path.  TEST INFRASTRUCTURE ONLY.

Every function cites the reference file:line (relative to /root/reference) whose
behaviour it restates.  Logical tensors used throughout:

    iw  uint8  [K, N]    4-bit integer weights (0..15), K = in_features, N = out_features
    s   fp16   [K/G, N]  per-group scales
    z   uint8  [K/G, N]  per-group integer zero points (0..15)

Two packed formats are described here:

* "cuda order"   -- exactly what the reference's ``WQLinear_QUICK.from_linear`` emits
                    (quick/awq/modules/linear/quick.py:88-150), i.e. the state_dict format;
* "mi355x order" -- this repository's re-derivation of the interleave for the
                    v_mfma_f32_16x16x32_f16 A-operand fragment; same tensor
                    shapes/dtypes, different element order.
"""
from __future__ import annotations

import numpy as np

__all__ = [
    "quantize_intweight", "pack_cuda_order", "unpack_cuda_order", "pack_mi355x", "unpack_mi355x",
    "unpack_mi355x_columns", "dequantize", "gemm_fp32acc", "w4a16_forward", "gemm_splitk_fp16_partials", "quick_cat_cuda_order",
    "algorithmic_bytes", "algorithmic_flops", "make_synthetic",
]


# --------------------------------------------------------------------------------------------
# integer quantisation done by from_linear
# --------------------------------------------------------------------------------------------
def quantize_intweight(weight: np.ndarray, scales: np.ndarray, zeros: np.ndarray, group_size: int) -> np.ndarray:
    """``intweight[k, n] = round((W[n, k] + z[n, g] * s[n, g]) / s[n, g])`` in fp16 arithmetic.

    Restates quick.py:67-81: ``scale_zeros = zeros * scales`` (fp16), then per input column
    ``torch.round((weight[:, idx] + scale_zeros[:, idx // G]) / scales[:, idx // G]).to(int)``,
    transposed to [K, N].  ``weight`` is [N, K] fp16, ``scales``/``zeros`` are [N, K/G] (the
    layout AwqQuantizer passes for version 'QUICK', quantizer.py:154-174).
    torch.round and np.rint both round half to even.
    """
    w = np.asarray(weight, dtype=np.float16)
    s = np.asarray(scales, dtype=np.float16)
    zf = np.asarray(zeros).astype(np.float16)
    sz = (zf * s).astype(np.float16)                       # quick.py:67
    s_full = np.repeat(s, group_size, axis=1)              # [N, K]
    sz_full = np.repeat(sz, group_size, axis=1)
    q = ((w + sz_full).astype(np.float16) / s_full).astype(np.float16)  # quick.py:78 (two fp16 ops)
    iw = np.rint(q.astype(np.float32)).astype(np.int32)    # torch.round(...).to(torch.int)
    return np.ascontiguousarray(iw.T)                      # quick.py:80 -> [K, N]


# --------------------------------------------------------------------------------------------
# "cuda order": the reference's packed format (closed form of quick.py:88-150)
# --------------------------------------------------------------------------------------------
def _cuda_weight_index(K: int, N: int):
    """Flat dword index and nibble position of logical weight (k, n) in the reference qweight.

    Derived from the kernel pointer math (csrc/gemm_cuda_quick.cu:1257-1262, 1272), the fragment
    use in compute_gemm (gemm_cuda_quick.cu:20-455), the nibble order of
    dequantize_s4_to_fp16x2_fused (csrc/dequantize_quick.cuh:15-63) and the packer
    (quick.py:88-119).  Uses the kernel's *flat* indexing, which coincides with the packer's
    row/column arithmetic wherever the packer works at all (N == 128 or N % 256 == 0).
    """
    k = np.arange(K, dtype=np.int64)[:, None]
    n = np.arange(N, dtype=np.int64)[None, :]
    kt, half, r = k // 32, (k % 32) // 16, k % 16
    l4, hi, odd = (r % 8) // 2, r // 8, r % 2
    bx, ty, chunk, t, j = n // 128, (n // 64) % 2, (n % 64) // 16, (n % 16) // 8, n % 8
    idx = kt * (4 * N) + ((2 * ty + j // 4) * (N // 8) + 16 * bx + 4 * (j % 4) + l4) * 8 + 4 * half + chunk
    nib = 4 * odd + hi + 2 * t
    return idx + np.zeros_like(k + n), nib + np.zeros_like(k + n)


def _cuda_sz_slot(N: int) -> np.ndarray:
    """Slot x(n) shared by the reference scale/zero permutation (quick.py:121-150; kernel 1260-1261,
    1273-1277): scales[g, 2x] = scales[g, 2x+1] = s[g, n]; nibbles x%4 and x%4+4 of
    qzeros[g, x/4] = z[g, n]."""
    n = np.arange(N, dtype=np.int64)
    bx, ty, chunk, t, j = n // 128, (n // 64) % 2, (n % 64) // 16, (n % 16) // 8, n % 8
    return ((2 * ty + j // 4) * (N // 32) + 4 * bx + (j % 4)) * 8 + 2 * chunk + t


def _scatter_nibbles(n_dwords: int, idx: np.ndarray, nib: np.ndarray, val: np.ndarray) -> np.ndarray:
    out = np.zeros(n_dwords, dtype=np.uint32)
    np.bitwise_or.at(out, idx.ravel(), (val.ravel().astype(np.uint32) & 15) << (4 * nib.ravel()).astype(np.uint32))
    return out


def pack_cuda_order(iw: np.ndarray, s: np.ndarray, z: np.ndarray):
    """(iw [K,N], s [K/G,N], z [K/G,N]) -> (qweight int32 [K/4, N/2], scales fp16 [K/G, 2N],
    qzeros int32 [K/G, N/4]) exactly as quick.py:88-150 produces them."""
    K, N = iw.shape
    NG = s.shape[0]
    assert K % 32 == 0 and N % 128 == 0
    idx, nib = _cuda_weight_index(K, N)
    qweight = _scatter_nibbles(K * N // 8, idx, nib, iw).view(np.int32).reshape(K // 4, N // 2)
    x = _cuda_sz_slot(N)
    qscales = np.zeros((NG, 2 * N), dtype=np.float16)
    qscales[:, 2 * x] = s
    qscales[:, 2 * x + 1] = s
    zz = np.asarray(z).astype(np.uint32) & 15
    qz = np.zeros((NG, N // 4), dtype=np.uint32)
    for rep in (0, 4):
        np.bitwise_or.at(qz, (np.arange(NG)[:, None], (x // 4)[None, :]), zz << (4 * (x % 4 + rep)).astype(np.uint32)[None, :])
    return qweight, qscales, qz.view(np.int32)


def unpack_cuda_order(qweight: np.ndarray, qscales: np.ndarray, qzeros: np.ndarray):
    """Inverse of :func:`pack_cuda_order`: recover (iw uint8 [K,N], s fp16 [K/G,N], z uint8 [K/G,N])."""
    K, N = qweight.shape[0] * 4, qweight.shape[1] * 2
    idx, nib = _cuda_weight_index(K, N)
    flat = np.ascontiguousarray(qweight).view(np.uint32).ravel()
    iw = ((flat[idx] >> (4 * nib).astype(np.uint32)) & 15).astype(np.uint8)
    x = _cuda_sz_slot(N)
    s = np.ascontiguousarray(qscales[:, 2 * x]).astype(np.float16)
    qz = np.ascontiguousarray(qzeros).view(np.uint32)
    z = ((qz[:, x // 4] >> (4 * (x % 4)).astype(np.uint32)[None, :]) & 15).astype(np.uint8)
    return iw, s, z


# --------------------------------------------------------------------------------------------
# "mi355x order": this repository's packed format (see DESIGN.md, section "Data layout")
# --------------------------------------------------------------------------------------------
def _mi355x_weight_index(K: int, N: int):
    """Weight (k, n) -> (flat dword index, nibble) in the MI355X-order qweight.

    tile (nt = n/16, kt = k/128) is 1 KiB contiguous; inside it lane = (n%16) + 16*((k%32)/8)
    owns 16 bytes = dwords t = (k%128)/32; inside a dword the 8 consecutive k of the lane sit at
    nibble p = 4*(j%2) + j/2 (j = k%8), so that (q & 0x000f000f), (q & 0x00f000f0), ((q>>8) & ...)
    yield the fp16 pairs (k0,k1),(k2,k3),(k4,k5),(k6,k7) in MFMA A-operand register order.
    """
    k = np.arange(K, dtype=np.int64)[:, None]
    n = np.arange(N, dtype=np.int64)[None, :]
    nt, kt, t = n // 16, k // 128, (k % 128) // 32
    lane = (n % 16) + 16 * ((k % 32) // 8)
    j = k % 8
    idx = ((nt * (K // 128) + kt) * 64 + lane) * 4 + t
    nib = 4 * (j % 2) + j // 2
    return idx + np.zeros_like(k + n), nib + np.zeros_like(k + n)


def _mi355x_group_word_index(NG: int, N: int) -> np.ndarray:
    """32-bit word index of the (scale, zero point) pair of (group g, channel n) inside the scales tensor viewed as
    uint32 [NG * N]: the pairs of one 16-channel block are contiguous over all groups -- ((n/16) * NG + g) * 16 + n%16."""
    g = np.arange(NG, dtype=np.int64)[:, None]
    n = np.arange(N, dtype=np.int64)[None, :]
    return ((n // 16) * NG + g) * 16 + (n % 16)


def pack_mi355x(iw: np.ndarray, s: np.ndarray, z: np.ndarray):
    """(iw, s, z) -> (qweight int32 [K/4, N/2], scales fp16 [K/G, 2N], qzeros int32 [K/G, N/4]).

    The scales tensor holds one 32-bit word per (group, channel): fp16 scale in the low half, zero point (0..15) in the
    high half, at word ((n/16) * NG + g) * 16 + n%16 -- the reference's 2N halves per group row are exactly that many
    bytes (its duplicate slots are not needed on gfx950), so the kernels fetch scale and zero point with one load and the
    constants of a 16-channel block stream contiguously along K.  qzeros keeps a plain copy (nibble n%8 of dword n/8,
    dwords N/8..N/4-1 zero) that the GEMM kernels do not read.
    """
    K, N = iw.shape
    NG = s.shape[0]
    assert K % 128 == 0 and N % 16 == 0
    idx, nib = _mi355x_weight_index(K, N)
    qweight = _scatter_nibbles(K * N // 8, idx, nib, iw).view(np.int32).reshape(K // 4, N // 2)
    words = np.zeros(NG * N, dtype=np.uint32)
    words[_mi355x_group_word_index(NG, N)] = (np.asarray(s, dtype=np.float16).view(np.uint16).astype(np.uint32)
                                              | ((np.asarray(z).astype(np.uint32) & 15) << 16))
    qscales = words.view(np.float16).reshape(NG, 2 * N)
    n = np.arange(N)
    qz = np.zeros((NG, N // 4), dtype=np.uint32)
    np.bitwise_or.at(qz, (np.arange(NG)[:, None], (n // 8)[None, :]),
                     (np.asarray(z).astype(np.uint32) & 15) << (4 * (n % 8)).astype(np.uint32)[None, :])
    return qweight, qscales, qz.view(np.int32)


def unpack_mi355x(qweight: np.ndarray, qscales: np.ndarray, qzeros: np.ndarray):
    K, N = qweight.shape[0] * 4, qweight.shape[1] * 2
    idx, nib = _mi355x_weight_index(K, N)
    flat = np.ascontiguousarray(qweight).view(np.uint32).ravel()
    iw = ((flat[idx] >> (4 * nib).astype(np.uint32)) & 15).astype(np.uint8)
    NG = qscales.shape[0]
    words = np.ascontiguousarray(qscales).view(np.uint32).ravel()[_mi355x_group_word_index(NG, N)]
    s = (words & 0xffff).astype(np.uint16).view(np.float16)
    z = ((words >> 16) & 15).astype(np.uint8)
    return iw, s, z


def unpack_mi355x_columns(qweight: np.ndarray, qscales: np.ndarray, qzeros: np.ndarray, cols):
    """(iw[K, len(cols)], s[K/G, len(cols)], z[K/G, len(cols)]) of the output channels `cols` only -- the same closed
    form as unpack_mi355x, evaluated for a handful of columns, so that a test can check sampled channels of a layer that
    is far too large to unpack (or to dequantise on the CPU) whole."""
    K, N = qweight.shape[0] * 4, qweight.shape[1] * 2
    cols = np.asarray(cols, dtype=np.int64)
    k = np.arange(K, dtype=np.int64)[:, None]
    n = cols[None, :]
    idx = (((n // 16) * (K // 128) + k // 128) * 64 + (n % 16) + 16 * ((k % 32) // 8)) * 4 + (k % 128) // 32
    j = k % 8
    nib = (4 * (j % 2) + j // 2) + np.zeros_like(n)
    flat = np.ascontiguousarray(qweight).view(np.uint32).ravel()
    iw = ((flat[idx] >> (4 * nib).astype(np.uint32)) & 15).astype(np.uint8)
    NG = qscales.shape[0]
    g = np.arange(NG, dtype=np.int64)[:, None]
    words = np.ascontiguousarray(qscales).view(np.uint32).ravel()[((n // 16) * NG + g) * 16 + (n % 16)]
    return iw, (words & 0xffff).astype(np.uint16).view(np.float16), ((words >> 16) & 15).astype(np.uint8)


# --------------------------------------------------------------------------------------------
# dequantisation + GEMM numerics
# --------------------------------------------------------------------------------------------
def dequantize(iw: np.ndarray, s: np.ndarray, z: np.ndarray, group_size: int) -> np.ndarray:
    """W_deq[k, n] = fp16( (iw - z) * s ), the integer difference being exact and the product
    rounded once to fp16.

    This is what both reference implementations compute per weight:
    * CPU path ``dequantize_gemm`` (quick/awq/utils/packing_utils.py:82-97): int8 (iw - iz) times
      fp16 scales -> fp16;
    * CUDA kernel: (1024+w) - (1024+z) with ``sub.f16x2`` (exact) then ``mul.rn.f16x2`` by the scale
      (csrc/dequantize_quick.cuh:15-63, csrc/gemm_cuda_quick.cu:52-60).
    """
    d = iw.astype(np.int16) - np.repeat(z.astype(np.int16), group_size, axis=0)
    return (d.astype(np.float16) * np.repeat(s.astype(np.float16), group_size, axis=0)).astype(np.float16)


def gemm_fp32acc(x: np.ndarray, w_deq: np.ndarray) -> np.ndarray:
    """y = fp16( sum_k fp32(x) * fp32(w_deq) ): fp16 operands, fp32 accumulation, one final rounding.

    Restates the reference CPU path ``torch.matmul(x, dequantize_gemm(...))``
    (quick/awq/modules/linear/gemm.py:173-181; CPU fp16 matmul accumulates in fp32) and the
    accumulate-in-fp32 contract of mma.sync.m16n8k16.f32.f16.f16.f32 (gemm_cuda_quick.cu:62-75).
    Summation order is unspecified on both sides, hence the 1e-2 relative tolerance of the parity
    tests rather than bit equality.
    """
    return (x.astype(np.float32) @ w_deq.astype(np.float32)).astype(np.float16)


def w4a16_forward(x: np.ndarray, iw: np.ndarray, s: np.ndarray, z: np.ndarray, group_size: int,
                  bias: np.ndarray | None = None) -> np.ndarray:
    """WQLinear_QUICK.forward (quick.py:158-166) on logical tensors: y = x @ dequant(W) (+ bias)."""
    x2 = x.reshape(-1, x.shape[-1])
    y = gemm_fp32acc(x2, dequantize(iw, s, z, group_size))
    if bias is not None:
        y = (y + bias.astype(np.float16)).astype(np.float16)     # quick.py:165, fp16 add
    return y.reshape(x.shape[:-1] + (iw.shape[1],))


def gemm_splitk_fp16_partials(x: np.ndarray, w_deq: np.ndarray, split_k: int) -> np.ndarray:
    """What the CUDA host function returns: split-K slice i takes k-tiles kt with kt % split_k == i
    (gemm_cuda_quick.cu:1221-1222), each partial is rounded to fp16 (``__float22half2_rn``, 1236-1241
    / 1287) and the partials are summed by ``_out_feats.sum(0)`` in fp16 (1515).  Kept to document
    how far the reference's own GPU numerics sit from the single-rounding result."""
    M, K = x.shape
    kt = np.arange(K) // 32
    parts = []
    for i in range(split_k):
        sel = (kt % split_k) == i
        parts.append((x[:, sel].astype(np.float32) @ w_deq[sel].astype(np.float32)).astype(np.float16))
    acc = np.zeros_like(parts[0], dtype=np.float32)          # torch sum over fp16 accumulates in fp32
    for p in parts:
        acc += p.astype(np.float32)
    return acc.astype(np.float16)


# --------------------------------------------------------------------------------------------
# QUICK_cat (packed-space concatenation along N), reference behaviour
# --------------------------------------------------------------------------------------------
def quick_cat_cuda_order(layers, options: str) -> np.ndarray:
    """quick/awq/utils/fused_utils.py:119-159: reshape every [H, W] input to
    qweight (H/2, 2W) / qzeros, scales (4H, W/4), concatenate along dim 1, reshape to (H, -1)."""
    H, W = layers[0].shape
    for l in layers[1:]:
        if l.shape != layers[0].shape:
            raise ValueError("All input layers must have the same shape")
    dims = {"qweight": (H // 2, W * 2), "qzeros": (H * 4, W // 4), "scales": (H * 4, W // 4)}[options]
    return np.concatenate([l.reshape(dims) for l in layers], axis=1).reshape(H, -1)


# --------------------------------------------------------------------------------------------
# bookkeeping used by bench.py / tests
# --------------------------------------------------------------------------------------------
def algorithmic_bytes(M: int, K: int, N: int, G: int) -> int:
    """int4 weights + fp16 scales + int4 zeros (both un-duplicated) + A + C."""
    return K * N // 2 + (K // G) * N * 2 + (K // G) * N // 2 + 2 * M * K + 2 * M * N


def algorithmic_flops(M: int, K: int, N: int) -> int:
    return 2 * M * K * N


def make_synthetic(M: int, K: int, N: int, G: int, seed: int = 0):
    """SURVEY.md section 8(d) synthetic inputs: iw, z ~ U{0..15}; s ~ U(0.005, 0.025) fp16; x ~ N(0,1) fp16."""
    rng = np.random.default_rng(seed)
    iw = rng.integers(0, 16, size=(K, N), dtype=np.uint8)
    z = rng.integers(0, 16, size=(K // G, N), dtype=np.uint8)
    s = rng.uniform(0.005, 0.025, size=(K // G, N)).astype(np.float16)
    x = rng.standard_normal(size=(M, K)).astype(np.float16)
    return x, iw, s, z
