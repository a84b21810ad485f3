"""Host-side packing for WQLinear_QUICK (torch ops, any device).

Counterpart of the Python-loop packer in the reference's ``WQLinear_QUICK.from_linear``
(quick/awq/modules/linear/quick.py:76-150), vectorised, and extended with the MI355X order that
the gfx950 kernels consume.  Logical tensors:

    iw [K, N] int (0..15)     s [K/G, N] fp16     z [K/G, N] int (0..15)

"cuda order"   = byte-for-byte the reference's packed tensors / checkpoint format.
"mi355x order" = same shapes and dtypes, elements permuted for the v_mfma_f32_16x16x32_f16 A-operand
                 fragment (DESIGN.md "Data layout"): tile (n/16, k/128) is 1 KiB contiguous, lane
                 (n%16) + 16*((k%32)/8) owns 16 bytes = the dwords of 4 consecutive k-steps, and inside
                 a dword k-offset j sits at nibble 4*(j%2) + j/2.
"""
import torch

__all__ = ["quantize_intweight", "pack_cuda_order", "unpack_cuda_order", "pack_mi355x", "unpack_mi355x",
           "cuda_to_mi355x", "mi355x_to_cuda"]


def quantize_intweight(weight, scales, zeros, group_size):
    """[N, K] fp16 weight + [N, K/G] scales/zeros -> integer weights [K, N] (quick.py:67-81, same fp16
    arithmetic: round((W + z*s) / s)), done for all columns at once instead of one column per Python step."""
    scales = scales.to(torch.float16)
    scale_zeros = (zeros * scales).to(weight.dtype)
    s_full = scales.repeat_interleave(group_size, dim=1)
    q = (weight + scale_zeros.repeat_interleave(group_size, dim=1)) / s_full
    return torch.round(q).to(torch.int32).t().contiguous()


def _arange(n, dev):
    return torch.arange(n, device=dev, dtype=torch.int64)


def _cuda_weight_pos(K, N, dev):
    k = _arange(K, dev)[:, None]
    n = _arange(N, dev)[None, :]
    kt, half, r = k // 32, (k % 32) // 16, k % 16
    l4, hi, odd = (r % 8) // 2, r // 8, r % 2
    bx, ty, chunk, t, j = n // 128, (n // 64) % 2, (n % 64) // 16, (n % 16) // 8, n % 8
    idx = kt * (4 * N) + ((2 * ty + j // 4) * (N // 8) + 16 * bx + 4 * (j % 4) + l4) * 8 + 4 * half + chunk
    nib = 4 * odd + hi + 2 * t
    return idx.expand(K, N), nib.expand(K, N)


def _cuda_slot(N, dev):
    n = _arange(N, dev)
    bx, ty, chunk, t, j = n // 128, (n // 64) % 2, (n % 64) // 16, (n % 16) // 8, n % 8
    return ((2 * ty + j // 4) * (N // 32) + 4 * bx + (j % 4)) * 8 + 2 * chunk + t


def _mi355x_weight_pos(K, N, dev):
    k = _arange(K, dev)[:, None]
    n = _arange(N, dev)[None, :]
    lane = (n % 16) + 16 * ((k % 32) // 8)
    idx = (((n // 16) * (K // 128) + k // 128) * 64 + lane) * 4 + (k % 128) // 32
    j = k % 8
    nib = 4 * (j % 2) + j // 2
    return idx.expand(K, N), nib.expand(K, N)


def _scatter_nibbles(numel, idx, nib, val):
    """OR 4-bit values into a flat int64 accumulator (every nibble slot is hit exactly once, so add == or)."""
    out = torch.zeros(numel, dtype=torch.int64, device=val.device)
    out.scatter_add_(0, idx.reshape(-1), (val.reshape(-1).to(torch.int64) & 15) << (4 * nib.reshape(-1)))
    return _to_i32(out)


def _to_i32(x64):
    return torch.where(x64 >= 2 ** 31, x64 - 2 ** 32, x64).to(torch.int32)


def _as_u32(x32):
    return x32.to(torch.int64) & 0xFFFFFFFF


def _check(K, N, mi355x):
    if N % 128 != 0:
        raise ValueError("OC is not multiple of cta_N = 128")
    if K % 32 != 0:
        raise ValueError("in_features must be a multiple of 32")
    if mi355x and K % 128 != 0:
        raise ValueError("in_features must be a multiple of 128 for the MI355X weight order")


def pack_cuda_order(iw, s, z):
    """-> (qweight int32 [K/4, N/2], scales fp16 [K/G, 2N], qzeros int32 [K/G, N/4]), reference order."""
    K, N = iw.shape
    _check(K, N, False)
    dev = iw.device
    idx, nib = _cuda_weight_pos(K, N, dev)
    qweight = _scatter_nibbles(K * N // 8, idx, nib, iw).reshape(K // 4, N // 2)
    x = _cuda_slot(N, dev)
    NG = s.shape[0]
    qscales = torch.zeros((NG, 2 * N), dtype=torch.float16, device=dev)
    qscales[:, 2 * x] = s.to(torch.float16)
    qscales[:, 2 * x + 1] = s.to(torch.float16)
    zz = z.to(torch.int64) & 15
    qz = torch.zeros((NG, N // 4), dtype=torch.int64, device=dev)
    col = (x // 4)[None, :].expand(NG, N)
    qz.scatter_add_(1, col, (zz << (4 * (x % 4))[None, :]) | (zz << (4 * (x % 4) + 16)[None, :]))
    return qweight, qscales, _to_i32(qz)


def unpack_cuda_order(qweight, qscales, qzeros):
    K, N = qweight.shape[0] * 4, qweight.shape[1] * 2
    _check(K, N, False)
    dev = qweight.device
    idx, nib = _cuda_weight_pos(K, N, dev)
    flat = _as_u32(qweight.reshape(-1))
    iw = ((flat[idx] >> (4 * nib)) & 15).to(torch.uint8)
    x = _cuda_slot(N, dev)
    s = qscales[:, 2 * x].contiguous()
    z = ((_as_u32(qzeros)[:, x // 4] >> (4 * (x % 4))[None, :]) & 15).to(torch.uint8)
    return iw, s, z


def _mi355x_group_word_pos(NG, N, dev):
    """32-bit word of the (scale, zero point) pair of (g, n) in the scales tensor viewed as int32 [NG * N]."""
    g = _arange(NG, dev)[:, None]
    n = _arange(N, dev)[None, :]
    return ((n // 16) * NG + g) * 16 + (n % 16)


def pack_mi355x(iw, s, z):
    """-> (qweight, scales, qzeros) in MI355X order (same shapes / dtypes as the reference buffers).  The scales tensor
    carries one 32-bit word per (group, channel) -- fp16 scale | zero point << 16 -- block-contiguous along K; qzeros
    keeps a plain copy of the zero points that the GEMM kernels do not read (oracle/w4a16.py::pack_mi355x)."""
    K, N = iw.shape
    _check(K, N, True)
    dev = iw.device
    idx, nib = _mi355x_weight_pos(K, N, dev)
    qweight = _scatter_nibbles(K * N // 8, idx, nib, iw).reshape(K // 4, N // 2)
    NG = s.shape[0]
    sbits = s.to(torch.float16).contiguous().view(torch.int16).to(torch.int64) & 0xffff
    words = torch.zeros(NG * N, dtype=torch.int64, device=dev)
    words[_mi355x_group_word_pos(NG, N, dev).reshape(-1)] = (sbits | ((z.to(torch.int64) & 15) << 16)).reshape(-1)
    qscales = _to_i32(words).view(torch.float16).reshape(NG, 2 * N)
    n = _arange(N, dev)
    qz = torch.zeros((NG, N // 4), dtype=torch.int64, device=dev)
    qz.scatter_add_(1, (n // 8)[None, :].expand(NG, N), (z.to(torch.int64) & 15) << (4 * (n % 8))[None, :])
    return qweight, qscales, _to_i32(qz)


def unpack_mi355x(qweight, qscales, qzeros):
    K, N = qweight.shape[0] * 4, qweight.shape[1] * 2
    _check(K, N, True)
    dev = qweight.device
    idx, nib = _mi355x_weight_pos(K, N, dev)
    flat = _as_u32(qweight.reshape(-1))
    iw = ((flat[idx] >> (4 * nib)) & 15).to(torch.uint8)
    NG = qscales.shape[0]
    words = _as_u32(qscales.contiguous().view(torch.int32).reshape(-1))[_mi355x_group_word_pos(NG, N, dev)]
    s = (words & 0xffff).to(torch.int32).to(torch.int16).view(torch.float16)
    z = ((words >> 16) & 15).to(torch.uint8)
    return iw, s, z


def random_mi355x(K, N, G, device, generator=None, zero_point=None, scale_lo=0.005, scale_span=0.02):
    """Random packed tensors in MI355X order (benchmarks, synthetic models): uniform 4-bit weights, scales in
    [scale_lo, scale_lo + scale_span), zero points uniform in 0..15 or the given constant."""
    NG = K // G
    qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 4, N // 2), dtype=torch.int32, device=device, generator=generator)
    s = (torch.rand((NG, N), device=device, generator=generator) * scale_span + scale_lo).half()
    if zero_point is None:
        z = torch.randint(0, 16, (NG, N), dtype=torch.int64, device=device, generator=generator)
    else:
        z = torch.full((NG, N), int(zero_point), dtype=torch.int64, device=device)
    sbits = s.view(torch.int16).to(torch.int64) & 0xffff
    words = torch.zeros(NG * N, dtype=torch.int64, device=device)
    words[_mi355x_group_word_pos(NG, N, device).reshape(-1)] = (sbits | (z << 16)).reshape(-1)
    n = _arange(N, device)
    qz = torch.zeros((NG, N // 4), dtype=torch.int64, device=device)
    qz.scatter_add_(1, (n // 8)[None, :].expand(NG, N), z << (4 * (n % 8))[None, :])
    return qw, _to_i32(words).view(torch.float16).reshape(NG, 2 * N), _to_i32(qz)


def cuda_to_mi355x(qweight, qscales, qzeros):
    """Reference-order packed tensors -> MI355X order (torch ops; the GPU path uses the HIP repack kernels)."""
    return pack_mi355x(*unpack_cuda_order(qweight, qscales, qzeros))


def mi355x_to_cuda(qweight, qscales, qzeros):
    return pack_cuda_order(*unpack_mi355x(qweight, qscales, qzeros))
