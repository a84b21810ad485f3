"""The reference's only CPU-executable W4A16 linear, restated with torch CPU ops.  TEST INFRASTRUCTURE
(checker + the timed `cpu_baseline` leg of bench.py); never imported by the product.

The reference falls back to this when its `awq_ext` CUDA extension is absent:
``WQLinear_GEMM.forward`` -> ``dequantize_gemm`` + ``torch.matmul`` (quick/awq/modules/linear/gemm.py:173-181,
quick/awq/utils/packing_utils.py:82-97).  It works on the AWQ "GEMM" packing ([K, N/8] int32, eight output
channels per dword in AWQ order), not on the QUICK packing, and re-does the whole dequantisation on every
call -- which is what dominates its run time and what the baseline must therefore include.
"""
import numpy as np
import torch

AWQ_ORDER = [0, 2, 4, 6, 1, 3, 5, 7]          # gemm.py:119-127 / packing_utils.py:4
AWQ_REVERSE_ORDER = [0, 4, 1, 5, 2, 6, 3, 7]  # packing_utils.py:5


def pack_gemm_format(iw: np.ndarray, z: np.ndarray):
    """Logical (iw [K,N], z [K/G,N]) -> (qweight int32 [K, N/8], qzeros int32 [K/G, N/8]); nibble i of dword c
    holds channel 8c + AWQ_ORDER[i] (WQLinear_GEMM.from_linear, gemm.py:112-148)."""
    def pack(a):
        a = a.astype(np.uint32).reshape(a.shape[0], -1, 8)[:, :, AWQ_ORDER]
        return (a << (4 * np.arange(8, dtype=np.uint32))).sum(axis=2, dtype=np.uint32).view(np.int32)
    return pack(iw), pack(z)


def dequantize_gemm(qweight: torch.Tensor, qzeros: torch.Tensor, scales: torch.Tensor, group_size: int) -> torch.Tensor:
    """packing_utils.py:82-97 (unpack_awq 10-26, reverse_awq_order 29-44): shift-unpack to int8, undo the AWQ
    order, mask to 4 bits, expand the groups, (w - z) * s in fp16."""
    shifts = torch.arange(0, 32, 4)
    iw = torch.bitwise_right_shift(qweight[:, :, None], shifts[None, None, :]).to(torch.int8).view(qweight.shape[0], -1)
    iz = torch.bitwise_right_shift(qzeros[:, :, None], shifts[None, None, :]).to(torch.int8).view(qzeros.shape[0], -1)
    order = torch.arange(iz.shape[-1], dtype=torch.int32).view(-1, 8)[:, AWQ_REVERSE_ORDER].reshape(-1)
    iw, iz = iw[:, order], iz[:, order]
    iw, iz = torch.bitwise_and(iw, 15), torch.bitwise_and(iz, 15)
    s = scales.repeat_interleave(group_size, dim=0)
    iz = iz.repeat_interleave(group_size, dim=0)
    return (iw - iz) * s


@torch.no_grad()
def forward(x: torch.Tensor, qweight: torch.Tensor, qzeros: torch.Tensor, scales: torch.Tensor, group_size: int) -> torch.Tensor:
    """gemm.py:152-187 with AWQ_INSTALLED == False: dequantise, then fp16 torch.matmul on the host cores."""
    out_shape = x.shape[:-1] + (qweight.shape[1] * 8,)
    w = dequantize_gemm(qweight, qzeros, scales, group_size)
    return torch.matmul(x.reshape(-1, x.shape[-1]), w).reshape(out_shape)
