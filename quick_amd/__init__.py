"""quick_amd -- MI355X-native W4A16 GEMM behind the SqueezeBits/QUICK operator interface.

    from quick_amd import WQLinear_QUICK, gemm_forward_cuda_quick

The arithmetic lives in the in-tree HIP library (quick_amd/lib/libquick_amd.so, built by
``python -m quick_amd.build``); importing this package does not need a GPU, calling the GEMM does.
"""
from .kernels import (dequantize_mi355x, gemm_forward, gemm_forward_cuda_quick, repack_cuda_to_mi355x,  # noqa: F401
                      repack_mi355x_to_cuda)
from .linear import WQLinear_QUICK  # noqa: F401
from .fused_utils import QUICK_cat, fuse_qkv_quick  # noqa: F401
from .quantize import pseudo_quantize_tensor, quantize_linear, quantize_module_linears  # noqa: F401

__version__ = "0.1.0"
