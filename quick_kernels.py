"""Drop-in for the reference's ``quick_kernels`` extension module (csrc/pybind.cpp:5-8): the reference's
``from quick_kernels import gemm_forward_cuda_quick`` (quick/awq/modules/linear/quick.py:4,
quick/awq/modules/fused/mlp.py:6) resolves to the MI355X implementation."""
from quick_amd.kernels import gemm_forward_cuda_quick  # noqa: F401
