"""ctypes binding of libquick_amd.so (include/quick_amd.h).  Fails loudly: there is no fallback."""
import ctypes
import os

from .build import LIB as _DEFAULT_LIB

LIB = _DEFAULT_LIB   # the shared object in use (see load())

_P, _I, _Z = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t


class GemmFusion(ctypes.Structure):
    """struct quick_gemm_fusion (include/quick_amd.h)."""
    _fields_ = [("bias", _P), ("residual", _P), ("rmsnorm_weight", _P), ("rmsnorm_eps", ctypes.c_float), ("silu_mul", _I)]


_SIGNATURES = {
    "quick_amd_abi_version": (_I, []),
    "quick_amd_last_error": (ctypes.c_char_p, []),
    "quick_w4a16_gemm_f16": (_I, [_P, _P, _P, _P, _P, _P, _Z, _I, _I, _I, _I, _I, _P]),
    "quick_w4a16_workspace_bytes": (_Z, [_I, _I, _I, _I, _I]),
    "quick_w4a16_workspace_check": (_I, [_P, _Z, _P]),
    "quick_w4a16_gemm_f16_ex": (_I, [_P, _P, _P, _P, _P, _P, _P, _Z, _I, _I, _I, _I, _I, _I, _P]),
    "quick_w4a16_workspace_bytes_ex": (_Z, [_I, _I, _I, _I, _I, _I]),
    "quick_w4a16_gemm_profile": (_I, [_P, _P, _P, _P, _I, _P, _P, _Z, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "quick_w4a16_gemm_span": (_I, [_P, _P, _P, _P, _I, _P, _P, _Z, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "quick_w4a16_gemm_f16_fused": (_I, [_P, _P, _P, _P, _P, _P, _P, _Z, _I, _I, _I, _I, _I, _I, _P]),
    "quick_w4a16_can_fuse_rmsnorm": (_I, [_I, _I, _I, _I]),
    "quick_w4a16_plan_describe": (_I, [_I, _I, _I, _I, _I, _I, ctypes.c_char_p, _Z]),
    "quick_rmsnorm_f16": (_I, [_P, _P, _P, _I, _I, ctypes.c_float, _P]),
    "quick_rope_kv_append_f16": (_I, [_P] * 7 + [_I] * 5 + [_P]),
    "quick_rope_kv_write_f16": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "quick_decode_attention_f16": (_I, [_P] * 5 + [_I] * 5 + [ctypes.c_float, _P]),
    "quick_decode_rope_attention_f16": (_I, [_P] * 7 + [_I] * 5 + [ctypes.c_float, _P]),
    "quick_silu_mul_f16": (_I, [_P, _P, _I, _I, _P]),
    "quick_lm_head_workspace_bytes": (_Z, [_I]),
    "quick_lm_head_argmax_f16": (_I, [_P, _P, ctypes.c_float, _P, _P, _P, _P, _P, _Z, _I, _I, _I, _P]),
    "quick_amd_dispatch_floor": (_I, [_I, _P, _P]),
    "quick_repack_cuda_to_mi355x": (_I, [_P] * 6 + [_I, _I, _I, _P]),
    "quick_repack_mi355x_to_cuda": (_I, [_P] * 6 + [_I, _I, _I, _P]),
    "quick_padded_in_features": (_I, [_I, _I]),
    "quick_repack_cuda_to_mi355x_padded": (_I, [_P] * 6 + [_I, _I, _I, _P]),
    "quick_dequantize_mi355x_f16": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
}
EXPORTS = tuple(_SIGNATURES)
_TOOLS_ONLY = {"quick_prefetch": (_I, [_P, _Z, _I, _P])}   # measurement aids of libquick_amd_tools.so (-DQUICK_AMD_TOOLS), absent from the product library

_lib = None


def load():
    """Load the in-tree shared object; raise if it is missing (build it with `python -m quick_amd.build`)."""
    global _lib, LIB
    if _lib is None:
        LIB = os.environ.get("QUICK_AMD_LIB_OVERRIDE") or _DEFAULT_LIB   # override: A/B timing of kernel builds (tools/ab.sh)
        if not os.path.exists(LIB):
            raise ImportError(
                f"{LIB} is missing: the HIP extension has not been built. "
                "Run `python -m quick_amd.build` (needs hipcc); there is no CPU fallback for the W4A16 GEMM.")
        lib = ctypes.CDLL(LIB)
        older = LIB != _DEFAULT_LIB      # an A/B library built from an earlier tree (tools/audit_vs_r03.py) may lack later entry points
        for name, (res, args) in _SIGNATURES.items():
            if older and not hasattr(lib, name):
                continue
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        for name, (res, args) in _TOOLS_ONLY.items():
            if hasattr(lib, name):
                fn = getattr(lib, name)
                fn.restype, fn.argtypes = res, args
        if lib.quick_amd_abi_version() != 1:
            raise ImportError("libquick_amd.so ABI version mismatch; rebuild with `python -m quick_amd.build --force`")
        _lib = lib
    return _lib


def load_other(path):
    """A second library with the same C ABI (tests: the -amdgpu-waitcnt-forcezero build of the product sources, quick_amd.build.FORCEZERO_LIB)."""
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: build it with `python -m quick_amd.build --forcezero`")
    lib = ctypes.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


def last_error():
    return load().quick_amd_last_error().decode()
