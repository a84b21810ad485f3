"""Builds the compiled `quick_kernels` torch extension (quick_amd/csrc/quick_kernels_ext.cpp) in-tree and returns the
module: the pybind11 face of libquick_amd.so with the reference's symbol (csrc/pybind.cpp:5-8).

    python -m quick_amd.build_ext          # -> quick_amd/lib/quick_kernels_ext/quick_kernels_ext.so

The ctypes shim `quick_kernels.py` at the repository root needs no compiler at install time and is what the package uses by
default; this is the same boundary as a compiled module, for deployments that want one.  Needs the torch headers and a
host C++ compiler; no GPU.
"""
import os

from .build import LIBDIR, build

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
INCLUDE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")


def build_quick_kernels_ext(verbose=False):
    from torch.utils import cpp_extension
    build()                                   # the C-ABI library the extension links against
    out = os.path.join(LIBDIR, "quick_kernels_ext")
    os.makedirs(out, exist_ok=True)
    return cpp_extension.load(
        name="quick_kernels_ext", sources=[os.path.join(CSRC, "quick_kernels_ext.cpp")], extra_include_paths=[INCLUDE],
        extra_cflags=["-O2", "-D__HIP_PLATFORM_AMD__=1"],
        extra_ldflags=[f"-L{LIBDIR}", "-lquick_amd", f"-Wl,-rpath,{LIBDIR}"],
        with_cuda=True, build_directory=out, verbose=verbose)


if __name__ == "__main__":
    m = build_quick_kernels_ext(verbose=True)
    print(m.__file__)
