"""Builds libquick_amd.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python -m quick_amd.build [--force] [--tools | --forcezero]

The shared object lands in quick_amd/lib/ (git-ignored, but it travels to the GPU box with the
working-tree snapshot).  Cross-compiles without a GPU.  Every translation unit is compiled to its own
object (in parallel, rebuilt only when it or a header changed) and the objects are linked.

--tools builds tools/bin/libquick_amd_tools.so instead: the same library plus the timing-experiment
kernels (ablations with wrong results, phase stamps) that tools/*.py ask for through kernel-id bits 16-20
(-DQUICK_AMD_TOOLS).  The product library contains none of them.

--forcezero builds quick_amd/lib/libquick_amd_forcezero.so: the product sources with `-mllvm -amdgpu-waitcnt-forcezero` (hipcc waits for every
counter in front of every instruction of the code IT schedules; the generated asm loops are what they are).  Not a product: the comparison
partner of tests/test_gemm_gpu.py::test_hipcc_scheduled_kernels_equal_their_forcezero_build -- twice, in r04 and r05, a build computed wrong results
that this flag cured, so every round's GPU suite demands bit-equal outputs from both libraries.
"""
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libquick_amd.so")
TOOLS_DIR = os.path.join(os.path.dirname(PKG), "tools", "bin")   # measurement builds live with the tools, not with the product
TOOLS_LIB = os.path.join(TOOLS_DIR, "libquick_amd_tools.so")
FORCEZERO_LIB = os.path.join(LIBDIR, "libquick_amd_forcezero.so")
SOURCES = ["w4a16_gemm.hip", "w4a16_xk.hip", "w4a16_xw.hip", "w4a16_xm.hip", "w4a16_lean.hip", "w4a16_lean_a.hip", "w4a16_lean_b.hip", "w4a16_lean_c.hip", "repack.hip", "decode_ops.hip"]
HEADERS = ["w4a16_common.hpp", "w4a16_args.hpp", "w4a16_wide.hpp", "w4a16_xk.hpp", "w4a16_xk_host.hpp", "w4a16_xw.hpp", "w4a16_xw_host.hpp", "w4a16_xw_loop.inc", "w4a16_xm.hpp", "w4a16_xm_host.hpp", "w4a16_xm_loop.inc", "w4a16_lean.hpp", "w4a16_lean_host.hpp", "w4a16_lean_inst.hpp",
           os.path.join("..", "..", "include", "quick_amd.h")]
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc"]
LDFLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc"]
# per-source extras: the lean small-M kernels take their leading arguments preloaded into SGPRs (w4a16_lean.hpp)
_PRELOAD = ["-mllvm", "-amdgpu-kernarg-preload-count=16"]
EXTRA_CFLAGS = {"w4a16_xw.hip": _PRELOAD, "w4a16_xm.hip": _PRELOAD, "w4a16_xk.hip": _PRELOAD, "w4a16_lean_a.hip": _PRELOAD, "w4a16_lean_b.hip": _PRELOAD, "w4a16_lean_c.hip": _PRELOAD}
FLAGS = CFLAGS + ["-shared"]   # (what the library is built with, for the record)


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (looked at $HIPCC, /opt/rocm/bin/hipcc, PATH)")


def _sha(paths, extra=""):
    h = hashlib.sha256()
    for f in paths:
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(extra.encode())
    return h.hexdigest()


def _digest(tools=False, forcezero=False):
    return _sha([os.path.join(CSRC, f) for f in SOURCES + HEADERS], " ".join(FLAGS) + repr(sorted(EXTRA_CFLAGS.items())) + (" tools" if tools else "") + (" forcezero" if forcezero else ""))


def _compile(src, obj, flags, verbose):
    cmd = [_hipcc()] + flags + EXTRA_CFLAGS.get(src, []) + ["-c", "-o", obj, os.path.join(CSRC, src)]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed on %s:\n%s%s" % (src, r.stdout, r.stderr))


def build(force=False, verbose=False, tools=False, forcezero=False):
    assert not (tools and forcezero)
    os.makedirs(TOOLS_DIR if tools else LIBDIR, exist_ok=True)
    lib = TOOLS_LIB if tools else (FORCEZERO_LIB if forcezero else LIB)
    stamp = lib + ".sha256"
    dig = _digest(tools, forcezero)
    if not force and os.path.exists(lib) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return lib
    objdir = os.path.join(TOOLS_DIR, "obj_tools") if tools else os.path.join(LIBDIR, "obj_forcezero" if forcezero else "obj")
    os.makedirs(objdir, exist_ok=True)
    flags = CFLAGS + (["-DQUICK_AMD_TOOLS"] if tools else []) + (["-mllvm", "-amdgpu-waitcnt-forcezero"] if forcezero else [])
    hdr = [os.path.join(CSRC, h) for h in HEADERS]
    jobs, objs = [], []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        want = _sha([os.path.join(CSRC, src)] + hdr, " ".join(flags + EXTRA_CFLAGS.get(src, [])))
        ostamp = obj + ".sha256"
        if force or not os.path.exists(obj) or not os.path.exists(ostamp) or open(ostamp).read().strip() != want:
            jobs.append((src, obj, ostamp, want))
    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        futs = [(j, ex.submit(_compile, j[0], j[1], flags, verbose)) for j in jobs]
        for j, f in futs:
            f.result()
            with open(j[2], "w") as fh:
                fh.write(j[3])
    cmd = [_hipcc()] + LDFLAGS + ["-o", lib] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + r.stdout + r.stderr)
    with open(stamp, "w") as f:
        f.write(dig)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, tools="--tools" in sys.argv, forcezero="--forcezero" in sys.argv))
