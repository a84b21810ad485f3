// Kernel argument block and the helpers every W4A16 kernel family shares: in-kernel span stamps and the K-split slabs
// ("last arriver finishes the tile").  Included by every translation unit that defines kernels (w4a16_gemm.hip, w4a16_xk.hip).
#pragma once
#include "w4a16_common.hpp"

namespace quick_amd {

struct GemmArgs {
  const half_t* X;
  const u32x4* QW;
  const half_t* S;
  const uint32_t* QZ;
  const half_t* bias;      // [N] or null
  const half_t* residual;  // [M, N] added in the epilogue, or null
  int silu_mul;            // epilogue: y[m, 8t+i] = silu(acc[m, 16t+i]) * acc[m, 16t+8+i]  (gate/up interleaved by 8), Y is [M, N/2]
  half_t* Y;
  float* slabs;        // ksplit > 1: fp32 partial tiles, [tile][slice][slab]
  unsigned* counters;  // ksplit > 1: one arrival counter per output tile (zero on entry, zero again on exit)
  int M, K, N, G;
  int tpg;     // G / 128 (group mode 1)
  int ksplit;  // K slices across workgroups
  int kt_per_split;
  int xcd_gm;  // tiled: XCD-aware tile order -- the 8 XCDs form an xcd_gm x (8/xcd_gm) grid over (token, channel) blocks; 0 = plain order
  unsigned long long* dbg;  // ablation bit 16: per-wave phase cycle totals [workgroup][wave][8]
  const half_t* ln_w;  // deferred-zero skinny kernel: RMSNorm weight [K] applied to x on its way into LDS, or null
  float ln_eps;
  unsigned long long* span;  // measurement aid: per-wave start / end stamps in s_memrealtime ticks (100 MHz), see span_stamp; or null
};

// [r05] The exchange-K and four-wave kernels take what their first requests depend on as leading SCALAR parameters (13 dwords) and the rest as
// this by-value block: their translation units are built with -amdgpu-kernarg-preload-count=16, so the scalars arrive in SGPRs with the
// wave instead of through dependent s_load round trips to memory the host has just written (see w4a16_lean.hpp).
struct XwRest {
  const half_t* bias;
  const half_t* residual;
  half_t* Y;
  float* slabs;
  unsigned* counters;
  unsigned long long* dbg;
  unsigned long long* span;
  int silu_mul, G;
};
__device__ __forceinline__ GemmArgs xw_args(const half_t* aX, const u32x4* aQW, const half_t* aS, int aM, int aK, int aN, int a_tpg, int a_ksplit, int a_kps,
                                            int a_xcd_gm, const XwRest& rest) {
  GemmArgs a;
  a.X = aX; a.QW = aQW; a.S = aS; a.QZ = nullptr; a.bias = rest.bias; a.residual = rest.residual; a.silu_mul = rest.silu_mul; a.Y = rest.Y;
  a.slabs = rest.slabs; a.counters = rest.counters; a.M = aM; a.K = aK; a.N = aN; a.G = rest.G; a.tpg = a_tpg; a.ksplit = a_ksplit;
  a.kt_per_split = a_kps; a.xcd_gm = a_xcd_gm; a.dbg = rest.dbg; a.ln_w = nullptr; a.ln_eps = 0.f; a.span = rest.span;
  return a;
}

// In-kernel wall-clock span of a launch: first wave's start -> last wave's end on the constant 100 MHz counter.  The
// dispatch-duration clock (event pair / rocprofv3) cannot read below ~4.2 us -- an EMPTY kernel reads that -- so for the
// microsecond-scale small-M launches this is the clock that can see the kernel (quick_w4a16_gemm_span, bench.py
// roofline.frac_inkernel).  Every wave stores its own two stamps into its own slot (plain 8-byte stores: 2048 atomics on
// one word would serialise for tens of microseconds and be the thing measured); the host takes min / max.  Costs one
// scalar compare per wave when off.
constexpr unsigned kSpanWaves = 1u << 16;  // slots per launch: [kSpanWaves starts][kSpanWaves ends]
__device__ __forceinline__ void span_stamp(unsigned long long* span, int end) {
  if (span != nullptr) {  // wave-uniform branch on purpose (all 64 lanes store the same word): a lane-0 branch at kernel entry
    // made hipcc treat the buffer descriptors built after it as divergent (VGPRs, "invalid operand" in the LDS-DMA asm).
    // Only the SPAN = true instantiations contain this at all: even switched off, the branch and the longer kernarg cost
    // the M = 1 launch 7 % in an A/B session [r02].
    const unsigned w = (((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (blockDim.x >> 6) +
                        (unsigned)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) & (kSpanWaves - 1);
    span[(end ? kSpanWaves : 0u) + w] = __builtin_amdgcn_s_memrealtime();
  }
}

// ------------------------------------------------------------------------------------------------
// K split across workgroups, reduced inside the launch ("last arriver finishes the tile").
// Every slice writes its fp32 partial tile to its slab with WRITE-THROUGH (sc1) 16-byte stores, drains them
// (per-wave vmcnt(0)), and after a workgroup barrier ONE lane draws a ticket from the tile's agent-scope
// counter.  The workgroup that draws the last ticket resets the counter (the workspace is handed back zeroed)
// and adds the other slices' slabs -- read with sc1 loads, which bypass the reader's possibly stale L1 -- to the
// partial it still holds in registers.  No fences (a release would write back the whole XCD L2: 2-7 us under
// load), no spinning (so no co-residency requirement), and nothing depends on which XCD or CU a slice ran on
// (cdna_hip_programming.md G16, form R1).  Replaces the reference's fp16 `[split_k, M, N]` scratch + torch
// `.sum(0)` (csrc/gemm_cuda_quick.cu:1468, 1515).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ __amdgpu_buffer_rsrc_t slab_rsrc(float* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(base, 0, bytes, 0x00020000);
}
__device__ __forceinline__ void slab_store(__amdgpu_buffer_rsrc_t r, unsigned byte_off, floatx4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, byte_off, 0, /*sc1*/ 16);
}
__device__ __forceinline__ floatx4 slab_load(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  return __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, /*sc1*/ 16));
}
__device__ __forceinline__ bool splitk_arrive(unsigned* counter, int nslices, unsigned* lds_word) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave: its write-through stores have landed
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = t == (unsigned)nslices - 1u;
    if (last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *lds_word = last ? 1u : 0u;
  }
  __syncthreads();
  return *lds_word != 0u;
}

}  // namespace quick_amd
