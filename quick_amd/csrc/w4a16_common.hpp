// Shared device helpers for the gfx950 W4A16 kernels.  CDNA4 only: 64-lane wavefronts,
// v_mfma_f32_16x16x32_f16, packed-f16 VALU.  No other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

namespace quick_amd {
// Raise a kernel's dynamic-LDS limit once per (kernel, DEVICE): `done` is the call site's own static word, one bit per device ordinal
// (the attribute is per device; a plain process-wide flag left a second GPU's first launch without it -- ADVICE r04).  Concurrent first
// launches may both set it (idempotent).  A refusal is not cached: it stays in hipGetLastError() for the launch check behind the
// launch (QUICK_ERR_LAUNCH) and the next call tries again.
inline bool lds_limit_once(std::atomic<unsigned long long>& done, const void* kfn, int bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  const unsigned long long bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return true;
  if (hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
  done.fetch_or(bit, std::memory_order_release);
  return true;
}
}  // namespace quick_amd

namespace quick_amd {

typedef _Float16 half_t;
typedef half_t half2_t __attribute__((ext_vector_type(2)));
typedef half_t half4_t __attribute__((ext_vector_type(4)));
typedef half_t half8_t __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define QA_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define QA_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ half2_t as_h2(uint32_t u) { return __builtin_bit_cast(half2_t, u); }
__device__ __forceinline__ uint32_t as_u32(half2_t h) { return __builtin_bit_cast(uint32_t, h); }

// (a & mask) | orv in ONE VALU op.  gfx950 VOP3 takes no literals, so hipcc splits the C expression into v_and + v_or (two literal-carrying VOP2s)
// when it can see the constants; with the mask in an SGPR and the magic number in a VGPR whose values it cannot see it selects v_and_or_b32 -- the AMD
// counterpart of the `lop3` in the reference's dequantize_s4_to_fp16x2_fused (csrc/dequantize_quick.cuh:37-51).
// [r06] r01-r05 wrote the instruction itself as inline asm.  An instruction inside an asm statement is invisible to hipcc's hazard recognizer: gfx950
// does not interlock the matrix core against the vector ALU (measured table: tools/mfma_hazard_lint.py, profiles/r06_mfma_hazards.txt), hipcc pads
// those pairs only for instructions it knows to be VALU, and a build whose register allocation put such an asm next to the wrong MFMA computed tiles
// wrong (the unconditional-request chunk loop of r05, DESIGN.md 9.6: cured by this form, by a trailing s_nop inside the asm, and by
// -amdgpu-waitcnt-forcezero; not by any wait on memory).  Now the two EMPTY asm statements only hide the constants' values; the instruction is hipcc's
// own, with its hazards seen.  QA_ANDOR_ASM=1 (A/B builds) restores the asm form.
__device__ __forceinline__ uint32_t and_or(uint32_t a, uint32_t mask, uint32_t orv) {
#if defined(QA_ANDOR_ASM) && QA_ANDOR_ASM == 1
  uint32_t r;
  asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(mask), "v"(orv));
  return r;
#elif defined(QA_ANDOR_ASM) && QA_ANDOR_ASM == 2   // (... with a wait state behind it: cures the r05 build; ... == 3, in front of it: does not)
  uint32_t r;
  asm("v_and_or_b32 %0, %1, %2, %3\n\ts_nop 0" : "=v"(r) : "v"(a), "s"(mask), "v"(orv));
  return r;
#elif defined(QA_ANDOR_ASM) && QA_ANDOR_ASM == 3
  uint32_t r;
  asm("s_nop 0\n\tv_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(mask), "v"(orv));
  return r;
#else
  uint32_t m = mask, o = orv;
  asm("" : "+s"(m));
  asm("" : "+v"(o));
  return (a & m) | o;
#endif
}

// Per-(group, output channel) dequantisation constants held by a lane.
struct GroupQ {
  half2_t s2;    // (s, s)
  half2_t nzlo;  // -(1024 + z) twice: subtracts the magic bias and the zero point exactly
  half2_t nzhi;  // -(64 + z) twice: same for the nibbles read 16x too large
};

__device__ __forceinline__ GroupQ make_group(half_t s, uint32_t z /*0..15*/) {
  GroupQ g;
  g.s2 = half2_t{s, s};
  const uint32_t zz = z | (z << 16);
  g.nzlo = as_h2(0xE400E400u | zz);         // fp16 bits of -(1024 + z): 0xE400 | z
  g.nzhi = as_h2(0xD400D400u | (zz << 4));  // fp16 bits of -(64 + z):   0xD400 | z << 4
  return g;
}

// One packed dword (8 weights of one output channel, 8 consecutive k, MI355X nibble order
// p = 4*(j%2) + j/2) -> the v_mfma_f32_16x16x32_f16 A-operand fragment of this lane.
// Every weight is fp16((w - z) * s): (1024+w) - (1024+z) and (1024+16w)/16 - (64+z) are exact in
// fp16, the multiply by s rounds once -- bit-identical to the reference's sub.f16x2 + mul.rn.f16x2
// (csrc/gemm_cuda_quick.cu:52-60).  13 VALU ops: 1 shift, 4 and_or, 2 pk_add, 2 pk_fma, 4 pk_mul.
__device__ __forceinline__ half8_t dequant8(uint32_t q, const GroupQ& g) {
  const uint32_t magic = 0x64006400u;  // fp16 1024.0 twice
  const half2_t sixteenth = {(half_t)0.0625f, (half_t)0.0625f};
  const uint32_t q8 = q >> 8;
  const half2_t h0 = (as_h2(and_or(q, 0x000f000fu, magic)) + g.nzlo) * g.s2;               // k0, k1
  const half2_t h1 = (as_h2(and_or(q, 0x00f000f0u, magic)) * sixteenth + g.nzhi) * g.s2;   // k2, k3
  const half2_t h2 = (as_h2(and_or(q8, 0x000f000fu, magic)) + g.nzlo) * g.s2;              // k4, k5
  const half2_t h3 = (as_h2(and_or(q8, 0x00f000f0u, magic)) * sixteenth + g.nzhi) * g.s2;  // k6, k7
  half8_t r;
  r[0] = h0[0]; r[1] = h0[1]; r[2] = h1[0]; r[3] = h1[1];
  r[4] = h2[0]; r[5] = h2[1]; r[6] = h3[0]; r[7] = h3[1];
  return r;
}

// One packed dword -> eight fp16 values, the weight still carrying its bias: (1024 + w) for k % 8 in {0,1,4,5}, where
// the nibble lands in the mantissa of 1024.0 (ulp 1), and (64 + w) for k % 8 in {2,3,6,7}, where it lands four bits
// higher in the mantissa of 64.0 (ulp 1/16).  Exact, 5 VALU ops; the deferred-zero kernel feeds this to the MFMA and
// removes bias, zero point and scale once per group in fp32 (see w4a16_gemm.hip).
__device__ __forceinline__ half8_t biased8(uint32_t q) {
  const uint32_t m1024 = 0x64006400u, m64 = 0x54005400u;
  const uint32_t q8 = q >> 8;
  const u32x4 r = {and_or(q, 0x000f000fu, m1024), and_or(q, 0x00f000f0u, m64), and_or(q8, 0x000f000fu, m1024),
                   and_or(q8, 0x00f000f0u, m64)};
  return __builtin_bit_cast(half8_t, r);
}

// v + (v of the lane the DPP control selects); CTRL: 0xB1 = quad_perm[1,0,3,2], 0x4E = quad_perm[2,3,0,1],
// 0x141 = row_half_mirror, 0x140 = row_mirror
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// sum over aligned groups of L = 4, 8 or 16 lanes, delivered to every lane of the group
template <int L>
__device__ __forceinline__ float lanes_sum(float v) {
  v = dpp_add<0xB1>(v);
  v = dpp_add<0x4E>(v);
  if constexpr (L >= 8) v = dpp_add<0x141>(v);
  if constexpr (L >= 16) v = dpp_add<0x140>(v);
  return v;
}

// Group constants as they come out of memory: ONE 32-bit word per (group, channel) -- fp16 scale in the low half, zero
// point (0..15) in the high half -- that a lane fetches for its own channel with a plain dword load.  make_group() is
// applied at the point of use so that a prefetch of the next k-tile's constants carries no dependent ALU work (which would
// make the compiler wait for the whole in-order load queue right after issuing it).
struct GroupRaw {
  uint32_t sz;
};

// word index of (g, n) in the scales tensor viewed as uint32 [NG * N]: the words of a 16-channel block are contiguous
// over all groups, so a wave walking along K for fixed channels streams them (64 B per group) next to its weights
__host__ __device__ __forceinline__ size_t group_word_index(int g, int n, int NG) {
  return ((size_t)(n >> 4) * NG + g) * 16 + (n & 15);
}


// group index of k-step t of 128-k tile kt.  GM: 0 -> G == 128, 1 -> G % 128 == 0 (tpg = G / 128),
// 2 -> G == 64, 3 -> G == 32, 4 -> any other multiple of 32 (runtime division).
template <int GM>
__device__ __forceinline__ int group_index(int kt, int t, int tpg, int G) {
  if constexpr (GM == 0) return kt;
  else if constexpr (GM == 1) return kt / tpg;
  else if constexpr (GM == 2) return kt * 2 + (t >> 1);
  else if constexpr (GM == 3) return kt * 4 + t;
  else return (kt * 128 + 32 * t) / G;
}
// distinct groups touched by one 128-k tile
template <int GM>
constexpr int groups_per_tile() { return GM <= 1 ? 1 : (GM == 2 ? 2 : 4); }
template <int GM>
__device__ __forceinline__ int group_slot(int t) { return GM <= 1 ? 0 : (GM == 2 ? (t >> 1) : t); }

__device__ __forceinline__ GroupRaw load_group_raw(const half_t* __restrict__ S, const uint32_t* __restrict__ /*QZ*/,
                                                   int g, int n, int N, int NG) {
  (void)N;
  return GroupRaw{((const uint32_t*)S)[group_word_index(g, n, NG)]};
}
__device__ __forceinline__ float group_scale_f32(const GroupRaw& r) { return (float)as_h2(r.sz)[0]; }
__device__ __forceinline__ float group_zero_f32(const GroupRaw& r) { return (float)(r.sz >> 16); }
// 4 VALU: v_perm (scale pair), v_lshrrev + v_lshl_or (zero point twice), 2 v_or (bias constants) -- minus what hipcc folds
__device__ __forceinline__ GroupQ make_group(const GroupRaw& r) {
  GroupQ g;
  g.s2 = as_h2(__builtin_amdgcn_perm(r.sz, r.sz, 0x01000100u));
  const uint32_t z = r.sz >> 16;
  const uint32_t zz = z | (z << 16);
  g.nzlo = as_h2(0xE400E400u | zz);
  g.nzhi = as_h2(0xD400D400u | (zz << 4));
  return g;
}

__device__ __forceinline__ floatx4 mfma16(half8_t a, half8_t b, floatx4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
// silu(g) * u with torch's fp16 rounding points: silu rounded to fp16, product rounded to fp16
__device__ __forceinline__ half_t silu_mul_f16(half_t g, half_t u) {
  const float x = (float)g;
  return (half_t)((float)(half_t)(x / (1.f + __expf(-x))) * (float)u);
}

// wave-uniform values the compiler cannot prove uniform (anything derived from threadIdx)
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

}  // namespace quick_amd
