// Chained small-batch GEMMs: up to kChainMax dependent W4A16 GEMMs (M <= 16 tokens) in ONE launch.
//
// Why: at batch 1 a decoder layer is four weight streams of 8..45 MB with a kernel boundary between each pair, and the
// boundary costs more than the small streams themselves -- the grid drains, the next grid is dispatched, its first HBM
// requests go out ~1 us later and come back ~2 us after that, while HBM sits idle.  Here one
// persistent grid (one 8-wave workgroup per CU, all co-resident) walks the tasks in order; between two tasks stands a
// grid barrier, and a workgroup issues the first weight chunk of its NEXT task before it waits at that barrier -- the
// weights do not depend on the previous task, only x does -- so the HBM stream runs through the boundary.
//
// The barrier is the cheap kind (cdna_hip_programming.md G16): results are stored WRITE-THROUGH (sc1), drained with
// vmcnt(0), then one lane per workgroup draws a ticket from an agent-scope counter and polls it; everything a later task
// reads of an earlier task's output (x, the residual) is read with sc1 loads.  No release / acquire fences, i.e. no L2
// write-back or invalidate (2-7 us each under load).  The counter is zero on entry and zero again on exit.
//
// Co-residency is what makes polling safe: the host launches at most one workgroup per CU (512 threads, <= 128 VGPRs,
// <= 160 KB LDS always fit an empty CU) and refuses anything else.  The poll is bounded all the same: a grid that does
// not meet within ~seconds traps (a loud queue error) instead of hanging the device.
//
// Each task is computed by exactly the code of the single launch the planner picks at M = 1 -- the table deferred-zero
// skinny flavour: x (optionally RMS-normalised) in LDS with the unit sums tabulated, weights straight from HBM into the
// MFMA A operand, the waves of a workgroup split K -- so a chain returns bit for bit what the same GEMMs return when
// launched one by one (tests/test_gemm_gpu.py::test_chain_*).
#pragma once

namespace quick_amd {

constexpr int kChainMax = 6;
struct ChainArgs {
  GemmArgs t[kChainMax];
  int n;
  unsigned* barrier;  // agent-scope arrival counter: zero on entry, zero again on exit
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t chain_rsrc(const void* base, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (unsigned)bytes, 0x00020000);
}

// Every workgroup: "my part of task `done` is in memory".  Stores were write-through; drain them, then one ticket.
__device__ __forceinline__ void chain_arrive(unsigned* counter, unsigned total_at_exit) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == total_at_exit - 1u) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__device__ __forceinline__ void chain_wait(unsigned* counter, unsigned target) {
  if (threadIdx.x == 0) {
    unsigned polls = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++polls > (1u << 23)) __builtin_trap();  // seconds: the grid is not co-resident (the host refuses such launches)
    }
  }
  __syncthreads();
}

// skinny_finish of the table deferred-zero flavour (TR, NTW = 1, K not split across workgroups) with write-through
// stores and sc1 loads of the residual: same arithmetic, same order.
template <int WAVES>
__device__ __forceinline__ void chain_finish(const GemmArgs& a, floatx4 (&acc)[1], floatx4* red, int nb, int lane, int wave) {
  const int n16 = lane & 15, q = lane >> 4;
  float* rf = (float*)(red + wave * 64) + 64 * q + n16;
#pragma unroll
  for (int r = 0; r < 4; ++r) rf[16 * r] = acc[0][r];
  acc[0] = floatx4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  constexpr int TPW = 16 / WAVES;  // tokens per wave
  const int t = min(wave * TPW + q, 15), m = t;
  if (wave * TPW >= min(16, a.M)) return;  // whole waves: the shuffle below stays wave-wide
  const float* src = (const float*)red + t * 16 + n16;
  float v = 0.f;
#pragma unroll
  for (int w = 0; w < WAVES; ++w) v += src[w * 256];
  const bool live = q < TPW && m < a.M;
  if (a.silu_mul) {
    const float up = __shfl_xor(v, 8);  // channels 0..7 gate, 8..15 up
    if (live && n16 < 8) {
      const __amdgpu_buffer_rsrc_t ry = chain_rsrc(a.Y, (size_t)a.M * (a.N >> 1) * 2);
      const half_t o = silu_mul_f16((half_t)v, (half_t)up);
      __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, o), ry,
                                            (unsigned)((m * (a.N >> 1) + nb * 8 + n16) * 2), 0, /*sc1*/ 16);
    }
    return;
  }
  if (live) {
    const int n = nb * 16 + n16;
    const unsigned off = (unsigned)((m * a.N + n) * 2);
    if (a.bias) v += (float)a.bias[n];
    if (a.residual) {
      const __amdgpu_buffer_rsrc_t rr = chain_rsrc(a.residual, (size_t)a.M * a.N * 2);
      v += (float)__builtin_bit_cast(half_t, (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rr, off, 0, /*sc1*/ 16));
    }
    const __amdgpu_buffer_rsrc_t ry = chain_rsrc(a.Y, (size_t)a.M * a.N * 2);
    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (half_t)v), ry, off, 0, /*sc1*/ 16);
  }
}

template <int GM>
__global__ __launch_bounds__(512) void w4a16_chain_kernel(const ChainArgs ca) {
  constexpr int NTW = 1, WAVES = 8, U = 4;
  constexpr int NG = groups_per_tile<GM>();
  constexpr int L = 16 / NG;  // lanes (16-byte chunks) per unit
  extern __shared__ __attribute__((aligned(16))) char smem[];
  floatx4* red = (floatx4*)smem;  // [2][WAVES][64]
  char* xlds = smem + 2 * (WAVES * 64 * sizeof(floatx4));

  const int lane = threadIdx.x & 63;
  const int wave = uniform(threadIdx.x >> 6);
  const int n16 = lane & 15, q = lane >> 4;
  const int gx = (int)gridDim.x;
  const int bx = (gx & 7) == 0 ? ((int)blockIdx.x & 7) * (gx >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  const LaneSel ls = lane_sel(n16);

  floatx4 acc[NTW];
  acc[0] = floatx4{0.f, 0.f, 0.f, 0.f};
  SkinnyChunk<NTW, GM, U, true> cA, cB;
  int parity = 0;

  // state of the task in hand (wave-uniform)
  GemmArgs a = ca.t[0];
  SkinnyBufs bufs;
  int nblocks, KT, kt_begin, kt_end, kt_last;
  int nb_cur, kt_cur, nb_nxt, kt_nxt;
#define QA_CHAIN_SETUP()                                                                                           \
  do {                                                                                                             \
    bufs = skinny_bufs(a, lane);                                                                                   \
    nblocks = a.N / 16;                                                                                            \
    KT = a.K >> 7;                                                                                                 \
    kt_begin = KT * wave / WAVES;                                                                                  \
    kt_end = KT * (wave + 1) / WAVES;                                                                              \
    kt_last = max(kt_end - 1, kt_begin);                                                                           \
    nb_cur = bx;                                                                                                   \
    kt_cur = kt_begin;                                                                                             \
    nb_nxt = nb_cur;                                                                                               \
    kt_nxt = kt_cur;                                                                                               \
  } while (0)
#define QA_CHAIN_LOAD(c) skinny_load<NTW, GM, U, true, false>(c, kt_nxt, kt_last, bufs, nb_nxt * NTW, nullptr, a)
#define QA_CHAIN_ADVANCE(nb, kt)                                                                                   \
  do {                                                                                                             \
    kt += U;                                                                                                       \
    if (kt >= kt_end) {                                                                                            \
      kt = kt_begin;                                                                                               \
      nb += gx;                                                                                                    \
    }                                                                                                              \
  } while (0)
#define QA_CHAIN_COMPUTE(ccomp)                                                                                    \
  skinny_compute_dz<NTW, GM, U>(ccomp, kt_cur, kt_end, xl, tab, ls, acc);                                          \
  if (kt_cur + U >= kt_end) {                                                                                      \
    if (nb_cur < nblocks) { /* a workgroup beyond the task's channel blocks computed zeros: nothing to store */    \
      chain_finish<WAVES>(a, acc, red + parity * (WAVES * NTW * 64), nb_cur, lane, wave);                          \
      parity = 1 - parity;                                                                                         \
    }                                                                                                              \
  }
#define QA_CHAIN_STEP(cload, ccomp)                                                                                \
  if (nb_nxt >= nblocks) { /* the chunk in hand is this workgroup's last of the task */                            \
    QA_CHAIN_COMPUTE(ccomp);                                                                                       \
    break;                                                                                                         \
  }                                                                                                                \
  QA_CHAIN_LOAD(cload);                                                                                            \
  __builtin_amdgcn_sched_barrier(0);                                                                               \
  QA_CHAIN_COMPUTE(ccomp);                                                                                         \
  nb_cur = nb_nxt;                                                                                                 \
  kt_cur = kt_nxt;                                                                                                 \
  QA_CHAIN_ADVANCE(nb_nxt, kt_nxt)

  QA_CHAIN_SETUP();
  QA_CHAIN_LOAD(cA);  // HBM requests first
  QA_CHAIN_ADVANCE(nb_nxt, kt_nxt);
  __builtin_amdgcn_sched_barrier(0);

  for (int t = 0;;) {
    // ---- x[rows, K] of this task -> LDS [rows][pitch] (+ RMSNorm), unit sums tabulated on the way.  sc1 loads: for
    // t > 0 the rows were written during this launch by other workgroups, maybe on other XCDs.
    const int rows = min(16, a.M);
    const int kc = KT * 16;  // 16-byte chunks per row
    const int pitch = KT * 256 + 16;
    const char* xl = xlds + min(n16, rows - 1) * pitch + q * 16;
    float* tab0 = (float*)(xlds + rows * pitch);
    const float* tab = tab0 + 4 * q;
    const __amdgpu_buffer_rsrc_t rx = chain_rsrc(a.X, (size_t)a.M * a.K * 2);
    if (a.ln_w) {  // pass 1 copies x raw and sums its squares per row; pass 2 finds x in LDS
      float* ssq = (float*)smem;  // [rows][WAVES], in the still unused reduction buffer (parity side irrelevant: rewritten later)
      for (int r = 0; r < rows; ++r) {
        float ss = 0.f;
        for (int c = threadIdx.x; c < kc; c += WAVES * 64) {
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rx, (unsigned)(r * a.K * 2 + c * 16), 0, /*sc1*/ 16);
          *(u32x4*)(xlds + r * pitch + c * 16) = v;
#pragma unroll
          for (int i = 0; i < 4; ++i) ss = __builtin_amdgcn_fdot2(as_h2(v[i]), as_h2(v[i]), ss, false);
        }
        ss = wave_sum(ss);
        if (lane == 0) ssq[r * WAVES + wave] = ss;
      }
      __syncthreads();
    }
    for (int r = 0; r < rows; ++r) {
      float inv = 0.f;
      if (a.ln_w) {
        float ss = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) ss += ((const float*)smem)[r * WAVES + w];
        inv = rsqrtf(ss / (float)a.K + a.ln_eps);
      }
      for (int c = threadIdx.x; c < kc; c += WAVES * 64) {  // kc % 16 == 0: rows of 16 lanes are all in or all out
        u32x4 v;
        if (a.ln_w) {  // fp16(fp16(x * inv) * weight): the rounding points of quick_rmsnorm_f16 (and of torch)
          const half8_t xv = *(const half8_t*)(xlds + r * pitch + c * 16), gv = *(const half8_t*)(a.ln_w + c * 8);
          half8_t o;
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = (half_t)((half_t)((float)xv[j] * inv) * gv[j]);
          v = __builtin_bit_cast(u32x4, o);
        } else {
          v = __builtin_amdgcn_raw_buffer_load_b128(rx, (unsigned)(r * a.K * 2 + c * 16), 0, /*sc1*/ 16);
        }
        *(u32x4*)(xlds + r * pitch + c * 16) = v;
        const half2_t one2 = {(half_t)1.f, (half_t)1.f};
        const float lo = __builtin_amdgcn_fdot2(as_h2(v[0]), one2, __builtin_amdgcn_fdot2(as_h2(v[2]), one2, 0.f, false), false);
        const float hi = __builtin_amdgcn_fdot2(as_h2(v[1]), one2, __builtin_amdgcn_fdot2(as_h2(v[3]), one2, 0.f, false), false);
        const float sa = lanes_sum<L>(lo + hi);
        const float sc = lanes_sum<L>(1024.f * lo + 64.f * hi);
        if ((lane & (L - 1)) == 0) {
          float* tp = tab0 + (c / L) * 32 + r;
          tp[0] = sa;
          tp[16] = -sc;
        }
      }
    }
    __syncthreads();

    while (true) {
      QA_CHAIN_STEP(cB, cA);
      QA_CHAIN_STEP(cA, cB);
    }
    acc[0] = floatx4{0.f, 0.f, 0.f, 0.f};  // (a workgroup beyond the task's channel blocks leaves without a finish)

    ++t;
    chain_arrive(ca.barrier, (unsigned)(ca.n * gx));
    if (t >= ca.n) break;
    a = ca.t[t];
    QA_CHAIN_SETUP();
    QA_CHAIN_LOAD(cA);  // the next task's first weight chunk goes out before the wait: weights do not depend on task t-1
    QA_CHAIN_ADVANCE(nb_nxt, kt_nxt);
    __builtin_amdgcn_sched_barrier(0);
    chain_wait(ca.barrier, (unsigned)(t * gx));
  }
#undef QA_CHAIN_STEP
#undef QA_CHAIN_COMPUTE
#undef QA_CHAIN_ADVANCE
#undef QA_CHAIN_LOAD
#undef QA_CHAIN_SETUP
}

}  // namespace quick_amd
