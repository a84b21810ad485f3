// Chained small-batch GEMMs: up to kChainMax dependent W4A16 GEMMs (M <= 16 tokens) in ONE launch.
//
// Why: at batch 1 a decoder layer is four weight streams of 8..45 MB with a kernel boundary between each pair, and the
// boundary costs more than the small streams themselves -- the grid drains, the next grid is dispatched, its first HBM
// requests go out ~1 us later and come back ~2 us after that, while HBM sits idle.  Here one
// persistent grid (one 8-wave workgroup per CU, all co-resident) walks the tasks in order, and a workgroup requests the
// first weight chunk of its NEXT task before it looks for that task's x -- the weights do not depend on the previous task,
// only x does -- so the HBM stream runs through the boundary.
//
// How a task finds its x: task t + 1 reads the y of task t, all of it, from all workgroups.  A grid barrier built from an
// arrival counter costs four dependent trips through the memory system per boundary (stores acknowledged -> ticket ->
// poll -> read x: ~3.5 us measured, tools/chain_trace.py) -- no better than the kernel boundary it replaces.  Instead the
// DATA CARRIES ITS OWN FLAG: besides the ordinary y (for whoever runs after the launch), the workgroup that finishes a
// 16-channel block of a row writes a CELL into a scratch area -- 16 bytes holding a tag, then the block's fp16 values --
// with ONE write-through store instruction covering <= 64 contiguous bytes of one 64-byte line: one memory transaction,
// visible all or nothing.  A consumer stages x by reading the cells with sc1 loads (one instruction fetches a cell's tag
// and data together) and repeats until every tag is the one this launch writes for that task: one one-way trip + one
// round trip per boundary.  Tags are (launch epoch + 1 + task), the epoch lives in the scratch header and moves on by 8
// per launch, so a cell left by an earlier launch never matches.  Residuals that are the y of an earlier task are read
// from its cells the same way.  Nothing in the chain reads an ordinary y that the chain itself wrote.
//
// Co-residency is what makes polling safe: the host launches at most one workgroup per CU (512 threads, <= 128 VGPRs,
// <= 160 KB LDS always fit an empty CU) and refuses anything else.  The poll is bounded all the same: cells that do not
// appear within ~seconds make the kernel trap (a loud queue error) instead of hanging the device.
//
// Each task is computed by exactly the code of the single launch the planner picks at M = 1 -- the table deferred-zero
// skinny flavour: x (optionally RMS-normalised) in LDS with the unit sums tabulated, weights straight from HBM into the
// MFMA A operand, the waves of a workgroup split K -- and the cells hold exactly the fp16 values of y, so a chain returns
// bit for bit what the same GEMMs return when launched one by one (tests/test_gemm_gpu.py::test_chain_*).
#pragma once

namespace quick_amd {

constexpr int kChainMax = 6;
constexpr unsigned kChainHeaderBytes = 256;  // scratch: [epoch word, padding][cells of task 0][cells of task 1]...
struct ChainLink {
  int x_src;           // -1: x is an ordinary tensor (written before the launch); else the earlier task whose cells hold it
  int res_src;         // the same for the residual
  unsigned cells_off;  // byte offset of this task's cells in the scratch; row m, block nb at + (m * (N / 16) + nb) * cell bytes
  unsigned x_cells_off, x_cell_bytes;  // x_src >= 0: where that task's cells are, and their size (64, or 32 behind SiLU*mul)
  unsigned res_cells_off;              // res_src >= 0: where that task's cells are (always 64-byte cells)
};
struct ChainArgs {
  GemmArgs t[kChainMax];
  ChainLink link[kChainMax];
  int n;
  unsigned* exits;  // agent-scope exit counter: zero on entry, zero again on exit (the last workgroup out moves the epoch on)
  char* scratch;
  unsigned long long* trace;  // TRACE builds: [workgroup][kChainMax][8] s_memrealtime stamps (tools/chain_trace.py), else unused
};

// A cell: piece 0 (16 bytes) = {tag, -, -, -}, then the block's fp16 values: 16 channels (32 bytes) in a 64-byte cell -- or,
// behind a SiLU*mul epilogue, the 8 outputs the block's 8 gate + 8 up channels make (16 bytes), in a 32-byte cell.

__device__ __forceinline__ __amdgpu_buffer_rsrc_t chain_rsrc(const void* base, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (unsigned)bytes, 0x00020000);
}

// Residual of (row, channel) for a block this wave will finish later, REQUESTED early so that the trip is hidden and
// consumed in chain_finish.  From an ordinary tensor: a plain load.  From an earlier task's cells: tag, then data, two sc1
// loads of the same line in this order (same wave, same line: served in order); the tag is checked at the point of use --
// it is right, that task finished before the one whose cells this workgroup has already staged, and the check proves it
// (a wrong tag falls back to polling).
struct ChainRes {
  unsigned tag;
  unsigned short bits;
};
template <int WAVES>
__device__ __forceinline__ ChainRes chain_residual_issue(const GemmArgs& a, int nb, int nblocks, int lane, int wave, int res_src,
                                                        __amdgpu_buffer_rsrc_t rcells) {
  constexpr int TPW = 16 / WAVES;
  const int n16 = lane & 15, q = lane >> 4;
  const int m = min(wave * TPW + q, 15);
  ChainRes r{0u, 0};
  if (a.residual == nullptr || a.silu_mul || nb >= nblocks || q >= TPW || m >= a.M) return r;
  if (res_src < 0) {
    r.bits = __builtin_bit_cast(unsigned short, a.residual[(size_t)m * a.N + nb * 16 + n16]);
    return r;
  }
  const unsigned cell = ((unsigned)m * ((unsigned)a.N >> 4) + (unsigned)nb) * 64u;
  r.tag = __builtin_amdgcn_raw_buffer_load_b32(rcells, cell, 0, /*sc1*/ 16);
  r.bits = __builtin_amdgcn_raw_buffer_load_b16(rcells, cell + 16u + (unsigned)n16 * 2u, 0, /*sc1*/ 16);
  return r;
}
template <int WAVES>
__device__ __forceinline__ float chain_residual_value(ChainRes r, const GemmArgs& a, int nb, int nblocks, int lane, int wave,
                                                      int res_src, __amdgpu_buffer_rsrc_t rcells, unsigned rtag) {
  constexpr int TPW = 16 / WAVES;
  const int n16 = lane & 15, q = lane >> 4;
  const int m = min(wave * TPW + q, 15);
  if (a.residual == nullptr || a.silu_mul || nb >= nblocks || q >= TPW || m >= a.M) return 0.f;
  if (res_src >= 0) {
    const unsigned cell = ((unsigned)m * ((unsigned)a.N >> 4) + (unsigned)nb) * 64u;
    unsigned polls = 0;
    while (r.tag != rtag) {
      __builtin_amdgcn_s_sleep(2);
      r.tag = __builtin_amdgcn_raw_buffer_load_b32(rcells, cell, 0, /*sc1*/ 16);
      r.bits = __builtin_amdgcn_raw_buffer_load_b16(rcells, cell + 16u + (unsigned)n16 * 2u, 0, /*sc1*/ 16);
      if (++polls > (1u << 22)) __builtin_trap();
    }
  }
  return (float)__builtin_bit_cast(half_t, r.bits);
}

// skinny_finish of the table deferred-zero flavour (TR, NTW = 1, K not split across workgroups), plus the cell.  `rv`:
// this lane's residual, fetched by the caller (no load in here: a load consumed on the spot would make the wave wait for
// the weight chunk it has just requested as well -- loads return in order).
template <int WAVES>
__device__ __forceinline__ void chain_finish(const GemmArgs& a, floatx4 (&acc)[1], floatx4* red, int nb, int lane, int wave,
                                             __amdgpu_buffer_rsrc_t cells, unsigned tag, float rv) {
  const int n16 = lane & 15, q = lane >> 4;
  float* rf = (float*)(red + wave * 64) + 64 * q + n16;
#pragma unroll
  for (int r = 0; r < 4; ++r) rf[16 * r] = acc[0][r];
  acc[0] = floatx4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  constexpr int TPW = 16 / WAVES;  // tokens per wave
  const int t = min(wave * TPW + q, 15), m = t;
  if (wave * TPW >= min(16, a.M)) return;  // whole waves: the shuffles below stay wave-wide
  const float* src = (const float*)red + t * 16 + n16;
  float v = 0.f;
#pragma unroll
  for (int w = 0; w < WAVES; ++w) v += src[w * 256];
  const bool live = q < TPW && m < a.M;
  const unsigned nblk = (unsigned)a.N >> 4;
  half_t o;
  unsigned cell, ndata;  // byte offset of the cell, data dwords in it
  if (a.silu_mul) {
    const float up = __shfl_xor(v, 8);  // channels 0..7 gate, 8..15 up
    o = silu_mul_f16((half_t)v, (half_t)up);
    if (live && n16 < 8) a.Y[(size_t)m * (a.N >> 1) + nb * 8 + n16] = o;
    cell = ((unsigned)m * nblk + (unsigned)nb) * 32u;
    ndata = 4;
  } else {
    if (live) {
      const int n = nb * 16 + n16;
      if (a.bias) v += (float)a.bias[n];
      if (a.residual) v += rv;
    }
    o = (half_t)v;
    if (live) a.Y[(size_t)m * a.N + nb * 16 + n16] = o;
    cell = ((unsigned)m * nblk + (unsigned)nb) * 64u;
    ndata = 8;
  }
  // the cell: even lanes hold two neighbouring channels, lane 1 of the 16 the tag -- one store instruction, one 64-byte line
  const unsigned mine = (unsigned)__builtin_bit_cast(unsigned short, o);
  const unsigned pair = mine | ((unsigned)__shfl_xor((int)mine, 1) << 16);
  const bool is_tag = n16 == 1, is_data = (n16 & 1) == 0 && (unsigned)(n16 >> 1) < ndata;
  if (live && (is_tag || is_data))
    __builtin_amdgcn_raw_buffer_store_b32(is_tag ? tag : pair, cells, cell + (is_tag ? 0u : 16u + (unsigned)(n16 >> 1) * 4u), 0, /*sc1*/ 16);
}

// The launch's arguments, copied from the kernarg segment into LDS once (one vector load per thread at kernel start): a
// task's GemmArgs fetched from there costs ~0.1 us, the scalar loads from the kernarg segment that hipcc sinks to the first
// use cost a trip to memory per 64-byte line -- 1-2 us in the middle of a task boundary (tools/chain_trace.py).
constexpr int kChainArgWords = (int)(sizeof(GemmArgs) / 4), kChainLinkWords = (int)(sizeof(ChainLink) / 4);
static_assert(sizeof(GemmArgs) % 4 == 0 && sizeof(ChainLink) % 4 == 0 && sizeof(ChainArgs) <= 2048, "LDS copy of the arguments");
__device__ __forceinline__ void chain_task_from_lds(const unsigned* words, int t, GemmArgs& g, ChainLink& l) {
  unsigned w[kChainArgWords], x[kChainLinkWords];
  const unsigned* gp = words + t * kChainArgWords;
  const unsigned* lp = words + (offsetof(ChainArgs, link) / 4) + t * kChainLinkWords;
#pragma unroll
  for (int i = 0; i < kChainArgWords; ++i) w[i] = (unsigned)__builtin_amdgcn_readfirstlane((int)gp[i]);
#pragma unroll
  for (int i = 0; i < kChainLinkWords; ++i) x[i] = (unsigned)__builtin_amdgcn_readfirstlane((int)lp[i]);
  __builtin_memcpy(&g, w, sizeof(GemmArgs));
  __builtin_memcpy(&l, x, sizeof(ChainLink));
}

template <int GM, bool TRACE = false>
__global__ __launch_bounds__(512) void w4a16_chain_kernel(const ChainArgs ca) {
#define QA_CHAIN_STAMP(task, i)                                                                                    \
  do {                                                                                                             \
    if constexpr (TRACE) {                                                                                         \
      if (threadIdx.x == 0) ca.trace[((size_t)blockIdx.x * kChainMax + (task)) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    }                                                                                                              \
  } while (0)
  constexpr int NTW = 1, WAVES = 8, U = 4;
  constexpr int NG = groups_per_tile<GM>();
  constexpr int L = 16 / NG;  // lanes (16-byte chunks) per unit
  constexpr int GPRE = 2;     // RMSNorm weight chunks per thread requested ahead (K <= 8192 has no more)
  extern __shared__ __attribute__((aligned(16))) char smem_all[];
  unsigned* arg_words = (unsigned*)smem_all;  // [2 KiB] copy of the kernel arguments
  char* smem = smem_all + 2048;
  floatx4* red = (floatx4*)smem;  // [2][WAVES][64]
  char* xlds = smem + 2 * (WAVES * 64 * sizeof(floatx4));
  if (threadIdx.x < sizeof(ChainArgs) / 4)
    arg_words[threadIdx.x] = ((const unsigned*)__builtin_amdgcn_kernarg_segment_ptr())[threadIdx.x];

  const int lane = threadIdx.x & 63;
  const int wave = uniform(threadIdx.x >> 6);
  const int n16 = lane & 15, q = lane >> 4;
  const int gx = (int)gridDim.x;
  const int bx = (gx & 7) == 0 ? ((int)blockIdx.x & 7) * (gx >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  const LaneSel ls = lane_sel(n16);
  const unsigned epoch = (unsigned)__builtin_amdgcn_readfirstlane((int)*(const volatile unsigned*)ca.scratch);

  floatx4 acc[NTW];
  acc[0] = floatx4{0.f, 0.f, 0.f, 0.f};
  SkinnyChunk<NTW, GM, U, true> cA, cB, cP;  // two sets alternate within a task; cP receives the next task's first chunk
  int parity = 0;

  // The task in hand and the one after it (all wave-uniform).  The NEXT task's arguments are fetched while this one runs
  // (a kernarg fetch is a trip to memory too), and its first weight chunk and RMSNorm weights are requested before this
  // task's last chunk is computed: by the time the workgroup looks for the next x they have landed.
  GemmArgs a = ca.t[0], an = ca.t[0];
  ChainLink lk = ca.link[0], lkn = ca.link[0];
  SkinnyBufs bufs, bufs_n;
  int nblocks, KT, kt_begin, kt_end, kt_last;
  int ktb_n, kte_n, ktl_n;
  int nb_cur, kt_cur, nb_nxt, kt_nxt;
  half8_t gpre[GPRE];
#define QA_CHAIN_NEXT_CONTEXT(args)                                                                                \
  do {                                                                                                             \
    bufs_n = skinny_bufs(args, lane);                                                                              \
    bufs_n.wstride_bytes = (unsigned)uniform((int)bufs_n.wstride_bytes);                                           \
    bufs_n.sstride_bytes = (unsigned)uniform((int)bufs_n.sstride_bytes);                                           \
    const int ktn = uniform((args).K >> 7);                                                                        \
    ktb_n = uniform(ktn * wave / WAVES);                                                                           \
    kte_n = uniform(ktn * (wave + 1) / WAVES);                                                                     \
    ktl_n = uniform(max(kte_n - 1, ktb_n));                                                                        \
  } while (0)
#define QA_CHAIN_PREFETCH(c, args)                                                                                 \
  do {                                                                                                             \
    skinny_load<NTW, GM, U, true, false>(c, ktb_n, ktl_n, bufs_n, bx * NTW, nullptr, args);                        \
    if ((args).ln_w) {                                                                                             \
      _Pragma("unroll") for (int i = 0; i < GPRE; ++i)                                                             \
        gpre[i] = *(const half8_t*)((args).ln_w + min((int)threadIdx.x + i * WAVES * 64, ((args).K >> 3) - 1) * 8); \
    }                                                                                                              \
  } while (0)
  // uniform(): these are wave-uniform by construction, but carried around the task loop hipcc moves some of them into
  // VGPRs and then wraps every buffer load that takes them as its scalar offset in a readfirstlane loop
#define QA_CHAIN_ENTER()  /* the prefetched task becomes the task in hand; its first chunk is already in flight */  \
  do {                                                                                                             \
    bufs = bufs_n;                                                                                                 \
    bufs.wstride_bytes = (unsigned)uniform((int)bufs.wstride_bytes);                                               \
    bufs.sstride_bytes = (unsigned)uniform((int)bufs.sstride_bytes);                                               \
    nblocks = uniform(a.N / 16);                                                                                   \
    KT = uniform(a.K >> 7);                                                                                        \
    kt_begin = uniform(ktb_n);                                                                                     \
    kt_end = uniform(kte_n);                                                                                       \
    kt_last = uniform(ktl_n);                                                                                      \
    nb_cur = bx;                                                                                                   \
    kt_cur = kt_begin;                                                                                             \
    nb_nxt = bx;                                                                                                   \
    kt_nxt = kt_begin;                                                                                             \
    QA_CHAIN_ADVANCE(nb_nxt, kt_nxt);                                                                              \
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
#define QA_CHAIN_RESIDUAL(nb) chain_residual_issue<WAVES>(a, nb, nblocks, lane, wave, lk.res_src, scr_res)
#define QA_CHAIN_RESIDUAL_VALUE(r, nb) chain_residual_value<WAVES>(r, a, nb, nblocks, lane, wave, lk.res_src, scr_res, res_tag)
#define QA_CHAIN_COMPUTE(ccomp)                                                                                    \
  skinny_compute_dz<NTW, GM, U>(ccomp, kt_cur, kt_end, xl, tab, ls, acc);                                          \
  if (kt_cur + U >= kt_end) {                                                                                      \
    QA_CHAIN_STAMP(t, 7); /* (last write wins: the last block's MFMAs are done) */                                 \
    if (nb_cur < nblocks) { /* a workgroup beyond the task's channel blocks computed zeros: nothing to store */    \
      /* residuals of this workgroup's (at most two: the host checks) blocks sit in registers -- a load in here, even on a \
         path never taken, makes hipcc drain the load queue, i.e. wait for the chunk just requested, at every finish */ \
      const float rnow = nb_cur == bx ? rvf0 : rvf1;                                                               \
      chain_finish<WAVES>(a, acc, red + parity * (WAVES * NTW * 64), nb_cur, lane, wave, scr_out, out_tag, rnow);  \
      parity = 1 - parity;                                                                                         \
    }                                                                                                              \
  }
#define QA_CHAIN_STEP(cload, ccomp)                                                                                \
  nb_nxt = uniform(nb_nxt);                                                                                        \
  kt_nxt = uniform(kt_nxt);                                                                                        \
  nb_cur = uniform(nb_cur);                                                                                        \
  kt_cur = uniform(kt_cur);                                                                                        \
  if (nb_nxt >= nblocks) { /* the chunk in hand is this workgroup's last of the task: request the next task's first */ \
    if (has_next) QA_CHAIN_PREFETCH(cP, an);                                                                       \
    QA_CHAIN_STAMP(t, 6);                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    QA_CHAIN_COMPUTE(ccomp);                                                                                       \
    break;                                                                                                         \
  }                                                                                                                \
  QA_CHAIN_LOAD(cload);                                                                                            \
  __builtin_amdgcn_sched_barrier(0);                                                                               \
  QA_CHAIN_COMPUTE(ccomp);                                                                                         \
  nb_cur = nb_nxt;                                                                                                 \
  kt_cur = kt_nxt;                                                                                                 \
  QA_CHAIN_ADVANCE(nb_nxt, kt_nxt)

  QA_CHAIN_NEXT_CONTEXT(a);
  QA_CHAIN_PREFETCH(cP, a);  // HBM requests first
  QA_CHAIN_ENTER();
  __builtin_amdgcn_sched_barrier(0);

  for (int t = 0;;) {
    QA_CHAIN_STAMP(t, 0);  // task begins: its first weights are on their way
    const bool has_next = t + 1 < ca.n;
    const int rows = min(16, a.M);
    const int kc = KT * 16;  // 16-byte chunks per row
    const int pitch = KT * 256 + 16;
    const char* xl = xlds + min(n16, rows - 1) * pitch + q * 16;
    float* tab0 = (float*)(xlds + rows * pitch);
    const float* tab = tab0 + 4 * q;
    // cells this task writes / reads its residual from (descriptor offsets stay below 2^31: the scratch is a few MB)
    const __amdgpu_buffer_rsrc_t scr_out = chain_rsrc(ca.scratch + lk.cells_off, 0x7fffffffu);
    const unsigned out_tag = epoch + 1u + (unsigned)t;
    const __amdgpu_buffer_rsrc_t scr_res = chain_rsrc(ca.scratch + lk.res_cells_off, 0x7fffffffu);
    const unsigned res_tag = epoch + 1u + (unsigned)max(lk.res_src, 0);

    // ---- step 0, x from an earlier task: its cells -> raw x rows in LDS.  A cell of LPC 16-byte pieces is fetched by LPC
    // neighbouring lanes of ONE load instruction (piece 0 = tag), and fetched again until its tag is the one this launch
    // writes for that task.  While the producers are still at work a wave polls ONE instruction's worth of cells (1 KiB),
    // with growing pauses -- the early workgroups' polling competes with the late ones' weight stream -- and fetches the
    // rest of its share once those have arrived.
    if (lk.x_src >= 0) {
      const unsigned cb = lk.x_cell_bytes, lpc = cb >> 4, cpi = 64u / lpc;  // lanes per cell, cells per wave instruction
      const unsigned ncell = (unsigned)a.K >> (cb == 64u ? 4 : 3);          // cells per row (= producer blocks): 16 or 8 channels each
      const __amdgpu_buffer_rsrc_t scr_x = chain_rsrc(ca.scratch + lk.x_cells_off, 0x7fffffffu);
      const unsigned want = epoch + 1u + (unsigned)lk.x_src;
      const unsigned piece = (unsigned)lane & (lpc - 1u), cil = (unsigned)lane / lpc;
      const unsigned nbatch = (ncell + cpi - 1u) / cpi;
      constexpr int GRP = 6;  // batches in flight per wave and round
      {  // sentinel: the last batch of this wave's share of row 0 (the tail of a row is written by the workgroups that finish last)
        const unsigned bs = min(nbatch - 1u, nbatch - 1u - (unsigned)wave);
        const unsigned c = min(bs * cpi + cil, ncell - 1u);
        unsigned polls = 0, pause = 1;
        while (true) {
          const unsigned tg = __builtin_amdgcn_raw_buffer_load_b32(scr_x, c * cb, 0, /*sc1*/ 16);
          if (__builtin_amdgcn_ballot_w64(tg != want) == 0ull) break;
          for (unsigned i = 0; i < pause; ++i) __builtin_amdgcn_s_sleep(8);
          pause = min(pause * 2u, 8u);
          if (++polls > (1u << 21)) __builtin_trap();  // seconds: the grid is not co-resident (the host refuses such launches)
        }
      }
      for (int r = 0; r < rows; ++r)
        for (unsigned b0 = (unsigned)wave * GRP; b0 < nbatch; b0 += WAVES * GRP) {
          unsigned polls = 0;
          while (true) {
            u32x4 v[GRP];
            bool ok = true;
#pragma unroll
            for (int g = 0; g < GRP; ++g) {
              const unsigned c = (b0 + g) * cpi + cil;
              v[g] = __builtin_amdgcn_raw_buffer_load_b128(scr_x, min(c, ncell - 1u) * cb + (unsigned)r * ncell * cb + piece * 16u, 0, /*sc1*/ 16);
            }
#pragma unroll
            for (int g = 0; g < GRP; ++g) {
              const unsigned c = (b0 + g) * cpi + cil;
              const unsigned tg = (unsigned)__shfl((int)v[g][0], (int)((unsigned)lane - piece));
              ok = ok && (c >= ncell || tg == want);
            }
            if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) {
#pragma unroll
              for (int g = 0; g < GRP; ++g) {
                const unsigned c = (b0 + g) * cpi + cil;
                if (c < ncell && piece >= 1u && piece < (cb == 64u ? 3u : 2u))
                  *(u32x4*)(xlds + r * pitch + (c * (cb == 64u ? 2u : 1u) + piece - 1u) * 16u) = v[g];
              }
              break;
            }
            __builtin_amdgcn_s_sleep(8);
            if (++polls > (1u << 22)) __builtin_trap();
          }
        }
      __syncthreads();
    }
    QA_CHAIN_STAMP(t, 1);  // x of an earlier task has arrived
    // residuals of this workgroup's first two blocks: requested now, turned into values once x is staged -- the main loop
    // then has no load of its own to wait for besides the weight chunks
    const ChainRes rv0 = QA_CHAIN_RESIDUAL(bx), rv1 = QA_CHAIN_RESIDUAL(bx + gx);

    // ---- x[rows, K] -> LDS [rows][pitch] (+ RMSNorm), unit sums tabulated on the way: w4a16_skinny_kernel's staging,
    // chunk for chunk and thread for thread (same partial sums, same bits); the source is the ordinary tensor or the raw
    // rows step 0 left in place.
    const bool in_lds = lk.x_src >= 0;
    if (a.ln_w) {  // pass 1 copies x raw and sums its squares per row; pass 2 finds x in LDS
      float* ssq = (float*)smem;  // [rows][WAVES], in the reduction buffer (idle between tasks)
      for (int r = 0; r < rows; ++r) {
        const half_t* src = a.X + (size_t)r * a.K;
        float ss = 0.f;
        for (int c = threadIdx.x; c < kc; c += WAVES * 64) {
          u32x4 v;
          if (in_lds) {
            v = *(const u32x4*)(xlds + r * pitch + c * 16);
          } else {
            v = *(const u32x4*)(src + c * 8);
            *(u32x4*)(xlds + r * pitch + c * 16) = v;
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) ss = __builtin_amdgcn_fdot2(as_h2(v[i]), as_h2(v[i]), ss, false);
        }
        ss = wave_sum(ss);
        if (lane == 0) ssq[r * WAVES + wave] = ss;
      }
      __syncthreads();
    }
    for (int r = 0; r < rows; ++r) {
      const half_t* src = a.X + (size_t)r * a.K;
      float inv = 0.f;
      if (a.ln_w) {
        float ss = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) ss += ((const float*)smem)[r * WAVES + w];
        inv = rsqrtf(ss / (float)a.K + a.ln_eps);
      }
      int it = 0;
      for (int c = threadIdx.x; c < kc; c += WAVES * 64, ++it) {  // kc % 16 == 0: rows of 16 lanes are all in or all out
        u32x4 v;
        if (a.ln_w) {  // fp16(fp16(x * inv) * weight): the rounding points of quick_rmsnorm_f16 (and of torch)
          const half8_t xv = *(const half8_t*)(xlds + r * pitch + c * 16);
          half8_t gv;
          if (it == 0) gv = gpre[0];
          else if (it == 1) gv = gpre[1];
          else gv = *(const half8_t*)(a.ln_w + c * 8);
          half8_t o;
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = (half_t)((half_t)((float)xv[j] * inv) * gv[j]);
          v = __builtin_bit_cast(u32x4, o);
          *(u32x4*)(xlds + r * pitch + c * 16) = v;
        } else if (in_lds) {
          v = *(const u32x4*)(xlds + r * pitch + c * 16);
        } else {
          v = *(const u32x4*)(src + c * 8);
          *(u32x4*)(xlds + r * pitch + c * 16) = v;
        }
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
    QA_CHAIN_STAMP(t, 2);  // x staged
    const float rvf0 = QA_CHAIN_RESIDUAL_VALUE(rv0, bx), rvf1 = QA_CHAIN_RESIDUAL_VALUE(rv1, bx + gx);
    QA_CHAIN_STAMP(t, 4);
    if (has_next) {  // (the staging barriers above have made the LDS copy of the arguments visible)
      chain_task_from_lds(arg_words, t + 1, an, lkn);
      QA_CHAIN_NEXT_CONTEXT(an);
    }
    cA = cP;  // register moves, placed where the chunk has long landed
    QA_CHAIN_STAMP(t, 5);

    while (true) {
      QA_CHAIN_STEP(cB, cA);
      QA_CHAIN_STEP(cA, cB);
    }
    acc[0] = floatx4{0.f, 0.f, 0.f, 0.f};  // (a workgroup beyond the task's channel blocks leaves without a finish)
    QA_CHAIN_STAMP(t, 3);  // last block finished: cells and y stored

    ++t;
    if (t >= ca.n) break;
    __syncthreads();  // every wave is done with this task's x rows, table and reduction buffers
    a = an;
    lk = lkn;
    QA_CHAIN_ENTER();
  }
  // the last workgroup out hands the exit counter back zeroed and moves the launch epoch on (nobody waits for this)
  if (threadIdx.x == 0) {
    const unsigned e = __hip_atomic_fetch_add(ca.exits, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (e == (unsigned)gx - 1u) {
      __hip_atomic_store(ca.exits, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store((unsigned*)ca.scratch, epoch + 8u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
#undef QA_CHAIN_STAMP
#undef QA_CHAIN_STEP
#undef QA_CHAIN_COMPUTE
#undef QA_CHAIN_RESIDUAL_VALUE
#undef QA_CHAIN_RESIDUAL
#undef QA_CHAIN_ADVANCE
#undef QA_CHAIN_LOAD
#undef QA_CHAIN_ENTER
#undef QA_CHAIN_PREFETCH
#undef QA_CHAIN_NEXT_CONTEXT
}

}  // namespace quick_amd
