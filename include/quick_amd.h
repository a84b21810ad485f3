/*
 * quick_amd.h -- C ABI of libquick_amd.so: the MI355X (gfx950) W4A16 GEMM behind the
 * SqueezeBits/QUICK operator `quick_kernels.gemm_forward_cuda_quick`.
 *
 * Plain pointers and sizes only: no torch types, no C++ in the signatures.  All pointers are
 * DEVICE pointers unless stated otherwise; `hip_stream` is a hipStream_t passed as void*
 * (NULL = the null stream).  Every entry point is asynchronous with respect to the host and
 * returns a status code; the text of the last error on the calling thread is available through
 * quick_amd_last_error().
 *
 * Reference interfaces replaced (paths relative to the reference repository root):
 *   quick_w4a16_gemm_f16          <- torch::Tensor gemm_forward_cuda_quick(Tensor, Tensor, Tensor,
 *                                    Tensor, int)           csrc/gemm_cuda_quick.h:3-8,
 *                                    host function          csrc/gemm_cuda_quick.cu:1456-1517,
 *                                    exported by            csrc/pybind.cpp:5-8
 *   quick_w4a16_workspace_bytes   <- the `torch::empty({split_k_iters, M, N})` scratch
 *                                                            csrc/gemm_cuda_quick.cu:1468
 *   quick_repack_cuda_to_mi355x   <- the offline interleave of WQLinear_QUICK.from_linear
 *   quick_repack_mi355x_to_cuda      quick/awq/modules/linear/quick.py:88-150 (format bridge:
 *                                    the reference's packed order <-> the MFMA-fragment order)
 */
#ifndef QUICK_AMD_H
#define QUICK_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QUICK_AMD_ABI_VERSION 1

/* status codes */
#define QUICK_OK 0
#define QUICK_ERR_INVALID_ARGUMENT 1 /* shape rule violated: the reference throws std::invalid_argument
                                        (csrc/gemm_cuda_quick.cu:1479-1484) -> Python ValueError */
#define QUICK_ERR_WORKSPACE 2        /* workspace missing or too small */
#define QUICK_ERR_LAUNCH 3           /* HIP launch error */
#define QUICK_ERR_UNSUPPORTED 4      /* valid for the reference but outside this library's envelope */

/* kernel selection for quick_w4a16_gemm_f16_ex (tests / bench); 0 = library heuristic */
#define QUICK_KERNEL_AUTO 0
#define QUICK_KERNEL_SKINNY 1 /* M-tiles straight from L2 to VGPRs, k split over the waves of a workgroup */
#define QUICK_KERNEL_TILED 2  /* activations staged through LDS, MFMA-bound regime */
#define QUICK_KERNEL_WIDE 3   /* large M: 32x32x16 MFMA, one wave per SIMD, operands by LDS-DMA (G % 128 == 0; else TILED runs) */
#define QUICK_KERNEL_XK 4     /* 64- / 128-token tiles, eight waves, the K slices of a tile on different CUs exchange partial tiles
                                 (G % 128 == 0; else TILED runs) */
#define QUICK_KERNEL_XW 5     /* 256 x 256 / 128 x 256 / 128 x 128 / 64 x 128 tiles, four waves (one per SIMD) each owning all tokens and a quarter
                                 of the channels, generated hand-placed K loop; 1 / 2 / 4 K slices of a tile on different CUs exchange fp16
                                 parts and give up on partners that are not there (256 x 256: one slice)
                                 (G / 128 a power of two, N % 256 == 0 for the 256-channel tiles; else TILED runs) */

#define QUICK_KERNEL_LEAN 6   /* [r05] 1..16 tokens: one workgroup per 16 tokens x 16 channels, its waves split K; x by LDS-DMA FIRST in the
                                 memory queue, every weight tile of a wave requested up front, unit sums computed under the weights'
                                 flight (G % 128 == 0, K / 128 between waves and 16 * waves; else QUICK_ERR_UNSUPPORTED when forced) */
#define QUICK_KERNEL_XM 7     /* [r06] 17..128 tokens: one workgroup per 32 / 64 tokens x 32..96 channels for all of K, eight waves splitting K, each
                                 with its own x ring (LDS-DMA) and weight queue in a generated loop without barriers; every dequantised
                                 fragment feeds all token blocks (G a power-of-two multiple of 128, K / 128 >= 8) */

int quick_amd_abi_version(void);
const char* quick_amd_last_error(void);

/*
 * y[M, N] (fp16, row-major) = x[M, K] (fp16, row-major, contiguous) @ dequant(qweight, scales, qzeros)
 *
 *   qweight  int32 [K/4, N/2]   4-bit weights, MI355X order (DESIGN.md "Data layout")
 *   scales   fp16  [K/G, 2N]    as uint32 [K/G * N]: word ((n/16) * (K/G) + g) * 16 + n%16 = fp16 scale | zero point << 16
 *   qzeros   int32 [K/G, N/4]   plain copy of the zero points (nibble n%8 of dword n/8 of row g); not read by the GEMM
 *
 * Pointers: every tensor base 16-byte aligned (rows then are, N and K being multiples of 128); torch allocations are
 * 256-byte aligned.  The kernels move x, y, bias and the residual in 16-byte pieces.
 *
 * Dequantised weight = fp16((w - z) * s) exactly as the reference computes it
 * (csrc/dequantize_quick.cuh:15-63 + sub/mul.rn.f16x2 in csrc/gemm_cuda_quick.cu:52-60);
 * accumulation in fp32, one rounding to fp16 at the end.
 *
 * split_k_iters keeps the reference's argument (number of K slices, >= 1).  The reference uses it
 * as a tuning knob for NVIDIA parts; here it is validated and otherwise treated as a hint -- the
 * library picks its own K partitioning and always returns the fully reduced [M, N] result.
 *
 * Errors (mirroring csrc/gemm_cuda_quick.cu:1479-1484): N % 128 != 0 or N % 8 != 0 or
 * G % 32 != 0 -> QUICK_ERR_INVALID_ARGUMENT.  Additional envelope of this library:
 * K % 128 == 0 and K % G == 0 (every shape in BASELINE.json satisfies it) -> else QUICK_ERR_UNSUPPORTED.
 *
 * workspace: device scratch of at least quick_w4a16_workspace_bytes(...) bytes (may be NULL when
 * that function returns 0).  It must stay valid until the work enqueued on `hip_stream` completes,
 * must be ZERO-FILLED before its first use, and is handed back zero-filled where it matters -- layout:
 * [64 KiB arrival counters of the last-arriver split-K reduction, which reset themselves][16 MiB exchange
 * zone: the mailboxes through which the K slices of an exchange-K tile swap partial sums; every consumer
 * zeroes what it has read][fp32 partial tiles of the last-arriver kernels, overwritten before they are read]
 * -- so one zeroed buffer can be reused by every later call on the same stream, whatever its shape.  Do
 * not share it between streams that run concurrently.  (A launch that is aborted -- device reset --
 * may leave the first two regions dirty: zero the buffer again before reusing it.)
 * The K slices of a tile that meet through the exchange zone (QUICK_KERNEL_XK / QUICK_KERNEL_XW launches with more than
 * one slice) need NOT be co-resident since r04: a wave that has polled for a partner longer than the poll limit (41 us)
 * gives its part up -- own share to its self box, a flag bit in the part's state word in the counter region -- and leaves;
 * the partner whose shares reach memory last finds the flag, claims the part (the protocol's only atomic, a compare-and-swap) and
 * finishes it from the boxes.  The fast path writes no state word.  No kernel of the library spins without bound or traps.
 */
int quick_w4a16_gemm_f16(const void* x, const void* qweight, const void* scales, const void* qzeros,
                         void* y, void* workspace, size_t workspace_bytes,
                         int M, int K, int N, int group_size, int split_k_iters, void* hip_stream);

size_t quick_w4a16_workspace_bytes(int M, int K, int N, int group_size, int split_k_iters);

/* Is the workspace in the state every launch expects -- arrival counters, part state words and the exchange zone all-zero?  One pass
 * over the guarded regions (64 KiB + 16 MiB, or the whole buffer if smaller) on `hip_stream`, then a stream synchronise: QUICK_OK, or
 * QUICK_ERR_WORKSPACE with the first dirty byte offset in quick_amd_last_error().  Why it exists: the K slices of an exchange launch
 * hand their partial tiles over in 16-byte granules that are their own arrival flags (no dword of a payload granule is ever 0), and the
 * payload -- eight fp16 partial sums -- has no spare bits for a launch epoch; a separate epoch word per box would need the sender's
 * stores acknowledged before it may be written, the round trip the protocol exists to avoid.  So a granule left behind by an aborted
 * launch cannot be told from a partner's, and the contract is "zero between launches": the library keeps it on every path (also when
 * waves give up), this call verifies it, and QUICK_AMD_CHECK_WORKSPACE=1 in the environment runs it in front of every
 * launch that uses the regions (debugging: it synchronises).  Python: quick_amd.kernels zeroes its per-stream workspace again
 * whenever a launch returns an error.  The reference's split-K scratch + `.sum(0)` (csrc/gemm_cuda_quick.cu:1468,1515) holds no state
 * between launches; this is the price of reducing inside the launch. */
int quick_w4a16_workspace_check(const void* workspace, size_t workspace_bytes, void* hip_stream);

/* Same as quick_w4a16_gemm_f16 with an explicit kernel choice, an optional fp16 bias[N] added in the
 * epilogue (NULL = none; replaces the separate torch add of quick/awq/modules/linear/quick.py:165), and a
 * forced K split across workgroups (0 = heuristic).
 *
 * `kernel`: 0 = the library's planner (what every product path passes).  Otherwise, for tests and tuning, a
 * bit field -- every field 0 = "planner's choice within the family":
 *   bits 0-3    family, QUICK_KERNEL_*
 *   bits 4-7    SKINNY: channel tiles of 16 per workgroup (1, 2, 4; 8 = [r05] the straight-line eight-tile fragment kernel where it is built --
 *               9..16 tokens, G = 128, K / 128 = 8 waves x slices x {2, 4, 7, 8} k tiles, no RMSNorm prologue -- else 4; 7 = [r06] the same kernel with seven tiles per
 *               workgroup: K = 8192, one slice, N % 112 == 0 -- else 4); TILED: token tiles of 16 per workgroup (2, 4, 8);
 *               WIDE / XK / XW: token tiles of 32 per workgroup (WIDE 2, 4, 8; XK 2, 4; XW 2, 4, 8 -- 0 = 4; 8 = the 256 x 256 tile; [r06] WIDE 8 x 2 pairs runs
 *               128 x 256 tiles: r02's hipcc-scheduled 256 x 256 tile spilled registers and lives on in tools builds only);
 *               XM: 32-channel pairs per workgroup (1..3; 0 = the fewest that cover the layer in one round)
 *   bits 8-11   SKINNY / TILED / LEAN: waves per workgroup / 4 ([r06] SKINNY: sixteen waves only with one channel tile per workgroup); XM (bits 8-9): 1 = 32-token tiles, 2 = 64-token tiles; WIDE: 32-channel pairs per wave (1, 2); XK: K slices per tile (1, 2, 4, 8; 15 = half
 *               the planner's count); XW: K slices per tile (1, 2, 4 <= token tiles)
 *   bit 12      SKINNY: no LDS copy of x; WIDE: the double-buffered kernel at every tile size (no ring); XW: 128-channel tiles (implied by 2 token tiles)
 *   bit 13      TILED: retired in r06 (r01's 32x32x16 flavour: QUICK_ERR_INVALID_ARGUMENT)         bit 14  TILED / WIDE / XK: plain (not XCD-aware) tile order
 *   bit 15      TILED: 2 x 4 wave grid; WIDE: eight waves per workgroup (ring kernel)
 *   bits 16-20  timing experiments (wrong results on purpose, phase stamps): only in a QUICK_AMD_TOOLS build of the library
 *               (`python -m quick_amd.build --tools`); the product library answers QUICK_ERR_INVALID_ARGUMENT
 *   bit 21      SKINNY: flip the persistence default
 *   bits 22-24  SKINNY: persistent slots per CU; WIDE: LDS ring slots; XK: x ring slots; XW (bits 22-26): log2 of the exchange poll limit in
 *               ticks of 10 ns (0 = default 2^12; tests pass 1: every wave gives its part up at once)
 *   bit 25      SKINNY: exact per-weight dequantisation (no deferred zero point)
 *   bits 26-28  SKINNY: 26 force the table deferred-zero path, 28 no fragment deferred-zero path; TILED: 27 force 128 x 256
 *               four-wave tiles; XK: weight queue depth in stages (3..6)
 *   bits 29-30  TILED: force / forbid 256-channel tiles
 * Environment: QUICK_AMD_EXCHANGE_CUS=<n> -- a SPEED hint since r04: the K slices of an XK launch exchange fastest when tiles x slices
 * workgroups are co-resident, and the planner sizes the slice count for the device's CU count; a process whose queues see fewer CUs (a
 * CU mask) may say so (0: never split K this way).  Correctness does not depend on it: slices that are not there in time are given up
 * on and finished by the last arriver (see "workspace").  QUICK_AMD_EXCHANGE_POLL_LOG2=<n> overrides the poll limit (2^n ticks of 10 ns).
 * QUICK_AMD_XW256=0 -- the planner keeps r02's hipcc-scheduled kernel for the 256 x 256 tile (bit-identical results; the A/B switch).
 * A combination the library has no build for returns QUICK_ERR_UNSUPPORTED; results never depend on the field
 * beyond fp32 summation order (and bit 25's rounding). */
int quick_w4a16_gemm_f16_ex(const void* x, const void* qweight, const void* scales, const void* qzeros,
                            const void* bias, void* y, void* workspace, size_t workspace_bytes,
                            int M, int K, int N, int group_size, int kernel, int grid_split_k,
                            void* hip_stream);
size_t quick_w4a16_workspace_bytes_ex(int M, int K, int N, int group_size, int kernel, int grid_split_k);

/* What may be fused around the GEMM (all optional; zero-initialise the struct):
 *   rmsnorm_weight  fp16 [K]: the GEMM consumes RMSNorm(x) * weight (eps = rmsnorm_eps).  Where the kernel holds x whole
 *                   in LDS it normalises there with quick_rmsnorm_f16's rounding points (fp16(fp16(x * rstd) * weight));
 *                   where it takes x fragments straight from L2 it multiplies them by the weight in fp16 and applies
 *                   1 / rms to the fp32 result.  Only where quick_w4a16_can_fuse_rmsnorm() says so (small-M kernels, K not
 *                   split across workgroups) -- QUICK_ERR_UNSUPPORTED otherwise (run quick_rmsnorm_f16 first).
 *   bias            fp16 [N]      (replaces the torch add of quick/awq/modules/linear/quick.py:165)
 *   residual        fp16 [M, N], may alias y: the decoder block's `hidden + proj(...)`
 *   silu_mul        output channels are gate/up interleaved in blocks of 8 (16t+i gate, 16t+8+i up, i < 8) and the
 *                   epilogue writes y[M, N/2] with y[m, 8t+i] = silu(gate) * up -- the intent of the reference's unused
 *                   QuantFusedMLP (quick/awq/modules/fused/mlp.py:52-71).  Excludes bias and residual.
 * Everything is accumulated in fp32 and rounded to fp16 once (silu_mul rounds gate, up and silu as torch does). */
typedef struct quick_gemm_fusion {
  const void* bias;
  const void* residual;
  const void* rmsnorm_weight;
  float rmsnorm_eps;
  int silu_mul;
} quick_gemm_fusion;
int quick_w4a16_gemm_f16_fused(const void* x, const void* qweight, const void* scales, const void* qzeros,
                               const quick_gemm_fusion* fusion, void* y, void* workspace, size_t workspace_bytes,
                               int M, int K, int N, int group_size, int kernel, int grid_split_k, void* hip_stream);
int quick_w4a16_can_fuse_rmsnorm(int M, int K, int N, int group_size);

/* What a launch of this shape will run, as one line of text (host-only, no GPU needed): kernel family, tile shape,
 * grid, K split and workspace, e.g. "tiled tokens=64 channels=128 waves=8 grid=256x1 ksplit=1 xcd_rows=2 workspace=0" or
 * "skinny ntw=1 waves=8 x=lds dequant=deferred-zero-table grid=256x1x1 ksplit=1 workspace=0".  For logs and for tests of
 * the shape heuristics; the wording may grow fields, the leading family word will not change.  `kernel` and
 * `grid_split_k` as in quick_w4a16_gemm_f16_ex (0 = auto).  (No counterpart in the reference: its kernel has one shape.) */
int quick_w4a16_plan_describe(int M, int K, int N, int group_size, int kernel, int grid_split_k, char* text, size_t text_bytes);

/*
 * Measurement aid (bench.py): enqueue the GEMM `iters` times on `hip_stream`, cycling through `n_sets`
 * weight sets (host arrays of device pointers) so that consecutive launches do not hit in the 256 MiB
 * Infinity Cache, with a hipEvent pair bound to each main-kernel dispatch (hipExtLaunchKernelGGL);
 * synchronise, and write each dispatch's own duration in microseconds to the HOST array
 * kernel_us[iters].  Blocking; not part of the reference interface.
 */
int quick_w4a16_gemm_profile(const void* x, const void* const* qweights, const void* const* scales,
                             const void* const* qzeros, int n_sets, void* y, void* workspace,
                             size_t workspace_bytes, int M, int K, int N, int group_size, int kernel,
                             int grid_split_k, int iters, float* kernel_us, void* hip_stream);

/* Measurement aid: the IN-KERNEL wall-clock span of each of `iters` launches -- first wave's start to last wave's end on the
 * constant 100 MHz counter (s_memrealtime), written by the kernels themselves -- in microseconds to the HOST array
 * span_us[iters]; weight sets cycled as in quick_w4a16_gemm_profile.  The dispatch-duration clock reads ~4.2 us for an
 * EMPTY kernel, so launches of a few microseconds (small M) are only visible on this one.  Blocking. */
int quick_w4a16_gemm_span(const void* x, const void* const* qweights, const void* const* scales, const void* const* qzeros,
                          int n_sets, void* y, void* workspace, size_t workspace_bytes, int M, int K, int N, int group_size,
                          int kernel, int grid_split_k, int iters, float* span_us, void* hip_stream);

/* Measurement aid: the same event-pair clock on `iters` dispatches of an EMPTY kernel with the GEMM's launch
 * shape (256 workgroups x 512 threads): the fixed part of every "kernel duration" reading on this stack. */
int quick_amd_dispatch_floor(int iters, float* kernel_us, void* hip_stream);

/*
 * Decode-step glue around the GEMMs (callers of the hot path; counterparts of the out-of-tree kernels the
 * reference's fused runtime uses: awq_ext.layernorm_forward_cuda, quick/awq/modules/fused/norm.py:18;
 * awq_ft_ext.single_query_attention, quick/awq/modules/fused/attn.py:217).  fp16 tensors, fp32 arithmetic.
 *   quick_rmsnorm_f16          y[r,:] = x[r,:] * rsqrt(mean(x[r,:]^2) + eps) * weight
 *   quick_rope_kv_append_f16   one new token per sequence: rotate q, k of qkv[B, (nh+2nkv)*D] by table row *pos
 *                              (rotate-half convention), q -> q_out[B, nh, D], k/v -> caches [B, nkv, L, D] at *pos
 *   quick_decode_attention_f16 single-query attention over cache positions 0..*pos, GQA aware, D == 128
 *   quick_silu_mul_f16         y[m, 8t+i] = silu(gate_up[m, 16t+i]) * gate_up[m, 16t+8+i]  (gate/up interleaved by 8)
 * `pos` is a DEVICE pointer to one int64 (so a captured hipGraph can advance it); the caller keeps 0 <= *pos < cache_len
 * (a device value: the library cannot check it) and batch <= 65535.
 */
int quick_rmsnorm_f16(const void* x, const void* weight, void* y, int rows, int hidden, float eps, void* hip_stream);
int quick_rope_kv_append_f16(const void* qkv, const void* cos_table, const void* sin_table, const void* pos,
                             void* q_out, void* k_cache, void* v_cache, int batch, int n_heads, int n_kv_heads,
                             int head_dim, int cache_len, void* hip_stream);
/* prefill form: `tokens` consecutive positions *pos0 .. per sequence; qkv [batch * tokens, (nh + 2 nkv) * D];
 * q_out [batch, nh, tokens, D] */
int quick_rope_kv_write_f16(const void* qkv, const void* cos_table, const void* sin_table, const void* pos0, void* q_out,
                            void* k_cache, void* v_cache, int batch, int tokens, int n_heads, int n_kv_heads, int head_dim,
                            int cache_len, void* hip_stream);
int quick_decode_attention_f16(const void* q, const void* k_cache, const void* v_cache, const void* pos, void* out,
                               int batch, int n_heads, int n_kv_heads, int head_dim, int cache_len, float scale,
                               void* hip_stream);
/* quick_rope_kv_append_f16 + quick_decode_attention_f16 in one launch, reading the qkv GEMM output directly.  One online-softmax sweep
 * over the cache, software-pipelined (r05); rows at or behind *pos may hold anything (they are never read into a sum).  Grouped-query
 * models with >= 256 (sequence, KV head) pairs and 4 or 8 query heads per KV head take their scores from the matrix core
 * (v_mfma_f32_16x16x32_f16 on 16 cache rows x the heads of the group); QUICK_AMD_ATTN_MFMA=0 keeps the vector-ALU sweep (A/B switch,
 * same results within fp32 summation order). */
int quick_decode_rope_attention_f16(const void* qkv, const void* cos_table, const void* sin_table, const void* pos,
                                    void* k_cache, void* v_cache, void* out, int batch, int n_heads, int n_kv_heads,
                                    int head_dim, int cache_len, float scale, void* hip_stream);
int quick_silu_mul_f16(const void* gate_up, void* y, int rows, int intermediate, void* hip_stream);

/* lm_head of a decode step with the greedy arg-max folded in (the reference: torch linear + max on the fp16 layer AWQ leaves
 * unquantised, examples/benchmark.py:54-57): h = RMSNorm(x[batch, hidden]) * norm_weight (quick_rmsnorm_f16's rounding; norm_weight
 * may be null: h = x), logits[b, v] = h[b, :] . weight[v, :] (fp16 [vocab, hidden], fp32 sums, rounded to fp16),
 * next_token[b] (int64) = the lowest index among the largest fp16 logits.  hidden_out [batch, hidden] and logits [batch, vocab]
 * are optional outputs (null: not written).  batch <= 4, hidden % 512 == 0, batch * hidden <= 36864; QUICK_ERR_UNSUPPORTED
 * otherwise (the caller falls back to its GEMM library).  workspace: quick_lm_head_workspace_bytes(batch), any content. */
size_t quick_lm_head_workspace_bytes(int batch);
int quick_lm_head_argmax_f16(const void* x, const void* norm_weight, float eps, const void* weight, void* hidden_out, void* logits,
                             void* next_token, void* workspace, size_t workspace_bytes, int batch, int vocab, int hidden,
                             void* hip_stream);

#ifdef QUICK_AMD_TOOLS /* only in `python -m quick_amd.build --tools` libraries (libquick_amd_tools.so) */
/* Measurement aid (tools/prefetch_probe.py): pull [ptr, ptr + bytes) through HBM into the memory-side cache -- one
 * dword read per 128-byte line, results unused -- with `workgroups` (low 16 bits; 0 = 64) workgroups of 256 threads; bits 16..
 * choose the touch density (0 / 1: one dword per line, 2, 4, 32 = every byte).  No library path calls it. */
int quick_prefetch(const void* ptr, size_t bytes, int workgroups, void* hip_stream);
#endif

/*
 * Format bridge.  "cuda order" is byte-for-byte what the reference's WQLinear_QUICK.from_linear
 * writes (quick/awq/modules/linear/quick.py:88-150) and what its checkpoints hold; "mi355x order"
 * is what quick_w4a16_gemm_f16 consumes.  All six buffers have the reference's shapes; in and out
 * must not alias.
 */
int quick_repack_cuda_to_mi355x(const void* qweight_in, const void* scales_in, const void* qzeros_in,
                                void* qweight_out, void* scales_out, void* qzeros_out,
                                int K, int N, int group_size, void* hip_stream);
int quick_repack_mi355x_to_cuda(const void* qweight_in, const void* scales_in, const void* qzeros_in,
                                void* qweight_out, void* scales_out, void* qzeros_out,
                                int K, int N, int group_size, void* hip_stream);

/* in_features that the MI355X weight order cannot tile (K % 128 != 0; the reference accepts K % 32 == 0,
 * csrc/gemm_cuda_quick.cu:1479-1484 -- only possible with a group size that is not a multiple of 128): the layer runs on a copy
 * padded along K to quick_padded_in_features(K, G) = the next multiple of lcm(128, G), with weights 0, zero points 0 and
 * scales 0 in the added rows / groups (they contribute exactly 0), and on activations zero-padded to that width by the
 * caller.  quick_repack_cuda_to_mi355x_padded writes that copy: outputs are sized for the padded K
 * (qweight [Kp/4, N/2], scales [Kp/G, 2N], qzeros [Kp/G, N/4]).  quick_padded_in_features returns 0 for shapes the
 * reference rejects too. */
int quick_padded_in_features(int K, int group_size);
int quick_repack_cuda_to_mi355x_padded(const void* qweight_in, const void* scales_in, const void* qzeros_in, void* qweight_out,
                                       void* scales_out, void* qzeros_out, int K, int N, int group_size, void* hip_stream);

/* Dequantise an MI355X-order layer to a dense fp16 [K, N] row-major matrix (debug / parity aid;
 * counterpart of the reference CPU path quick/awq/utils/packing_utils.py:82-97). */
int quick_dequantize_mi355x_f16(const void* qweight, const void* scales, const void* qzeros,
                                void* w_out, int K, int N, int group_size, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* QUICK_AMD_H */
