// r04 microbenchmark: what ONE filler instruction of a given kind costs beside v_mfma_f32_32x32x16_f16 with one wave per SIMD.
// Each workgroup = 4 waves; a wave runs REP x { 8 MFMAs on 8 different accumulators, each followed by N fillers of kind K }, all asm,
// fillers on rotating independent registers (no dependent pairs closer than 8 instructions).  Prints shader clocks per MFMA.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_filler_cost tools/mfma_filler_cost.hip && tools/bin/mfma_filler_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

#define MF(i) "v_mfma_f32_32x32x16_f16 %" #i ", %[a], %[b], %" #i "\n\t"

// filler strings: operate on v[200:215] (clobbered), constants in s40, s41 (set by the asm), LDS address v216
#define F_NONE ""
#define F_ANDOR(r) "v_and_or_b32 v" #r ", v" #r ", s40, v217\n\t"
#define F_LSHR(r) "v_lshrrev_b32 v" #r ", 8, v" #r "\n\t"
#define F_PKADD(r) "v_pk_add_f16 v" #r ", v" #r ", v217\n\t"
#define F_PKMUL(r) "v_pk_mul_f16 v" #r ", v" #r ", v217\n\t"
#define F_PKFMA(r) "v_pk_fma_f16 v" #r ", v" #r ", s41, v217\n\t"
#define F_ADD16(r) "v_add_f16 v" #r ", v" #r ", v217\n\t"
#define F_MIXLO(r) "v_fma_mixlo_f16 v" #r ", v" #r ", v217, 0 op_sel_hi:[1,1,0]\n\t"
#define F_FMA32(r) "v_fma_f32 v" #r ", v" #r ", v217, v217\n\t"
#define F_MOV(r) "v_mov_b32 v" #r ", v217\n\t"
#define F_DSRD(r) "ds_read_b128 v[" #r ":" #r "+3], v216\n\t"
#define F_SALU(r) "s_add_u32 s42, s42, 1\n\t"
#define F_PKADD32(r) "v_pk_add_f32 v[" #r ":" #r "+1], v[" #r ":" #r "+1], v[218:219]\n\t"

// N fillers behind MFMA i, on registers 200 + 2 * ((i * N + j) % 8)
#define FILL0(F, i)
#define FILL1(F, i) F(200)
#define FILL2(F, i) F(200) F(202)
#define FILL3(F, i) F(200) F(202) F(204)
#define FILL4(F, i) F(200) F(202) F(204) F(206)
#define FILL5(F, i) F(200) F(202) F(204) F(206) F(208)
#define FILL6(F, i) F(200) F(202) F(204) F(206) F(208) F(210)
#define FILL4B(F, i) F(208) F(210) F(212) F(214)
#define FILL3B(F, i) F(208) F(210) F(212)
#define FILL2B(F, i) F(208) F(210)
#define FILL1B(F, i) F(208)
#define FILL5B(F, i) F(208) F(210) F(212) F(214) F(200)
#define FILL6B(F, i) F(208) F(210) F(212) F(214) F(200) F(202)
#define FILL0B(F, i)

// realistic mixes per 8 MFMAs: A = 26 VALU + 4 reads (128 x 64 per wave), B = 26 VALU + 8 reads (128 x 32), C = 52 VALU + 8 reads (64 x 32)
#define V3 F_PKMUL(200) F_ANDOR(202) F_PKADD(204)
#define V4 F_PKMUL(200) F_ANDOR(202) F_PKADD(204) F_LSHR(206)
#define V6 F_PKMUL(200) F_ANDOR(202) F_PKADD(204) F_LSHR(206) F_PKFMA(208) F_ANDOR(210)
#define V7 F_PKMUL(200) F_ANDOR(202) F_PKADD(204) F_LSHR(206) F_PKFMA(208) F_ANDOR(210) F_PKMUL(212)
#define RD(r) "ds_read_b128 v[" #r ":" #r "+3], v216\n\t"
#define RDA(r) "ds_read_b128 a[" #r ":" #r "+3], v216\n\t"
#define MIXA MF(0) RD(220) V3 MF(1) RD(224) V3 MF(2) RD(228) V3 MF(3) RD(232) V3 MF(4) V3 MF(5) V4 MF(6) V3 MF(7) V4
#define MIXB MF(0) RD(220) V3 MF(1) RD(224) V3 MF(2) RD(228) V3 MF(3) RD(232) V3 MF(4) RD(236) V3 MF(5) RD(240) V4 MF(6) RD(244) V3 MF(7) RD(220) V4
#define MIXC MF(0) RD(220) V6 MF(1) RD(224) V7 MF(2) RD(228) V6 MF(3) RD(232) V7 MF(4) RD(236) V6 MF(5) RD(240) V7 MF(6) RD(244) V6 MF(7) RD(220) V7
#define MIXBA MF(0) RDA(128) V3 MF(1) RDA(132) V3 MF(2) RDA(136) V3 MF(3) RDA(140) V3 MF(4) RDA(144) V3 MF(5) RDA(148) V4 MF(6) RDA(152) V3 MF(7) RDA(156) V4
#define V1 F_PKMUL(200)
#define V2 F_ANDOR(202) F_PKADD(204)
// D = 13 VALU + 4 reads per 8 MFMAs (256 x 64 per wave: a k16 step is 16 MFMAs, 26 VALU, 8 reads)
#define MIXD MF(0) RD(220) V2 MF(1) RD(224) V1 MF(2) RD(228) V2 MF(3) RD(232) V1 MF(4) V2 MF(5) V2 MF(6) V1 MF(7) V2
// the same mixes on ROTATING random operands (four A, eight B fragments of N(0, 0.5)-like numbers): what the part's power management makes of
// real data -- constant operands toggle nothing between consecutive MFMAs and run at a clock no real launch sees
#define MR(i, x, y) "v_mfma_f32_32x32x16_f16 %" #i ", %[ra" #x "], %[rb" #y "], %" #i "\n\t"
#define MIXAR MR(0, 0, 0) RD(220) V3 MR(1, 0, 1) RD(224) V3 MR(2, 0, 2) RD(228) V3 MR(3, 0, 3) RD(232) V3 MR(4, 1, 0) V3 MR(5, 1, 1) V4 MR(6, 1, 2) V3 MR(7, 1, 3) V4 \
              MR(0, 2, 4) RD(220) V3 MR(1, 2, 5) RD(224) V3 MR(2, 2, 6) RD(228) V3 MR(3, 2, 7) RD(232) V3 MR(4, 3, 4) V3 MR(5, 3, 5) V4 MR(6, 3, 6) V3 MR(7, 3, 7) V4
#define MIXDR MR(0, 0, 0) RD(220) V2 MR(1, 0, 1) RD(224) V1 MR(2, 0, 2) RD(228) V2 MR(3, 0, 3) RD(232) V1 MR(4, 0, 4) RD(236) V2 MR(5, 0, 5) RD(240) V2 MR(6, 0, 6) RD(244) V1 MR(7, 0, 7) RD(220) V2 \
              MR(0, 1, 0) V2 MR(1, 1, 1) V1 MR(2, 1, 2) V2 MR(3, 1, 3) V1 MR(4, 1, 4) V2 MR(5, 1, 5) V2 MR(6, 1, 6) V1 MR(7, 1, 7) V2
#define MIXNR MR(0, 0, 0) MR(1, 0, 1) MR(2, 0, 2) MR(3, 0, 3) MR(4, 1, 4) MR(5, 1, 5) MR(6, 1, 6) MR(7, 1, 7) MR(0, 2, 0) MR(1, 2, 1) MR(2, 2, 2) MR(3, 2, 3) MR(4, 3, 4) MR(5, 3, 5) MR(6, 3, 6) MR(7, 3, 7)
#define BODY(F, N) MF(0) FILL##N(F, 0) MF(1) FILL##N##B(F, 1) MF(2) FILL##N(F, 2) MF(3) FILL##N##B(F, 3) MF(4) FILL##N(F, 4) MF(5) FILL##N##B(F, 5) MF(6) FILL##N(F, 6) MF(7) FILL##N##B(F, 7)

template <int KIND, int N>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void bench(unsigned long long* out, int rep) {
  extern __shared__ char smem[];
  floatx16 c[8];
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f); b[i] = (_Float16)(i * 0.01f); }
  const unsigned lds = (threadIdx.x & 63) * 16;
  half8 ra[4], rb[8];
  {
    unsigned st = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return (_Float16)(((int)(st >> 9) % 2001 - 1000) * 0.001f); };   // ~U(-1, 1)
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 8; ++e) ra[j][e] = rnd();
    for (int j = 0; j < 8; ++j) for (int e = 0; e < 8; ++e) rb[j][e] = rnd();
  }
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define RUN(F, NN)                                                                                                                  \
  asm volatile("s_mov_b32 s40, 0x000f000f\n\ts_mov_b32 s41, 0x2c002c00\n\ts_mov_b32 s42, 0\n\tv_mov_b32 v217, 0x3c003c00\n\tv_mov_b32 v216, %[l]\n\t" \
               "v_mov_b32 v218, 1.0\n\tv_mov_b32 v219, 1.0\n\t"                                                                     \
               "v_mov_b32 v200, 0\n\tv_mov_b32 v201, 0\n\tv_mov_b32 v202, 0\n\tv_mov_b32 v203, 0\n\tv_mov_b32 v204, 0\n\tv_mov_b32 v205, 0\n\tv_mov_b32 v206, 0\n\tv_mov_b32 v207, 0\n\t" \
               "v_mov_b32 v208, 0\n\tv_mov_b32 v209, 0\n\tv_mov_b32 v210, 0\n\tv_mov_b32 v211, 0\n\tv_mov_b32 v212, 0\n\tv_mov_b32 v213, 0\n\tv_mov_b32 v214, 0\n\tv_mov_b32 v215, 0\n\t" \
               "s_mov_b32 s43, %[rep]\n\t"                                                                                          \
               ".Lloop_%=:\n\t" BODY(F, NN) "s_waitcnt lgkmcnt(0)\n\ts_sub_u32 s43, s43, 1\n\ts_cmp_lg_u32 s43, 0\n\ts_cbranch_scc1 .Lloop_%=\n\ts_nop 15\n\t"  \
               : "+a"(c[0]), "+a"(c[1]), "+a"(c[2]), "+a"(c[3]), "+a"(c[4]), "+a"(c[5]), "+a"(c[6]), "+a"(c[7])                     \
               : [a] "v"(a), [b] "v"(b), [l] "v"(lds), [rep] "s"(rep)                                                               \
               : "memory", "scc", "s40", "s41", "s42", "s43", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", \
                 "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219")
#define RUNK(NN)                                                        \
  if constexpr (KIND == 0) RUN(F_MOV, NN);                              \
  else if constexpr (KIND == 1) RUN(F_ANDOR, NN);                       \
  else if constexpr (KIND == 2) RUN(F_LSHR, NN);                        \
  else if constexpr (KIND == 3) RUN(F_PKADD, NN);                       \
  else if constexpr (KIND == 4) RUN(F_PKMUL, NN);                       \
  else if constexpr (KIND == 5) RUN(F_PKFMA, NN);                       \
  else if constexpr (KIND == 6) RUN(F_ADD16, NN);                       \
  else if constexpr (KIND == 7) RUN(F_MIXLO, NN);                       \
  else if constexpr (KIND == 8) RUN(F_FMA32, NN);                       \
  else if constexpr (KIND == 9) RUN(F_DSRD, NN);                        \
  else if constexpr (KIND == 10) RUN(F_SALU, NN);                       \
  else RUN(F_PKADD32, NN);
#define RUNMIX(MIX)                                                                                                                 \
  asm volatile("s_mov_b32 s40, 0x000f000f\n\ts_mov_b32 s41, 0x2c002c00\n\ts_mov_b32 s42, 0\n\tv_mov_b32 v217, 0x3c003c00\n\tv_mov_b32 v216, %[l]\n\t" \
               "v_mov_b32 v200, 0\n\tv_mov_b32 v202, 0\n\tv_mov_b32 v204, 0\n\tv_mov_b32 v206, 0\n\tv_mov_b32 v208, 0\n\tv_mov_b32 v210, 0\n\tv_mov_b32 v212, 0\n\t" \
               "s_mov_b32 s43, %[rep]\n\t"                                                                                          \
               ".Lloop_%=:\n\t" MIX "s_waitcnt lgkmcnt(0)\n\ts_sub_u32 s43, s43, 1\n\ts_cmp_lg_u32 s43, 0\n\ts_cbranch_scc1 .Lloop_%=\n\ts_nop 15\n\t"  \
               : "+a"(c[0]), "+a"(c[1]), "+a"(c[2]), "+a"(c[3]), "+a"(c[4]), "+a"(c[5]), "+a"(c[6]), "+a"(c[7])                     \
               : [a] "v"(a), [b] "v"(b), [l] "v"(lds), [rep] "s"(rep)                                                               \
               : "memory", "scc", "s40", "s41", "s42", "s43", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", \
                 "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", \
                 "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", \
                 "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", \
                 "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159")
#define RUNMIXR(MIX)                                                                                                                \
  asm volatile("s_mov_b32 s40, 0x000f000f\n\ts_mov_b32 s41, 0x2c002c00\n\ts_mov_b32 s42, 0\n\tv_mov_b32 v217, 0x3c003c00\n\tv_mov_b32 v216, %[l]\n\t" \
               "v_mov_b32 v200, 0\n\tv_mov_b32 v202, 0\n\tv_mov_b32 v204, 0\n\tv_mov_b32 v206, 0\n\tv_mov_b32 v208, 0\n\tv_mov_b32 v210, 0\n\tv_mov_b32 v212, 0\n\t" \
               "s_mov_b32 s43, %[rep]\n\t"                                                                                          \
               ".Lloop_%=:\n\t" MIX "s_waitcnt lgkmcnt(0)\n\ts_sub_u32 s43, s43, 1\n\ts_cmp_lg_u32 s43, 0\n\ts_cbranch_scc1 .Lloop_%=\n\ts_nop 15\n\t"  \
               : "+a"(c[0]), "+a"(c[1]), "+a"(c[2]), "+a"(c[3]), "+a"(c[4]), "+a"(c[5]), "+a"(c[6]), "+a"(c[7])                     \
               : [ra0] "v"(ra[0]), [ra1] "v"(ra[1]), [ra2] "v"(ra[2]), [ra3] "v"(ra[3]), [rb0] "v"(rb[0]), [rb1] "v"(rb[1]), [rb2] "v"(rb[2]),   \
                 [rb3] "v"(rb[3]), [rb4] "v"(rb[4]), [rb5] "v"(rb[5]), [rb6] "v"(rb[6]), [rb7] "v"(rb[7]), [l] "v"(lds), [rep] "s"(rep / 2)      \
               : "memory", "scc", "s40", "s41", "s42", "s43", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", \
                 "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", \
                 "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247")
  if constexpr (KIND == 30) { RUNMIXR(MIXAR); }
  else if constexpr (KIND == 31) { RUNMIXR(MIXDR); }
  else if constexpr (KIND == 32) { RUNMIXR(MIXNR); }
  else if constexpr (KIND == 20) { RUNMIX(MIXA); }
  else if constexpr (KIND == 21) { RUNMIX(MIXB); }
  else if constexpr (KIND == 22) { RUNMIX(MIXC); }
  else if constexpr (KIND == 23) { RUNMIX(MIXBA); }
  else if constexpr (KIND == 24) { RUNMIX(MIXD); }
  else if constexpr (N == 0) { RUNK(0) }
  else if constexpr (N == 1) { RUNK(1) }
  else if constexpr (N == 2) { RUNK(2) }
  else if constexpr (N == 3) { RUNK(3) }
  else if constexpr (N == 4) { RUNK(4) }
  else if constexpr (N == 5) { RUNK(5) }
  else { RUNK(6) }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += c[i][0];
  if (threadIdx.x % 64 == 0) {
    out[(blockIdx.x * 4 + threadIdx.x / 64) * 2] = t1 - t0;
    out[(blockIdx.x * 4 + threadIdx.x / 64) * 2 + 1] = r1 - r0;   // 100 MHz ticks
  }
  if (s == 123.456f) out[0] = 1;
}

static const char* names[] = {"v_mov_b32", "v_and_or_b32", "v_lshrrev_b32", "v_pk_add_f16", "v_pk_mul_f16", "v_pk_fma_f16(sgpr)", "v_add_f16", "v_fma_mixlo_f16",
                              "v_fma_f32", "ds_read_b128", "s_add_u32", "v_pk_add_f32"};

template <int KIND, int N>
void run(unsigned long long* d, int blocks) {
  const int rep = 512;
  std::vector<unsigned long long> h(blocks * 8);
  for (int it = 0; it < 3; ++it) {
    hipLaunchKernelGGL((bench<KIND, N>), dim3(blocks), dim3(256), 1024 * 16, 0, d, rep);
    hipDeviceSynchronize();
  }
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  double sum = 0, rsum = 0;
  for (int i = 0; i < blocks * 4; ++i) { sum += (double)h[i * 2]; rsum += (double)h[i * 2 + 1]; }
  static const char* mixes[] = {"mix A: 26 VALU + 4 ds_read_b128 / 8 MFMA", "mix B: 26 VALU + 8 reads", "mix C: 52 VALU + 8 reads", "mix B, reads into AGPRs",
                                "mix D: 13 VALU + 4 reads"};
  static const char* rmixes[] = {"random operands, mix A (128 x 64 per wave)", "random operands, mix D (256 x 64 per wave)", "random operands, MFMAs only"};
  const double clk = sum / (blocks * 4) / (rep * 8.0), ns = rsum * 10.0 / (blocks * 4) / (rep * 8.0);
  printf("  %-20s N=%d: %6.2f clocks per MFMA  %6.2f ns per MFMA (%.3f GHz)\n", KIND >= 30 ? rmixes[KIND - 30] : (KIND >= 20 ? mixes[KIND - 20] : names[KIND]), N, clk, ns, clk / ns);
}
template <int KIND>
void run_all(unsigned long long* d, int blocks) {
  run<KIND, 0>(d, blocks); run<KIND, 1>(d, blocks); run<KIND, 2>(d, blocks); run<KIND, 3>(d, blocks); run<KIND, 4>(d, blocks); run<KIND, 5>(d, blocks); run<KIND, 6>(d, blocks);
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 256;
  unsigned long long* d;
  hipMalloc(&d, 4096 * 8 * 8);
  printf("blocks = %d (4 waves each, one per SIMD)\n", blocks);
  run<20, 0>(d, blocks); run<21, 0>(d, blocks); run<22, 0>(d, blocks); run<23, 0>(d, blocks); run<24, 0>(d, blocks); run<0, 0>(d, blocks);
  run<30, 0>(d, blocks); run<31, 0>(d, blocks); run<32, 0>(d, blocks); run<30, 0>(d, blocks); run<31, 0>(d, blocks); run<32, 0>(d, blocks);
  if (argc > 2) return 0;
  run_all<0>(d, blocks); run_all<1>(d, blocks); run_all<2>(d, blocks); run_all<3>(d, blocks); run_all<4>(d, blocks); run_all<5>(d, blocks);
  run_all<6>(d, blocks); run_all<7>(d, blocks); run_all<8>(d, blocks); run_all<9>(d, blocks); run_all<10>(d, blocks); run_all<11>(d, blocks);
  return 0;
}
