// Stand-alone reproducer (r06; not product): a VALU instruction WRITES a register an MFMA in flight still has business with -- its
// destination (write after write: the MFMA's result lands later and wins) or its SrcC (write after read: the MFMA reads SrcC passes after it
// issued).  gfx950 does not interlock either; hipcc's hazard recognizer pads them -- for VALU instructions it can see.  One inside an asm statement it
// cannot: w4a16_common.hpp's and_or() (v_and_or_b32 as inline asm, the heart of dequant8 / biased8) is exactly that, and in the four-tile skinny
// chunk loop with unconditional requests (profiles/r05_skinny_variants.txt) register allocation put the next tile's weight fragment into the
// accumulator of the chain just issued:
//     v_mfma_f32_16x16x32_f16 v[84:87], v[46:49], v[118:121], v[84:87]
//     v_mfma_f32_16x16x32_f16 v[66:69], v[42:45], v[66:69], v[84:87]
//     v_and_or_b32 v84, v63, s41, v110            <- asm, 1 wait state behind an MFMA that reads v84 (SrcC), 2 behind one that writes it
// -> tiles 0 and 1 of every four non-finite, every run; -amdgpu-waitcnt-forcezero (an s_waitcnt in front of everything = wait states) cured it.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_valu_write_hazard tools/mfma_valu_write_hazard.hip && tools/bin/mfma_valu_write_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
#define N1 "s_nop 0\n\t"
#define R0(x)
#define R1(x) x
#define R2(x) x x
#define R3(x) R2(x) x
#define R4(x) R2(x) R2(x)
#define R5(x) R4(x) x
#define R6(x) R4(x) R2(x)
#define R7(x) R6(x) x
#define R8(x) R4(x) R4(x)
#define R9(x) R8(x) x
#define R10(x) R8(x) R2(x)
#define R11(x) R10(x) x
#define R12(x) R8(x) R4(x)
#define R13(x) R12(x) x
#define R14(x) R12(x) R2(x)
#define R15(x) R14(x) x
#define R16(x) R8(x) R8(x)
#define R18(x) R16(x) R2(x)
#define R20(x) R16(x) R4(x)
#define R40(x) R20(x) R20(x)
#define OUT4(r0, r1, r2, r3) "v_mov_b32 %[r0], " r0 "\n\tv_mov_b32 %[r1], " r1 "\n\tv_mov_b32 %[r2], " r2 "\n\tv_mov_b32 %[r3], " r3 "\n\t"
#define OUTS [r0] "=&v"(r[0]), [r1] "=&v"(r[1]), [r2] "=&v"(r[2]), [r3] "=&v"(r[3])
#define CLOB "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", \
             "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79"
#define SETC4 "v_mov_b32 v44, %[c0]\n\tv_mov_b32 v45, %[c1]\n\tv_mov_b32 v46, %[c2]\n\tv_mov_b32 v47, %[c3]\n\ts_nop 7\n\t"
#define JUNK4 "v_mov_b32 v44, 0x7149f2ca\n\tv_mov_b32 v45, 0x7149f2ca\n\tv_mov_b32 v46, 0x7149f2ca\n\tv_mov_b32 v47, 0x7149f2ca\n\t"   /* 1e30 */
// MODE 0 WAW 16x16x32   1 WAW 32x32x16   2 WAR SrcC 16x16x32   3 WAR SrcC 32x32x16 (first and last register of SrcC)   4 WAR SrcB 16x16x32
//      5 / 6 the kernel's shape: MFMA 1 writes v[44:47], MFMA 2 reads them as SrcC into v[40:43], GAP, VALU writes v44..v47 (5: MFMA 2's result, 6: the VALU's values)
#define KERN(NAME, GAP, MODE)                                                                                                          \
  __global__ void NAME(const half8* a, const half8* b, const floatx16* c, floatx4* out) {                                              \
    const int l = threadIdx.x;                                                                                                         \
    half8 a1 = a[l], b1 = b[l], a2 = a[64 + l], b2 = b[64 + l];                                                                        \
    floatx16 c16 = c[l];                                                                                                               \
    floatx4 c0 = {c16[0], c16[1], c16[2], c16[3]}, r;                                                                                  \
    const floatx4 bw = __builtin_bit_cast(floatx4, b1);                                                                                \
    if (MODE == 0)                                                                                                                     \
      asm volatile("s_nop 7\n\tv_mfma_f32_16x16x32_f16 v[40:43], %[a1], %[b1], %[c0]\n\t" GAP "v_mov_b32 v40, 0x42280000\n\tv_mov_b32 v43, 0x42280000\n\t" R40(N1) \
                   OUT4("v40", "v43", "v40", "v43") : OUTS : [a1] "v"(a1), [b1] "v"(b1), [c0] "v"(c0) : CLOB);                          \
    else if (MODE == 1)                                                                                                                \
      asm volatile("s_nop 7\n\tv_mfma_f32_32x32x16_f16 v[40:55], %[a1], %[b1], %[c16]\n\t" GAP "v_mov_b32 v40, 0x42280000\n\tv_mov_b32 v55, 0x42280000\n\t" R40(N1) \
                   OUT4("v40", "v55", "v40", "v55") : OUTS : [a1] "v"(a1), [b1] "v"(b1), [c16] "v"(c16) : CLOB);                        \
    else if (MODE == 2)                                                                                                                \
      asm volatile(SETC4 "v_mfma_f32_16x16x32_f16 v[40:43], %[a1], %[b1], v[44:47]\n\t" GAP JUNK4 R40(N1) OUT4("v40", "v41", "v42", "v43") \
                   : OUTS : [a1] "v"(a1), [b1] "v"(b1), [c0] "v"(c0[0]), [c1] "v"(c0[1]), [c2] "v"(c0[2]), [c3] "v"(c0[3]) : CLOB);     \
    else if (MODE == 3)                                                                                                                \
      asm volatile("v_mov_b32 v44, %[q0]\n\tv_mov_b32 v45, %[q1]\n\tv_mov_b32 v46, %[q2]\n\tv_mov_b32 v47, %[q3]\n\tv_mov_b32 v48, %[q4]\n\tv_mov_b32 v49, %[q5]\n\t" \
                   "v_mov_b32 v50, %[q6]\n\tv_mov_b32 v51, %[q7]\n\tv_mov_b32 v52, %[q8]\n\tv_mov_b32 v53, %[q9]\n\tv_mov_b32 v54, %[q10]\n\tv_mov_b32 v55, %[q11]\n\t" \
                   "v_mov_b32 v56, %[q12]\n\tv_mov_b32 v57, %[q13]\n\tv_mov_b32 v58, %[q14]\n\tv_mov_b32 v59, %[q15]\n\ts_nop 7\n\t"          \
                   "v_mfma_f32_32x32x16_f16 v[64:79], %[a1], %[b1], v[44:59]\n\t" GAP "v_mov_b32 v44, 0x7149f2ca\n\tv_mov_b32 v59, 0x7149f2ca\n\t" R40(N1) \
                   OUT4("v64", "v67", "v76", "v79") : OUTS                                                                             \
                   : [a1] "v"(a1), [b1] "v"(b1), [q0] "v"(c16[0]), [q1] "v"(c16[1]), [q2] "v"(c16[2]), [q3] "v"(c16[3]), [q4] "v"(c16[4]), [q5] "v"(c16[5]), \
                     [q6] "v"(c16[6]), [q7] "v"(c16[7]), [q8] "v"(c16[8]), [q9] "v"(c16[9]), [q10] "v"(c16[10]), [q11] "v"(c16[11]), [q12] "v"(c16[12]),     \
                     [q13] "v"(c16[13]), [q14] "v"(c16[14]), [q15] "v"(c16[15]) : CLOB);                                               \
    else if (MODE == 4)                                                                                                                \
      asm volatile(SETC4 "v_mfma_f32_16x16x32_f16 v[40:43], %[a1], v[44:47], %[cc]\n\t" GAP JUNK4 R40(N1) OUT4("v40", "v41", "v42", "v43") \
                   : OUTS : [a1] "v"(a1), [cc] "v"(c0), [c0] "v"(bw[0]), [c1] "v"(bw[1]), [c2] "v"(bw[2]), [c3] "v"(bw[3]) : CLOB);     \
    else if (MODE == 5)                                                                                                                \
      asm volatile("s_nop 7\n\tv_mfma_f32_16x16x32_f16 v[44:47], %[a2], %[b2], %[c0]\n\tv_mfma_f32_16x16x32_f16 v[40:43], %[a1], %[b1], v[44:47]\n\t" GAP JUNK4 R40(N1) \
                   OUT4("v40", "v41", "v42", "v43") : OUTS : [a1] "v"(a1), [b1] "v"(b1), [a2] "v"(a2), [b2] "v"(b2), [c0] "v"(c0) : CLOB); \
    else /* 6: the same pair; does what the VALU wrote into MFMA 1's vDst SURVIVE (MFMA 1's result may land behind it)? */              \
      asm volatile("s_nop 7\n\tv_mfma_f32_16x16x32_f16 v[44:47], %[a2], %[b2], %[c0]\n\tv_mfma_f32_16x16x32_f16 v[40:43], %[a1], %[b1], v[44:47]\n\t" GAP JUNK4 R40(N1) \
                   OUT4("v44", "v45", "v46", "v47") : OUTS : [a1] "v"(a1), [b1] "v"(b1), [a2] "v"(a2), [b2] "v"(b2), [c0] "v"(c0) : CLOB); \
    out[l] = r;                                                                                                                        \
  }
#define FAM(G) KERN(k0_##G, R##G(N1), 0) KERN(k1_##G, R##G(N1), 1) KERN(k2_##G, R##G(N1), 2) KERN(k3_##G, R##G(N1), 3) KERN(k4_##G, R##G(N1), 4) KERN(k5_##G, R##G(N1), 5) KERN(k6_##G, R##G(N1), 6)
FAM(0) FAM(1) FAM(2) FAM(3) FAM(4) FAM(5) FAM(6) FAM(7) FAM(8) FAM(9) FAM(10) FAM(11) FAM(12) FAM(13) FAM(14) FAM(15) FAM(16) FAM(18) FAM(20) FAM(40)
typedef void (*kern_t)(const half8*, const half8*, const floatx16*, floatx4*);
int main() {
  std::vector<_Float16> ha(128 * 8), hb(128 * 8);
  std::vector<float> hc(64 * 16);
  unsigned s = 4242;
  auto rnd = [&] { s = s * 1664525u + 1013904223u; return (int)((s >> 20) % 15) - 7; };
  for (auto& v : ha) v = (_Float16)rnd();
  for (auto& v : hb) v = (_Float16)(rnd() * 0.5f);
  for (auto& v : hc) v = (float)rnd();
  half8 *a, *b; floatx16* c; floatx4* out;
  (void)hipMalloc(&a, 2048); (void)hipMalloc(&b, 2048); (void)hipMalloc(&c, 4096); (void)hipMalloc(&out, 1024);
  (void)hipMemcpy(a, ha.data(), 2048, hipMemcpyHostToDevice); (void)hipMemcpy(b, hb.data(), 2048, hipMemcpyHostToDevice); (void)hipMemcpy(c, hc.data(), 4096, hipMemcpyHostToDevice);
  struct Row { int gap; kern_t k[7]; };
#define ROW(G) {G, {k0_##G, k1_##G, k2_##G, k3_##G, k4_##G, k5_##G, k6_##G}}
  Row rows[] = {ROW(0), ROW(1), ROW(2), ROW(3), ROW(4), ROW(5), ROW(6), ROW(7), ROW(8), ROW(9), ROW(10), ROW(11), ROW(12), ROW(13), ROW(14), ROW(15), ROW(16), ROW(18), ROW(20)};
  kern_t refk[7] = {k0_40, k1_40, k2_40, k3_40, k4_40, k5_40, k6_40};
  std::vector<float> ref[7], got(256);
  for (int m = 0; m < 7; ++m) {
    ref[m].resize(256);
    hipLaunchKernelGGL(refk[m], dim3(1), dim3(64), 0, 0, a, b, c, out);
    (void)hipMemcpy(ref[m].data(), out, 1024, hipMemcpyDeviceToHost);
  }
  printf("MFMA, GAP wait states (s_nop), then a VALU WRITES a register the MFMA writes (WAW) or reads as SrcC / SrcB (WAR); values (64 lanes x 4) that differ from the GAP = 40 run, worst of 100 launches\n");
  printf("(last two columns: MFMA 1 writes v[44:47], MFMA 2 takes them as SrcC into other registers, GAP, a VALU writes v44..v47 -- MFMA 2's result / what the VALU wrote)\n");
  printf("%4s %20s %20s %22s %22s %22s %34s %34s\n", "gap", "WAW vDst 16x16x32", "WAW vDst 32x32x16", "WAR SrcC 16x16x32", "WAR SrcC 32x32x16", "WAR SrcB 16x16x32", "chain: MFMA 2's result", "chain: the VALU's values");
  for (auto& r : rows) {
    printf("%4d", r.gap);
    for (int m = 0; m < 7; ++m) {
      int worst = 0, launches = 0;
      for (int it = 0; it < 100; ++it) {
        (void)hipMemset(out, 0, 1024);
        hipLaunchKernelGGL(r.k[m], dim3(1), dim3(64), 0, 0, a, b, c, out);
        (void)hipMemcpy(got.data(), out, 1024, hipMemcpyDeviceToHost);
        int nb = 0;
        for (int i = 0; i < 256; ++i) nb += memcmp(&got[i], &ref[m][i], 4) != 0;
        worst = nb > worst ? nb : worst;
        launches += nb != 0;
      }
      printf(m >= 5 ? " %24d (%3d/100)" : (m >= 2 ? " %12d (%3d/100)" : " %10d (%3d/100)"), worst, launches);
    }
    printf("\n");
  }
  return 0;
}
