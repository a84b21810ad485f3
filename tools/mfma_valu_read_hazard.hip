// Stand-alone reproducer (r06; not product): how many wait states does gfx950 need between an MFMA and a VALU instruction that
// READS its result?  The hardware does not interlock this pair: software (hipcc's hazard recognizer, or whoever writes the asm) pads it.  In the
// four-tile skinny chunk loop with unconditional requests (profiles/r05_skinny_variants.txt) hipcc's allocation ends a tile's MFMA chain with
//     v_mfma_f32_16x16x32_f16 v[34:37], ...   /   v_cvt_f32_f16 v82, v113   /   s_nop 6   /   v_pk_fma_f32 v[36:37], v[80:81], v[104:105], v[36:37]
// i.e. 8 wait states between the MFMA and the first read of its result, and tiles 0 and 1 of every four come out non-finite, every run;
// -amdgpu-waitcnt-forcezero (an s_waitcnt behind every instruction = more wait states) cures it.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_valu_read_hazard tools/mfma_valu_read_hazard.hip && tools/bin/mfma_valu_read_hazard
// The destination is poisoned, the MFMA issued, GAP wait states pass, a VALU copies the destination out; compared with GAP = 40.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
#define N1 "s_nop 0\n\t"
#define V1 "v_mov_b32 v60, v61\n\t"
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
#define R14(x) R12(x) R2(x)
#define R16(x) R8(x) R8(x)
#define R18(x) R16(x) R2(x)
#define R20(x) R16(x) R4(x)
#define R40(x) R20(x) R20(x)
#define POISON4 "v_mov_b32 v40, 0x7fc00000\n\tv_mov_b32 v41, 0x7fc00000\n\tv_mov_b32 v42, 0x7fc00000\n\tv_mov_b32 v43, 0x7fc00000\n\ts_nop 7\n\t"
// SHAPE 16: v_mfma_f32_16x16x32_f16 (4 result registers: all four read); SHAPE 32: v_mfma_f32_32x32x16_f16 (16: the first two and the last two read);
// SHAPE 17: 16x16x32 whose SrcC is the result of an MFMA issued right in front (the end of a chain, as in the kernel)
#define KERN(NAME, GAP, SHAPE)                                                                                                       \
  __global__ void NAME(const half8* a, const half8* b, const floatx16* c, floatx4* out) {                                            \
    const int l = threadIdx.x;                                                                                                       \
    half8 a1 = a[l], b1 = b[l], a2 = a[64 + l], b2 = b[64 + l];                                                                      \
    floatx16 c16 = c[l];                                                                                                             \
    floatx4 c0 = {c16[0], c16[1], c16[2], c16[3]}, r;                                                                                \
    const floatx4 bw = __builtin_bit_cast(floatx4, b1);                                                                               \
    if (SHAPE == 16)                                                                                                                 \
      asm volatile(POISON4 "v_mfma_f32_16x16x32_f16 v[40:43], %[a1], %[b1], %[c0]\n\t" GAP                                           \
                   "v_mov_b32 %[r0], v40\n\tv_mov_b32 %[r1], v41\n\tv_mov_b32 %[r2], v42\n\tv_mov_b32 %[r3], v43\n\t" R40(N1)        \
                   : [r0] "=&v"(r[0]), [r1] "=&v"(r[1]), [r2] "=&v"(r[2]), [r3] "=&v"(r[3]) : [a1] "v"(a1), [b1] "v"(b1), [c0] "v"(c0) \
                   : "v40", "v41", "v42", "v43", "v60", "v61");                                                                      \
    else if (SHAPE == 17)                                                                                                            \
      asm volatile(POISON4 "v_mfma_f32_16x16x32_f16 v[44:47], %[a2], %[b2], %[c0]\n\tv_mfma_f32_16x16x32_f16 v[40:43], %[a1], %[b1], v[44:47]\n\t" GAP \
                   "v_mov_b32 %[r0], v40\n\tv_mov_b32 %[r1], v41\n\tv_mov_b32 %[r2], v42\n\tv_mov_b32 %[r3], v43\n\t" R40(N1)        \
                   : [r0] "=&v"(r[0]), [r1] "=&v"(r[1]), [r2] "=&v"(r[2]), [r3] "=&v"(r[3])                                           \
                   : [a1] "v"(a1), [b1] "v"(b1), [c0] "v"(c0), [a2] "v"(a2), [b2] "v"(b2)                                             \
                   : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v60", "v61");                                          \
    else if (SHAPE == 18) /* the other direction: four VALU instructions write the SrcB registers, GAP, the MFMA reads them */                      \
      asm volatile("v_mov_b32 v44, 0x7fc00000\n\tv_mov_b32 v45, 0x7fc00000\n\tv_mov_b32 v46, 0x7fc00000\n\tv_mov_b32 v47, 0x7fc00000\n\ts_nop 7\n\t"  \
                   "v_mov_b32 v44, %[s0]\n\tv_mov_b32 v45, %[s1]\n\tv_mov_b32 v46, %[s2]\n\tv_mov_b32 v47, %[s3]\n\t" GAP                               \
                   "v_mfma_f32_16x16x32_f16 v[40:43], %[a1], v[44:47], %[c0]\n\t" R40(N1)                                                               \
                   "v_mov_b32 %[r0], v40\n\tv_mov_b32 %[r1], v41\n\tv_mov_b32 %[r2], v42\n\tv_mov_b32 %[r3], v43\n\t"                                    \
                   : [r0] "=&v"(r[0]), [r1] "=&v"(r[1]), [r2] "=&v"(r[2]), [r3] "=&v"(r[3])                                                              \
                   : [a1] "v"(a1), [c0] "v"(c0), [s0] "v"(bw[0]), [s1] "v"(bw[1]), [s2] "v"(bw[2]), [s3] "v"(bw[3])                                       \
                   : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v60", "v61");                                                               \
    else if (SHAPE == 19) /* ... the SrcC registers */                                                                                          \
      asm volatile("v_mov_b32 v44, 0x7fc00000\n\tv_mov_b32 v45, 0x7fc00000\n\tv_mov_b32 v46, 0x7fc00000\n\tv_mov_b32 v47, 0x7fc00000\n\ts_nop 7\n\t"  \
                   "v_mov_b32 v44, %[s0]\n\tv_mov_b32 v45, %[s1]\n\tv_mov_b32 v46, %[s2]\n\tv_mov_b32 v47, %[s3]\n\t" GAP                               \
                   "v_mfma_f32_16x16x32_f16 v[40:43], %[a1], %[b1], v[44:47]\n\t" R40(N1)                                                               \
                   "v_mov_b32 %[r0], v40\n\tv_mov_b32 %[r1], v41\n\tv_mov_b32 %[r2], v42\n\tv_mov_b32 %[r3], v43\n\t"                                    \
                   : [r0] "=&v"(r[0]), [r1] "=&v"(r[1]), [r2] "=&v"(r[2]), [r3] "=&v"(r[3])                                                              \
                   : [a1] "v"(a1), [b1] "v"(b1), [s0] "v"(c0[0]), [s1] "v"(c0[1]), [s2] "v"(c0[2]), [s3] "v"(c0[3])                                       \
                   : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v60", "v61");                                                               \
    else                                                                                                                             \
      asm volatile("v_mov_b32 v40, 0x7fc00000\n\tv_mov_b32 v41, 0x7fc00000\n\tv_mov_b32 v54, 0x7fc00000\n\tv_mov_b32 v55, 0x7fc00000\n\ts_nop 7\n\t" \
                   "v_mfma_f32_32x32x16_f16 v[40:55], %[a1], %[b1], %[c16]\n\t" GAP                                                  \
                   "v_mov_b32 %[r0], v40\n\tv_mov_b32 %[r1], v41\n\tv_mov_b32 %[r2], v54\n\tv_mov_b32 %[r3], v55\n\t" R40(N1)        \
                   : [r0] "=&v"(r[0]), [r1] "=&v"(r[1]), [r2] "=&v"(r[2]), [r3] "=&v"(r[3]) : [a1] "v"(a1), [b1] "v"(b1), [c16] "v"(c16) \
                   : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v60", "v61"); \
    out[l] = r;                                                                                                                      \
  }
#define FAM(G) KERN(n16_##G, R##G(N1), 16) KERN(v16_##G, R##G(V1), 16) KERN(c16_##G, R##G(N1), 17) KERN(n32_##G, R##G(N1), 32) KERN(v32_##G, R##G(V1), 32) KERN(wb_##G, R##G(N1), 18) KERN(wc_##G, R##G(N1), 19)
FAM(0) FAM(1) FAM(2) FAM(3) FAM(4) FAM(5) FAM(6) FAM(7) FAM(8) FAM(9) FAM(10) FAM(11) FAM(12) FAM(14) FAM(16) FAM(18) FAM(20) FAM(40)
typedef void (*kern_t)(const half8*, const half8*, const floatx16*, floatx4*);
int main() {
  std::vector<_Float16> ha(128 * 8), hb(128 * 8);
  std::vector<float> hc(64 * 16);
  unsigned s = 777;
  auto rnd = [&] { s = s * 1664525u + 1013904223u; return (int)((s >> 20) % 15) - 7; };
  for (auto& v : ha) v = (_Float16)rnd();
  for (auto& v : hb) v = (_Float16)(rnd() * 0.5f);
  for (auto& v : hc) v = (float)rnd();
  half8 *a, *b; floatx16* c; floatx4* out;
  (void)hipMalloc(&a, 2048); (void)hipMalloc(&b, 2048); (void)hipMalloc(&c, 4096); (void)hipMalloc(&out, 1024);
  (void)hipMemcpy(a, ha.data(), 2048, hipMemcpyHostToDevice); (void)hipMemcpy(b, hb.data(), 2048, hipMemcpyHostToDevice); (void)hipMemcpy(c, hc.data(), 4096, hipMemcpyHostToDevice);
  struct Row { int gap; kern_t k[7]; };
#define ROW(G) {G, {n16_##G, v16_##G, c16_##G, n32_##G, v32_##G, wb_##G, wc_##G}}
  Row rows[] = {ROW(0), ROW(1), ROW(2), ROW(3), ROW(4), ROW(5), ROW(6), ROW(7), ROW(8), ROW(9), ROW(10), ROW(11), ROW(12), ROW(14), ROW(16), ROW(18), ROW(20)};
  kern_t refk[7] = {n16_40, v16_40, c16_40, n32_40, v32_40, wb_40, wc_40};
  std::vector<float> ref[7], got(256);
  for (int m = 0; m < 7; ++m) {
    ref[m].resize(256);
    hipLaunchKernelGGL(refk[m], dim3(1), dim3(64), 0, 0, a, b, c, out);
    (void)hipMemcpy(ref[m].data(), out, 1024, hipMemcpyDeviceToHost);
  }
  printf("MFMA -> GAP wait states -> VALU reads the result; registers (of 64 lanes x 4) that differ from the GAP = 40 run, worst of 100 launches (launches with a difference)\n");
  printf("(last two columns, the other direction: VALU writes SrcB / SrcC of a 16x16x32 MFMA, GAP wait states, the MFMA)\n");
  printf("%4s %24s %24s %30s %24s %24s %24s %24s\n", "gap", "16x16x32_f16, s_nop gap", "16x16x32_f16, VALU gap", "16x16x32 end of a chain, s_nop", "32x32x16_f16, s_nop gap", "32x32x16_f16, VALU gap", "VALU -> MFMA SrcB", "VALU -> MFMA SrcC");
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
      printf(m == 2 ? " %20d (%3d/100)" : " %14d (%3d/100)", worst, launches);
    }
    printf("\n");
  }
  return 0;
}
