// Stand-alone reproducer (r06; not product): the exact instruction shape of the failing skinny build around its first finding --
//     v_mfma_f32_16x16x32_f16 v[84:87], ..., v[84:87]          x L   (an accumulate chain on one vDst, issued back to back)
//     v_mfma_f32_16x16x32_f16 v[66:69], v[42:45], v[66:69], v[84:87]  (SrcC = the chain's vDst, vDst = its own SrcB)
//     v_and_or_b32 v84, ...                                      (a VALU overwrites a register of the chain's vDst at once)
// Questions: is the last MFMA's result right, and does what the VALU wrote survive?  L = 1, 2, 3; GAP wait states in front of the VALU.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_chain_waw_probe tools/mfma_chain_waw_probe.hip && tools/bin/mfma_chain_waw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
#define N1 "s_nop 0\n\t"
#define R0(x)
#define R1(x) x
#define R2(x) x x
#define R3(x) x x x
#define R4(x) R2(x) R2(x)
#define R6(x) R4(x) R2(x)
#define R8(x) R4(x) R4(x)
#define R12(x) R8(x) R4(x)
#define R40(x) R8(x) R8(x) R8(x) R8(x) R8(x)
#define ACC "v_mfma_f32_16x16x32_f16 v[44:47], %[a2], %[b2], v[44:47]\n\t"
#define CH1 "v_mfma_f32_16x16x32_f16 v[44:47], %[a2], %[b2], %[c0]\n\t"
#define CH2 CH1 ACC
#define CH3 CH1 ACC ACC
#define CH4 CH1 ACC ACC ACC
#define SETB "v_mov_b32 v40, %[s0]\n\tv_mov_b32 v41, %[s1]\n\tv_mov_b32 v42, %[s2]\n\tv_mov_b32 v43, %[s3]\n\ts_nop 7\n\t"
#define JUNK "v_mov_b32 v44, 0x42280000\n\tv_mov_b32 v45, 0x42280000\n\tv_mov_b32 v46, 0x42280000\n\tv_mov_b32 v47, 0x42280000\n\t"
// WHAT 0: the last MFMA's result (v[40:43]); 1: the VALU's values (v[44:47])
#define KERN(NAME, CHAIN, GAP, WHAT)                                                                                               \
  __global__ void NAME(const half8* a, const half8* b, const floatx4* c, floatx4* out) {                                          \
    const int l = threadIdx.x & 63;                                                                                                \
    half8 a1 = a[l], b1 = b[l], a2 = a[64 + l], b2 = b[64 + l];                                                                    \
    floatx4 c0 = c[l], r;                                                                                                          \
    const floatx4 bw = __builtin_bit_cast(floatx4, b1);                                                                            \
    asm volatile(SETB CHAIN "v_mfma_f32_16x16x32_f16 v[40:43], %[a1], v[40:43], v[44:47]\n\t" GAP JUNK R40(N1)                      \
                 "v_mov_b32 %[r0], v%c[w0]\n\tv_mov_b32 %[r1], v%c[w1]\n\tv_mov_b32 %[r2], v%c[w2]\n\tv_mov_b32 %[r3], v%c[w3]\n\t"   \
                 : [r0] "=&v"(r[0]), [r1] "=&v"(r[1]), [r2] "=&v"(r[2]), [r3] "=&v"(r[3])                                           \
                 : [a1] "v"(a1), [a2] "v"(a2), [b2] "v"(b2), [c0] "v"(c0), [s0] "v"(bw[0]), [s1] "v"(bw[1]), [s2] "v"(bw[2]), [s3] "v"(bw[3]), \
                   [w0] "i"(WHAT ? 44 : 40), [w1] "i"(WHAT ? 45 : 41), [w2] "i"(WHAT ? 46 : 42), [w3] "i"(WHAT ? 47 : 43)             \
                 : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");                                                         \
    if (threadIdx.x < 64) out[l] = r;                                                                                              \
  }
#define FAM(L, G) KERN(k##L##_##G##_res, CH##L, R##G(N1), 0) KERN(k##L##_##G##_val, CH##L, R##G(N1), 1)
#define FAMS(G) FAM(1, G) FAM(2, G) FAM(3, G) FAM(4, G)
FAMS(0) FAMS(1) FAMS(2) FAMS(3) FAMS(4) FAMS(6) FAMS(8) FAMS(12) FAMS(40)
typedef void (*kern_t)(const half8*, const half8*, const floatx4*, floatx4*);
int main() {
  std::vector<_Float16> ha(128 * 8), hb(128 * 8);
  std::vector<float> hc(64 * 4);
  unsigned s = 31337;
  auto rnd = [&] { s = s * 1664525u + 1013904223u; return (int)((s >> 20) % 15) - 7; };
  for (auto& v : ha) v = (_Float16)rnd();
  for (auto& v : hb) v = (_Float16)(rnd() * 0.5f);
  for (auto& v : hc) v = (float)rnd();
  half8 *a, *b; floatx4 *c, *out;
  (void)hipMalloc(&a, 2048); (void)hipMalloc(&b, 2048); (void)hipMalloc(&c, 1024); (void)hipMalloc(&out, 1024);
  (void)hipMemcpy(a, ha.data(), 2048, hipMemcpyHostToDevice); (void)hipMemcpy(b, hb.data(), 2048, hipMemcpyHostToDevice); (void)hipMemcpy(c, hc.data(), 1024, hipMemcpyHostToDevice);
  struct Row { int gap; kern_t k[8]; };
#define ROW(G) {G, {k1_##G##_res, k1_##G##_val, k2_##G##_res, k2_##G##_val, k3_##G##_res, k3_##G##_val, k4_##G##_res, k4_##G##_val}}
  Row rows[] = {ROW(0), ROW(1), ROW(2), ROW(3), ROW(4), ROW(6), ROW(8), ROW(12)};
  kern_t refk[8] = {k1_40_res, k1_40_val, k2_40_res, k2_40_val, k3_40_res, k3_40_val, k4_40_res, k4_40_val};
  std::vector<float> ref[8], got(256);
  for (int blocks : {1, 256}) {
    for (int m = 0; m < 8; ++m) {
      ref[m].resize(256);
      hipLaunchKernelGGL(refk[m], dim3(1), dim3(64), 0, 0, a, b, c, out);
      (void)hipMemcpy(ref[m].data(), out, 1024, hipMemcpyDeviceToHost);
    }
    printf("%s; per chain length L: wrong values in the last MFMA's result / in what the VALU wrote (of 256), worst of 100 launches\n", blocks == 1 ? "one wave" : "8 waves per CU x 256");
    printf("%4s %22s %22s %22s %22s\n", "gap", "L = 1", "L = 2", "L = 3", "L = 4");
    for (auto& r : rows) {
      printf("%4d", r.gap);
      for (int m = 0; m < 8; ++m) {
        int worst = 0;
        for (int it = 0; it < 100; ++it) {
          (void)hipMemset(out, 0, 1024);
          hipLaunchKernelGGL(r.k[m], dim3(blocks), dim3(blocks == 1 ? 64 : 512), 0, 0, a, b, c, out);
          (void)hipMemcpy(got.data(), out, 1024, hipMemcpyDeviceToHost);
          int nb = 0;
          for (int i = 0; i < 256; ++i) nb += memcmp(&got[i], &ref[m][i], 4) != 0;
          worst = nb > worst ? nb : worst;
        }
        printf(m % 2 == 0 ? " %14d /" : " %5d", worst);
      }
      printf("\n");
    }
  }
  return 0;
}
