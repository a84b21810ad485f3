// Stand-alone reproducer (r06; not product): an accumulate chain of L MFMAs on ONE vDst issued back to back, GAP wait states, a VALU
// reads the result.  Does the distance a reader has to keep from the LAST MFMA of the chain grow with the chain's length?  (hipcc pads MFMA -> VALU
// read by the instruction's pass count, counted from the last MFMA's issue.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_chain_read_probe tools/mfma_chain_read_probe.hip && tools/bin/mfma_chain_read_probe
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
#define R5(x) R4(x) x
#define R6(x) R4(x) R2(x)
#define R7(x) R6(x) x
#define R8(x) R4(x) R4(x)
#define R10(x) R8(x) R2(x)
#define R12(x) R8(x) R4(x)
#define R40(x) R8(x) R8(x) R8(x) R8(x) R8(x)
#define ACC "v_mfma_f32_16x16x32_f16 v[44:47], %[a2], %[b2], v[44:47]\n\t"
#define CH1 "v_mfma_f32_16x16x32_f16 v[44:47], %[a2], %[b2], %[c0]\n\t"
#define CH2 CH1 ACC
#define CH3 CH1 ACC ACC
#define CH4 CH1 ACC ACC ACC
#define CH8 CH4 ACC ACC ACC ACC
#define KERN(NAME, CHAIN, GAP)                                                                                                     \
  __global__ void NAME(const half8* a, const half8* b, const floatx4* c, floatx4* out) {                                          \
    const int l = threadIdx.x & 63;                                                                                                \
    half8 a2 = a[64 + l], b2 = b[64 + l];                                                                                          \
    floatx4 c0 = c[l], r;                                                                                                          \
    asm volatile("v_mov_b32 v44, 0x7fc00000\n\tv_mov_b32 v45, 0x7fc00000\n\tv_mov_b32 v46, 0x7fc00000\n\tv_mov_b32 v47, 0x7fc00000\n\ts_nop 7\n\t" \
                 CHAIN GAP "v_mov_b32 %[r0], v44\n\tv_mov_b32 %[r1], v45\n\tv_mov_b32 %[r2], v46\n\tv_mov_b32 %[r3], v47\n\t" R40(N1)  \
                 : [r0] "=&v"(r[0]), [r1] "=&v"(r[1]), [r2] "=&v"(r[2]), [r3] "=&v"(r[3]) : [a2] "v"(a2), [b2] "v"(b2), [c0] "v"(c0)  \
                 : "v44", "v45", "v46", "v47");                                                                                     \
    if (threadIdx.x < 64) out[l] = r;                                                                                              \
  }
#define FAMS(G) KERN(k1_##G, CH1, R##G(N1)) KERN(k2_##G, CH2, R##G(N1)) KERN(k3_##G, CH3, R##G(N1)) KERN(k4_##G, CH4, R##G(N1)) KERN(k8_##G, CH8, R##G(N1))
FAMS(0) FAMS(1) FAMS(2) FAMS(3) FAMS(4) FAMS(5) FAMS(6) FAMS(7) FAMS(8) FAMS(10) FAMS(12) FAMS(40)
typedef void (*kern_t)(const half8*, const half8*, const floatx4*, floatx4*);
int main() {
  std::vector<_Float16> ha(128 * 8), hb(128 * 8);
  std::vector<float> hc(64 * 4);
  unsigned s = 271828;
  auto rnd = [&] { s = s * 1664525u + 1013904223u; return (int)((s >> 20) % 15) - 7; };
  for (auto& v : ha) v = (_Float16)rnd();
  for (auto& v : hb) v = (_Float16)(rnd() * 0.5f);
  for (auto& v : hc) v = (float)rnd();
  half8 *a, *b; floatx4 *c, *out;
  (void)hipMalloc(&a, 2048); (void)hipMalloc(&b, 2048); (void)hipMalloc(&c, 1024); (void)hipMalloc(&out, 1024);
  (void)hipMemcpy(a, ha.data(), 2048, hipMemcpyHostToDevice); (void)hipMemcpy(b, hb.data(), 2048, hipMemcpyHostToDevice); (void)hipMemcpy(c, hc.data(), 1024, hipMemcpyHostToDevice);
  struct Row { int gap; kern_t k[5]; };
#define ROW(G) {G, {k1_##G, k2_##G, k3_##G, k4_##G, k8_##G}}
  Row rows[] = {ROW(0), ROW(1), ROW(2), ROW(3), ROW(4), ROW(5), ROW(6), ROW(7), ROW(8), ROW(10), ROW(12)};
  kern_t refk[5] = {k1_40, k2_40, k3_40, k4_40, k8_40};
  std::vector<float> ref[5], got(256);
  for (int blocks : {1, 256}) {
    for (int m = 0; m < 5; ++m) {
      ref[m].resize(256);
      hipLaunchKernelGGL(refk[m], dim3(1), dim3(64), 0, 0, a, b, c, out);
      (void)hipMemcpy(ref[m].data(), out, 1024, hipMemcpyDeviceToHost);
    }
    printf("%s; chain of L v_mfma_f32_16x16x32_f16 on one vDst, GAP wait states, VALU reads it: wrong values of 256, worst of 100 launches\n", blocks == 1 ? "one wave" : "8 waves per CU x 256");
    printf("%4s %8s %8s %8s %8s %8s\n", "gap", "L = 1", "L = 2", "L = 3", "L = 4", "L = 8");
    for (auto& r : rows) {
      printf("%4d", r.gap);
      for (int m = 0; m < 5; ++m) {
        int worst = 0;
        for (int it = 0; it < 100; ++it) {
          (void)hipMemset(out, 0, 1024);
          hipLaunchKernelGGL(r.k[m], dim3(blocks), dim3(blocks == 1 ? 64 : 512), 0, 0, a, b, c, out);
          (void)hipMemcpy(got.data(), out, 1024, hipMemcpyDeviceToHost);
          int nb = 0;
          for (int i = 0; i < 256; ++i) nb += memcmp(&got[i], &ref[m][i], 4) != 0;
          worst = nb > worst ? nb : worst;
        }
        printf(" %8d", worst);
      }
      printf("\n");
    }
  }
  return 0;
}
