// Stand-alone reproducer (r06; not product): is "MFMA 2 reads as SrcC the vDst of MFMA 1 and writes a DIFFERENT vDst" safe on gfx950
// without software wait states, and how many does it need?  This is the one structural difference between the four-tile skinny chunk loop that
// computes right (guarded requests) and the one that computes tiles wrong from run to run (unconditional requests, profiles/r05_skinny_variants.txt):
// in the failing build hipcc's register allocation ends a tile's MFMA chain with
//     v_mfma_f32_16x16x32_f16 v[34:37], v[18:21], v[34:37], v[74:77]      ; SrcC = the previous MFMA's vDst, vDst = SrcB
// behind a handful of independent VALU instructions, where the working build ends every chain on the accumulator it started on.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_srcc_hazard tools/mfma_srcc_hazard.hip && tools/bin/mfma_srcc_hazard
// For every gap G (s_nop wait states, or independent VALU instructions, between the two MFMAs) and every destination choice the second result is
// compared bit for bit with the same pair issued 40 wait states apart into fresh registers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

#define NOP1 "s_nop 0\n\t"
#define VALU1 "v_mov_b32 %[t], %[t]\n\t"
template <int N> struct Rep {
  static constexpr const char* nop() { return ""; }
};
// gap text is built at compile time from string literal concatenation: GAPS(n) expands n copies
#define R0(x)
#define R1(x) x
#define R2(x) x x
#define R3(x) x x x
#define R4(x) R2(x) R2(x)
#define R6(x) R4(x) R2(x)
#define R8(x) R4(x) R4(x)
#define R10(x) R8(x) R2(x)
#define R12(x) R8(x) R4(x)
#define R16(x) R8(x) R8(x)
#define R20(x) R16(x) R4(x)
#define R40(x) R20(x) R20(x)

// MODE 0: vDst2 = SrcB2 (the failing build's shape); 1: vDst2 = SrcA2; 2: vDst2 fresh; 3: vDst2 = SrcC2 = vDst1 (the ordinary accumulate chain)
#define KERNEL16(NAME, GAP, MODE)                                                                                              \
  __global__ void NAME(const half8* a, const half8* b, const floatx4* c, floatx4* out) {                                       \
    const int l = threadIdx.x;                                                                                                 \
    half8 a1 = a[l], a2 = a[64 + l], b1 = b[l], b2 = b[64 + l];                                                                \
    floatx4 c0 = c[l], d1, d2;                                                                                                 \
    unsigned t = l;                                                                                                            \
    if (MODE == 0)                                                                                                             \
      asm volatile("v_mfma_f32_16x16x32_f16 %[d1], %[a1], %[b1], %[c0]\n\t" GAP "v_mfma_f32_16x16x32_f16 %[b2], %[a2], %[b2], %[d1]\n\t" R40(NOP1) \
                   : [d1] "=&v"(d1), [b2] "+v"(b2), [t] "+v"(t) : [a1] "v"(a1), [b1] "v"(b1), [c0] "v"(c0), [a2] "v"(a2));    \
    else if (MODE == 1)                                                                                                        \
      asm volatile("v_mfma_f32_16x16x32_f16 %[d1], %[a1], %[b1], %[c0]\n\t" GAP "v_mfma_f32_16x16x32_f16 %[a2], %[a2], %[b2], %[d1]\n\t" R40(NOP1) \
                   : [d1] "=&v"(d1), [a2] "+v"(a2), [t] "+v"(t) : [a1] "v"(a1), [b1] "v"(b1), [c0] "v"(c0), [b2] "v"(b2));    \
    else if (MODE == 2)                                                                                                        \
      asm volatile("v_mfma_f32_16x16x32_f16 %[d1], %[a1], %[b1], %[c0]\n\t" GAP "v_mfma_f32_16x16x32_f16 %[d2], %[a2], %[b2], %[d1]\n\t" R40(NOP1) \
                   : [d1] "=&v"(d1), [d2] "=&v"(d2), [t] "+v"(t) : [a1] "v"(a1), [b1] "v"(b1), [c0] "v"(c0), [a2] "v"(a2), [b2] "v"(b2)); \
    else                                                                                                                       \
      asm volatile("v_mfma_f32_16x16x32_f16 %[d1], %[a1], %[b1], %[c0]\n\t" GAP "v_mfma_f32_16x16x32_f16 %[d1], %[a2], %[b2], %[d1]\n\t" R40(NOP1) \
                   : [d1] "=&v"(d1), [t] "+v"(t) : [a1] "v"(a1), [b1] "v"(b1), [c0] "v"(c0), [a2] "v"(a2), [b2] "v"(b2));    \
    floatx4 r = MODE == 0 ? __builtin_bit_cast(floatx4, b2) : (MODE == 1 ? __builtin_bit_cast(floatx4, a2) : (MODE == 2 ? d2 : d1));           \
    out[l] = r;                                                                                                                \
    if (t == 0xffffffffu) out[0] = d1;                                                                                         \
  }
#define FAMILY(G, RG)                                                                                \
  KERNEL16(k_nop_##G##_m0, RG(NOP1), 0) KERNEL16(k_nop_##G##_m1, RG(NOP1), 1) KERNEL16(k_nop_##G##_m2, RG(NOP1), 2) \
  KERNEL16(k_nop_##G##_m3, RG(NOP1), 3) KERNEL16(k_valu_##G##_m0, RG(VALU1), 0) KERNEL16(k_valu_##G##_m2, RG(VALU1), 2)
FAMILY(0, R0) FAMILY(1, R1) FAMILY(2, R2) FAMILY(3, R3) FAMILY(4, R4) FAMILY(6, R6) FAMILY(8, R8) FAMILY(10, R10) FAMILY(12, R12) FAMILY(16, R16) FAMILY(20, R20)
KERNEL16(k_ref, R40(NOP1), 2)


typedef void (*kern_t)(const half8*, const half8*, const floatx4*, floatx4*);
int main() {
  std::vector<_Float16> ha(128 * 8), hb(128 * 8);
  std::vector<float> hc(64 * 4);
  unsigned s = 12345;
  auto rnd = [&] { s = s * 1664525u + 1013904223u; return (int)((s >> 20) % 15) - 7; };
  for (auto& v : ha) v = (_Float16)rnd();
  for (auto& v : hb) v = (_Float16)(rnd() * 0.5f);
  for (auto& v : hc) v = (float)rnd();
  half8 *a, *b; floatx4 *c, *out;
  (void)hipMalloc(&a, 128 * 16); (void)hipMalloc(&b, 128 * 16); (void)hipMalloc(&c, 64 * 16); (void)hipMalloc(&out, 64 * 16);
  (void)hipMemcpy(a, ha.data(), 128 * 16, hipMemcpyHostToDevice); (void)hipMemcpy(b, hb.data(), 128 * 16, hipMemcpyHostToDevice);
  (void)hipMemcpy(c, hc.data(), 64 * 16, hipMemcpyHostToDevice);
  std::vector<float> ref(256), got(256);
  hipLaunchKernelGGL(k_ref, dim3(1), dim3(64), 0, 0, a, b, c, out);
  (void)hipMemcpy(ref.data(), out, 1024, hipMemcpyDeviceToHost);
  struct Row { const char* name; int gap; kern_t k[6]; };
#define ROW(G) {#G, G, {k_nop_##G##_m0, k_nop_##G##_m1, k_nop_##G##_m2, k_nop_##G##_m3, k_valu_##G##_m0, k_valu_##G##_m2}}
  Row rows[] = {ROW(0), ROW(1), ROW(2), ROW(3), ROW(4), ROW(6), ROW(8), ROW(10), ROW(12), ROW(16), ROW(20)};
  printf("v_mfma_f32_16x16x32_f16 pair, SrcC of the second = vDst of the first; wrong lanes out of 64 x 4 registers, 200 launches each (0 = bit-equal to the pair 40 wait states apart)\n");
  printf("%4s %18s %18s %18s %22s %22s %22s\n", "gap", "s_nop: vDst=SrcB", "s_nop: vDst=SrcA", "s_nop: vDst fresh", "s_nop: vDst=SrcC (chain)", "VALU gap: vDst=SrcB", "VALU gap: vDst fresh");
  for (auto& r : rows) {
    printf("%4d", r.gap);
    for (int m = 0; m < 6; ++m) {
      int bad = 0, launches_bad = 0;
      for (int it = 0; it < 200; ++it) {
        (void)hipMemset(out, 0, 1024);
        hipLaunchKernelGGL(r.k[m], dim3(1), dim3(64), 0, 0, a, b, c, out);
        (void)hipMemcpy(got.data(), out, 1024, hipMemcpyDeviceToHost);
        int nb = 0;
        for (int i = 0; i < 256; ++i) nb += memcmp(&got[i], &ref[i], 4) != 0;
        bad = nb > bad ? nb : bad;
        launches_bad += nb != 0;
      }
      printf(m < 3 ? " %10d (%3d/200)" : " %14d (%3d/200)", bad, launches_bad);
    }
    printf("\n");
  }
  return 0;
}
