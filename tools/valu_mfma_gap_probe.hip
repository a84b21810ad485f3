// Stand-alone reproducer (r06; not product): VALU writes a source register of an MFMA, ONE filler instruction, the MFMA.  Which fillers
// make the wait state the pair needs?  (tools/mfma_valu_read_hazard.hip: with nothing in between the MFMA reads the old value; one s_nop 0 is enough.)
// In the failing skinny build every v_and_or_b32 (inline asm) that writes the last register of a B fragment is followed by ONE hipcc-visible
// instruction -- v_lshrrev_b32, v_cvt_f32_u32_sdwa, v_cvt_f32_f16, s_nop -- and then the MFMA; a trailing s_nop inside the asm statement cures the build.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/valu_mfma_gap_probe tools/valu_mfma_gap_probe.hip && tools/bin/valu_mfma_gap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
#define N1 "s_nop 0\n\t"
#define R20(x) x x x x x x x x x x x x x x x x x x x x
#define KERN(NAME, WRITER, FILL)                                                                                                   \
  __global__ void NAME(const half8* a, const half8* b, const floatx4* c, floatx4* out, unsigned mask) {                            \
    const int l = threadIdx.x & 63;                                                                                                \
    half8 a1 = a[l], b1 = b[l];                                                                                                    \
    floatx4 c0 = c[l], r;                                                                                                          \
    const floatx4 bw = __builtin_bit_cast(floatx4, b1);                                                                            \
    unsigned magic = 0;                                                                                                            \
    asm volatile("v_mov_b32 v44, 0x7fc00000\n\tv_mov_b32 v45, 0x7fc00000\n\tv_mov_b32 v46, 0x7fc00000\n\tv_mov_b32 v47, 0x7fc00000\n\tv_mov_b32 v60, %[s3]\n\t" \
                 "v_mov_b32 v61, 0\n\ts_nop 7\n\t"                                                                                 \
                 "v_mov_b32 v44, %[s0]\n\tv_mov_b32 v45, %[s1]\n\tv_mov_b32 v46, %[s2]\n\t" WRITER FILL                            \
                 "v_mfma_f32_16x16x32_f16 v[40:43], %[a1], v[44:47], %[c0]\n\t" R20(N1) R20(N1)                                     \
                 "v_mov_b32 %[r0], v40\n\tv_mov_b32 %[r1], v41\n\tv_mov_b32 %[r2], v42\n\tv_mov_b32 %[r3], v43\n\t"                   \
                 : [r0] "=&v"(r[0]), [r1] "=&v"(r[1]), [r2] "=&v"(r[2]), [r3] "=&v"(r[3])                                           \
                 : [a1] "v"(a1), [c0] "v"(c0), [s0] "v"(bw[0]), [s1] "v"(bw[1]), [s2] "v"(bw[2]), [s3] "v"(bw[3]), [m] "s"(mask), [g] "v"(magic) \
                 : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v60", "v61", "v62", "v63");                             \
    if (threadIdx.x < 64) out[l] = r;                                                                                              \
  }
#define W_MOV "v_mov_b32 v47, v60\n\t"
#define W_ANDOR "v_and_or_b32 v47, v60, %[m], %[g]\n\t"      /* mask = all ones, magic = 0: the same value */
KERN(k_ref, W_MOV, R20(N1))
KERN(k_mov_none, W_MOV, "") KERN(k_mov_nop, W_MOV, N1) KERN(k_mov_valu, W_MOV, "v_mov_b32 v62, v61\n\t") KERN(k_mov_shift, W_MOV, "v_lshrrev_b32 v62, 8, v61\n\t")
KERN(k_mov_sdwa, W_MOV, "v_cvt_f32_u32_sdwa v62, v61 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t") KERN(k_mov_cvt, W_MOV, "v_cvt_f32_f16 v62, v61\n\t")
KERN(k_mov_salu, W_MOV, "s_mov_b32 m0, m0\n\t") KERN(k_mov_war, W_MOV, "v_lshrrev_b32 v60, 8, v60\n\t")
KERN(k_ao_none, W_ANDOR, "") KERN(k_ao_nop, W_ANDOR, N1) KERN(k_ao_valu, W_ANDOR, "v_mov_b32 v62, v61\n\t") KERN(k_ao_shift, W_ANDOR, "v_lshrrev_b32 v62, 8, v61\n\t")
KERN(k_ao_sdwa, W_ANDOR, "v_cvt_f32_u32_sdwa v62, v61 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t") KERN(k_ao_cvt, W_ANDOR, "v_cvt_f32_f16 v62, v61\n\t")
KERN(k_ao_salu, W_ANDOR, "s_mov_b32 m0, m0\n\t") KERN(k_ao_war, W_ANDOR, "v_lshrrev_b32 v60, 8, v60\n\t")
typedef void (*kern_t)(const half8*, const half8*, const floatx4*, floatx4*, unsigned);
int main() {
  std::vector<_Float16> ha(64 * 8), hb(64 * 8);
  std::vector<float> hc(64 * 4);
  unsigned s = 99;
  auto rnd = [&] { s = s * 1664525u + 1013904223u; return (int)((s >> 20) % 15) - 7; };
  for (auto& v : ha) v = (_Float16)rnd();
  for (auto& v : hb) v = (_Float16)(rnd() * 0.5f);
  for (auto& v : hc) v = (float)rnd();
  half8 *a, *b; floatx4 *c, *out;
  (void)hipMalloc(&a, 1024); (void)hipMalloc(&b, 1024); (void)hipMalloc(&c, 1024); (void)hipMalloc(&out, 1024);
  (void)hipMemcpy(a, ha.data(), 1024, hipMemcpyHostToDevice); (void)hipMemcpy(b, hb.data(), 1024, hipMemcpyHostToDevice); (void)hipMemcpy(c, hc.data(), 1024, hipMemcpyHostToDevice);
  std::vector<float> ref(256), got(256);
  hipLaunchKernelGGL(k_ref, dim3(1), dim3(64), 0, 0, a, b, c, out, 0xffffffffu);
  (void)hipMemcpy(ref.data(), out, 1024, hipMemcpyDeviceToHost);
  struct Row { const char* name; kern_t k; };
  Row rows[] = {{"v_mov_b32 writer, nothing in between", k_mov_none}, {"v_mov_b32 writer, s_nop 0", k_mov_nop}, {"v_mov_b32 writer, independent v_mov_b32", k_mov_valu},
                {"v_mov_b32 writer, independent v_lshrrev_b32", k_mov_shift}, {"v_mov_b32 writer, independent v_cvt_f32_u32_sdwa", k_mov_sdwa}, {"v_mov_b32 writer, independent v_cvt_f32_f16", k_mov_cvt},
                {"v_mov_b32 writer, s_mov_b32", k_mov_salu}, {"v_mov_b32 writer, v_lshrrev_b32 overwriting the writer's source", k_mov_war},
                {"v_and_or_b32 writer, nothing in between", k_ao_none}, {"v_and_or_b32 writer, s_nop 0", k_ao_nop}, {"v_and_or_b32 writer, independent v_mov_b32", k_ao_valu},
                {"v_and_or_b32 writer, independent v_lshrrev_b32", k_ao_shift}, {"v_and_or_b32 writer, independent v_cvt_f32_u32_sdwa", k_ao_sdwa}, {"v_and_or_b32 writer, independent v_cvt_f32_f16", k_ao_cvt},
                {"v_and_or_b32 writer, s_mov_b32", k_ao_salu}, {"v_and_or_b32 writer, v_lshrrev_b32 overwriting the writer's source", k_ao_war}};
  for (int blocks : {1, 256})
    for (auto& r : rows) {
      int worst = 0, launches = 0;
      for (int it = 0; it < 100; ++it) {
        (void)hipMemset(out, 0, 1024);
        hipLaunchKernelGGL(r.k, dim3(blocks), dim3(blocks == 1 ? 64 : 512), 0, 0, a, b, c, out, 0xffffffffu);
        (void)hipMemcpy(got.data(), out, 1024, hipMemcpyDeviceToHost);
        int nb = 0;
        for (int i = 0; i < 256; ++i) nb += memcmp(&got[i], &ref[i], 4) != 0;
        worst = nb > worst ? nb : worst;
        launches += nb != 0;
      }
      printf("%-72s %s: %3d wrong values (%3d / 100 launches)\n", r.name, blocks == 1 ? "one wave      " : "8 waves x 256 ", worst, launches);
    }
  return 0;
}
