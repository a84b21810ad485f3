#!/usr/bin/env python3
"""Generates quick_amd/csrc/w4a16_xw_loop.inc: the hand-placed K loop of the 128 x 256 four-wave kernel (w4a16_xw.hpp) as ONE inline-asm
statement -- prologue, a peeled first k16 step (accumulators start from the constant 0), and the loop unrolled over four stages
(x ring slot and weight queue set are static per copy).

    python tools/gen_xw_loop.py            # rewrites the .inc (checked in; the build does not run this script)

Why asm: one wave per SIMD hides at most ~5 issue slots under a 32-cycle v_mfma_f32_32x32x16_f16 (MI355X_MICROARCH.md, "one wave per SIMD"),
and the work next to an MFMA here is 3.25 VALU of dequantisation + 0.5 ds_read_b128 + ~0.5 vector-memory / scalar instructions.  hipcc's
schedule of the same work (w4a16_wide_kernel<4, 2>) carries 42 s_nop and 24 s_waitcnt per 64 MFMAs, 5.9 issue slots per MFMA, and measures
45.8 clocks per MFMA [r04, profiles/r04_base_phases.txt].  Here every filler has its slot, dependent VALU pairs are >= 2 instructions
apart (no pads needed), M0 / soffset values come from scalar adds placed a slot ahead of their use.

Pipeline per wave (wave wn owns all 128 tokens x channels 64 wn .. 64 wn + 63 of the tile; a stage is 128 k = 8 k16 steps x 8 MFMAs):
  x        LDS-DMA into a ring of four 32 KiB slots; stage s issues X(s + 3) (8 pieces) into the slot stage s - 1 released;
  weights  HBM -> VGPR queue of four sets (lo / hi dwordx4 per 32-channel pair + the (scale, zero) word), stage s issues W(s + 3) behind its
           x pieces, so the counted wait "X(s + 2) has landed" is vmcnt(6 + 14) and never waits for HBM-cold weights issued later;
  one s_waitcnt vmcnt(20) + s_barrier per stage, in the middle of k16 step 7 (four MFMAs are in the pipe while the waves meet);
  B fragments (tokens) ds_read_b128 one k16 step ahead, across the stage boundary; A fragments (weights) dequantised one step ahead.

Register plan (fixed registers are CLOBBERS of the one statement: nothing of them lives outside it):
  v[96:167]   weight queue, set j at 96 + 18 j: lo0[4] hi0[4] lo1[4] hi1[4] sz0 sz1
  v[168:199]  B fragments [buf 2][mt 4][4]          v[200:215]  A fragments [buf 2][pair 2][4]
  v[216:227]  group constants [set 2][pair 2]{s2, nzlo, nzhi}
  v[228:231]  temps           v[232:239] / v[240:247]  LDS read addresses per k16 step (slots 0-1 / slots 2-3)      v248  0x64006400
  s[52:71]    masks, constants, stage counter, offsets
Operands (symbolic): a0-a7 accumulators acc[pair][mt] ("=&a"), rsx rsw rss x / weight / group-word descriptors, xv0-xv7 x piece offsets, wv
weight offset, sv group-word offset, xrd LDS read address of the lane, xdst LDS-DMA destination of piece 0 in slot 0, ktlo / kthi first / end k
tile (ktlo carries log2(k tiles per group) in bits 24..), wps / sps byte strides between the wave's two pairs (weights / group words);
stamped build: t0, t1 ("=s", 64 bit: s_memrealtime behind the prologue's barrier / at the end of the loop), clk (shader clocks between the two).
"""
import os

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "quick_amd", "csrc", "w4a16_xw_loop.inc")

# ---- registers
def Q(j, what, p=0, t=0):
    base = 96 + 18 * j
    if what == "lo": return base + 8 * p + t
    if what == "hi": return base + 8 * p + 4 + t
    if what == "sz": return base + 16 + p
    raise ValueError(what)
def BF(b, mt): return 168 + (b * 4 + mt) * 4
def AF(b, p): return 200 + (b * 2 + p) * 4
def GC(c, p, i): return 216 + (c * 2 + p) * 3 + i   # i: 0 s2, 1 nzlo, 2 nzhi
T = [228, 229]
ZT = [230, 231]
RA = [232 + k for k in range(8)]
RA2 = [240 + k for k in range(8)]
MAGIC = 248
S_MLO, S_MHI, S_SIXT, S_D400, S_PERM = 52, 53, 54, 55, 56
S_S, S_N, S_KT, S_XSO, S_WSO0, S_WSO1, S_SSO0, S_SSO1, S_KTMAX, S_G = 57, 58, 59, 60, 61, 62, 63, 64, 65, 66
STAMP0, STAMP1 = 68, 70   # s[68:69], s[70:71]
S_KTLO, S_TPG, CLK0, CLK1 = 67, 72, 74, 76

ACC = lambda p, mt: "%%[a%d]" % (p * 4 + mt)
RSX, RSW, RSS = "%[rsx]", "%[rsw]", "%[rss]"
XV = lambda i: "%%[xv%d]" % i
WV, SV, XRD, XDST, KTLO, KTHI, WPS, SPS = "%[wv]", "%[sv]", "%[xrd]", "%[xdst]", "%[ktlo]", "%[kthi]", "%[wps]", "%[sps]"

def v(r, n=1): return "v%d" % r if n == 1 else "v[%d:%d]" % (r, r + n - 1)
def s(r): return "s%d" % r

# ---- instruction records: (text, kind, writes, reads) with register sets for the distance check
class I:
    def __init__(self, text, kind, w=(), r=()):
        self.text, self.kind, self.w, self.r = text, kind, set(w), set(r)

def dequant_chain(qreg, gc, out, tmp):
    """13 VALU ops: packed dword v[qreg] -> 4 registers v[out:out+3] (8 fp16 weights = fp16((w - z) * s)), constants gc = (s2, nzlo, nzhi)."""
    s2, nzlo, nzhi = gc
    o = [out + i for i in range(4)]
    ops = [
        I(f"v_lshrrev_b32 {v(tmp)}, 8, {v(qreg)}", "valu", [tmp], [qreg]),
        I(f"v_and_or_b32 {v(o[0])}, {v(qreg)}, {s(S_MLO)}, {v(MAGIC)}", "valu", [o[0]], [qreg]),
        I(f"v_and_or_b32 {v(o[1])}, {v(qreg)}, {s(S_MHI)}, {v(MAGIC)}", "valu", [o[1]], [qreg]),
        I(f"v_and_or_b32 {v(o[2])}, {v(tmp)}, {s(S_MLO)}, {v(MAGIC)}", "valu", [o[2]], [tmp]),
        I(f"v_and_or_b32 {v(o[3])}, {v(tmp)}, {s(S_MHI)}, {v(MAGIC)}", "valu", [o[3]], [tmp]),
        I(f"v_pk_add_f16 {v(o[0])}, {v(o[0])}, {v(nzlo)}", "valu", [o[0]], [o[0], nzlo]),
        I(f"v_pk_fma_f16 {v(o[1])}, {v(o[1])}, {s(S_SIXT)}, {v(nzhi)}", "valu", [o[1]], [o[1], nzhi]),
        I(f"v_pk_add_f16 {v(o[2])}, {v(o[2])}, {v(nzlo)}", "valu", [o[2]], [o[2], nzlo]),
        I(f"v_pk_fma_f16 {v(o[3])}, {v(o[3])}, {s(S_SIXT)}, {v(nzhi)}", "valu", [o[3]], [o[3], nzhi]),
        I(f"v_pk_mul_f16 {v(o[0])}, {v(o[0])}, {v(s2)}", "valu", [o[0]], [o[0], s2]),
        I(f"v_pk_mul_f16 {v(o[1])}, {v(o[1])}, {v(s2)}", "valu", [o[1]], [o[1], s2]),
        I(f"v_pk_mul_f16 {v(o[2])}, {v(o[2])}, {v(s2)}", "valu", [o[2]], [o[2], s2]),
        I(f"v_pk_mul_f16 {v(o[3])}, {v(o[3])}, {v(s2)}", "valu", [o[3]], [o[3], s2]),
    ]
    return ops

def interleave(a, b):
    """a0 b0 a1 b1 ...: dependent ops of one chain end up two apart"""
    out = []
    for x, y in zip(a, b):
        out += [x, y]
    return out

def group_consts(szreg, c, p, zt):
    """(scale | zero << 16) word -> (s, s), -(1024 + z) twice, -(64 + z) twice: 5 VALU (w4a16_common.hpp make_group)"""
    s2, nzlo, nzhi = GC(c, p, 0), GC(c, p, 1), GC(c, p, 2)
    return [
        I(f"v_perm_b32 {v(s2)}, {v(szreg)}, {v(szreg)}, {s(S_PERM)}", "valu", [s2], [szreg]),
        I(f"v_lshrrev_b32 {v(zt)}, 16, {v(szreg)}", "valu", [zt], [szreg]),
        I(f"v_lshl_or_b32 {v(zt)}, {v(zt)}, 16, {v(zt)}", "valu", [zt], [zt]),
        I(f"v_or_b32 {v(nzlo)}, 0xe400e400, {v(zt)}", "valu", [nzlo], [zt]),
        I(f"v_lshl_or_b32 {v(nzhi)}, {v(zt)}, 4, {s(S_D400)}", "valu", [nzhi], [zt]),
    ]

def b_read(buf, mt, kk, slot):
    ra = (RA if slot < 2 else RA2)[kk]
    off = (slot % 2) * 32768 + mt * 8192
    return I(f"ds_read_b128 {v(BF(buf, mt), 4)}, {v(ra)} offset:{off}", "lds", range(BF(buf, mt), BF(buf, mt) + 4), [ra])

def mfma(p, mt, abuf, bbuf, zero_c=False):
    a, b = AF(abuf, p), BF(bbuf, mt)
    c = "0" if zero_c else ACC(p, mt)
    return I(f"v_mfma_f32_32x32x16_f16 {ACC(p, mt)}, {v(a, 4)}, {v(b, 4)}, {c}", "mfma", [], list(range(a, a + 4)) + list(range(b, b + 4)))

def x_piece(i, slot, pad=False):
    """two instructions: M0 <- LDS destination of piece i in `slot`, then the LDS-DMA (1 KiB: 4 token rows x 256 B); one wait state
    between the two (pad: an s_nop where no filler sits between them)"""
    return ([I(f"s_add_u32 m0, {XDST}, {slot * 32768 + i * 4096}", "salu", ["m0"], [])] + ([I("s_nop 0", "salu")] if pad else []) +
            [I(f"buffer_load_dwordx4 {XV(i)}, {RSX}, {s(S_XSO)} offen lds", "vmem", [], ["m0"])])

def w_loads(j):
    return [
        I(f"buffer_load_dwordx4 {v(Q(j, 'lo', 0), 4)}, {WV}, {RSW}, {s(S_WSO0)} offen", "vmem"),
        I(f"buffer_load_dwordx4 {v(Q(j, 'hi', 0), 4)}, {WV}, {RSW}, {s(S_WSO0)} offen offset:512", "vmem"),
        I(f"buffer_load_dwordx4 {v(Q(j, 'lo', 1), 4)}, {WV}, {RSW}, {s(S_WSO1)} offen", "vmem"),
        I(f"buffer_load_dwordx4 {v(Q(j, 'hi', 1), 4)}, {WV}, {RSW}, {s(S_WSO1)} offen offset:512", "vmem"),
        I(f"buffer_load_dword {v(Q(j, 'sz', 0))}, {SV}, {RSS}, {s(S_SSO0)} offen", "vmem"),
        I(f"buffer_load_dword {v(Q(j, 'sz', 1))}, {SV}, {RSS}, {s(S_SSO1)} offen", "vmem"),
    ]

def offsets(ahead):
    """scalar offsets of stage S_S + ahead (clamped to the last k tile of the slice: past the end the loads replay it, nobody uses them)"""
    return [
        I(f"s_add_u32 {s(S_KT)}, {s(S_S)}, {ahead}", "salu"),
        I(f"s_add_u32 {s(S_KT)}, {s(S_KT)}, {s(S_KTLO)}", "salu"),
        I(f"s_min_u32 {s(S_KT)}, {s(S_KT)}, {s(S_KTMAX)}", "salu"),
        I(f"s_lshl_b32 {s(S_XSO)}, {s(S_KT)}, 8", "salu"),
        I(f"s_lshl_b32 {s(S_WSO0)}, {s(S_KT)}, 10", "salu"),
        I(f"s_add_u32 {s(S_WSO1)}, {s(S_WSO0)}, {WPS}", "salu"),
        I(f"s_lshr_b32 {s(S_G)}, {s(S_KT)}, {s(S_TPG)}", "salu"),
        I(f"s_lshl_b32 {s(S_SSO0)}, {s(S_G)}, 6", "salu"),
        I(f"s_add_u32 {s(S_SSO1)}, {s(S_SSO0)}, {SPS}", "salu"),
    ]

def weight_dword(j, p, kk):
    t = kk >> 1
    return Q(j, "lo" if kk % 2 == 0 else "hi", p, t)

def step(J, kk, zero_c=False, first_of_kernel=False):
    """k16 step kk of a stage whose x slot / weight set is J (= stage % 4), constants set C = J % 2.
    Returns the instruction list: 8 MFMAs, each followed by its fillers."""
    C = J % 2
    cb, nb = kk % 2, (kk + 1) % 2
    nJ, nC = (J + 1) % 4, 1 - C
    # --- fillers
    if kk < 7:
        reads = [b_read(nb, mt, kk + 1, J) for mt in range(4)]
        dq = interleave(dequant_chain(weight_dword(J, 0, kk + 1), [GC(C, 0, i) for i in range(3)], AF(nb, 0), T[0]),
                        dequant_chain(weight_dword(J, 1, kk + 1), [GC(C, 1, i) for i in range(3)], AF(nb, 1), T[1]))
    else:  # prepare step 0 of the next stage: its slot / set / constants
        reads = [b_read(nb, mt, 0, nJ) for mt in range(4)]
        dq = interleave(dequant_chain(weight_dword(nJ, 0, 0), [GC(nC, 0, i) for i in range(3)], AF(nb, 0), T[0]),
                        dequant_chain(weight_dword(nJ, 1, 0), [GC(nC, 1, i) for i in range(3)], AF(nb, 1), T[1]))
    extras = [[] for _ in range(8)]   # per gap, placed behind the dequantisation ops of the gap
    if first_of_kernel:
        # the peeled first step also issues what the prologue left out -- X(2), W(2) -- so that the prologue's own issue (a CU's
        # vector-memory path moves 64 B per clock) is over before its first wait: then X(3)'s first two pieces as every step 0
        e = []
        for i in range(8): e += x_piece(i, 2, True)
        e += w_loads(2) + offsets(3) + x_piece(0, 3, True) + x_piece(1, 3, True)
        for i, op in enumerate(e):
            extras[i * 8 // len(e)] += [op]
    elif kk <= 3:      # X(s + 3): two pieces per step, into the slot stage s - 1 released = (J + 3) % 4
        pa, pb = x_piece(2 * kk, (J + 3) % 4), x_piece(2 * kk + 1, (J + 3) % 4)
        extras[0] += [pa[0]]; extras[1] += [pa[1]]; extras[2] += [pb[0]]; extras[3] += [pb[1]]
    elif kk in (4, 5):   # W(s + 3) into set (J + 3) % 4, behind the x pieces
        wl = w_loads((J + 3) % 4)[(kk - 4) * 3:(kk - 4) * 3 + 3]
        extras[1] += [wl[0]]; extras[3] += [wl[1]]; extras[5] += [wl[2]]
    elif kk == 6:    # the next stage's group constants (its weight set landed before the barrier of the previous stage)
        g = interleave(group_consts(Q(nJ, "sz", 0), nC, 0, ZT[0]), group_consts(Q(nJ, "sz", 1), nC, 1, ZT[1]))
        for i, op in enumerate(g):
            extras[i * 8 // len(g)] += [op]
    else:            # kk == 7: stage counter, the next stage's load offsets, the counted wait + barrier
        sal = [I(f"s_add_u32 {s(S_S)}, {s(S_S)}, 1", "salu")] + offsets(3)
        extras[0] += sal[0:2]; extras[1] += sal[2:4]; extras[2] += sal[4:6]
        extras[3] += [I("s_waitcnt vmcnt(20)", "wait"), I("s_barrier", "wait")]
        extras[4] += sal[6:8]; extras[5] += sal[8:10]
    # dequantisation ops per gap (26): chain a (pair 0) must be complete two slots before the next step's first MFMA
    per_gap = [2, 2, 2, 2, 4, 4, 5, 5]
    out = []
    di = 0
    if kk == 6:   # W(s + 1) has landed: behind it X(s + 2) W(s + 2) X(s + 3) W(s + 3) = 28 (its group constants are made in this step)
        out.append(I("s_waitcnt vmcnt(28)", "wait"))
    for g in range(8):
        p, mt = g // 4, g % 4
        if g == 0 and not first_of_kernel:
            out.append(I("s_waitcnt lgkmcnt(0)", "wait"))   # this step's B fragments (requested during the previous step)
        out.append(mfma(p, mt, cb, cb, zero_c))
        if g < 4:
            out.append(reads[g])
        out += dq[di:di + per_gap[g]]
        di += per_gap[g]
        out += extras[g]
    assert di == len(dq) == 26
    return out

def check(seq, name):
    """dependent VALU pairs at least two instructions apart; an M0 write at least one instruction before its LDS-DMA; a VALU result at
    least two instructions before an MFMA reads it (the wait states hipcc would pad: the asm string gets none)"""
    last_w = {}
    for idx, ins in enumerate(seq):
        for r in ins.r:
            if r in last_w:
                d = idx - last_w[r][0]
                need = 3 if ins.kind == "mfma" else 2
                if last_w[r][1] == "lds":
                    continue   # covered by s_waitcnt lgkmcnt
                assert d >= need, f"{name}: '{seq[last_w[r][0]].text}' -> '{ins.text}' only {d} apart"
        for r in ins.w:
            last_w[r] = (idx, ins.kind)

def emit(seq):
    return "".join('  "%s\\n\\t"\n' % i.text for i in seq)

def build(stamped):
    L = []   # list of (label or None, [instructions])
    pro = []
    pro += [I(f"s_mov_b32 {s(S_MLO)}, 0x000f000f", "salu"), I(f"s_mov_b32 {s(S_MHI)}, 0x00f000f0", "salu"),
            I(f"s_mov_b32 {s(S_SIXT)}, 0x2c002c00", "salu"), I(f"s_mov_b32 {s(S_D400)}, 0xd400d400", "salu"),
            I(f"s_mov_b32 {s(S_PERM)}, 0x01000100", "salu"), I(f"s_mov_b32 {s(S_S)}, 0", "salu"),
            I(f"s_lshr_b32 {s(S_TPG)}, {KTLO}, 24", "salu"), I(f"s_and_b32 {s(S_KTLO)}, {KTLO}, 0xffffff", "salu"),
            I(f"s_sub_u32 {s(S_N)}, {KTHI}, {s(S_KTLO)}", "salu"), I(f"s_sub_u32 {s(S_KTMAX)}, {KTHI}, 1", "salu")]
    # W(0) X(0) X(1) W(1): what stage 0 needs first (X(2) W(2) follow from the peeled step: issue order W(0) X(0) X(1) W(1) X(2) W(2) X(3) ...)
    def pro_offsets(q):
        return offsets(q)
    pro += pro_offsets(0) + w_loads(0)
    for i in range(8): pro += x_piece(i, 0, True)
    pro += pro_offsets(1)
    for i in range(8): pro += x_piece(i, 1, True)
    pro += w_loads(1)
    pro += pro_offsets(2)
    # LDS read addresses per k16 step, the magic constant (while the loads fly)
    pro += [I(f"v_mov_b32 {v(MAGIC)}, 0x64006400", "valu", [MAGIC])]
    for kk in range(8):
        pro += [I(f"v_xor_b32 {v(RA[kk])}, {kk << 5}, {XRD}", "valu", [RA[kk]])]
    for kk in range(8):
        pro += [I(f"v_add_u32 {v(RA2[kk])}, 0x10000, {v(RA[kk])}", "valu", [RA2[kk]], [RA[kk]])]
    pro += [I("s_waitcnt vmcnt(6)", "wait"), I("s_barrier", "wait")]   # W(0), X(0), X(1) have landed (W(1) may be on its way), in every wave
    if stamped:
        pro += [I(f"s_memrealtime s[{STAMP0}:{STAMP0 + 1}]", "salu"), I(f"s_memtime s[{CLK0}:{CLK0 + 1}]", "salu")]
    pro += interleave(group_consts(Q(0, "sz", 0), 0, 0, ZT[0]), group_consts(Q(0, "sz", 1), 0, 1, ZT[1]))
    pro += [b_read(0, mt, 0, 0) for mt in range(4)]
    pro += interleave(dequant_chain(weight_dword(0, 0, 0), [GC(0, 0, i) for i in range(3)], AF(0, 0), T[0]),
                      dequant_chain(weight_dword(0, 1, 0), [GC(0, 1, i) for i in range(3)], AF(0, 1), T[1]))
    pro += [I("s_nop 1", "salu"), I("s_waitcnt lgkmcnt(0)", "wait")]
    check(pro, "prologue")
    text = emit(pro)
    peel = step(0, 0, zero_c=True, first_of_kernel=True)
    check(pro[-40:] + peel, "peel")
    text += emit(peel)
    text += '  "s_branch .Lxw_j0k1_%=\\n\\t"\n'
    text += '  ".Lxw_loop_%=:\\n\\t"\n'
    prev = peel
    for J in range(4):
        for kk in range(8):
            st = step(J, kk)
            check(prev[-30:] + st, f"J{J} k{kk}")
            if J == 0 and kk == 1:
                text += '  ".Lxw_j0k1_%=:\\n\\t"\n'
            text += emit(st)
            prev = st
        # end of the stage: done?
        text += '  "s_cmp_ge_u32 %s, %s\\n\\t"\n' % (s(S_S), s(S_N))
        if J < 3:
            text += '  "s_cbranch_scc1 .Lxw_done_%=\\n\\t"\n'
        else:
            text += '  "s_cbranch_scc0 .Lxw_loop_%=\\n\\t"\n'
    # wrap-around check: the last step of J = 3 followed by the first of J = 0
    check(step(3, 7)[-30:] + step(0, 0), "wrap")
    text += '  ".Lxw_done_%=:\\n\\t"\n'
    if stamped:
        text += '  "s_memrealtime s[%d:%d]\\n\\t"\n  "s_memtime s[%d:%d]\\n\\t"\n' % (STAMP1, STAMP1 + 1, CLK1, CLK1 + 1)
    text += '  "s_waitcnt vmcnt(0) lgkmcnt(0)\\n\\t"\n'
    if stamped:
        text += '  "s_mov_b64 %%[t0], s[%d:%d]\\n\\t"\n  "s_mov_b64 %%[t1], s[%d:%d]\\n\\t"\n' % (STAMP0, STAMP0 + 1, STAMP1, STAMP1 + 1)
        text += '  "s_sub_u32 %%[clk], s%d, s%d\\n\\t"\n' % (CLK1, CLK0)
    text += '  "s_nop 15\\n\\t"\n  "s_nop 7"\n'   # the last MFMAs' results -> whoever reads the accumulators next
    return text

def clobbers():
    c = ['"memory"', '"scc"', '"m0"'] if False else ['"memory"', '"scc"']
    c += ['"v%d"' % r for r in range(96, 250)]
    c += ['"s%d"' % r for r in range(52, 78)]
    return ", ".join(c)

def main():
    hdr = ("// GENERATED by tools/gen_xw_loop.py -- do not edit.  The K loop of w4a16_xw_kernel as one inline-asm statement\n"
           "// (operands and register plan: see the generator's docstring and w4a16_xw.hpp).\n")
    body = hdr
    body += "#define QA_XW_LOOP_ASM \\\n" + "".join(l + " \\\n" for l in build(False).rstrip("\n").split("\n")) + "\n"
    body += "#define QA_XW_LOOP_ASM_STAMPED \\\n" + "".join(l + " \\\n" for l in build(True).rstrip("\n").split("\n")) + "\n"
    body += "#define QA_XW_LOOP_CLOBBERS " + clobbers() + "\n"
    with open(OUT, "w") as f:
        f.write(body)
    n = build(False).count("\n")
    print("wrote", OUT, n, "lines of asm")

if __name__ == "__main__":
    main()
