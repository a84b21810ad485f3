#!/usr/bin/env python3
"""Generates quick_amd/csrc/w4a16_xw_loop.inc: the hand-placed K loops of the four-wave, one-wave-per-SIMD kernels (w4a16_xw.hpp), each as
ONE inline-asm statement -- prologue, a peeled first super-step (accumulators start from the constant 0), and the loop unrolled over NS
stages (x ring slot and weight queue set are static per copy).

    python tools/gen_xw_loop.py            # rewrites the .inc (checked in; the build does not run this script)

Configurations (MB 32-token blocks x PAIRS 32-channel pairs per WAVE; the workgroup tile is MB * 32 tokens x PAIRS * 128 channels):
    (8, 2)  256 x 256   1.6  VALU + 0.5 ds_read_b128 per MFMA    ring / queue of 2 stages (64 KiB slots), super-steps of 16 MFMAs, bar = 1
    (4, 2)  128 x 256   3.25 VALU + 0.5 ds_read_b128 per MFMA    ring / queue of 4 stages
    (4, 1)  128 x 128   3.25 VALU + 1   ds_read_b128 per MFMA    ring / queue of 4 stages
    (2, 1)   64 x 128   6.5  VALU + 1   ds_read_b128 per MFMA    ring / queue of 8 stages, two accumulator sets (even / odd k16 steps: a wave
                                                                  owns only two 32 x 32 tiles, and an MFMA on the accumulator of the one
                                                                  before last would wait for it)

Why asm: one wave per SIMD hides ~5 issue slots under a 32-cycle v_mfma_f32_32x32x16_f16 (MI355X_MICROARCH.md, "one wave per SIMD").  hipcc's
schedule of the (4, 2) work (w4a16_wide_kernel<4, 2>) carries 42 s_nop and 24 s_waitcnt per 64 MFMAs and measures 45.8 clocks per MFMA; this
loop measures 36.5 [r04, tools/xk_phases.py; the microbenchmark tools/mfma_filler_cost.hip prices the same filler mix at 35.6, bare MFMAs at
33.1, and one ds_read_b128 per MFMA at +7].  Every filler has its slot, dependent VALU pairs are >= 2 instructions apart (no pads), M0 and
soffset values come from scalar adds placed a slot ahead of their use, every wait is counted.

Structure.  A stage is 128 k = 8 k16 steps; a SUPER-STEP is 8 MFMAs = SS = 8 / (MB * PAIRS) k16 steps.  While the MFMAs of a super-step run,
the fillers prepare the next one: NRD = MB * SS B fragments (tokens, ds_read_b128, one per gap) and NCH = PAIRS * SS A fragments (weights,
13-op dequantisation chains, interleaved), double-buffered; across the stage boundary the same.  Per stage and wave: X(s + NS - 1) (2 MB
LDS-DMA pieces into the slot stage s - 1 released) then W(s + NS - 1) (lo / hi dwordx4 + the (scale, zero) word per pair, HBM -> VGPR queue);
two counted waits -- "W(s + 1) has landed" before its group constants are made, "X(s + 2) has landed" before the one barrier of the stage,
which sits inside the last super-step with MFMAs in the pipe.

Register plan (fixed registers are CLOBBERS of the one statement: nothing of them lives outside it), from v48 up:
  weight queue NS sets [lo[4] hi[4]] per pair + sz per pair (padded to even) | B fragments [2][NRD][4] | A fragments [2][NCH][4] |
  group constants [2][PAIRS]{s2, nzlo, nzhi} | temps | LDS read addresses per k16 step (two banks of 64 KiB) | 0x64006400;  s[52:77] scalars.
Operands (symbolic): a0.. accumulators ("=&a"; index (kp * PAIRS + pair) * MB + block), rsx rsw rss x / weight / group-word descriptors,
xv0.. x piece offsets, wv weight offset, sv group-word offset, xrd LDS read address of the lane, xdst LDS-DMA destination of piece 0 in slot 0,
ktlo first k tile | log2(k tiles per group) << 24, kthi end k tile, wps / sps byte strides between the wave's pairs (weights / group words);
stamped builds: t0, t1 ("=s", 64 bit: s_memrealtime behind the prologue's barrier / at the end of the loop), clk (shader clocks between them).
"""
import os

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "quick_amd", "csrc", "w4a16_xw_loop.inc")

S_MLO, S_MHI, S_SIXT, S_D400, S_PERM = 52, 53, 54, 55, 56
S_S, S_N, S_KT, S_XSO, S_KTMAX, S_G, S_KTLO = 57, 58, 59, 60, 65, 66, 67
S_WSO = [61, 62]
S_SSO = [63, 64]
STAMP0, STAMP1, S_TPG, CLK0, CLK1 = 68, 70, 72, 74, 76
S_KTHI, S_WPS, S_SPS, S_K2 = 78, 79, 80, 81     # (256-token tiles: end k tile, pair strides and the row pitch of x, derived from the k tile count)
RSX, RSW, RSS = "%[rsx]", "%[rsw]", "%[rss]"
WV, SV, XRD, XDST, KTLO, KTHI, WPS, SPS = "%[wv]", "%[sv]", "%[xrd]", "%[xdst]", "%[ktlo]", "%[kthi]", "%[wps]", "%[sps]"


def v(r, n=1): return "v%d" % r if n == 1 else "v[%d:%d]" % (r, r + n - 1)
def s(r): return "s%d" % r


class I:
    def __init__(self, text, kind, w=(), r=()):
        self.text, self.kind, self.w, self.r = text, kind, set(w), set(r)


def even(n): return n + (n & 1)


class Cfg:
    def __init__(self, MB, PAIRS, NS, KP, exp=0, KW=1, bar=0):
        self.MB, self.PAIRS, self.NS, self.KP, self.KW = MB, PAIRS, NS, KP, KW
        # bar = 1: the barrier of a stage sits in FRONT of the last super-step (whose fillers are the first reads of the next stage), so
        # the only counted wait of a stage is "W(s + 1) has landed" (behind X(s + 1)): a load has a whole stage longer to arrive
        self.bar = bar
        self.exp = exp   # timing experiments (tools builds, wrong results): 1 no barrier, 2 no vector memory in the loop, 4 no dequantisation, 8 no B reads
        self.MPS = MB * PAIRS
        assert self.MPS in (2, 4, 8, 16) and NS % 2 == 0 and (KP == 1 or PAIRS == 1)
        self.SSM = max(8, self.MPS)        # MFMAs per super-step
        self.SS = self.SSM // self.MPS     # k16 steps per super-step
        self.big = MB == 8                 # 256-token tiles: ring of TWO 64 KiB slots (needs bar = 1), x piece offsets in registers
        assert not self.big or (NS == 2 and bar == 1 and KW == 1)
        self.KST = 8 // KW                 # k16 steps of a stage that are this wave's (KW = 2: those of its parity)
        assert KW in (1, 2) and self.KST % self.SS == 0
        self.NSU = self.KST // self.SS     # super-steps per stage
        self.NCH, self.NRD = PAIRS * self.SS, MB * self.SS
        self.NX, self.LW = 2 * MB, (3 if KW == 1 else 2) * PAIRS
        self.L = self.NX + self.LW
        self.QW = 8 // KW                  # packed dwords per pair and stage
        self.QS = even((self.QW + 1) * PAIRS)
        self.SLOTB = MB * 8192
        self.G = self.SSM * self.NSU       # MFMAs (= gaps) per stage
        self.NACC = KP * PAIRS * MB
        b = 48
        self.VQ = b; b += NS * self.QS
        self.VBF = b; b += 2 * self.NRD * 4
        self.VAF = b; b += 2 * self.NCH * 4
        self.VGC = b; b += even(2 * PAIRS * 3)
        self.VT = b; b += even(self.NCH)
        self.VZT = b; b += even(PAIRS)
        self.VRA = b; b += 8
        self.VRA2 = b; b += 8
        self.VMAGIC = b; b += 1
        self.VX = b; b += (self.NX + 1 if self.big else 0)   # byte offsets of the x pieces (rows clamped to M - 1) + a temporary
        self.VEND = b
        assert self.VEND <= 253, self.VEND
        assert NS * self.SLOTB <= 128 * 1024
        self.name = "%d%d%s" % (MB, PAIRS, "k" if KW == 2 else "")

    # registers
    def Q(self, j, what, p=0, t=0):
        base = self.VQ + self.QS * j
        if what == "lo": return base + self.QW * p + t
        if what == "hi" and self.KW == 1: return base + 8 * p + 4 + t
        if what == "sz": return base + self.QW * self.PAIRS + p
        raise ValueError(what)
    def BF(self, b, rd): return self.VBF + (b * self.NRD + rd) * 4
    def AF(self, b, ch): return self.VAF + (b * self.NCH + ch) * 4
    def GC(self, c, p, i): return self.VGC + (c * self.PAIRS + p) * 3 + i
    def ACC(self, kk, p, mt): return "%%[a%d]" % (((kk % self.KP) * self.PAIRS + p) * self.MB + mt)
    def XV(self, i): return "%%[xv%d]" % i

    def dequant_chain(self, qreg, gc, out, tmp):
        """13 VALU ops: packed dword v[qreg] -> v[out:out+3] (8 fp16 weights = fp16((w - z) * s)), constants gc = (s2, nzlo, nzhi)"""
        s2, nzlo, nzhi = gc
        o = [out + i for i in range(4)]
        M = self.VMAGIC
        return [
            I(f"v_lshrrev_b32 {v(tmp)}, 8, {v(qreg)}", "valu", [tmp], [qreg]),
            I(f"v_and_or_b32 {v(o[0])}, {v(qreg)}, {s(S_MLO)}, {v(M)}", "valu", [o[0]], [qreg]),
            I(f"v_and_or_b32 {v(o[1])}, {v(qreg)}, {s(S_MHI)}, {v(M)}", "valu", [o[1]], [qreg]),
            I(f"v_and_or_b32 {v(o[2])}, {v(tmp)}, {s(S_MLO)}, {v(M)}", "valu", [o[2]], [tmp]),
            I(f"v_and_or_b32 {v(o[3])}, {v(tmp)}, {s(S_MHI)}, {v(M)}", "valu", [o[3]], [tmp]),
            I(f"v_pk_add_f16 {v(o[0])}, {v(o[0])}, {v(nzlo)}", "valu", [o[0]], [o[0], nzlo]),
            I(f"v_pk_fma_f16 {v(o[1])}, {v(o[1])}, {s(S_SIXT)}, {v(nzhi)}", "valu", [o[1]], [o[1], nzhi]),
            I(f"v_pk_add_f16 {v(o[2])}, {v(o[2])}, {v(nzlo)}", "valu", [o[2]], [o[2], nzlo]),
            I(f"v_pk_fma_f16 {v(o[3])}, {v(o[3])}, {s(S_SIXT)}, {v(nzhi)}", "valu", [o[3]], [o[3], nzhi]),
            I(f"v_pk_mul_f16 {v(o[0])}, {v(o[0])}, {v(s2)}", "valu", [o[0]], [o[0], s2]),
            I(f"v_pk_mul_f16 {v(o[1])}, {v(o[1])}, {v(s2)}", "valu", [o[1]], [o[1], s2]),
            I(f"v_pk_mul_f16 {v(o[2])}, {v(o[2])}, {v(s2)}", "valu", [o[2]], [o[2], s2]),
            I(f"v_pk_mul_f16 {v(o[3])}, {v(o[3])}, {v(s2)}", "valu", [o[3]], [o[3], s2]),
        ]

    def group_consts(self, szreg, c, p):
        """(scale | zero << 16) word -> (s, s), -(1024 + z) twice, -(64 + z) twice: 5 VALU (w4a16_common.hpp make_group)"""
        s2, nzlo, nzhi = self.GC(c, p, 0), self.GC(c, p, 1), self.GC(c, p, 2)
        zt = self.VZT + p
        return [
            I(f"v_perm_b32 {v(s2)}, {v(szreg)}, {v(szreg)}, {s(S_PERM)}", "valu", [s2], [szreg]),
            I(f"v_lshrrev_b32 {v(zt)}, 16, {v(szreg)}", "valu", [zt], [szreg]),
            I(f"v_lshl_or_b32 {v(zt)}, {v(zt)}, 16, {v(zt)}", "valu", [zt], [zt]),
            I(f"v_or_b32 {v(nzlo)}, 0xe400e400, {v(zt)}", "valu", [nzlo], [zt]),
            I(f"v_lshl_or_b32 {v(nzhi)}, {v(zt)}, 4, {s(S_D400)}", "valu", [nzhi], [zt]),
        ]

    def b_read(self, buf, rd, kk, mt, slot):
        off = slot * self.SLOTB + mt * 8192
        ra = self.VRA + kk if off < 65536 else self.VRA2 + kk
        return I(f"ds_read_b128 {v(self.BF(buf, rd), 4)}, {v(ra)} offset:{off % 65536}", "lds", range(self.BF(buf, rd), self.BF(buf, rd) + 4), [ra])

    def mfma(self, kk, p, mt, buf, ch, rd, zero_c):
        a, b = self.AF(buf, ch), self.BF(buf, rd)
        acc = self.ACC(kk, p, mt)
        return I(f"v_mfma_f32_32x32x16_f16 {acc}, {v(a, 4)}, {v(b, 4)}, {'0' if zero_c else acc}", "mfma", [], list(range(a, a + 4)) + list(range(b, b + 4)))

    def x_piece(self, i, slot, pad=False):
        """M0 <- LDS destination of piece i in `slot`, then the LDS-DMA (1 KiB: 4 token rows x 256 B); one wait state between the two"""
        return ([I(f"s_add_u32 m0, {XDST}, {slot * self.SLOTB + i * 4096}", "salu", ["m0"], [])] + ([I("s_nop 0", "salu")] if pad else []) +
                [I(f"buffer_load_dwordx4 {v(self.VX + i) if self.big else self.XV(i)}, {RSX}, {s(S_XSO)} offen lds", "vmem", [], ["m0"])])

    def w_loads(self, j):
        out = []
        for p in range(self.PAIRS):
            out += [I(f"buffer_load_dwordx4 {v(self.Q(j, 'lo', p), 4)}, {WV}, {RSW}, {s(S_WSO[p])} offen", "vmem")]
            if self.KW == 1:
                out += [I(f"buffer_load_dwordx4 {v(self.Q(j, 'hi', p), 4)}, {WV}, {RSW}, {s(S_WSO[p])} offen offset:512", "vmem")]
        for p in range(self.PAIRS):
            out += [I(f"buffer_load_dword {v(self.Q(j, 'sz', p))}, {SV}, {RSS}, {s(S_SSO[p])} offen", "vmem")]
        return out

    def offsets(self, ahead):
        """scalar offsets of stage S_S + ahead (clamped to the last k tile of the slice: past the end the loads replay it, nobody uses them)"""
        out = [
            I(f"s_add_u32 {s(S_KT)}, {s(S_S)}, {ahead}", "salu"),
            I(f"s_add_u32 {s(S_KT)}, {s(S_KT)}, {s(S_KTLO)}", "salu"),
            I(f"s_min_u32 {s(S_KT)}, {s(S_KT)}, {s(S_KTMAX)}", "salu"),
            I(f"s_lshl_b32 {s(S_XSO)}, {s(S_KT)}, 8", "salu"),
            I(f"s_lshl_b32 {s(S_WSO[0])}, {s(S_KT)}, 10", "salu"),
            I(f"s_lshr_b32 {s(S_G)}, {s(S_KT)}, {s(S_TPG)}", "salu"),
            I(f"s_lshl_b32 {s(S_SSO[0])}, {s(S_G)}, 6", "salu"),
        ]
        if self.PAIRS == 2:
            wps, sps = (s(S_WPS), s(S_SPS)) if self.big else (WPS, SPS)
            out += [I(f"s_add_u32 {s(S_WSO[1])}, {s(S_WSO[0])}, {wps}", "salu"), I(f"s_add_u32 {s(S_SSO[1])}, {s(S_SSO[0])}, {sps}", "salu")]
        return out

    def weight_dword(self, j, p, kk):
        if self.KW == 2: return self.Q(j, "lo", p, kk)   # (the wave's own half of the 32 bytes: wv points at it)
        return self.Q(j, "lo" if kk % 2 == 0 else "hi", p, kk >> 1)

    def prep(self, J, u):
        """fillers that prepare super-step u of the stage whose slot / set is J, into buffer u % 2: (reads, interleaved dequantisation ops)"""
        buf, C = u % 2, J % 2
        reads, chains = [], []
        for rd in range(self.NRD):
            st, mt = rd // self.MB, rd % self.MB
            reads.append(self.b_read(buf, rd, u * self.SS + st, mt, J))
        for ch in range(self.NCH):
            st, p = ch // self.PAIRS, ch % self.PAIRS
            chains.append(self.dequant_chain(self.weight_dword(J, p, u * self.SS + st), [self.GC(C, p, i) for i in range(3)], self.AF(buf, ch), self.VT + ch))
        dq = []
        for i in range(13):
            for c in chains:
                dq.append(c[i])
        return reads, dq

    def superstep(self, J, u, peel=False):
        """the SSM MFMAs of super-step u of stage J, each followed by its fillers"""
        G, NS, SSM = self.G, self.NS, self.SSM
        last = u == self.NSU - 1
        nJ, nu = ((J + 1) % NS, 0) if last else (J, u + 1)
        reads, dq = self.prep(nJ, nu)
        buf = u % 2
        extras = [[] for _ in range(SSM)]
        pre = [[] for _ in range(SSM)]   # placed in front of the gap's dequantisation ops (waits)
        base = u * SSM                   # stage gap index of this super-step's gap 0
        def at(gs):                      # local gap of stage gap gs, or None
            return gs - base if base <= gs < base + SSM else None
        # X(s + NS - 1): piece i, M0 at stage gap gx, DMA at gx + 1, into the slot stage s - 1 released
        xs = (J + NS - 1) % NS
        for i in range(self.NX):
            gx = (i * (G // 2)) // self.NX
            pc = self.x_piece(i, xs)
            if at(gx) is not None: extras[at(gx)] += [pc[0]]
            if at(gx + 1) is not None: extras[at(gx + 1)] += [pc[1]]
        # W(s + NS - 1) behind the x pieces
        wl = self.w_loads(xs)
        for j, op in enumerate(wl):
            gw = G // 2 + (j * (G // 4)) // self.LW
            if at(gw) is not None: extras[at(gw)] += [op]
        # how many of this stage's vector-memory instructions have been issued before the fillers of stage gap g
        def issued_before(g):
            n = 0
            for i in range(self.NX):
                if (i * (G // 2)) // self.NX + 1 < g: n += 1
            for j in range(self.LW):
                if G // 2 + (j * (G // 4)) // self.LW < g: n += 1
            return n
        # the next stage's group constants in the four gaps before the last super-step, behind "W(s + 1) has landed"
        gc0 = G - SSM - 4
        if at(gc0) is not None:
            nxt = (J + 1) % NS
            cnt = (NS - 3) * self.L + issued_before(gc0)
            pre[at(gc0)] += [I(f"s_waitcnt vmcnt({cnt})", "wait")]
            ops = []
            per = [self.group_consts(self.Q(nxt, "sz", p), 1 - J % 2, p) for p in range(self.PAIRS)]
            for i in range(5):
                for c in per:
                    ops.append(c[i])
            for i, op in enumerate(ops):
                extras[at(gc0) + (i * 4) // len(ops)] += [op]
        # "X(s + 2) has landed" + the barrier, then the stage counter and the next stage's load offsets
        sal = [I(f"s_add_u32 {s(S_S)}, {s(S_S)}, 1", "salu")] + self.offsets(NS - 1)
        if self.bar:
            gb = G - SSM - 1
            if at(gb) is not None:
                extras[at(gb)] += [I("s_waitcnt lgkmcnt(0)", "wait"), I("s_barrier", "wait")]   # (nobody still reads the slot the next stage's X goes to)
            for i, op in enumerate(sal):
                gs = G - SSM + (i * 4) // len(sal)
                if at(gs) is not None: extras[at(gs)] += [op]
        else:
            gb = G - 5
            if at(gb) is not None:
                assert issued_before(gb) == self.L
                extras[at(gb)] += [I(f"s_waitcnt vmcnt({self.LW + (NS - 3) * self.L})", "wait"), I("s_barrier", "wait")]
                for i, op in enumerate(sal):
                    extras[at(gb) + 1 + (i * 4) // len(sal)] += [op]
        # dequantisation ops per gap
        nd = len(dq)
        if SSM == 16: per_gap = [1] * 8 + [2, 2, 2, 2, 2, 3, 3, 2]
        elif self.NRD == 4: per_gap = [2, 2, 2, 2, 4, 4, 5, 5]
        elif nd == 26: per_gap = [3, 3, 3, 3, 3, 3, 4, 4]
        else: per_gap = [6, 6, 6, 6, 7, 7, 7, 7]
        assert sum(per_gap) == nd
        out, di, waited, zeroed = [], 0, -1, set()
        for g in range(SSM):
            st, gi = g // self.MPS, g % self.MPS
            p, mt = gi // self.MB, gi % self.MB
            kk = u * self.SS + st
            rd, ch = st * self.MB + mt, st * self.PAIRS + p
            if rd > waited and not peel:
                out.append(I(f"s_waitcnt lgkmcnt({(self.NRD - 1 - rd) + min(g, self.NRD)})", "wait"))
                waited = rd
            acc = self.ACC(kk, p, mt)
            out.append(self.mfma(kk, p, mt, buf, ch, rd, peel and acc not in zeroed))
            zeroed.add(acc)
            if g < self.NRD:
                out.append(reads[g])
            out += pre[g]
            out += dq[di:di + per_gap[g]]
            di += per_gap[g]
            out += extras[g]
        assert di == nd
        if self.exp:
            def keep(i):
                t = i.text
                if (self.exp & 1) and t == "s_barrier": return False
                if (self.exp & 2) and (i.kind == "vmem" or t.startswith("s_waitcnt vmcnt") or t.startswith("s_add_u32 m0")): return False
                if (self.exp & 4) and i.kind == "valu": return False
                if (self.exp & 8) and (i.kind == "lds" or t.startswith("s_waitcnt lgkmcnt")): return False
                if (self.exp & 16) and t.startswith("s_waitcnt vmcnt"): return False
                return True
            out = [i for i in out if keep(i)]
        return pad_deps(out)


def check(seq, name):
    """dependent VALU pairs at least two instructions apart; an M0 write at least one instruction before its LDS-DMA; a VALU result at
    least two instructions before an MFMA reads it (the wait states hipcc would pad: the asm string gets none)"""
    last_w = {}
    for idx, ins in enumerate(seq):
        for r in ins.r:
            if r in last_w and last_w[r][1] != "lds":
                d = idx - last_w[r][0]
                need = 3 if ins.kind == "mfma" else 2
                assert d >= need, f"{name}: '{seq[last_w[r][0]].text}' -> '{ins.text}' only {d} apart"
        for r in ins.w:
            last_w[r] = (idx, ins.kind)


def pad_deps(seq):
    """insert s_nop 0 where two instructions would violate check()'s distances (single dequantisation / constant chains: PAIRS == 1)"""
    out, last_w = [], {}
    for ins in seq:
        while True:
            bad = False
            for r in ins.r:
                if r in last_w and last_w[r][1] != "lds" and len(out) - last_w[r][0] < (3 if ins.kind == "mfma" else 2):
                    bad = True
            if not bad:
                break
            out.append(I("s_nop 0", "salu"))
        for r in ins.w:
            last_w[r] = (len(out), ins.kind)
        out.append(ins)
    return out


def emit(seq):
    return "".join('  "%s\\n\\t"\n' % i.text for i in seq)


def build(c, stamped):
    pro = [I(f"s_mov_b32 {s(S_MLO)}, 0x000f000f", "salu"), I(f"s_mov_b32 {s(S_MHI)}, 0x00f000f0", "salu"),
           I(f"s_mov_b32 {s(S_SIXT)}, 0x2c002c00", "salu"), I(f"s_mov_b32 {s(S_D400)}, 0xd400d400", "salu"),
           I(f"s_mov_b32 {s(S_PERM)}, 0x01000100", "salu"), I(f"s_mov_b32 {s(S_S)}, 0", "salu"),
           I(f"s_lshr_b32 {s(S_TPG)}, {KTLO}, 24", "salu"), I(f"s_and_b32 {s(S_KTLO)}, {KTLO}, 0xffffff", "salu")]
    kthi = KTHI
    if c.big:
        # the operand list is at clang's limit of 30: the k tile count rides in the upper half of kthi, and the pair strides of the weights
        # and group words and the row pitch of x are derived from it; the x piece offsets (rows clamped to M - 1) are made here, once
        kthi = s(S_KTHI)
        pro += [I(f"s_and_b32 {s(S_KTHI)}, {KTHI}, 0xffff", "salu"), I(f"s_lshr_b32 {s(S_K2)}, {KTHI}, 16", "salu"),
                I(f"s_lshl_b32 {s(S_WPS)}, {s(S_K2)}, 11", "salu"), I(f"s_lshr_b32 {s(S_SPS)}, {s(S_K2)}, {s(S_TPG)}", "salu"),
                I(f"s_lshl_b32 {s(S_SPS)}, {s(S_SPS)}, 7", "salu"), I(f"s_lshl_b32 {s(S_K2)}, {s(S_K2)}, 8", "salu")]
        t = c.VX + c.NX
        for i in range(c.NX):
            pro += [I(f"v_add_u32 {v(t)}, {16 * i}, %[xr]", "valu", [t]),
                    I(f"v_min_u32 {v(t)}, %[mlast], {v(t)}", "valu", [t], [t]),
                    I(f"v_mad_u32_u24 {v(c.VX + i)}, {v(t)}, {s(S_K2)}, %[xc]", "valu", [c.VX + i], [t])]
    pro += [I(f"s_sub_u32 {s(S_N)}, {kthi}, {s(S_KTLO)}", "salu"), I(f"s_sub_u32 {s(S_KTMAX)}, {kthi}, 1", "salu")]
    # the ring and the queue as NS - 1 stages of the steady state would have left them: X(q) W(q), q = 0 .. NS - 2
    for q in range(c.NS - 1):
        pro += c.offsets(q)
        for i in range(c.NX): pro += c.x_piece(i, q, True)
        pro += c.w_loads(q)
    pro += c.offsets(c.NS - 1)   # what stage 0 itself issues
    pro += [I(f"v_mov_b32 {v(c.VMAGIC)}, 0x64006400", "valu", [c.VMAGIC])]
    for kk in range(c.KST):   # (KW = 2: the wave's j-th step is k16 step 2 j + parity of the stage; xrd carries the parity)
        pro += [I(f"v_xor_b32 {v(c.VRA + kk)}, {(kk * c.KW) << 5}, {XRD}", "valu", [c.VRA + kk])]
    for kk in range(c.KST):
        pro += [I(f"v_add_u32 {v(c.VRA2 + kk)}, 0x10000, {v(c.VRA + kk)}", "valu", [c.VRA2 + kk], [c.VRA + kk])]
    # X(0), W(0), X(1) have landed, in every wave: behind them W(1) and NS - 3 whole stages
    pro += [I(f"s_waitcnt vmcnt({(c.NS - 2) * c.L if c.bar else c.LW + (c.NS - 3) * c.L})", "wait"), I("s_barrier", "wait")]
    if stamped:
        pro += [I(f"s_memrealtime s[{STAMP0}:{STAMP0 + 1}]", "salu"), I(f"s_memtime s[{CLK0}:{CLK0 + 1}]", "salu"), I("s_waitcnt lgkmcnt(0)", "wait")]
    per = [c.group_consts(c.Q(0, "sz", p), 0, p) for p in range(c.PAIRS)]
    for i in range(5):
        for ch in per:
            pro.append(ch[i])
    reads, dq = c.prep(0, 0)
    pro += reads + dq
    pro += [I("s_nop 1", "salu"), I("s_waitcnt lgkmcnt(0)", "wait")]
    pro = pad_deps(pro)
    check(pro, "prologue")
    text = emit(pro)
    peel = c.superstep(0, 0, peel=True)
    check(pro[-60:] + peel, "peel")
    text += emit(peel)
    text += '  "s_branch .Lxw_j0u1_%=\\n\\t"\n'
    text += '  ".Lxw_loop_%=:\\n\\t"\n'
    prev = peel
    nins = len(pro) + len(peel)
    for J in range(c.NS):
        for u in range(c.NSU):
            st = c.superstep(J, u)
            check(prev[-40:] + st, f"J{J} u{u}")
            if J == 0 and u == 1:
                text += '  ".Lxw_j0u1_%=:\\n\\t"\n'
            text += emit(st)
            nins += len(st)
            prev = st
        text += '  "s_cmp_ge_u32 %s, %s\\n\\t"\n' % (s(S_S), s(S_N))
        text += '  "s_cbranch_scc1 .Lxw_done_%=\\n\\t"\n' if J < c.NS - 1 else '  "s_cbranch_scc0 .Lxw_loop_%=\\n\\t"\n'
    check(c.superstep(c.NS - 1, c.NSU - 1)[-40:] + c.superstep(0, 0), "wrap")
    text += '  ".Lxw_done_%=:\\n\\t"\n'
    if stamped:
        text += '  "s_memrealtime s[%d:%d]\\n\\t"\n  "s_memtime s[%d:%d]\\n\\t"\n' % (STAMP1, STAMP1 + 1, CLK1, CLK1 + 1)
    text += '  "s_waitcnt vmcnt(0) lgkmcnt(0)\\n\\t"\n'
    if stamped:
        text += '  "s_mov_b64 %%[t0], s[%d:%d]\\n\\t"\n  "s_mov_b64 %%[t1], s[%d:%d]\\n\\t"\n' % (STAMP0, STAMP0 + 1, STAMP1, STAMP1 + 1)
        if not c.big:
            text += '  "s_sub_u32 %%[clk], s%d, s%d\\n\\t"\n' % (CLK1, CLK0)
    text += '  "s_nop 15\\n\\t"\n  "s_nop 7"\n'   # the last MFMAs' results -> whoever reads the accumulators next
    return text, nins


def run_macro(c):
    """the whole asm statement: names of the kernel's variables (accr[], b, xrd, xdst, kt_lo, kt_hi; stamped: t0, t1, clk; 256-token tiles:
    xr = first row of the lane's piece 0, xc = its swizzled 16-byte chunk, kt_hi carries K / 128 in its upper half, m_last = M - 1)"""
    outs = ", ".join('[a%d] "=&a"(accr[%d])' % (i, i) for i in range(c.NACC))
    ins = ['[rsx] "s"(b.x)', '[rsw] "s"(b.w)', '[rss] "s"(b.s)']
    if c.big:
        ins += ['[xr] "v"(xr)', '[xc] "v"(xc)']
    else:
        ins += ['[xv%d] "v"(b.x_voff[%d])' % (i, i) for i in range(c.NX)]
    ins += ['[wv] "v"(b.w_voff)', '[sv] "v"(b.s_voff)', '[xrd] "v"(xrd)', '[xdst] "s"(xdst)', '[ktlo] "s"(kt_lo)', '[kthi] "s"(kt_hi)']
    ins += ['[mlast] "s"(m_last)'] if c.big else ['[wps] "s"(b.w_pstride)', '[sps] "s"(b.s_pstride)']
    cl = ['"memory"', '"scc"', '"m0"'] + ['"v%d"' % r for r in range(48, c.VEND)] + ['"s%d"' % r for r in range(52, 82 if c.big else 78)]
    stamps = '[t0] "=&s"(t0), [t1] "=&s"(t1)' + ('' if c.big else ', [clk] "=&s"(clk)')
    body = "#define QA_XW_RUN_%s() asm volatile(QA_XW_ASM_%s : %s : %s : %s)\n" % (c.name, c.name, outs, ", ".join(ins), ", ".join(cl))
    body += ("#define QA_XW_RUN_STAMPED_%s() asm volatile(QA_XW_ASM_STAMPED_%s : %s, %s : %s : %s)\n"
             % (c.name, c.name, outs, stamps, ", ".join(ins), ", ".join(cl)))
    return body


CONFIGS = [Cfg(4, 2, 4, 1), Cfg(4, 1, 4, 1), Cfg(2, 1, 8, 2), Cfg(8, 2, 2, 1, bar=1)]
# Tried and not kept [r04, profiles/r04_loop_variants.txt] -- the options stay in the generator for the record:
#   Cfg(4, 2, 4, 1, KW=2)   128 x 128 tile of (4, 2)-shaped waves, the two wave pairs splitting the k16 steps of a stage by parity (half the
#                           B reads per MFMA of (4, 1)): 40.6-41.3 clocks per MFMA, the same as (4, 1)'s 41.3-41.6 -- what the 128 x 128
#                           tile pays for is its x traffic per MFMA (twice that of 128 x 256), not its LDS reads;
#   bar=1                   the stage's barrier in front of the last super-step, one counted wait per stage: same time (+-0.5 %);
#   exp=16                  no counted vector-memory waits at all (wrong results): same time -- the loops do not wait for memory.
EXPERIMENTS = [(2, 1, 8, 2, e) for e in (1, 2, 4, 8, 15)] + [(4, 1, 4, 1, e) for e in (1, 2, 4, 8)]


def main():
    body = ("// GENERATED by tools/gen_xw_loop.py -- do not edit.  The K loops of w4a16_xw_kernel, one inline-asm statement per tile shape\n"
            "// (operands and register plan: the generator's docstring and w4a16_xw.hpp).\n")
    for c in CONFIGS:
        for stamped in (False, True):
            text, n = build(c, stamped)
            body += "#define QA_XW_ASM_%s%s \\\n" % ("STAMPED_" if stamped else "", c.name)
            body += "".join(l + " \\\n" for l in text.rstrip("\n").split("\n")) + "\n"
        body += run_macro(c)
        print("config (%d, %d): %d instructions, VGPRs v48..v%d, ring / queue of %d, LDS %d KiB" % (c.MB, c.PAIRS, n, c.VEND - 1, c.NS, c.NS * c.SLOTB // 1024))
    body += "#ifdef QUICK_AMD_TOOLS\n"
    for (mb, pairs, ns, kp, e) in EXPERIMENTS:
        c = Cfg(mb, pairs, ns, kp, e & ~64, bar=1 if e & 64 else 0)   # (64: not an omission -- the early barrier, right results)
        text, n = build(c, True)
        nm = "%s_E%d" % (c.name, e)
        body += "#define QA_XW_ASM_STAMPED_%s \\\n" % nm
        body += "".join(l + " \\\n" for l in text.rstrip("\n").split("\n")) + "\n"
        body += run_macro(c).split("\n")[1].replace("QA_XW_RUN_STAMPED_%s" % c.name, "QA_XW_RUN_STAMPED_%s" % nm).replace("QA_XW_ASM_STAMPED_%s " % c.name, "QA_XW_ASM_STAMPED_%s " % nm) + "\n"
    body += "#endif\n"
    with open(OUT, "w") as f:
        f.write(body)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
