#!/usr/bin/env python3
"""Generates quick_amd/csrc/w4a16_xm_loop.inc: the K loops of the mid-token kernels (w4a16_xm.hpp, 17..64 tokens), each as ONE inline-asm
statement per (MB 32-token blocks, PR 32-channel pairs per workgroup).

    python tools/gen_xm_loop.py            # rewrites the .inc (checked in; the build does not run this script)

What the loop is.  A workgroup owns MB * 32 tokens x PR * 32 channels for a K slice; its EIGHT waves split the slice's k tiles and every
wave runs this loop on its own k tiles -- its own x ring in LDS (filled by LDS-DMA, nobody else reads it), its own weight queue in
registers, its own accumulators -- so there is no workgroup barrier anywhere in the loop, and every dequantised weight fragment feeds all
MB token blocks (the property of the reference's compute_gemm_x2, csrc/gemm_cuda_quick.cu:458-1196).  The waves' partial tiles meet in LDS
after the loop (w4a16_xm.hpp).

A STAGE is one k tile (128 k = 8 k16 steps), a QUARTER two k16 steps (32 k: 64 bytes of every token row).  Per stage and wave:
  x      2 halves x MB blocks x 4 LDS-DMA instructions of 8 rows x 128 B -- whole cache lines: pieces of 16 rows x 64 B measured 67-107 GB/s per
         workgroup where these reach 116-136 (tools/x_dma_bw.hip) -- into the wave's ring of two half-stage slots.  Half h is refilled with the
         pieces of stage s + 1 during quarter 2 h + 1 of stage s (its last fragments were read at the head of quarter 2 h), ONE instruction
         every few MFMAs -- a burst holds the wave at the vector-memory issue
  W      PR x (lo, hi dwordx4 + the (scale, zero) word), HBM -> VGPR queue of D stages; one pair: W(s + D) leaves at the end of stage s; two
         and three pairs: W(s + 1) leaves in quarter 1 of stage s behind that quarter's x pieces (Cfg.jitq)
  reads  MB x 2 ds_read_b128 per quarter, one quarter ahead, double-buffered, into the last accumulator registers (MFMA B operands)
  math   8 PR dequantisation chains (13 VALU, exact form: fp16((w - z) s) as the reference's dequantize_s4_to_fp16x2_fused + sub + mul),
         each feeding MB v_mfma_f32_32x32x16_f16; chain n + 1 is interleaved with the MFMAs of chain n
Vector-memory loads return in issue order (one counter): "X has landed" = s_waitcnt vmcnt(loads issued behind X's last one).  The counts are
not written by hand: resolve() walks the dynamic instruction stream (prologue, the peeled first stage, a dozen stages of the loop) and every
static wait gets the smallest count over its dynamic instances.  The same order is why an x piece queued behind a weight load cannot be SEEN
before that load's trip to HBM is over: the weight loads of a stage leave behind the x pieces they would delay most, and with D >= the
wave's stage count (K = 4096: four) none leaves inside the loop at all.  Loads past the wave's last k tile go through a descriptor of zero
records: no traffic, the counts stay what they are.

Registers: fixed VGPRs from VBASE and SGPRs s52.. are CLOBBERS of the statement, and so are the accumulator registers that hold the x
fragments; accumulators are "=&a" operands a0.. (index p * MB + blk, the layout of the four-wave kernels: lane (token rho, h), register
i = channel 8 (i / 4) + 4 h + i % 4 of the pair).
Operands: rsx rsw rss + rsx2 rsw2 rss2 descriptor halves (x rows of the token tile / the workgroup's channel pairs / their group words),
xr xc0 xc1 mlast k2 the lane's row and swizzled chunks within an x piece, the tile's last row and the row pitch (the piece offsets are made
in the prologue), wv sv weight / group-word lane offsets, xrd LDS read address (k16 step 0 of a quarter),
xdst the wave's ring, kb | log2(k tiles per group) << 24 and ke the wave's k tiles, wps / sps byte strides between pairs.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_xw_loop import I, v, s, pad_deps, check, emit, even  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "quick_amd", "csrc", "w4a16_xm_loop.inc")

S_MLO, S_MHI, S_SIXT, S_D400, S_PERM = 52, 53, 54, 55, 56
S_S, S_T, S_KB, S_TPG, S_KT, S_G, S_XSO = 57, 58, 59, 60, 61, 62, 63
S_NRX, S_NRW, S_NRS = 64, 65, 66
S_DX, S_DW, S_DS = 68, 72, 76           # private copies of the three descriptors (word 2 = records: switched to 0 for dead stages)
S_WSO = [80, 81, 82]
S_SSO = [83, 84, 85]
S_XQ = [86, 87]                         # row offsets of the two halves of the stage being fetched
S_END = 88
S_STAMP = 90                            # stamped builds: six s_memrealtime pairs s[90:101] (x(0, 0) landed, W(0) landed, end of stages 0..3)
RSX, RSW, RSS = "%[rsx]", "%[rsw]", "%[rss]"
WV, SV, XRD, XDST, KB, KE, WPS, SPS, TV = "%[wv]", "%[sv]", "%[xrd]", "%[xdst]", "%[kb]", "%[ke]", "%[wps]", "%[sps]", "%[tv]"
TOUCH = os.environ.get("XM_TOUCH", "0") != "0"       # line touches in the prologue (A/B switch of the generator; measured slower: the touches fill the CU's in-order memory pipe)
BARRIER = os.environ.get("XM_BARRIER", "0") != "0"   # prologue barrier between x(0) and the weight requests (A/B switch)
WFIRST = os.environ.get("XM_WFIRST", "1") != "0"     # W(0) in front of x(0) in the prologue (A/B switch)
DQ = os.environ.get("XM_D", "")                      # queue depths of the six configurations, e.g. 2,2,2,2,2,2 (A/B switch)
W_NT = " nt" if os.environ.get("XM_W_NT", "1") != "0" else ""   # the weights are read once: streaming cache policy (as the lean kernels)
if os.environ.get("XM_W_POLICY") is not None:                    # A/B switch: any combination of sc0 / sc1 / nt on the weight loads, e.g. XM_W_POLICY="sc1 nt"
    W_NT = (" " + os.environ["XM_W_POLICY"].strip()) if os.environ["XM_W_POLICY"].strip() else ""
X_POLICY = (" " + os.environ["XM_X_POLICY"].strip()) if os.environ.get("XM_X_POLICY", "").strip() else ""   # A/B switch: cache policy bits on the x pieces (LDS-DMA)
JITQ_ENV = os.environ.get("XM_JITQ", "")                # "q": every configuration requests W(s + 1) in quarter q of stage s; "-1": none does (A/B switch; default: Cfg.jitq)
SBAR = os.environ.get("XM_SBAR", "")                   # quarters at whose head every stage has a workgroup barrier (e.g. "0" or "1,3"): phase-locks the waves' requests (A/B switch;
                                                       # only for shapes whose waves own equally many k tiles)
BURST = os.environ.get("XM_BURST", "0") != "0"         # x pieces of a half in one burst at the head of its quarter instead of one every few MFMAs (A/B switch)
EXP = int(os.environ.get("XM_EXP", "0"))   # timing experiments (wrong results; never checked in): 1 no MFMAs, 2 no vector memory, 4 no dequantisation,
                                            # 8 no fragment reads, 16 no weight loads, 32 no x pieces (16 / 32: and no counted waits)


def sr(r, n): return "s[%d:%d]" % (r, r + n - 1)
def a(r, n): return "a[%d:%d]" % (r, r + n - 1)


def need(tag):
    i = I("NEED", "wait")
    i.need = tag
    return i


class Cfg:
    def __init__(self, MB, PR, D, VBASE=24, RS=1, jitq=-1):
        # jitq >= 0: W(s + 1) leaves in quarter jitq of stage s, right behind that quarter's x pieces, into a queue of two slots -- instead of
        # W(s + D) at the end of stage s.  A wave's loads return in issue order, so an x piece waits for every weight request in front of it:
        # weights requested D stages ahead hold back x that is needed half a stage ahead.  Measured (profiles/r06_xm_anatomy.txt): 3-9 % off the
        # launches of two and three channel pairs, level or 2-5 % slower with one pair (its whole weight set is in flight from the prologue).
        if JITQ_ENV:
            jitq = int(JITQ_ENV)
        if jitq >= 0 and not DQ:
            D = 2
        assert MB in (1, 2) and PR in (1, 2, 3) and D >= 2 and RS in (1, 2) and D % RS == 0
        self.MB, self.PR, self.D, self.VBASE, self.RS, self.jitq = MB, PR, D, VBASE, RS, jitq   # RS: stages of the wave's x ring
        self.NX = 4 * MB                    # LDS-DMA instructions per half stage (8 rows x 128 B each)
        self.LW = 3 * PR                    # weight-side loads per stage
        self.HB = MB * 4096                 # bytes of a half-stage slot
        self.RING = 2 * self.HB * RS        # the wave's ring: RS stages
        self.NACC = PR * MB
        self.QS = even(9 * PR)              # queue slot: per pair lo[4] hi[4], then sz per pair
        self.AXF = 128 - 2 * MB * 2 * 4     # x fragments [buf][blk][uu][4] live in the LAST accumulator registers (ds_read writes, MFMA reads them)
        b = VBASE                           # first fixed VGPR (below it: the statement's operands and whatever hipcc keeps live across it)
        self.VQ = b; b += D * self.QS
        self.VWF = b; b += 2 * 4            # [buf][4]
        self.VGC = b; b += even(3 * PR)
        self.VT = b; b += 2                 # shift temporaries of two chains in flight
        self.VZT = b; b += even(PR)
        self.VRA = b; b += 4                # read addresses of the four k16 steps of a half
        self.VX = b; b += self.NX           # byte offsets of the x pieces (rows clamped to the tile's last one)
        self.VMAGIC = b; b += 1
        self.VEND = b
        assert self.VEND <= 128 and 16 * MB * PR <= self.AXF, self.VEND
        self.name = "%d%d" % (MB, PR)

    def Q(self, d, what, p=0, t=0):
        base = self.VQ + self.QS * d
        if what == "lo": return base + 8 * p + t
        if what == "hi": return base + 8 * p + 4 + t
        if what == "sz": return base + 8 * self.PR + p
        raise ValueError(what)
    def XF(self, buf, blk, uu): return self.AXF + ((buf * self.MB + blk) * 2 + uu) * 4
    def WF(self, buf): return self.VWF + buf * 4
    def GC(self, p, i): return self.VGC + p * 3 + i
    def ACC(self, p, blk): return "%%[a%d]" % (p * self.MB + blk)

    def dequant_chain(self, qreg, gc, out, tmp):
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

    def group_consts(self, szreg, p):
        s2, nzlo, nzhi = self.GC(p, 0), self.GC(p, 1), self.GC(p, 2)
        zt = self.VZT + p
        return [
            I(f"v_perm_b32 {v(s2)}, {v(szreg)}, {v(szreg)}, {s(S_PERM)}", "valu", [s2], [szreg]),
            I(f"v_lshrrev_b32 {v(zt)}, 16, {v(szreg)}", "valu", [zt], [szreg]),
            I(f"v_lshl_or_b32 {v(zt)}, {v(zt)}, 16, {v(zt)}", "valu", [zt], [zt]),
            I(f"v_or_b32 {v(nzlo)}, 0xe400e400, {v(zt)}", "valu", [nzlo], [zt]),
            I(f"v_lshl_or_b32 {v(nzhi)}, {v(zt)}, 4, {s(S_D400)}", "valu", [nzhi], [zt]),
        ]

    def weight_dword(self, d, p, u):
        return self.Q(d, "lo" if u % 2 == 0 else "hi", p, u >> 1)

    def x_reads(self, q, buf, rs=0):
        """the B fragments of quarter q (k16 steps 2 q, 2 q + 1 of the stage in ring slot rs: half q / 2 of it) into buffer buf: MB x 2 ds_read_b128"""
        out = []
        for blk in range(self.MB):
            for uu in range(2):
                r = self.XF(buf, blk, uu)
                ra = self.VRA + (q % 2) * 2 + uu
                out.append(I(f"ds_read_b128 {a(r, 4)}, {v(ra)} offset:{rs * 2 * self.HB + (q // 2) * self.HB + blk * 4096}", "lds", [("a", x) for x in range(r, r + 4)], [ra]))
        return out

    def x_dma(self, h, tag=None, rs=0):
        """the pieces of half h of the stage whose row offsets are in S_XQ (descriptor S_DX: live or dead) into slot h, as (M0 write, LDS-DMA)
        pairs; the caller puts at least one instruction between the two"""
        out = []
        for i in range(self.NX):
            ld = I(f"buffer_load_dwordx4 {v(self.VX + i)}, {sr(S_DX, 4)}, {s(S_XQ[h])} offen{X_POLICY} lds", "vmem", [], ["m0"])
            ld.tag = tag
            out.append((I(f"s_add_u32 m0, {XDST}, {rs * 2 * self.HB + h * self.HB + i * 1024}", "salu", ["m0"], []), ld))
        return out

    def x_dma_flat(self, h, tag=None, rs=0):
        out = []
        for m, ld in self.x_dma(h, tag, rs):
            out += [m, I("s_nop 0", "salu"), ld]
        return out

    def w_loads(self, d, tag=None):
        out = []
        for p in range(self.PR):
            out += [I(f"buffer_load_dwordx4 {v(self.Q(d, 'lo', p), 4)}, {WV}, {sr(S_DW, 4)}, {s(S_WSO[p])} offen{W_NT}", "vmem"),
                    I(f"buffer_load_dwordx4 {v(self.Q(d, 'hi', p), 4)}, {WV}, {sr(S_DW, 4)}, {s(S_WSO[p])} offen offset:512{W_NT}", "vmem")]
        for p in range(self.PR):
            out += [I(f"buffer_load_dword {v(self.Q(d, 'sz', p))}, {SV}, {sr(S_DS, 4)}, {s(S_SSO[p])} offen", "vmem")]
        for o in out:
            o.tag = tag
        return out

    def x_offsets(self, ahead, quarters):
        """row offsets (one SGPR per half in `quarters`) and live / dead descriptor of the x pieces of stage S_S + ahead"""
        out = [
            I(f"s_add_u32 {s(S_KT)}, {s(S_S)}, {ahead}", "salu"),
            I(f"s_cmp_lt_u32 {s(S_KT)}, {s(S_T)}", "salu"),
            I(f"s_cselect_b32 {s(S_DX + 2)}, {s(S_NRX)}, 0", "salu"),
            I(f"s_add_u32 {s(S_KT)}, {s(S_KT)}, {s(S_KB)}", "salu"),
            I(f"s_lshl_b32 {s(S_XSO)}, {s(S_KT)}, 8", "salu"),
        ]
        for h in quarters:
            out.append(I(f"s_add_u32 {s(S_XQ[h])}, {s(S_XSO)}, {128 * h}", "salu"))
        return out

    def w_offsets(self, ahead):
        out = [
            I(f"s_add_u32 {s(S_KT)}, {s(S_S)}, {ahead}", "salu"),
            I(f"s_cmp_lt_u32 {s(S_KT)}, {s(S_T)}", "salu"),
            I(f"s_cselect_b32 {s(S_DW + 2)}, {s(S_NRW)}, 0", "salu"),
            I(f"s_cselect_b32 {s(S_DS + 2)}, {s(S_NRS)}, 0", "salu"),
            I(f"s_add_u32 {s(S_KT)}, {s(S_KT)}, {s(S_KB)}", "salu"),
            I(f"s_lshl_b32 {s(S_WSO[0])}, {s(S_KT)}, 10", "salu"),
            I(f"s_lshr_b32 {s(S_G)}, {s(S_KT)}, {s(S_TPG)}", "salu"),
            I(f"s_lshl_b32 {s(S_SSO[0])}, {s(S_G)}, 6", "salu"),
        ]
        for p in range(1, self.PR):
            out += [I(f"s_add_u32 {s(S_WSO[p])}, {s(S_WSO[p - 1])}, {WPS}", "salu"), I(f"s_add_u32 {s(S_SSO[p])}, {s(S_SSO[p - 1])}, {SPS}", "salu")]
        return out

    def mfma(self, p, blk, wbuf, xbuf, uu, zero_c):
        wa, b = self.WF(wbuf), self.XF(xbuf, blk, uu)
        acc = self.ACC(p, blk)
        return I(f"v_mfma_f32_32x32x16_f16 {acc}, {v(wa, 4)}, {a(b, 4)}, {'0' if zero_c else acc}", "mfma", [], list(range(wa, wa + 4)) + [("a", x) for x in range(b, b + 4)])

    def stage(self, d, sa, first=False, stamp=None):
        """k tile number sa of the wave out of queue slot d = sa % D; first: the accumulators start from the constant 0.  Waits are NEED markers
        (the tag of what must have landed): their vmcnt values come out of resolve()."""
        MB, PR = self.MB, self.PR
        out = [need(("w", sa))]
        if stamp is not None and first:
            out.append(I(f"s_memrealtime {sr(S_STAMP + 2, 2)}", "salu"))
        per = [self.group_consts(self.Q(d, "sz", p), p) for p in range(PR)]
        for i in range(5):
            for c in per:
                out.append(c[i])
        RS, rs = self.RS, sa % self.RS
        out += self.x_offsets(RS, (0, 1))
        chains = []
        for u in range(8):
            for p in range(PR):
                n = u * PR + p
                chains.append(self.dequant_chain(self.weight_dword(d, p, u), [self.GC(p, i) for i in range(3)], self.WF(n % 2), self.VT + n % 2))
        N = len(chains)
        out += chains[0]
        nmf = 2 * PR * MB                     # MFMAs of a quarter
        pend, mi, npend = [], 0, 0
        for n in range(N):
            u, p = n // PR, n % PR
            q, uu = u // 2, u % 2
            if uu == 0 and p == 0:
                # head of quarter q: the next quarter's fragments (of this stage, or quarter 0 of the next); in quarters 1 and 3 the half whose
                # last fragments were read a quarter ago is refilled with the pieces of stage sa + 1, one LDS-DMA every few MFMAs
                out.append(need(("x", sa, (q + 1) // 2) if q < 3 else ("x", sa + 1, 0)))
                out += self.x_reads((q + 1) % 4, (q + 1) % 2, rs if q < 3 else (rs + 1) % RS)
                pend = self.x_dma(q // 2, ("x", sa + RS, q // 2), rs) if q % 2 == 1 else []
                if SBAR and str(q) in SBAR.split(","):
                    out.append(I("s_barrier", "wait"))
                if BURST and pend:
                    for m0w, ld in pend:
                        out += [m0w, I("s_nop 0", "salu"), ld]
                    pend = []
                if q == self.jitq:
                    out += self.w_offsets(1)
                    pend += [(I("s_nop 0", "salu"), ld) for ld in self.w_loads((d + 1) % self.D, ("w", sa + 1))]
                npend = len(pend)
                mi = 0
            nxt = chains[n + 1] if n + 1 < N else []
            per_gap = [len(nxt) // MB + (1 if i >= MB - len(nxt) % MB else 0) for i in range(MB)] if nxt else [0] * MB
            di = 0
            for blk in range(MB):
                out.append(self.mfma(p, blk, n % 2, q % 2, uu, first and u == 0))
                # this MFMA's share of the pending pieces (M0 write, some of the dequantisation ops, the LDS-DMA), then the rest of its ops
                npc = ((mi + 1) * npend) // nmf - (mi * npend) // nmf if npend else 0
                ops = nxt[di:di + per_gap[blk]]
                di += per_gap[blk]
                for k in range(npc):
                    m0w, ld = pend.pop(0)
                    out.append(m0w)
                    take = (len(ops) + (npc - k) - 1) // (npc - k) if ops else 0
                    out += ops[:take]
                    ops = ops[take:]
                    out.append(ld)
                out += ops
                mi += 1
            assert di == len(nxt)
            if uu == 1 and p == PR - 1:
                assert not pend
                out.append(I("s_waitcnt lgkmcnt(0)", "wait"))
        # W(sa + D) into this stage's queue slot (behind the x pieces of quarter 3: the weights' trip to HBM does not sit in front of them in the
        # in-order queue), then the counter
        if self.jitq < 0:
            out += self.w_offsets(self.D)
            out += self.w_loads(d, ("w", sa + self.D))
        out.append(I(f"s_add_u32 {s(S_S)}, {s(S_S)}, 1", "salu"))
        if stamp is not None:
            out.append(I(f"s_memrealtime {sr(S_STAMP + 4 + 2 * stamp, 2)}", "salu"))
        return pad_deps(out)

    def prologue(self, stamped=False):
        D = self.D
        pro = [I(f"s_mov_b32 {s(S_MLO)}, 0x000f000f", "salu"), I(f"s_mov_b32 {s(S_MHI)}, 0x00f000f0", "salu"),
               I(f"s_mov_b32 {s(S_SIXT)}, 0x2c002c00", "salu"), I(f"s_mov_b32 {s(S_D400)}, 0xd400d400", "salu"),
               I(f"s_mov_b32 {s(S_PERM)}, 0x01000100", "salu"), I(f"s_mov_b32 {s(S_S)}, 0", "salu"),
               I(f"s_lshr_b32 {s(S_TPG)}, {KB}, 24", "salu"), I(f"s_and_b32 {s(S_KB)}, {KB}, 0xffffff", "salu"),
               I(f"s_sub_u32 {s(S_T)}, {KE}, {s(S_KB)}", "salu"),
               I(f"s_mov_b64 {sr(S_DX, 2)}, {RSX}", "salu"), I(f"s_mov_b64 {sr(S_DW, 2)}, {RSW}", "salu"), I(f"s_mov_b64 {sr(S_DS, 2)}, {RSS}", "salu"),
               # (the upper halves of the descriptors travel as operands of their own: the loop switches the record count of its private copies)
               I(f"s_mov_b64 {sr(S_DX + 2, 2)}, %[rsx2]", "salu"), I(f"s_mov_b64 {sr(S_DW + 2, 2)}, %[rsw2]", "salu"), I(f"s_mov_b64 {sr(S_DS + 2, 2)}, %[rss2]", "salu"),
               I(f"s_mov_b32 {s(S_NRX)}, {s(S_DX + 2)}", "salu"), I(f"s_mov_b32 {s(S_NRW)}, {s(S_DW + 2)}", "salu"), I(f"s_mov_b32 {s(S_NRS)}, {s(S_DS + 2)}", "salu"),
               I(f"v_mov_b32 {v(self.VMAGIC)}, 0x64006400", "valu", [self.VMAGIC]),
               I(f"v_mov_b32 {v(self.VRA)}, {XRD}", "valu", [self.VRA]),
               I(f"v_xor_b32 {v(self.VRA + 1)}, 32, {XRD}", "valu", [self.VRA + 1]),
               I(f"v_xor_b32 {v(self.VRA + 2)}, 64, {XRD}", "valu", [self.VRA + 2]),
               I(f"v_xor_b32 {v(self.VRA + 3)}, 0x60, {XRD}", "valu", [self.VRA + 3])]
        # piece i = rows 8 i .. 8 i + 7 of the tile (clamped to its last row: rows past M replay it, never stored), lane p = row p / 8, chunk
        # (p % 8) ^ ((row / 2) % 8) of the row's 128 bytes -- the swizzle that makes the ds_read_b128 of the 32 x 16 fragments conflict-free
        tmp = self.VT
        for i in range(self.NX):
            pro += [I(f"v_add_u32 {v(tmp)}, {8 * i}, %[xr]", "valu", [tmp]),
                    I(f"v_min_u32 {v(tmp)}, %[mlast], {v(tmp)}", "valu", [tmp], [tmp]),
                    I(f"v_mad_u32_u24 {v(self.VX + i)}, {v(tmp)}, %[k2], %[xc{i % 2}]", "valu", [self.VX + i], [tmp])]
        # x of stage 0 first (it comes from L2, and everything behind it in the queue waits for it anyway), then W(0) .. W(D - 1); the first
        # fragments as soon as x(0, 0) is there, then -- slot 0 free -- the first pieces of stage 1
        xs = []
        for r in range(self.RS):
            xs += self.x_offsets(r, (0, 1))
            for h in range(2):
                xs += self.x_dma_flat(h, ("x", r, h), r)
        w0 = self.w_offsets(0) + self.w_loads(0, ("w", 0))
        # W(0) FIRST: its trip to HBM is the longest wait of the launch and everything queued behind it comes back behind it anyway
        pro += (w0 + xs) if WFIRST else xs
        if BARRIER:
            pro.append(I("s_barrier", "wait"))
        if TOUCH:
            pro += [I(f"s_lshl_b32 {s(S_KT)}, {s(S_KB)}, 10", "salu"), I(f"s_lshr_b32 {s(S_G)}, {WPS}, 1", "salu")]
            for j in range(2 * self.PR):
                ld = I(f"buffer_load_dword {v(self.VT)}, {TV}, {sr(S_DW, 4)}, {s(S_KT)} offen", "vmem")
                ld.tag = ("t", j)
                pro.append(ld)
                if j + 1 < 2 * self.PR:
                    pro.append(I(f"s_add_u32 {s(S_KT)}, {s(S_KT)}, {s(S_G)}", "salu"))
        for d in range(1 if WFIRST else 0, D if self.jitq < 0 else 1):
            pro += self.w_offsets(d)
            pro += self.w_loads(d, ("w", d))
        pro += [need(("x", 0, 0))]
        if stamped:
            pro.append(I(f"s_memrealtime {sr(S_STAMP, 2)}", "salu"))
        pro += self.x_reads(0, 0)
        pro += [I("s_waitcnt lgkmcnt(0)", "wait")]
        return pad_deps(pro)


def resolve(c, nsim, stamped=False):
    """Walks the dynamic stream: prologue, stage 0 (peeled), stages 1 .. nsim; every static wait of a loop copy gets the smallest count over
    its dynamic instances.  Returns the instruction lists with the NEED markers replaced: prologue, first stage, the D loop copies."""
    issued = []

    def walk(seq):
        counts = []
        for ins in seq:
            if ins.kind == "vmem":
                assert getattr(ins, "tag", None) is not None, ins.text
                issued.append(ins.tag)
            nd = getattr(ins, "need", None)
            if nd is not None:
                last = max(i for i, x in enumerate(issued) if x == nd)
                counts.append(len(issued) - 1 - last)
        return counts

    def fill(seq, counts):
        out, k = [], 0
        for ins in seq:
            if getattr(ins, "need", None) is not None:
                out.append(I(f"s_waitcnt vmcnt({min(counts[k], 63)})", "wait"))   # (the counter has six bits; a smaller count only waits for more)
                k += 1
            else:
                out.append(ins)
        assert k == len(counts)
        return out

    pro = c.prologue(stamped)
    pc = walk(pro)
    first = c.stage(0, 0, first=True, stamp=0 if stamped else None)
    fc = walk(first)
    copies = {d: None for d in range(c.D)}
    for sa in range(1, nsim + 1):
        cnt = walk(c.stage(sa % c.D, sa))
        d = sa % c.D
        copies[d] = cnt if copies[d] is None else [min(x, y) for x, y in zip(copies[d], cnt)]
    loop = [fill(c.stage(d, d if d else c.D, stamp=(min(3, d if d else c.D) if stamped else None)), copies[d]) for d in range(c.D)]
    return fill(pro, pc), fill(first, fc), loop, (pc, fc, copies)


def keep(i):
    t = i.text
    if (EXP & 1) and i.kind == "mfma": return False
    if (EXP & 2) and (i.kind == "vmem" or t.startswith("s_waitcnt vmcnt")): return False
    if (EXP & 4) and i.kind == "valu" and not t.startswith("v_mov") and not t.startswith("v_xor"): return False
    if (EXP & 8) and (i.kind == "lds" or t.startswith("s_waitcnt lgkmcnt")): return False
    if (EXP & 16) and i.kind == "vmem" and "lds" not in t: return False
    if (EXP & 32) and i.kind == "vmem" and " lds" in t: return False
    if (EXP & 48) and t.startswith("s_waitcnt vmcnt"): return False
    return True


def filt(seq):
    return [i for i in seq if keep(i)] if EXP else seq


def build(c, stamped=False):
    D = c.D
    pro, first, loop, counts = resolve(c, 4 * D + 3, stamped)
    check(pro, "prologue")
    text = emit(filt(pro))
    nins = len(pro)
    exit_check = '  "s_cmp_ge_u32 %s, %s\\n\\t"\n  "s_cbranch_scc1 .Lxm_done_%%=\\n\\t"\n' % (s(S_S), s(S_T))
    check(pro[-40:] + first, "first")
    text += emit(filt(first)) + exit_check
    nins += len(first)
    text += '  "s_branch .Lxm_d1_%=\\n\\t"\n'
    text += '  ".Lxm_loop_%=:\\n\\t"\n'
    prev = first
    for d in range(D):
        st = loop[d]
        check(prev[-40:] + st, f"stage {d}")
        if d == 1:
            text += '  ".Lxm_d1_%=:\\n\\t"\n'
        text += emit(filt(st))
        nins += len(st)
        prev = st
        text += '  "s_cmp_ge_u32 %s, %s\\n\\t"\n' % (s(S_S), s(S_T))
        text += '  "s_cbranch_scc1 .Lxm_done_%=\\n\\t"\n' if d < D - 1 else '  "s_cbranch_scc0 .Lxm_loop_%=\\n\\t"\n'
    check(loop[D - 1][-40:] + loop[0], "wrap")
    text += '  ".Lxm_done_%=:\\n\\t"\n'
    text += '  "s_waitcnt vmcnt(0) lgkmcnt(0)\\n\\t"\n'
    if stamped:   # the low words of the six stamps, two per 64-bit output
        for i in range(6):
            text += '  "s_mov_b32 s%d, s%d\\n\\t"\n' % (52 + i, S_STAMP + 2 * i)
        for i in range(3):
            text += '  "s_mov_b64 %%[t%d], s[%d:%d]\\n\\t"\n' % (i, 52 + 2 * i, 53 + 2 * i)
    text += '  "s_nop 15\\n\\t"\n  "s_nop 7"\n'   # the last MFMAs' results -> whoever reads the accumulators next
    return text, nins, counts


def run_macro(c):
    outs = ", ".join('[a%d] "=&a"(accr[%d])' % (i, i) for i in range(c.NACC))
    ins = ['[rsx] "s"(rsx_lo)', '[rsx2] "s"(rsx_hi)', '[rsw] "s"(rsw_lo)', '[rsw2] "s"(rsw_hi)', '[rss] "s"(rss_lo)', '[rss2] "s"(rss_hi)']
    ins += ['[xr] "v"(x_row)', '[xc0] "v"(x_chunk[0])', '[xc1] "v"(x_chunk[1])', '[mlast] "s"(m_last)', '[k2] "s"(k2)']
    ins += ['[wv] "v"(w_voff)', '[sv] "v"(s_voff)', '[xrd] "v"(xrd)', '[xdst] "s"(xdst)', '[kb] "s"(kb_tpg)', '[ke] "s"(ke)', '[wps] "s"(w_pstride)', '[sps] "s"(s_pstride)', '[tv] "v"(t_voff)']
    cl = (['"memory"', '"scc"', '"m0"'] + ['"v%d"' % r for r in range(c.VBASE, c.VEND)] + ['"a%d"' % r for r in range(c.AXF, 128)] +
          ['"s%d"' % r for r in range(52, S_END)])
    body = "#define QA_XM_RUN_%s() asm volatile(QA_XM_ASM_%s : %s : %s : %s)\n" % (c.name, c.name, outs, ", ".join(ins), ", ".join(cl))
    cls = cl + ['"s%d"' % r for r in range(S_END, S_STAMP + 12)]
    stamps = ", ".join('[t%d] "=&s"(xm_t[%d])' % (i, i) for i in range(3))
    body += "#define QA_XM_RUN_STAMPED_%s() asm volatile(QA_XM_ASM_STAMPED_%s : %s, %s : %s : %s)\n" % (c.name, c.name, outs, stamps, ", ".join(ins), ", ".join(cls))
    return body


_RS = int(os.environ.get("XM_RS", "1"))   # ring stages of the 32-token configurations (A/B switch; two measured level with one: the second stage's pieces
                                          # sit in front of the weights in the in-order queue and the first tile starts later by what the loop gains)
_D = [int(x) for x in DQ.split(",")] if DQ else [4, 4, 2, 4, 4, 2]
CONFIGS = [Cfg(2, 1, _D[0]), Cfg(2, 2, _D[1], jitq=1), Cfg(2, 3, _D[2], VBASE=16, jitq=1), Cfg(1, 1, _D[3], RS=_RS), Cfg(1, 2, _D[4], RS=_RS, jitq=1),
           Cfg(1, 3, _D[5], VBASE=16, RS=_RS, jitq=1)]


def generate():
    body = ("// GENERATED by tools/gen_xm_loop.py -- do not edit.  The per-wave K loops of w4a16_xm_kernel, one inline-asm statement per tile\n"
            "// shape (operands and register plan: the generator's docstring and w4a16_xm.hpp).\n")
    info = []
    for c in CONFIGS:
        text, n, counts = build(c)
        body += "#define QA_XM_ASM_%s \\\n" % c.name
        body += "".join(l + " \\\n" for l in text.rstrip("\n").split("\n")) + "\n"
        stext, _, _ = build(c, stamped=True)
        body += "#ifdef QUICK_AMD_TOOLS\n#define QA_XM_ASM_STAMPED_%s \\\n" % c.name
        body += "".join(l + " \\\n" for l in stext.rstrip("\n").split("\n")) + "\n#endif\n"
        body += run_macro(c)
        info.append("config (%d, %d): %d instructions, VGPRs v%d..v%d, queue of %d stages, ring %d KiB per wave (%d stages); vmcnt prologue %s first %s loop %s"
                    % (c.MB, c.PR, n, c.VBASE, c.VEND - 1, c.D, c.RING // 1024, c.RS, counts[0], counts[1], [counts[2][d] for d in range(c.D)]))
    return body, info


def main():
    body, info = generate()
    for l in info:
        print(l)
    out = OUT if not EXP else OUT.replace(".inc", "_exp%d.inc" % EXP).replace(os.path.join("quick_amd", "csrc"), os.path.join("tools", "bin"))
    out = os.environ.get("XM_OUT", out)
    with open(out, "w") as f:
        f.write(body)
    print("wrote", out)


if __name__ == "__main__":
    main()
