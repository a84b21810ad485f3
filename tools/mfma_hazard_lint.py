#!/usr/bin/env python3
"""Static check of the software-managed MFMA hazards of gfx950 kernels (r06).

gfx950 does not interlock a matrix-core instruction against the vector ALU (or memory / LDS instructions) around it: software keeps wait states
between them.  hipcc's hazard recognizer pads them -- for instructions it can see.  An instruction inside an asm statement it cannot (and this
repository has two kinds: w4a16_common.hpp's and_or(), a v_and_or_b32 as inline asm inside hipcc-scheduled code, and the generated K loops, whose
generators assert their own distances).  This checker trusts neither and reads the ISA.

The distances are MEASURED on the part, not taken from a manual (tools/mfma_valu_read_hazard.hip, tools/mfma_valu_write_hazard.hip,
profiles/r06_mfma_hazards.txt; one wave alone and eight per CU on every CU give the same table; a wait state = one s_nop 0 = one issued instruction):

    MFMA writes vDst   ->  anything but an MFMA READS it      4-pass (16x16x32_f16) >= 7     8-pass (32x32x16_f16) >= 10
    MFMA writes vDst   ->  anything but an MFMA WRITES it     4-pass >= 4                     8-pass >= 8
    MFMA reads SrcC    ->  anything WRITES it                 4-pass >= 0                     8-pass >= 3
    VALU writes a VGPR ->  MFMA reads it (SrcA / SrcB / SrcC) >= 1
    (MFMA -> MFMA through SrcC, same or another vDst: interlocked, 0.  MFMA reads SrcA / SrcB -> VALU writes them: 0.)
Other matrix-core shapes are priced by their pass count with the same offsets (passes + 3 / passes + 0 / passes - 5, floor 0); 16 passes and
more: + 2 each on top, untested here and conservative.

Method: per kernel, every straight-line window -- the layout order between unconditional branches, and across every branch edge the instructions in
front of the branch followed by those at its target -- is walked with the wait states counted (an s_nop N counts N + 1, everything else 1; an MFMA
that depends on the one being tracked through SrcC stalls until it is done and clears the hazard).

    python tools/mfma_hazard_lint.py <llvm-objdump -d output | hipcc -S output> [kernel-name-regex]
"""
import re
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from vmcnt_lint import parse, regs_of  # noqa: E402

WINDOW = 14


def passes_of(mn):
    """pass count of an MFMA mnemonic (4 cycles per pass)"""
    m = re.match(r"v_s?mfma[c]?_\w+?_(\d+)x(\d+)x(\d+)_?(\w*)", mn)
    if not m:
        return 16
    M, N, K = int(m.group(1)), int(m.group(2)), int(m.group(3))
    if (M, N) == (16, 16):
        return 4 if K <= 32 else 8 if K <= 64 else 16      # 16x16x32 f16 / bf16: 4; 16x16x64 / 128 (8-bit, f8f6f4): 8 (priced as such)
    if (M, N) == (32, 32):
        return 8 if K <= 16 else 16
    if (M, N) == (4, 4):
        return 2
    return 16


def nonmfma_read_ws(p): return p + 3 if p <= 4 else (p + 2 if p <= 8 else p + 4)
def nonmfma_write_ws(p): return p if p <= 8 else p + 2
def srcc_war_ws(p): return max(0, p - 5) if p <= 8 else p - 3


def split_ops(ins):
    ops = [o.strip() for o in ins.ops.split(",")] if ins.ops else []
    return ops


def is_mfma(ins): return ins.mn.startswith(("v_mfma", "v_smfmac"))


def dest_and_srcs(ins):
    """(registers written, registers read) of a non-MFMA instruction, conservatively: stores / LDS writes / exports read everything; LDS-DMA loads write
    nothing; everybody else writes operand 0 and reads the rest (an instruction that also reads its destination is covered by the write check)"""
    ops = split_ops(ins)
    mn = ins.mn
    if not ops:
        return [], []
    if mn.startswith(("buffer_store", "global_store", "flat_store", "scratch_store", "ds_write", "ds_store", "exp", "buffer_atomic", "global_atomic", "ds_add", "ds_max", "ds_min")):
        return [], regs_of(ins.ops)
    if mn.startswith(("buffer_load", "global_load", "flat_load")) and re.search(r"\blds\b", ins.ops):
        return [], regs_of(ins.ops)
    if mn.startswith(("s_", "v_cmp", "v_cmpx", "v_readlane", "v_readfirstlane")):
        return [], regs_of(ins.ops)
    if mn.startswith("v_swap"):
        return regs_of(ins.ops), regs_of(ins.ops)
    return regs_of(ops[0]), regs_of(",".join(ops[1:]))


def wait_states(ins):
    if ins.mn == "s_nop":
        m = re.match(r"(\d+|0x[0-9a-f]+)", ins.ops.strip())
        return (int(m.group(1), 0) if m else 0) + 1
    return 1


def check_window(name, seq, findings, seen):
    for i, ins in enumerate(seq):
        if is_mfma(ins):
            ops = split_ops(ins)
            p = passes_of(ins.mn)
            dst, srcc = set(regs_of(ops[0])), set(regs_of(ops[3])) if len(ops) > 3 else set()
            ws = 0
            for j in range(i + 1, min(i + 1 + WINDOW, len(seq))):
                nxt = seq[j]
                if j > 0 and (seq[j - 1].mn in ("s_branch", "s_endpgm") or seq[j - 1].mn.startswith(("s_setpc", "s_swappc"))) and not getattr(seq[j - 1], "edge", False):
                    break                                      # layout order is not execution order behind an unconditional branch (its edge is a window of its own)
                if is_mfma(nxt):
                    nops = split_ops(nxt)
                    if len(nops) > 3 and dst & set(regs_of(nops[3])):
                        break                                  # dependent through SrcC: the hardware holds it until the result is there
                    if dst & set(regs_of(nops[0])) and not (len(nops) > 3 and set(regs_of(nops[0])) <= set(regs_of(nops[3]))):
                        pass                                   # (an MFMA overwriting another's vDst: in order in the matrix pipe)
                    ws += wait_states(nxt)
                    continue
                w, r = dest_and_srcs(nxt)
                key = (ins.line, nxt.line)
                if key not in seen:
                    if dst & set(r) and ws < nonmfma_read_ws(p):
                        seen.add(key)
                        findings.append(f"{name}: line {nxt.line}: `{nxt.text}` READS the result of `{ins.text}` (line {ins.line}) after {ws} wait state(s); {p}-pass MFMA: >= {nonmfma_read_ws(p)}")
                    elif dst & set(w) and ws < nonmfma_write_ws(p):
                        seen.add(key)
                        findings.append(f"{name}: line {nxt.line}: `{nxt.text}` OVERWRITES the destination of `{ins.text}` (line {ins.line}) after {ws} wait state(s); {p}-pass MFMA: >= {nonmfma_write_ws(p)}")
                    elif srcc & set(w) and ws < srcc_war_ws(p):
                        seen.add(key)
                        findings.append(f"{name}: line {nxt.line}: `{nxt.text}` overwrites SrcC of `{ins.text}` (line {ins.line}) after {ws} wait state(s); {p}-pass MFMA: >= {srcc_war_ws(p)}")
                ws += wait_states(nxt)
                if ws >= 20:
                    break
        elif ins.mn.startswith("v_") and not ins.mn.startswith(("v_cmp", "v_readlane", "v_readfirstlane")) and i + 1 < len(seq) and is_mfma(seq[i + 1]):
            w, _ = dest_and_srcs(ins)
            nxt = seq[i + 1]
            nops = split_ops(nxt)
            used = set(regs_of(",".join(nops[1:])))
            if set(w) & used and (ins.line, nxt.line) not in seen:
                seen.add((ins.line, nxt.line))
                findings.append(f"{name}: line {nxt.line}: `{nxt.text}` reads a register `{ins.text}` wrote in the instruction right in front of it (>= 1 wait state)")


def check_kernel(name, items, max_findings=12):
    seq, labels = [], {}
    for kind, v in items:
        if kind == "label":
            labels[v] = len(seq)
        else:
            seq.append(v)
    findings, seen = [], set()
    # layout order, cut at unconditional control flow
    start = 0
    for i, ins in enumerate(seq):
        if ins.mn in ("s_branch", "s_endpgm") or ins.mn.startswith(("s_setpc", "s_swappc")):
            check_window(name, seq[start:i + 1], findings, seen)
            start = i + 1
    check_window(name, seq[start:], findings, seen)
    # across every branch edge
    for i, ins in enumerate(seq):
        if (ins.mn.startswith("s_cbranch") or ins.mn == "s_branch") and ins.target in labels:
            t = labels[ins.target]
            k = max(0, i - WINDOW)
            for j in range(i - 1, k - 1, -1):               # the window in front of the branch starts behind the last unconditional branch
                if seq[j].mn in ("s_branch", "s_endpgm") or seq[j].mn.startswith(("s_setpc", "s_swappc")):
                    k = j + 1
                    break
            ins.edge = True                                 # (this branch IS followed by its target here)
            check_window(name, seq[k:i + 1] + seq[t:t + WINDOW], findings, seen)
            ins.edge = False
    return findings[:max_findings], sum(1 for s in seq if is_mfma(s))


def main():
    path = sys.argv[1]
    pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
    total, bad, mfmas = 0, 0, 0
    for name, items in parse(path).items():
        if pat and not pat.search(name):
            continue
        if not any(k == "ins" and v.mn == "s_endpgm" for k, v in items):
            continue
        f, n = check_kernel(name, items)
        total += 1
        mfmas += n
        if f:
            bad += 1
            print("\n".join(f))
    print(f"{total} kernels checked ({mfmas} MFMAs), {bad} with findings")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
