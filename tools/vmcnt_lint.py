#!/usr/bin/env python3
"""Static check of the vector-memory wait counts of gfx950 kernels (r06): every instruction that touches a VGPR / AGPR which a
vector-memory LOAD still in flight is going to write must sit behind an `s_waitcnt vmcnt(N)` that covers that load.

Why it exists.  r04 and r05 each met a build whose results were wrong and varied from run to run and which `-mllvm -amdgpu-waitcnt-forcezero`
cured: hipcc's wait insertion (SIInsertWaitcnts) had lost track of requests whose results die on a loop's way out -- the
registers they were going to write were handed to other values behind the loop, with a wait that did not cover them, and the late data landed on
top.  The generated K loops (tools/gen_xw_loop.py, tools/gen_xm_loop.py) carry counted waits of their own.  This checker does not trust either:
it reads the ISA.

Model (gfx9 family: one counter, vmcnt, for vector-memory loads; they return data in issue order).  Abstract state per program point: for every
VGPR / AGPR with a load in flight, a lower bound on the number of LOADS issued behind that load ("younger").  `s_waitcnt vmcnt(N)` completes every
load with younger >= N.  A load: younger += 1 for everything pending, its destination registers become pending with younger = 0 (LDS-DMA loads
have no destination register).  Stores and atomics without return sit in the same counter and, on the gfx9 family (no separate store counter), in the
same in-order queue: they count as younger -- the assumption hipcc itself compiles with (SIInsertWaitcnts: one event type behind vmcnt before gfx10);
tools/vmcnt_order_probe.hip checks the in-order return across load classes on the part.  Joins take the union of the pending registers and the
minimum of younger; loops are iterated to a fixed point.  Any other touch of a pending register -- read or write, VALU, MFMA, LDS, store data,
address of another load -- is a finding.

Input: `llvm-objdump -d` of a code object, or hipcc's -S output.     python tools/vmcnt_lint.py <file> [kernel-name-regex]
"""
import re
import sys

LOAD_RE = re.compile(r"^(buffer_load|global_load|flat_load|scratch_load|tbuffer_load|image_load|image_sample|buffer_atomic|global_atomic|flat_atomic)")
STORE_RE = re.compile(r"^(buffer_store|global_store|flat_store|scratch_store|tbuffer_store|image_store|buffer_wbl2|buffer_inv|buffer_wbinvl1|buffer_gl)")
REG_RE = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")
TERMINATORS = ("s_endpgm",)


def regs_of(operand_text):
    out = []
    for m in REG_RE.finditer(operand_text):
        if m.group(1):
            out.append((m.group(1), int(m.group(2))))
        else:
            out += [(m.group(3), r) for r in range(int(m.group(4)), int(m.group(5)) + 1)]
    return out


class Ins:
    __slots__ = ("text", "mn", "ops", "line", "target", "edge")

    def __init__(self, text, line):
        self.text, self.line = text, line
        parts = text.split(None, 1)
        self.mn = parts[0]
        self.ops = parts[1] if len(parts) > 1 else ""
        self.target = None
        self.edge = False


def parse(path):
    """-> {kernel: [items]} where an item is ('label', name) or ('ins', Ins).  Understands hipcc -S output and llvm-objdump -d output."""
    kernels, cur, name = {}, None, None
    objdump_label = re.compile(r"^([0-9a-f]+) <([^>]+)>:")
    for ln, raw in enumerate(open(path, errors="replace"), 1):
        line = raw.rstrip("\n")
        m = objdump_label.match(line)
        if m:                                              # objdump: a symbol -- a kernel entry or a basic-block label (L<n> with --symbolize-operands)
            sym = m.group(2)
            if re.match(r"^L\d+$", sym) and cur is not None:
                cur.append(("label", sym))
            else:
                name, cur = sym, []
                kernels[name] = cur
            continue
        m = re.match(r"^<(L\d+)>:", line)
        if m and cur is not None:
            cur.append(("label", m.group(1)))
            continue
        m = re.match(r"^([A-Za-z_.$][\w.$]*):", line)     # -S: label in column 0
        if m:
            sym = m.group(1)
            if sym.startswith(".L") or sym.startswith("L"):
                if cur is not None:
                    cur.append(("label", sym))
            elif sym.startswith("_Z") or sym.startswith("w4a16") or sym.startswith("quick"):
                name, cur = sym, []
                kernels[name] = cur
            continue
        if cur is None:
            continue
        body = line.split("//")[0].split(";")[0].strip()
        if not body or body.startswith("."):
            continue
        body = re.sub(r"^\s*[0-9a-f]+:\s+", "", body)      # (objdump without --no-leading-addr)
        if not re.match(r"^[a-z_]+[a-z0-9_]*\b", body):
            continue
        ins = Ins(body, ln)
        if ins.mn.startswith("s_cbranch") or ins.mn == "s_branch":
            t = ins.ops.split()[-1] if ins.ops else ""
            ins.target = t.strip("<>")
        cur.append(("ins", ins))
    return kernels


def blocks_of(items):
    """basic blocks: list of (label or None, [Ins], successors-by-label, falls_through)"""
    blocks, cur, label = [], [], None
    for kind, v in items:
        if kind == "label":
            if cur or label is not None:
                blocks.append([label, cur, None, True])
            cur, label = [], v
        else:
            cur.append(v)
            if v.mn.startswith("s_cbranch") or v.mn == "s_branch" or v.mn in TERMINATORS or v.mn.startswith("s_setpc") or v.mn.startswith("s_swappc"):
                blocks.append([label, cur, v, v.mn.startswith("s_cbranch")])
                cur, label = [], None
    if cur or label is not None:
        blocks.append([label, cur, None, True])
    return blocks


def vmcnt_of(ins):
    if ins.mn != "s_waitcnt":
        return None
    m = re.search(r"vmcnt\((\d+)\)", ins.ops)
    if m:
        return int(m.group(1))
    m = re.match(r"^(0x[0-9a-f]+|\d+)$", ins.ops.strip())
    if m:                                                   # raw immediate: vmcnt = bits 3:0 and 15:14
        imm = int(m.group(1), 0)
        return (imm & 15) | (((imm >> 14) & 3) << 4)
    return None


def check_kernel(name, items, max_findings=8):
    blocks = blocks_of(items)
    index = {b[0]: i for i, b in enumerate(blocks) if b[0] is not None}
    succ = []
    for i, (label, body, term, falls) in enumerate(blocks):
        s = []
        if term is not None and term.target is not None and (term.mn.startswith("s_cbranch") or term.mn == "s_branch"):
            if term.target in index:
                s.append(index[term.target])
            else:
                return [f"{name}: branch target {term.target!r} not found (line {term.line})"], 0
        if term is None or term.mn.startswith("s_cbranch"):
            if i + 1 < len(blocks):
                s.append(i + 1)
        succ.append(s)
    state_in = [None] * len(blocks)
    state_in[0] = {}
    work, findings, seen = [0], [], set()
    loads = 0
    while work:
        b = work.pop()
        st = dict(state_in[b])
        for ins in blocks[b][1]:
            n = vmcnt_of(ins)
            if n is not None:
                st = {r: y for r, y in st.items() if y < n}
                continue
            is_load = bool(LOAD_RE.match(ins.mn))
            returns = is_load and not (ins.mn.startswith(("buffer_atomic", "global_atomic", "flat_atomic")) and " glc" not in " " + ins.ops and " sc0" not in " " + ins.ops)
            ops = ins.ops
            touched = regs_of(ops)
            if is_load and returns:
                lds = bool(re.search(r"\blds\b", ops))
                first = ops.split(",")[0]
                dest = [] if lds else regs_of(first)
                other = [r for r in touched if r not in dest] if not lds else touched
                for r in other:
                    if r in st and (ins.line, r) not in seen:
                        seen.add((ins.line, r))
                        findings.append(f"{name}: line {ins.line}: `{ins.text}` uses {r[0]}{r[1]} as an address while a load into it is in flight (younger >= {st[r]})")
                for r in st:
                    st[r] += 1
                for r in dest:
                    st[r] = 0
                loads += 1
                continue
            for r in touched:
                if r in st and (ins.line, r) not in seen:
                    seen.add((ins.line, r))
                    findings.append(f"{name}: line {ins.line}: `{ins.text}` touches {r[0]}{r[1]} while a load into it may still be in flight (at least {st[r]} load(s) were issued behind it; no covering s_waitcnt vmcnt)")
            if STORE_RE.match(ins.mn) or (is_load and not returns):
                for r in st:                                # (a store sits in the same in-order queue: it counts as younger -- what hipcc assumes on gfx9, see the docstring)
                    st[r] += 1
        for s in succ[b]:
            old = state_in[s]
            if old is None:
                state_in[s] = dict(st)
                work.append(s)
            else:
                changed = False
                for r, y in st.items():
                    if r not in old or y < old[r]:
                        old[r] = y
                        changed = True
                if changed:
                    work.append(s)
        if len(findings) >= max_findings:
            break
    return findings, loads


def main():
    path = sys.argv[1]
    pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
    kernels = parse(path)
    total, bad = 0, 0
    for name, items in kernels.items():
        if pat and not pat.search(name):
            continue
        if not any(k == "ins" and v.mn == "s_endpgm" for k, v in items):
            continue
        f, loads = check_kernel(name, items)
        total += 1
        if f:
            bad += 1
            print("\n".join(f))
    print(f"{total} kernels checked, {bad} with findings")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
