#!/usr/bin/env python3
"""Static check of the hand-counted loads in the persistent recurrence kernels.

The BPTT kernels issue their operand tiles with inline-asm `global_load_* ... sc1` and wait for
them with hand-written `s_waitcnt vmcnt(N)` (loads return in order).  The compiler does not know
that the destination registers of such a load are in flight: if its register allocator decides to
copy one of them (live-range split, a phi at a control-flow merge) it does so with a plain v_mov
that reads the register BEFORE the data has arrived.  Round 6 met exactly that -- a few stale
gradient elements per update that came and went with unrelated edits.

This tool reads the gfx950 assembly of a kernel (hipcc -S), cuts it into basic blocks and walks
every path of the control-flow graph with the queue of VMEM operations in flight: an inline-asm
load (between #ASMSTART / #ASMEND) puts its destination registers on the queue, compiler-issued
VMEM operations are counted as the hardware counts them, `s_waitcnt vmcnt(N)` retires all but the
youngest N entries, and an instruction that names a register which is still in flight on some
path is reported.  The walk follows one scalar idiom (the structuriser's duplicated paths selected
by a 0 / -1 flag, see check()); every other branch is taken both ways.

usage: check_inflight_loads.py file.s kernel_name_substring [...]
exit code 1 when a hazard is found, 2 when a named kernel is not in the file.
"""
import re
import sys

_REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")
_VMEM = re.compile(r"^\s*(global_load|global_store|global_atomic|buffer_load|buffer_store|buffer_atomic|flat_load|flat_store|flat_atomic|scratch_load|scratch_store)")


def regs_of(text):
    out = set()
    for m in _REG.finditer(text):
        if m.group(1):
            for i in range(int(m.group(2)), int(m.group(3)) + 1):
                out.add((m.group(1), i))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def kernel_bodies(path):
    """name -> list of (line number, text, inside inline asm) of every function in the file; labels are kept as text ending in ':'"""
    bodies, cur, in_asm = {}, None, False
    with open(path) as f:
        for no, line in enumerate(f, 1):
            if "#ASMSTART" in line:
                in_asm = True
            elif "#ASMEND" in line:
                in_asm = False
            s = line.split(";")[0].strip()
            if not s:
                continue
            m = re.match(r"^([A-Za-z_.$][\w$.]*):$", s)
            if m and not s.startswith(".L"):
                cur = []
                bodies[m.group(1)] = cur
                continue
            if cur is None:
                continue
            if s.startswith(".Lfunc_end"):
                cur = None
                continue
            if m:
                cur.append((no, s, False))
            elif not s.startswith("."):
                cur.append((no, s, in_asm))
    return bodies


def basic_blocks(body):
    """-> (blocks: list of instruction lists, succ: list of successor index lists)"""
    blocks, labels = [[]], {}
    for no, ins, in_asm in body:
        if ins.endswith(":"):
            if blocks[-1]:
                blocks.append([])
            labels[ins[:-1]] = len(blocks) - 1
            continue
        blocks[-1].append((no, ins, in_asm))
        if ins.split()[0].startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc")):
            blocks.append([])
    succ = []
    for i, blk in enumerate(blocks):
        nxt = [i + 1] if i + 1 < len(blocks) else []
        if blk:
            op = blk[-1][1].split()
            if op[0] == "s_branch":
                nxt = [labels[op[1]]] if op[1] in labels else []
            elif op[0].startswith("s_cbranch"):
                nxt = nxt + ([labels[op[1]]] if op[1] in labels else [])
            elif op[0] in ("s_endpgm", "s_setpc_b64"):
                nxt = []
        succ.append(nxt)
    return blocks, succ


def norm(queue):
    """oldest untracked operations do not matter: they retire first"""
    q = list(queue)
    while q and not q[0]:
        q.pop(0)
    return tuple(q[-63:])


_SPAIR = re.compile(r"^s\[(\d+):(\d+)\]$")


def _sdst(ins):
    """scalar registers an instruction writes (first operand): set of register numbers, 'vcc' in it when vcc is written"""
    parts = ins.split(None, 1)
    if len(parts) < 2:
        return set()
    op, first = parts[0], parts[1].split(",")[0].strip()
    out = set()
    m = _SPAIR.match(first)
    if m:
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    elif re.match(r"^s\d+$", first):
        out.add(int(first[1:]))
    elif first.startswith("vcc"):
        out.add("vcc")
    if op.startswith(("v_cmp", "v_div_scale", "v_add_co", "v_sub_co", "v_addc_co", "v_subb_co", "v_mad_u64", "v_mad_i64")) and ("vcc" in ins or op.endswith("_e32")):
        out.add("vcc")
    return out


def run_block(blk, queue, consts, vcc, hazards):
    """consts: known 64-bit scalar flags {(lo, hi): 0 | -1} (the structuriser's "which copy of the path runs" words); vcc: None | 'zero' | 'nonzero'"""
    queue, consts = list(queue), dict(consts)
    for no, ins, in_asm in blk:
        op = ins.split()[0]
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", ins)
            if m:
                keep = int(m.group(1))
                queue = queue[len(queue) - keep:] if keep < len(queue) else queue
                if keep == 0:
                    queue = []
            elif "cnt" not in ins.split(None, 1)[1]:
                queue = []        # raw immediate: treat as a full wait (not produced by this code base)
            continue
        inflight = frozenset().union(*queue) if queue else frozenset()
        if inflight:
            bad = regs_of(ins.split(None, 1)[1] if " " in ins else "") & inflight
            if bad:
                hazards[no] = (ins, sorted(bad))
        if _VMEM.match(ins):
            dst = frozenset()
            if in_asm and op.startswith("global_load") and "lds" not in op:      # the compiler tracks its own loads
                dst = frozenset(regs_of(ins.split(None, 1)[1].split(",")[0]))
            queue.append(dst)
        # scalar flags
        if op.startswith("s_cbranch") or op == "s_branch":
            continue
        m = re.match(r"^s_andn2_b64\s+vcc,\s*exec,\s*s\[(\d+):(\d+)\]$", ins)
        if m and (int(m.group(1)), int(m.group(2))) in consts:
            vcc = "nonzero" if consts[(int(m.group(1)), int(m.group(2)))] == 0 else "zero"       # (a wave that runs has a lane in exec)
            continue
        w = _sdst(ins) if not in_asm else set()
        if "vcc" in w:
            vcc = None
        m = re.match(r"^s_mov_b64\s+s\[(\d+):(\d+)\],\s*(-1|0)$", ins)
        for k in [k for k in consts if w & set(range(k[0], k[1] + 1))]:
            del consts[k]
        if m:
            consts[(int(m.group(1)), int(m.group(2)))] = int(m.group(3))
    return norm(queue), tuple(sorted(consts.items())), vcc


def check(body):
    """walk the paths of the control-flow graph with the queue of VMEM operations in flight; -> (hazards, inline-asm loads in the text).
    Path-sensitive for ONE idiom: the structuriser duplicates a path and selects the copy with a 64-bit scalar flag set to 0 / -1
    (s_mov_b64 ... ; s_andn2_b64 vcc, exec, flag ; s_cbranch_vccnz) -- the infeasible combination "both copies" is not walked."""
    blocks, succ = basic_blocks(body)
    hazards, seen, work = {}, set(), [(0, (), (), None)]
    tracked = sum(1 for b in blocks for no, ins, in_asm in b if in_asm and ins.startswith("global_load") and "lds" not in ins.split()[0])
    while work:
        i, q, consts, vcc = work.pop()
        if (i, q, consts, vcc) in seen:
            continue
        seen.add((i, q, consts, vcc))
        if len(seen) > 400000:
            raise RuntimeError("state explosion")
        q2, c2, v2 = run_block(blocks[i], q, consts, vcc, hazards)
        nxt = succ[i]
        if blocks[i] and len(nxt) == 2 and v2 is not None:
            term = blocks[i][-1][1].split()[0]
            if term in ("s_cbranch_vccnz", "s_cbranch_vccz"):
                taken = (v2 == "nonzero") == (term == "s_cbranch_vccnz")
                nxt = [nxt[1]] if taken else [nxt[0]]      # (basic_blocks lists the fall-through first, the branch target second)
        for j in nxt:
            work.append((j, q2, c2, v2))
    return [(no, ins, bad) for no, (ins, bad) in sorted(hazards.items())], tracked


def main(argv):
    if len(argv) < 3:
        print(__doc__)
        return 2
    bodies = kernel_bodies(argv[1])
    rc = 0
    for want in argv[2:]:
        names = [n for n in bodies if want in n]
        if not names:
            print("no function matching %r in %s" % (want, argv[1]))
            return 2
        for n in names:
            hz, tracked = check(bodies[n])
            print("%s: %d inline-asm loads followed, %d hazards" % (n, tracked, len(hz)))
            for no, ins, bad in hz[:20]:
                print("  line %d: %s   <- in flight: %s" % (no, ins, " ".join("%s%d" % b for b in bad)))
            if hz:
                rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv))
