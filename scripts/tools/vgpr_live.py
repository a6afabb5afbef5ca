#!/usr/bin/env python3
"""Which VGPRs are live at a given line of a gfx9 kernel's assembly (hipcc -S)?  A small backwards data-flow over the basic blocks:
good enough to see what the register allocator keeps across an inline-asm block that clobbers v96..v127 (csrc/fused_loop.h).
usage: vgpr_live.py kernel.s <line> [<line> ...]      (line numbers inside kernel.s, 1-based)"""
import re
import sys

STORE = re.compile(r"^(ds_write|ds_bpermute|scratch_store|buffer_store|global_store|flat_store|s_|v_cmp|v_cmpx|v_readlane|v_readfirstlane|ds_atomic|ds_add_u32|ds_or|ds_max|ds_min)")
REG = re.compile(r"v\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs(tok):
    out = []
    for m in REG.finditer(tok):
        if m.group(3) is not None:
            out.append(int(m.group(3)))
        else:
            out += list(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def parse(lines):
    ins = []          # (lineno, defs, uses, label, branch_target, falls_through)
    for no, raw in enumerate(lines, 1):
        s = raw.split(";")[0].strip()
        if not s:
            continue
        if s.endswith(":"):
            ins.append((no, [], [], s[:-1], None, True))
            continue
        if s.startswith("."):
            continue
        if s.startswith("v_add_f32_e32 ") or "\n" in s:
            pass
        op, _, rest = s.partition(" ")
        ops = [o.strip() for o in rest.split(",")] if rest else []
        defs, uses = [], []
        tgt, ft = None, True
        if op.startswith("s_cbranch"):
            tgt = ops[0]
        elif op == "s_branch":
            tgt, ft = ops[0], False
        elif op in ("s_endpgm",):
            ft = False
        if STORE.match(op) and not op.startswith("ds_bpermute"):
            for o in ops:
                uses += regs(o)
            if op.startswith(("v_readlane", "v_readfirstlane", "v_cmp")):
                pass
        else:
            if ops:
                defs = regs(ops[0])
                for o in ops[1:]:
                    uses += regs(o)
                if op.startswith(("v_fmac", "v_mac", "v_dot", "v_writelane")) or "dpp" in s or "sdwa" in s:
                    uses += defs
        ins.append((no, defs, uses, None, tgt, ft))
    return ins


def main():
    path = sys.argv[1]
    want = [int(x) for x in sys.argv[2:]]
    lines = open(path).read().splitlines()
    # inline asm blocks: treat "v_add_f32_e32 vX, vX, vY" etc. normally; they are ordinary lines in the .s
    ins = parse(lines)
    label_at = {i[3]: k for k, i in enumerate(ins) if i[3]}
    n = len(ins)
    succ = [[] for _ in range(n)]
    for k, (no, d, u, lab, tgt, ft) in enumerate(ins):
        if ft and k + 1 < n:
            succ[k].append(k + 1)
        if tgt and tgt in label_at:
            succ[k].append(label_at[tgt])
    live_in = [set() for _ in range(n)]
    changed = True
    while changed:
        changed = False
        for k in range(n - 1, -1, -1):
            out = set()
            for s_ in succ[k]:
                out |= live_in[s_]
            new = (out - set(ins[k][1])) | set(ins[k][2])
            if new != live_in[k]:
                live_in[k] = new
                changed = True
    by_line = {i[0]: k for k, i in enumerate(ins)}
    for w in want:
        k = by_line.get(w)
        while k is None and w < len(lines):
            w += 1
            k = by_line.get(w)
        lv = sorted(live_in[k])
        print("line %d: %d VGPRs live: %s" % (w, len(lv), " ".join("v%d" % r for r in lv)))


if __name__ == "__main__":
    main()
