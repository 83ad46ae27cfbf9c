"""Live VGPRs per instruction of a straight-line stretch of gfx950 assembly (backward scan, branches ignored).
usage: python tools/vgpr_live.py file.s first_line last_line [step]   (line numbers of file.s, 1-based)"""
import re
import sys

STORE_LIKE = ("global_store", "ds_write", "scratch_store", "buffer_store", "s_", "v_cmp", "global_atomic", "ds_max", "ds_add_u32")


def regs(tok):
    out = []
    for m in re.finditer(r"v\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.append(int(m.group(3)))
    return out


def main():
    lines = open(sys.argv[1]).read().split("\n")
    a, b = int(sys.argv[2]), int(sys.argv[3])
    step = int(sys.argv[4]) if len(sys.argv) > 4 else 50
    ins = []
    for n in range(a - 1, b):
        l = lines[n].split(";")[0].strip()
        if not l or l.endswith(":") or l.startswith("."):
            continue
        ins.append((n + 1, l))
    live = set()
    counts = []
    for n, l in reversed(ins):
        op, _, rest = l.partition(" ")
        toks = [t.strip() for t in rest.split(",")]
        if op.startswith(STORE_LIKE) and not op.startswith(("ds_add_rtn", "v_cmpx")):
            defs, uses = [], [r for t in toks for r in regs(t)]
        else:
            defs = regs(toks[0]) if toks else []
            uses = [r for t in toks[1:] for r in regs(t)]
            if op.startswith(("v_fmac", "v_mac", "v_cndmask_b32_dpp", "v_mov_b32_dpp")):
                uses += defs
        for r in defs:
            live.discard(r)
        live.update(uses)
        counts.append((n, len(live), op))
    counts.reverse()
    peak = max(c for _, c, _ in counts)
    print("peak live", peak)
    for i in range(0, len(counts), step):
        seg = counts[i:i + step]
        ops = {}
        for _, _, op in seg:
            k = op.split("_")[0] + "_" + op.split("_")[1] if "_" in op else op
            ops[k] = ops.get(k, 0) + 1
        top = sorted(ops.items(), key=lambda kv: -kv[1])[:4]
        print(f"lines {seg[0][0]:5d}-{seg[-1][0]:5d}  live min {min(c for _, c, _ in seg):3d} max {max(c for _, c, _ in seg):3d}  {top}")


main()
