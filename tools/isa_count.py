"""Static instruction counts per kernel and per loop of a gfx950 assembly listing
(hipcc -S --cuda-device-only).  usage: python tools/isa_count.py file.s [name-filter]"""
import collections
import re
import sys


def classify(op):
    if op.startswith(("v_log", "v_exp", "v_rcp", "v_sqrt", "v_rsq", "v_sin", "v_cos")):
        return "trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return op
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return op
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    text = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for m in re.finditer(r"\n(_Z\w+):[^\n]*\n(.*?)\n\.Lfunc_end", text, re.S):
        name, body = m.group(1), m.group(2)
        if flt not in name:
            continue
        lines = []
        labels = {}
        for raw in body.split("\n"):
            line = raw.split(";")[0].strip()
            if not line:
                continue
            if line.endswith(":"):
                labels[line[:-1]] = len(lines)
                continue
            if line.startswith("."):
                continue
            lines.append(line)
        total = collections.Counter(classify(l.split()[0]) for l in lines)
        print(name)
        print("  whole function:", dict(total))
        for i, l in enumerate(lines):
            parts = l.split()
            if parts[0].startswith(("s_cbranch", "s_branch")) and parts[-1] in labels and labels[parts[-1]] <= i:
                seg = lines[labels[parts[-1]]: i + 1]
                c = collections.Counter(classify(x.split()[0]) for x in seg)
                print(f"  loop {parts[-1]} ({len(seg)} instr):", dict(c))


main()
