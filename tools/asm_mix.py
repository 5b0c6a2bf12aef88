"""Instruction mix of the hottest basic blocks of a kernel in a hipcc -save-temps .s file.
usage: python tools/asm_mix.py file.s <substring of the mangled kernel name> [n_blocks]"""
import re
import sys
from collections import Counter


def kind(i):
    if i.startswith("v_mfma"):
        return "mfma"
    if i.startswith("v_accvgpr"):
        return "acc"
    if i.startswith("v_"):
        return "valu"
    if i.startswith("s_waitcnt"):
        return "wait"
    if i.startswith("s_barrier"):
        return "barrier"
    if i.startswith("s_"):
        return "salu"
    if i.startswith("ds_"):
        return "ds"
    if i.startswith(("buffer_", "global_", "scratch_", "flat_")):
        return "vmem"
    return i


def main():
    s = open(sys.argv[1]).read()
    pat = sys.argv[2]
    nb = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    for m in re.finditer(r"^(_Z\S*):\s*;\s*@\S+\n(.*?)\.Lfunc_end\d+:", s, re.S | re.M):
        if pat not in m.group(1):
            continue
        body = m.group(2)
        blocks = re.split(r"\n(?=\.LBB[0-9_]+:)", body)
        print(m.group(1), "blocks", len(blocks))
        for b in sorted(blocks, key=lambda b: -b.count("v_mfma"))[:nb]:
            lines = [l.strip() for l in b.split("\n")]
            ins = [l.split()[0] for l in lines if l and not l.startswith((".", ";", "/")) and not l.endswith(":")]
            c = Counter(kind(i) for i in ins)
            print("   ", lines[0][:20], dict(c))


main()
