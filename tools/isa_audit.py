#!/usr/bin/env python3
"""Instruction mix of one kernel of a hipcc -save-temps .s file, split into what the source's asm statements
emitted (A:) and what the compiler added around them (C:), plus the things a hand-owned register file must not
show: compiler v_accvgpr_* / scratch traffic outside the asm blocks, and compiler waits inside the tile loop.
usage: tools/isa_audit.py file.s [kernel-name-substring]"""
import sys


def main():
    s = open(sys.argv[1]).read()
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    for chunk in s.split(".end_amdhsa_kernel"):
        names = [l for l in chunk.split("\n") if l.startswith("_Z") and ":" in l]
        if not names or want not in names[0]:
            continue
        print("==", names[0])
        inasm = False
        cnt, outside, waits = {}, [], []
        regions, reg = [], [0, 0, 0]            # between s_barriers: [asm MFMAs, compiler s_nops, compiler v_movs]
        for i, l in enumerate(chunk.split("\n")):
            t = l.strip()
            if t.startswith(";;#ASMSTART"):
                inasm = True
                continue
            if t.startswith(";;#ASMEND"):
                inasm = False
                continue
            if not t or t[0] in ";." or t.split(";")[0].strip().endswith(":"):
                continue
            op = t.split()[0]
            key = ("A:" if inasm else "C:") + op
            cnt[key] = cnt.get(key, 0) + 1
            if not inasm and ("accvgpr" in op or "scratch" in op):
                outside.append((i, t))
            if not inasm and op == "s_waitcnt":
                waits.append((i, t))
            if op == "s_barrier":
                regions.append(tuple(reg))
                reg = [0, 0, 0]
            reg[0] += inasm and op.startswith("v_mfma")
            reg[1] += (not inasm) and op == "s_nop"
            reg[2] += (not inasm) and op.startswith("v_mov_b")
        for k in sorted(cnt, key=lambda x: -cnt[x])[:70]:
            print(f"  {k:34s}{cnt[k]}")
        print("compiler accvgpr/scratch outside asm:", len(outside), outside[:8])
        print("regions between barriers (asm MFMAs, compiler s_nops, compiler v_movs):", regions)
        print("compiler s_waitcnt:", len(waits))
        for w in waits:
            print("   ", w)


if __name__ == "__main__":
    main()
