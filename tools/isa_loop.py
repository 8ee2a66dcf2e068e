#!/usr/bin/env python3
"""Per-iteration instruction census of prefill_pw_kernel from a -save-temps .s: regions between the iteration-end
s_barrier statements; A: = from the source's asm statements, C: = added by the compiler."""
import sys
s = open(sys.argv[1]).read().split('.end_amdhsa_kernel')[0].split('\n')
bars = [i for i, l in enumerate(s) if 's_barrier' in l]
print("barriers at", bars)
for k in range(len(bars) - 1):
    inasm = False; cnt = {}
    for l in s[bars[k]:bars[k + 1]]:
        t = l.strip()
        if t.startswith(';;#ASMSTART'): inasm = True; continue
        if t.startswith(';;#ASMEND'): inasm = False; continue
        if not t or t[0] in ';.' or t.split(';')[0].strip().endswith(':'): continue
        op = t.split()[0]; key = ('A:' if inasm else 'C:') + op
        cnt[key] = cnt.get(key, 0) + 1
    tot = sum(cnt.values())
    if tot < 300: continue
    print(f"--- region {k}: lines {bars[k]}..{bars[k+1]}: {tot} instructions; asm {sum(v for x, v in cnt.items() if x[0]=='A')}, compiler {sum(v for x, v in cnt.items() if x[0]=='C')}")
    print("   ", {x: c for x, c in sorted(cnt.items(), key=lambda kv: -kv[1]) if x.startswith('C:')})
