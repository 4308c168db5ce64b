#!/usr/bin/env python3
"""Where do the gathers and their waits sit in a kernel's ISA?  usage: isa_waits.py file.s mangled-kernel-name-substring
Prints (line, multiply-adds so far, instruction) for every global load, vmcnt wait, scratch access and barrier."""
import re, sys
s = open(sys.argv[1]).read()
m = re.search(r'^(_Z\w*' + re.escape(sys.argv[2]) + r'\w*):', s, re.M)
a = m.end()
b = s.index('.Lfunc_end', a)
n_mad = 0
for i, l in enumerate(s[a:b].split('\n')):
    t = l.strip()
    if t.startswith('v_mad_u64_u32'):
        n_mad += 1
    if t.startswith('global_load') or ('s_waitcnt' in t and 'vmcnt' in t) or t.startswith('scratch_') or t.startswith('s_barrier'):
        print(i, n_mad, t[:80])
tail = s[b:b + 3000]
for k in ('NumVgprs', 'ScratchSize', 'Occupancy'):
    for m in re.finditer(r'.*' + k + '.*', tail):
        print(m.group(0))
