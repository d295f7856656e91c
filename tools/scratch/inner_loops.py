#!/usr/bin/env python3
"""Innermost loops of a function in hipcc -save-temps assembly (found by their back edges), with instruction mix.
usage: inner_loops.py file.s <demangled substring> [min instructions] [--dump LABEL]"""
import re
import subprocess
import sys

lines = open(sys.argv[1]).read().split('\n')
want = sys.argv[2]
minins = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 30
dump = sys.argv[sys.argv.index('--dump') + 1] if '--dump' in sys.argv else None
names = {i: re.match(r'^(_Z\w+):', l).group(1) for i, l in enumerate(lines) if re.match(r'^(_Z\w+):', l)}
dem = subprocess.run(['c++filt'], input='\n'.join(names.values()), capture_output=True, text=True).stdout.split('\n')
for (i, mangled), d in zip(names.items(), dem):
    if want not in d:
        continue
    end = next(j for j in range(i, len(lines)) if lines[j].startswith('.Lfunc_end'))
    body = lines[i:end]
    labels = {re.match(r'^\.L(BB\d+_\d+):', l).group(1): k for k, l in enumerate(body) if re.match(r'^\.L(BB\d+_\d+):', l)}
    loops = []
    for k, l in enumerate(body):
        m = re.search(r'\bs_cbranch_\w+\s+\.L(BB\d+_\d+)|\bs_branch\s+\.L(BB\d+_\d+)', l)
        if m:
            t = m.group(1) or m.group(2)
            if t in labels and labels[t] < k:
                loops.append((labels[t], k, t))
    # innermost = loops that contain no other loop
    inner = [a for a in loops if not any(b is not a and b[0] >= a[0] and b[1] <= a[1] and (b[0], b[1]) != (a[0], a[1]) for b in loops)]
    print(d[:110])
    for a, b, t in sorted(set(inner)):
        ins = [l for l in body[a:b + 1] if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
        if len(ins) < minins:
            continue
        c = lambda pat: sum(1 for l in ins if re.search(pat, l))
        print("  .L%s: %d instructions, fma/mul/add_f64 %d, ds_read %d, ds_write %d, s_waitcnt %d, permlane %d, global %d, scratch %d, cndmask %d, readlane %d, dpp %d" % (
            t, len(ins), c(r'v_(fma|fmac|mul|add)_f64'), c(r'ds_read'), c(r'ds_write'), c(r's_waitcnt'), c(r'permlane'), c(r'global_'), c(r'scratch_'),
            c(r'cndmask'), c(r'readlane'), c(r'_dpp')))
        if dump == t:
            print('\n'.join(body[a:b + 1]))
