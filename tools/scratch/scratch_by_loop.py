#!/usr/bin/env python3
"""Where the scratch (spill) instructions of a function sit: counts per enclosing loop, from hipcc -save-temps assembly.
usage: scratch_by_loop.py file.s mangled-or-demangled-substring"""
import collections
import re
import subprocess
import sys

lines = open(sys.argv[1]).read().split('\n')
want = sys.argv[2]
names = {}
for i, l in enumerate(lines):
    m = re.match(r'^(_Z\w+):', l)
    if m:
        names[i] = m.group(1)
dem = subprocess.run(['c++filt'], input='\n'.join(names.values()), capture_output=True, text=True).stdout.split('\n')
for (i, mangled), d in zip(names.items(), dem):
    if want not in d and want not in mangled:
        continue
    end = next(j for j in range(i, len(lines)) if lines[j].startswith('.Lfunc_end'))
    body = lines[i:end]
    # block -> (header, depth) from the comments the assembler printer leaves on block labels
    cur = ('-', 0)
    per = collections.Counter()
    tot = 0
    k = 0
    while k < len(body):
        l = body[k]
        if re.match(r'^(\.LBB\d+_\d+:|; %bb\.\d+:)', l):
            blk = l
            j = k + 1
            while j < len(body) and body[j].startswith(' ' * 30 + ';'):   # continuation comment lines
                blk += body[j]
                j += 1
            m = re.search(r'This (?:Inner )?Loop Header: Depth=(\d+)', blk)
            m1 = re.search(r'in Loop: Header=(BB\d+_\d+) Depth=(\d+)', blk)
            lab = re.match(r'^\.L(BB\d+_\d+):', l)
            if m:
                cur = (lab.group(1) if lab else '?', int(m.group(1)))
            elif m1:
                cur = (m1.group(1), int(m1.group(2)))
            else:
                cur = ('-', 0)
        if re.search(r'\bscratch_(load|store)', l):
            per[cur] += 1
            tot += 1
        k += 1
    print("%s: %d lines, %d scratch instructions" % (d[:100], end - i, tot))
    for (h, dep), v in sorted(per.items(), key=lambda x: (-x[0][1], -x[1])):
        print("   depth %d loop %-12s %4d" % (dep, h, v))
