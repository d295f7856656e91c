#!/usr/bin/env python3
"""Per-function register / scratch summary from `hipcc -Rpass-analysis=kernel-resource-usage` output (stdin or file)."""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
blocks = re.split(r'Function Name: ', txt)
for b in blocks[1:]:
    name = b.split(' [')[0].split('\n')[0].strip()

    def g(k):
        m = re.search(k + r': (\d+)', b)
        return int(m.group(1)) if m else -1
    n = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    print("V%4d S%4d scratch%6d spillV%4d spillS%4d occ%2d  %s" % (g('VGPRs'), g('TotalSGPRs'), g(r'ScratchSize \[bytes/lane\]'), g('VGPRs Spill'),
                                                               g('SGPRs Spill'), g(r'Occupancy \[waves/SIMD\]'), n[:110]))
