#!/usr/bin/env python3
"""Print per-kernel register / LDS / scratch usage from `make asm` remarks."""
import re, subprocess, sys
cur = None; d = {}
for l in sys.stdin:
    m = re.search(r'remark:\s+(Function Name|TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)', l)
    if not m: continue
    k, v = m.groups()
    if k == 'Function Name': cur = v; d[cur] = {}
    else: d[cur][k.split()[0] + ('Spill' if 'Spill' in k else '')] = v
for f, v in d.items():
    name = subprocess.run(['c++filt', f], capture_output=True, text=True).stdout.strip()
    print(f'{name[:58]:58s}', ' '.join(f'{k}={x}' for k, x in v.items()))
