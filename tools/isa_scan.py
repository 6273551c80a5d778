"""Scan device assembly (hipcc -S --cuda-device-only) for signs of loops the kernels expect to be unrolled:
dynamic register indexing (s_set_gpr_idx_on / v_movrel), scratch use; prints MFMA counts per kernel.
usage: python tools/isa_scan.py file.s [...]"""
import re, sys
for f in sys.argv[1:]:
    s = open(f).read()
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)s_endpgm', s, re.S | re.M):
        body = m.group(2)
        n = body.count('s_set_gpr_idx_on') + body.count('v_movrel')
        if n or 'v_mfma' in body or 'scratch_' in body:
            print(f"{f.split('/')[-1]:16s} {m.group(1)[-56:]:56s} gpr_idx {n:5d}  mfma {body.count('v_mfma'):5d}  scratch {len(re.findall(r'scratch_', body)):4d}  branches {len(re.findall(r's_cbranch', body)):4d}")
