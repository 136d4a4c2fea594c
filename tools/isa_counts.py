#!/usr/bin/env python3
"""Instruction-type counts per kernel from hipcc's assembly: tools/isa_counts.py file.s [name-substring]"""
import re
import sys
from collections import Counter
lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2] if len(sys.argv) > 2 else "kernel"
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
for k, (i, name) in enumerate(starts):
    if pat not in name:
        continue
    end = starts[k + 1][0] if k + 1 < len(starts) else len(lines)
    body = lines[i:end]
    c = Counter()
    for l in body:
        m = re.match(r"^\s+(global_load_\w+|global_store_\w+|ds_read\w*|ds_write\w*|v_mfma\w+|s_waitcnt|scratch_\w+|v_accvgpr_\w+|s_barrier|buffer_\w+|s_load_\w+|v_cvt_pk_bf16_f32|v_readlane\w*|v_writelane\w*)", l)
        if m:
            c[m.group(1)] += 1
    print(name)
    for kk, v in sorted(c.items()):
        print(f"    {kk:28s} {v}")
