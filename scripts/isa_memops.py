#!/usr/bin/env python3
"""Histogram of memory opcodes per kernel / function of a device ISA listing (hipcc -S --cuda-device-only):
flat_* left in a kernel means a pointer lost its address space (see kernels.hip.h "address spaces")."""
import collections
import re
import sys

cur, hist = None, {}
for l in open(sys.argv[1]):
    m = re.match(r"^(_Z\w+):", l)
    if m:
        cur = m.group(1)
        hist[cur] = collections.Counter()
        continue
    if l.startswith(".Lfunc_end"):
        cur = None
    if cur:
        m = re.match(r"^\s+((?:global|flat|buffer|ds|scratch)_(?:load|store|atomic|read|write|or|and|add|cmpst|bpermute|swizzle)\w*)", l)
        if m:
            op = re.sub(r"_(b\d+|dword(x\d)?|u?byte|u?short|u32|rtn.*)$", "", m.group(1))
            hist[cur][op.split("_")[0] + "_" + op.split("_")[1]] += 1
for fn, h in hist.items():
    if len(sys.argv) > 2 and sys.argv[2] not in fn:
        continue
    print(fn[:70], dict(h.most_common()))
