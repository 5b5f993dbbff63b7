#!/usr/bin/env python3
"""Where are a kernel's spills?  Reads the ISA of one function (hipcc -S --cuda-device-only), finds its loops
(backward branches) and reports, per loop nest, the scratch / writelane instructions inside it - a spill outside the
BCP loop costs nothing, one inside it costs a scratch round trip per step.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S -o k.s mi355sat.hip
    python3 scripts/isa_spill_map.py k.s _Z16ms_search_kernelILb0ELi4EEv8MsShared8MsLayoutPc8MsParams
"""
import re
import sys


def main():
    path, fn = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(fn + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    label_at = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            label_at[m.group(1)] = i
    loops = []   # (head, tail)
    for i, l in enumerate(body):
        m = re.search(r"\bs_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in label_at and label_at[m.group(1)] <= i:
            loops.append((label_at[m.group(1)], i))
    # merge loops with the same head (keep the widest)
    by_head = {}
    for h, t in loops:
        by_head[h] = max(by_head.get(h, t), t)
    loops = sorted(by_head.items())
    def depth(i):
        return sum(1 for h, t in loops if h <= i <= t)
    kinds = {"scratch_store": 0, "scratch_load": 0, "v_writelane": 0, "v_readlane": 0}
    per_depth = {}
    for i, l in enumerate(body):
        for k in kinds:
            if k in l:
                d = depth(i)
                per_depth.setdefault(d, dict.fromkeys(kinds, 0))[k] += 1
    print("function", fn, "lines", len(body), "loops", len(loops))
    for d in sorted(per_depth):
        print("  loop depth", d, per_depth[d])
    # the loops themselves: size, instruction mix
    for h, t in loops:
        seg = body[h:t + 1]
        n_ins = sum(1 for l in seg if l.startswith("\t") and not l.startswith("\t."))
        sc = sum(1 for l in seg if "scratch_" in l)
        wl = sum(1 for l in seg if "v_writelane" in l or "v_readlane" in l)
        vm = sum(1 for l in seg if re.search(r"\b(global|flat|buffer)_(load|store|atomic)", l))
        call = sum(1 for l in seg if "s_swappc" in l)
        print(f"  loop [{h},{t}] depth {depth(h)} instr {n_ins} scratch {sc} lane-spill {wl} vmem {vm} calls {call}")


if __name__ == "__main__":
    main()
