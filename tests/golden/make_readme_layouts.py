#!/usr/bin/env python3
"""Generates tests/golden/readme_layouts.json from the example transcript in the reference's
README (README.md:40-119): four printed layouts (18/17/16/15 supports of size 1x1) on a
21x16 terrain.  Cells: '▒' ceiling overhead, '█' ceiling with a 1x1 support, '░' empty.
These are the only solver outputs the reference publishes; they pin the validator
(PlatformLayout::validate with TERRAIN_SUPPORT_DISTANCE = 4).  Run in the build container only."""
import json
import os
import re

README = "/root/reference/README.md"
out = []
lines = open(README, encoding="utf-8").read().splitlines()
i = 0
while i < len(lines):
    m = re.match(r"Solution: \((\d+) marked\)", lines[i])
    if not m:
        i += 1
        continue
    marked = int(m.group(1))
    rows, supports = [], []
    i += 1
    while i < len(lines) and lines[i] and lines[i][0] in "▒█░":
        cells = lines[i].split()
        rows.append("".join("X" if c in "▒█" else " " for c in cells))
        supports += [(x, len(rows) - 1) for x, c in enumerate(cells) if c == "█"]
        i += 1
    assert len(supports) == marked
    out.append({"marked": marked, "terrain_rows": rows, "supports_xy": supports})
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "readme_layouts.json")
json.dump({"source": "reference README.md:40-119", "layouts": out}, open(path, "w"), indent=1)
print([o["marked"] for o in out], len(out[0]["terrain_rows"]), len(out[0]["terrain_rows"][0]))
