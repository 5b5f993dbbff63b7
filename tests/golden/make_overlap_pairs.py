#!/usr/bin/env python3
"""Generates tests/golden/overlap_pairs.json (run in the BUILD container only: it reads the reference tree).

The reference's own unit tests of `Platform::overlaps` (/root/reference/src/platform.rs:151-232) hold 40 placement
pairs with the expected answer: `platform_overlap_yes` (2 `test_case`s + an 8- and a 12-element `test_matrix` = 22
pairs) and `platform_overlap_no` (a 2x2, a 1x5 and a 1x9 `test_matrix` = 18 pairs).  This script extracts those
vectors - the `platform!(WxH @ x, y)` arguments of the attributes, nothing else - as data:

    {"a": [w, h, x, y], "b": [w, h, x, y], "overlap": true|false}

They pin the overlap geometry of the validators (product and oracle) and, through the CNF, the two overlap clause
families of the encoder (src/encoder.rs:559-596) for the default platform set.

    python3 tests/golden/make_overlap_pairs.py
"""
import json
import os
import re

SRC = "/root/reference/src/platform.rs"
PLAT = re.compile(r"platform!\(\s*(\d+)x(\d+)\s*@\s*(\d+)\s*,\s*(\d+)\s*\)")


def plats(text):
    return [[int(g) for g in m.groups()] for m in PLAT.finditer(text)]


def attributes(src, fn_name):
    """The #[test_case(..)] / #[test_matrix(..)] attributes stacked on `fn fn_name`, in source order."""
    end = src.index("fn " + fn_name)
    # attributes of this function start after the previous function body (or the module header)
    start = max(src.rfind("}\n\n", 0, end), src.rfind("// TODO", 0, end))
    block = src[start:end]
    out = []
    for m in re.finditer(r"#\[(test_case|test_matrix)\(", block):
        depth, i = 1, m.end()
        while depth:
            depth += {"(": 1, ")": -1}.get(block[i], 0)
            i += 1
        out.append((m.group(1), block[m.end():i - 1]))
    return out


def pairs_of(kind, body):
    if kind == "test_case":
        a, b = plats(body)
        return [(a, b)]
    # test_matrix([..], [..]) = cartesian product of the two bracketed lists
    depth, lists, cur = 0, [], None
    for i, ch in enumerate(body):
        if ch == "[":
            depth += 1
            if depth == 1:
                cur = i
        elif ch == "]":
            depth -= 1
            if depth == 0:
                lists.append(plats(body[cur:i]))
    assert len(lists) == 2, body
    return [(a, b) for a in lists[0] for b in lists[1]]


def main():
    src = open(SRC).read()
    out = []
    for fn, want in (("platform_overlap_yes", True), ("platform_overlap_no", False)):
        for kind, body in attributes(src, fn):
            for a, b in pairs_of(kind, body):
                out.append({"a": a, "b": b, "overlap": want})
    n_yes = sum(1 for p in out if p["overlap"])
    assert (n_yes, len(out) - n_yes) == (22, 18), (n_yes, len(out))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "overlap_pairs.json")
    with open(path, "w") as f:
        json.dump({"generator": "tests/golden/make_overlap_pairs.py",
                   "source": "src/platform.rs:151-232 (test_case / test_matrix vectors of platform_overlap_yes / _no)",
                   "format": "a, b = [width, height, x, y] (anchor = top-left tile); overlap = the reference's expected answer",
                   "pairs": out}, f, indent=1)
    print(len(out), "pairs,", n_yes, "overlapping")


if __name__ == "__main__":
    main()
