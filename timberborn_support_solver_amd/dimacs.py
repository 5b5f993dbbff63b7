"""DIMACS CNF in/out and DRUP proof reading — the wire format the reference's (commented-out)
debug dump used (crates/repl/src/main.rs:374-386); lets any external solver / checker arbitrate."""
import numpy as np


def write_dimacs(path, lits, offsets, n_vars):
    with open(path, "w") as f:
        f.write(f"p cnf {int(n_vars)} {len(offsets) - 1}\n")
        for i in range(len(offsets) - 1):
            f.write(" ".join(str(int(l)) for l in lits[offsets[i]:offsets[i + 1]]) + " 0\n")


def read_dimacs(path):
    """Returns (lits int32, offsets uint64, n_vars)."""
    lits, offsets, n_vars, cur = [], [0], 0, []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line or line[0] in "c%":
                continue
            if line[0] == "p":
                n_vars = int(line.split()[2])
                continue
            for tok in line.split():
                v = int(tok)
                if v == 0:
                    lits += cur
                    offsets.append(len(lits))
                    cur = []
                else:
                    cur.append(v)
                    n_vars = max(n_vars, abs(v))
    return np.asarray(lits, dtype=np.int32), np.asarray(offsets, dtype=np.uint64), n_vars


def read_drup(path, deletions=True):
    """DRUP text -> flat int32 array, clauses 0-terminated (the form a forward RUP checker takes); a deletion line
    ("d ...") becomes INT32_MIN followed by the clause, or is skipped with deletions=False."""
    out = []
    with open(path) as f:
        for line in f:
            toks = line.split()
            if toks and toks[0] == "d":
                if deletions:
                    out += [-2 ** 31] + [int(t) for t in toks[1:]]
                continue
            out += [int(t) for t in toks]
    return np.asarray(out, dtype=np.int64).astype(np.int32)
