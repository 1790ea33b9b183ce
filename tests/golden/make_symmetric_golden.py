"""Regenerates the symmetric Matrix Market fixtures and what the REFERENCE's own reader makes of them.

  sym_hand3.mtx, sym_spd40.mtx, sym_stencil8.mtx -- small `coordinate real symmetric` files (lower triangle +
      diagonal): a hand case, a random diagonally dominant (SPD) matrix, the 8 x 8 5-point stencil.
  symmetric_reader.json -- for each file the output of the reference's read_matrix_symtogen
      (/root/reference/src/io/io.cu:189-310), run here through oracle/_ref/libref_io.so, which oracle/Makefile
      compiles from that source where it lies: rows, cols, stored nnz, expanded nnz and the CSR arrays
      (columns inside a row in the reader's own, unsorted order).

The tests compare this repo's read_matrix_symtogen / load_matrix_market with the JSON everywhere, and with the
live libref_io.so where it exists. Run from the repo root:  python tests/golden/make_symmetric_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))

from oracle import oracle as O  # noqa: E402


def write(path, n, triples, comment=None):
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real symmetric\n")
        if comment:
            f.write(comment + "\n")
        f.write(f"{n} {n} {len(triples)}\n")
        for r, c, v in triples:
            f.write(f"{r + 1} {c + 1} {v!r}\n")


def main():
    O.build()
    assert O.ref_io_available(), "oracle/_ref/libref_io.so is needed (make -C oracle ref, with /root/reference present)"
    files = {}
    write(os.path.join(HERE, "sym_hand3.mtx"), 3, [(0, 0, 2.0), (1, 0, -1.0), (1, 1, 2.0), (2, 1, -1.0)])
    files["sym_hand3.mtx"] = None
    rng = np.random.default_rng(40)
    n, trip, rowsum = 40, [], np.zeros(40)
    for r in range(n):
        for c in rng.choice(r, size=min(r, 3), replace=False) if r else []:
            v = float(np.round(rng.uniform(-1.0, 1.0), 6))
            trip.append((r, int(c), v))
            rowsum[r] += abs(v)
            rowsum[int(c)] += abs(v)
    rng.shuffle(trip)  # file order is not row order
    trip += [(r, r, float(np.round(rowsum[r] + 1.0 + rng.uniform(0, 1), 6))) for r in range(n)]
    write(os.path.join(HERE, "sym_spd40.mtx"), n, trip)
    files["sym_spd40.mtx"] = None
    g, st = 8, []
    for i in range(g):
        for j in range(g):
            row = i * g + j
            st.append((row, row, 5.0))
            if j > 0:
                st.append((row, row - 1, -1.0))
            if i > 0:
                st.append((row, row - g, -1.0))
    write(os.path.join(HERE, "sym_stencil8.mtx"), g * g, st, comment="% STENCIL_GRID_SIZE 8")
    files["sym_stencil8.mtx"] = None
    out = {"generated_by": "tests/golden/make_symmetric_golden.py via oracle/_ref/libref_io.so (reference src/io/io.cu compiled in place)", "files": {}}
    for name in files:
        rows, cols, nnz, full, rp, ci, va = O.ref_read_matrix_symtogen(os.path.join(HERE, name))
        out["files"][name] = {"rows": rows, "cols": cols, "nnz_stored": nnz, "nnz_general": full, "row_ptr": rp.tolist(),
                              "col_idx": ci.tolist(), "values": va.tolist()}
    json.dump(out, open(os.path.join(HERE, "symmetric_reader.json"), "w"), indent=1)
    print("wrote", sorted(files), "and symmetric_reader.json")


if __name__ == "__main__":
    main()
