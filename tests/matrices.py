"""Small analytic matrices for known-answer tests, modelled on the ideas (not the code) of the
reference's tests/helpers/matrix_fixtures.cpp: identity 3x3 (sum y = 3), diagonal 5x5 (15),
tridiagonal 4x4 (2), upper-triangular 3x3 (21), plus seeded random and unbalanced-row matrices."""
import numpy as np

ENTRY_DTYPE = np.dtype([("row", np.int32), ("col", np.int32), ("value", np.float64)], align=True)


def entries(triples):
    e = np.zeros(len(triples), dtype=ENTRY_DTYPE)
    for k, (r, c, v) in enumerate(triples):
        e[k] = (r, c, v)
    return e


def identity(n=3):
    return entries([(i, i, 1.0) for i in range(n)]), n, n, float(n)


def diagonal(n=5):
    return entries([(i, i, float(i + 1)) for i in range(n)]), n, n, float(n * (n + 1) // 2)


def tridiagonal(n=4):
    t = []
    for i in range(n):
        t.append((i, i, 2.0))
        if i > 0:
            t.append((i, i - 1, -1.0))
        if i < n - 1:
            t.append((i, i + 1, -1.0))
    return entries(t), n, n, 2.0


def upper_triangular(n=3):
    t, v = [], 1.0
    for i in range(n):
        for j in range(i, n):
            t.append((i, j, v))
            v += 1.0
    return entries(t), n, n, float(sum(range(1, n * (n + 1) // 2 + 1)))


def random_sparse(rows, cols, per_row, seed=42, shuffle=True):
    """Distinct columns per row; entries in shuffled (file-like) order."""
    rng = np.random.default_rng(seed)
    t = []
    for r in range(rows):
        k = int(per_row(r, rng)) if callable(per_row) else per_row
        k = min(k, cols)
        for c in rng.choice(cols, size=k, replace=False):
            t.append((r, int(c), float(rng.uniform(-2.0, 2.0))))
    if shuffle:
        order = rng.permutation(len(t))
        t = [t[i] for i in order]
    return entries(t), rows, cols


def unbalanced(rows=200, cols=200, seed=7):
    """A few very long rows among short ones (and some empty rows)."""
    def per_row(r, rng):
        if r % 50 == 0:
            return 150
        if r % 7 == 0:
            return 0
        return int(rng.integers(1, 6))
    return random_sparse(rows, cols, per_row, seed)


def stencil_random_values(n, seed=3):
    """Complete 5-point structure in the writer's entry order (C,W,E,N,S) with random values."""
    rng = np.random.default_rng(seed)
    t = []
    for i in range(n):
        for j in range(n):
            idx = i * n + j
            t.append((idx, idx, float(rng.uniform(1.0, 9.0))))
            if j > 0:
                t.append((idx, idx - 1, float(rng.uniform(-2.0, 2.0))))
            if j < n - 1:
                t.append((idx, idx + 1, float(rng.uniform(-2.0, 2.0))))
            if i > 0:
                t.append((idx, idx - n, float(rng.uniform(-2.0, 2.0))))
            if i < n - 1:
                t.append((idx, idx + n, float(rng.uniform(-2.0, 2.0))))
    return entries(t), n * n, n * n


# ---- structured fixtures after the ideas of reference tests/helpers/matrix_fixtures.cpp:181-338 (own code, own values) ----
def from_arrays(rows, cols, vals, shuffle_seed=None):
    e = np.zeros(len(rows), dtype=ENTRY_DTYPE)
    e["row"], e["col"], e["value"] = rows, cols, vals
    if shuffle_seed is not None:
        e = e[np.random.default_rng(shuffle_seed).permutation(len(e))]
    return e


def stencil_9point(n, center=8.0, off=-1.0):
    """9-point stencil of an n x n grid (all eight neighbours). Carries grid_size = n, but is NOT the 5-point pattern:
    the stencil operators must notice and take the CSR loop. y = A * 1: sum = n^2*center + off*(number of neighbour pairs)."""
    i, j = np.divmod(np.arange(n * n), n)
    rows, cols, vals = [np.arange(n * n)], [np.arange(n * n)], [np.full(n * n, center)]
    for di in (-1, 0, 1):
        for dj in (-1, 0, 1):
            if di == 0 and dj == 0:
                continue
            ok = (i + di >= 0) & (i + di < n) & (j + dj >= 0) & (j + dj < n)
            rows.append(np.arange(n * n)[ok])
            cols.append(((i + di) * n + (j + dj))[ok])
            vals.append(np.full(int(ok.sum()), off))
    e = from_arrays(np.concatenate(rows), np.concatenate(cols), np.concatenate(vals), shuffle_seed=3)
    neighbours = len(e) - n * n
    return e, n * n, n * n, float(n * n * center + off * neighbours)


def banded(size, bandwidth, diagonal=4.0):
    """|i - j| <= bandwidth; diagonal value `diagonal`, off-diagonals -1 / (1 + |i - j|) -- strictly diagonally dominant for
    diagonal > 2 * H(bandwidth), hence SPD (symmetric). Returns entries, rows, cols, sum(A * 1)."""
    rows, cols, vals = [], [], []
    for d in range(-bandwidth, bandwidth + 1):
        r = np.arange(max(0, -d), min(size, size - d))
        rows.append(r)
        cols.append(r + d)
        vals.append(np.full(len(r), diagonal if d == 0 else -1.0 / (1 + abs(d))))
    e = from_arrays(np.concatenate(rows), np.concatenate(cols), np.concatenate(vals), shuffle_seed=4)
    return e, size, size, float(np.sum(e["value"]))


def dense_blocks(size, block, fill=0.5):
    """Dense block x block squares along the diagonal (the last one cut off at the matrix edge)."""
    rows, cols = [], []
    for b0 in range(0, size, block):
        b1 = min(b0 + block, size)
        r, c = np.meshgrid(np.arange(b0, b1), np.arange(b0, b1), indexing="ij")
        rows.append(r.ravel())
        cols.append(c.ravel())
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    e = from_arrays(rows, cols, np.full(len(rows), fill), shuffle_seed=5)
    return e, size, size, float(fill * len(rows))


def ill_conditioned(size, decades=12):
    """Tridiagonal, symmetric, diagonal falling exponentially over `decades` orders of magnitude, off-diagonals a fixed small
    fraction of the smaller neighbouring diagonal entry (so every row stays diagonally dominant)."""
    d = 10.0 ** (-decades * np.arange(size) / max(size - 1, 1))
    off = -0.25 * np.minimum(d[:-1], d[1:])
    rows = np.concatenate([np.arange(size), np.arange(size - 1), np.arange(1, size)])
    cols = np.concatenate([np.arange(size), np.arange(1, size), np.arange(size - 1)])
    vals = np.concatenate([d, off, off])
    return from_arrays(rows, cols, vals, shuffle_seed=6), size, size, float(np.sum(vals))
