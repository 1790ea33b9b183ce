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
