"""Regenerates the committed fixtures under tests/golden/.

  example81x81.mtx    -- the 81x81 5-point matrix in the reference's OLDER value convention
                         (centre -4.0, neighbours -1.0). Written by this repo's own writer
                         (spmv_amd_write_stencil5_values); when /root/reference is present the
                         script checks that the bytes equal the reference's shipped data file
                         matrix/example81x81.mtx, which is what the reference's CI runs on.
  known_answers.json  -- outputs of the CPU oracle (oracle/spmv_oracle.c) on the reference's
                         deterministic benchmark inputs (b = 1, x0 = 0, x = 1): SpMV checksums,
                         CG iteration counts, residual histories and solution checksums, next to
                         the values recorded in SURVEY.md section 8c (computed there by an
                         independent numpy/scipy restatement), which the oracle must reproduce.

Run from the repo root:  python tests/golden/make_golden.py [--with-10k] [--with-15k] [--with-20k] [--out=<json>]
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))

from oracle import oracle as O  # noqa: E402
import importlib.util  # noqa: E402

spec = importlib.util.spec_from_file_location("spmv_amd_binding", os.path.join(ROOT, "cuda-spmv-benchmark_amd", "binding.py"))
B = importlib.util.module_from_spec(spec)
spec.loader.exec_module(B)


def sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def case(n, center, off, with_cg=True, max_iters=1000):
    rp, ci, va = O.stencil5_csr(n, center, off)
    rows = n * n
    y = O.spmv_stencil5(rp, ci, va, np.ones(rows), n)
    out = {
        "n": n, "center": center, "off": off, "rows": rows, "nnz": int(rp[-1]),
        "sum_y": float(y.sum()), "norm2_y": float(np.sqrt((y * y).sum())),
        "row_ptr_head": [int(v) for v in rp[:6]],
    }
    if n >= 3:
        r = n + 1
        out["interior_row"] = r
        out["interior_offset"] = int(rp[r])
        out["interior_cols"] = [int(v) for v in ci[rp[r]:rp[r + 1]]]
    if with_cg:
        x, hist, res = O.cg(rp, ci, va, n, np.ones(rows), np.zeros(rows), max_iters=max_iters, tol=1e-6, device_form=True)
        out["cg"] = {
            "iterations": res.iterations, "converged": res.converged, "history": [float(v) for v in hist],
            "solution_sum": res.solution_sum, "solution_norm": res.solution_norm,
        }
    return out


def main():
    B.build()
    O.build()
    mtx = os.path.join(HERE, "example81x81.mtx")
    B.lib().spmv_amd_write_stencil5_values(81, mtx.encode(), b"-4.0", b"-1.0")
    ref = "/root/reference/matrix/example81x81.mtx"
    info = {"example81x81_sha256": sha(mtx)}
    if os.path.exists(ref):
        assert sha(ref) == info["example81x81_sha256"], "writer output differs from the reference's shipped file"
        info["matches_reference_file"] = True
    path = os.path.join(HERE, "known_answers.json")
    old = json.load(open(path)) if os.path.exists(path) else {}
    for a in sys.argv:
        if a.startswith("--out="):
            path = a[len("--out="):]
    cases = dict(old.get("cases", {}))
    for n, c in ((3, -4.0), (3, 5.0), (81, -4.0), (81, 5.0), (512, 5.0), (2000, 5.0)):
        cases[f"{n}:{c}"] = case(n, c, -1.0)
    if "--with-10k" in sys.argv:
        cases["10000:5.0"] = case(10000, 5.0, -1.0)
    if "--with-15k" in sys.argv:  # BASELINE config 5's grid; the reference publishes CG there too (docs/PROBLEM_SIZE_SCALING_RESULTS.md:31-38); ~25 GB of host memory
        cases["15000:5.0"] = case(15000, 5.0, -1.0)
    if "--with-20k" in sys.argv:  # needs ~45 GB of host memory and about a minute: run on the GPU box's host
        cases["20000:5.0"] = case(20000, 5.0, -1.0)
    info["cases"] = cases
    # values quoted in SURVEY.md section 8c from an independent restatement
    info["survey_8c"] = {
        "3:-4.0": {"sum_y": -60.0, "nnz": 33},
        "81:-4.0": {"sum_y": -52164.0, "norm2_y": 6.4424529489938845e+02, "cg_iterations": 40,
                    "solution_sum": -8.2608388842537738e+02, "solution_norm": 1.0206197049889337e+01},
        "81:5.0": {"sum_y": 6885.0, "norm2_y": 8.6838931361457924e+01, "cg_iterations": 18},
        "512:5.0": {"cg_iterations": 17},
        "2000:5.0": {"cg_iterations": 16},
        "10000:5.0": {"cg_iterations": 14, "solution_sum": 9.9975281007346511e+07, "solution_norm": 9.9978695581374704e+03,
                      "history": [1.0000000000e+04, 1.9990003898e+02, 6.6651104855e+01, 2.5023672721e+01, 9.6763393914e+00,
                                  4.4055499300e+00, 2.5479592812e+00, 1.0703501600e+00, 5.9744656333e-01, 2.7133517317e-01,
                                  1.4134168210e-01, 6.8795411595e-02, 3.4119920478e-02, 1.7334672540e-02, 8.3561478941e-03]},
    }
    json.dump(info, open(path, "w"), indent=1)
    print("wrote", path, "and", mtx)


if __name__ == "__main__":
    main()
