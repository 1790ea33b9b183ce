"""Comparator slot: the vendor libraries on the same inputs, as tests. The reference measures itself
against cuSPARSE and AmgX (external/benchmarks/amgx/BENCHMARK_RESULTS.md:43-50); here rocSPARSE's CSR SpMV and
rocALUTION's unpreconditioned CG (tools/rocsparse_compare.hip, tools/rocalution_cg.cpp, built by `make -C tools`) are the
only arithmetic in the tree that was NOT written by this repository -- an independent cross-check of the operators and
of the CG solver, next to the oracle."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "tools", "bin")


def tool(name):
    path = os.path.join(BIN, name)
    if not os.path.exists(path):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tools")], capture_output=True, text=True)
    assert os.path.exists(path), f"{path} missing: run `make -C tools` (rocSPARSE / rocALUTION ship with ROCm)"
    return path


def test_rocsparse_csr_spmv_agrees_with_every_operator(B, O, fresh_host_matrices, tmp_path):
    """Same synthetic 2048 x 2048 stencil, same random x: rocsparse_spmv's y against stencil5-csr, cusparse-csr (this
    repository's CSR kernels) and ellpack, element for element at 1e-12 relative to max|y|."""
    n = 2048
    rows = n * n
    x = np.random.default_rng(2048).standard_normal(rows)
    xf, yf = tmp_path / "x.bin", tmp_path / "y.bin"
    x.tofile(xf)
    out = subprocess.run([tool("rocsparse_compare"), str(n), str(xf), str(yf)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "wrote y" in out.stdout, out.stdout + out.stderr
    y_vendor = np.fromfile(yf, dtype=np.float64)
    assert len(y_vendor) == rows
    scale = np.max(np.abs(y_vendor))
    dx = B.DeviceVector.from_host(x)
    for mode in ("stencil5-csr", "cusparse-csr", "ellpack"):
        op = B.Operator(mode)
        assert op.init_synthetic(n) == 0
        dy = B.DeviceVector(rows, fill=0.0)
        assert op.run_device(dx, dy) == 0
        y = dy.to_host()
        assert np.max(np.abs(y - y_vendor)) <= 1e-12 * scale, mode
        dy.free()
        op.free()
    dx.free()


def test_rocalution_cg_agrees_with_the_slab_solver(B):
    """rocALUTION's CG (no preconditioner, b = 1, x0 = 0, relative tolerance 1e-6) on the 4096 x 4096 stencil: same
    iteration count, final residual within 1e-10 relative of the slab solver's. (Not 2048: rocALUTION 4.0 / ROCm 7.2
    returns a NaN residual from its first iteration at 2048 x 2048 and 2049 x 2049 on this GPU -- while 512, 4096,
    10 000 and 20 000 are fine -- a vendor-side defect this repository does not depend on.)"""
    n = 4096
    out = subprocess.run([tool("rocalution_cg"), str(n), "2"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    m = re.search(r"RESULT iterations=(\d+) residual=([0-9.eE+-]+)", out.stdout)
    assert m, out.stdout
    vendor_iterations, vendor_residual = int(m.group(1)), float(m.group(2))
    slab = B.CgSlab.stencil5(n)
    st = slab.solve()
    assert st.converged == 1 and st.iterations == vendor_iterations
    assert abs(st.residual_norm - vendor_residual) <= 1e-10 * vendor_residual
    slab.destroy()
