"""Shared fixtures. `-m "not gpu"` runs the oracle/golden/host-logic/ABI tests on CPU;
`-m gpu` runs the parity tests proper, all of them through the C-ABI of libspmv_amd.so."""
import importlib.util
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_binding():
    spec = importlib.util.spec_from_file_location("spmv_amd_binding", os.path.join(ROOT, "cuda-spmv-benchmark_amd", "binding.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def B():
    """The ctypes binding with the library built (hipcc cross-compiles without a GPU)."""
    mod = load_binding()
    if not os.path.exists(mod.LIB_PATH):
        mod.build()
    mod.lib()
    return mod


@pytest.fixture(scope="session")
def Blab(B):
    """The binding on the LAB build of the library (lib/libspmv_amd_lab.so: the product's sources compiled with -DSPMV_AMD_LAB,
    which adds the test and measurement hooks the product library does not have: stand-in slabs, self-neighbour communicators,
    loop options, fault injection). Everything else in the suite runs on the product library (`B`)."""
    mod = B.use_lab()
    mod.lib()
    assert mod.is_lab() and not B.is_lab()
    return mod


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle

    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def golden():
    return json.load(open(os.path.join(GOLDEN, "known_answers.json")))


@pytest.fixture()
def fresh_host_matrices(B):
    """build_csr_struct reuses csr_mat when (rows, nnz) match: start each test from a clean slate."""
    B.lib().spmv_amd_reset_host_matrices()
    yield
    B.lib().spmv_amd_reset_host_matrices()


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    den = np.maximum(np.abs(b), 1e-300)
    return float(np.max(np.abs(a - b) / den)) if len(a) else 0.0


def hist_err(hist, ref):
    """Relative error of a residual history against the oracle's. Entries that have fallen to
    rounding noise (below 1e-13 * ||r0||: exact convergence on tiny grids) cannot be compared
    relatively (noise vs noise); there both sides just have to be below 1e-12 * ||r0||."""
    hist, ref = np.asarray(hist, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    live = np.abs(ref) > 1e-13 * abs(ref[0])
    err = float(np.max(np.abs(hist[live] - ref[live]) / np.abs(ref[live])))
    if (~live).any() and float(np.max(np.abs(hist[~live]))) > 1e-12 * abs(ref[0]):
        return float("inf")
    return err
