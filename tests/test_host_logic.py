"""Host-side logic of libspmv_amd.so on CPU: Matrix Market I/O (against the reference's own
io.cu compiled into oracle/_ref where that exists), COO->CSR, ELLPACK build, partition and
offset arithmetic, harness statistics/metrics, and that the library loads and exports every
symbol include/spmv_amd/api.h declares. No GPU compute is called here."""
import ctypes as C
import filecmp
import json
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
import matrices as M


def test_library_loads_and_exports_every_declared_symbol(B):
    L = B.lib()
    assert b"gfx950" in L.spmv_amd_version()
    missing = [s for s in B.DECLARED_SYMBOLS + B.DECLARED_CXX_SYMBOLS if not hasattr(L, s)]
    assert not missing, missing


def test_header_and_binding_symbol_lists_agree(B):
    """Every function declared in the extern "C" parts of api.h is in the binding's list."""
    import re
    text = open(os.path.join(ROOT, "include", "spmv_amd", "api.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    cxx = text[text.index("/* C++ linkage") if "/* C++ linkage" in text else 0:]
    names = set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", text))
    names = {n for n in names if n.startswith("spmv_amd_") or n in B.DECLARED_SYMBOLS}
    names -= {"spmv_amd_comm_unique_id".upper()}
    typedef_like = {n for n in names if n.endswith("Fn")}
    undeclared = sorted(n for n in names - typedef_like if n not in B.DECLARED_SYMBOLS)
    assert not undeclared, undeclared


def test_operator_names_and_aliases(B):
    L = B.lib()
    for asked, canonical in (("stencil5-csr", "stencil5-csr"), ("stencil5", "stencil5-csr"), ("cusparse-csr", "cusparse-csr"),
                             ("csr", "cusparse-csr"), ("ellpack", "ellpack"), ("stencil5-ellpack", "stencil5-ellpack")):
        op = L.get_operator(asked.encode())
        assert op and op.contents.name.decode() == canonical
        assert op.contents.run_device  # every operator has the device-native entry point
    assert not L.get_operator(b"no-such-mode")


def test_compute_without_gpu_fails_loudly(B):
    """No CPU fallback: on a box without a GPU the binding refuses to compute."""
    if B.lib().spmv_amd_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(RuntimeError):
        B.Operator("stencil5-csr").init_synthetic(128)


def test_writer_matches_reference_writer_and_golden(B, O, tmp_path):
    mine = tmp_path / "mine.mtx"
    B.lib().write_matrix_market_stencil5(7, str(mine).encode())
    text = mine.read_text().split("\n")
    assert text[0] == "%%MatrixMarket matrix coordinate real general" and text[1] == "% STENCIL_GRID_SIZE 7"
    assert text[2] == "49 49 217" and text[3] == "1 1 5.0" and text[4] == "1 2 -1.0" and text[5] == "1 8 -1.0"
    if O.ref_io_available():  # byte-for-byte against the reference's own write_matrix_market_stencil5
        for n in (1, 2, 3, 7, 40):
            ref = tmp_path / f"ref{n}.mtx"
            got = tmp_path / f"got{n}.mtx"
            O.ref_write_stencil5(n, str(ref))
            B.lib().write_matrix_market_stencil5(n, str(got).encode())
            assert filecmp.cmp(ref, got, shallow=False), n
    old = tmp_path / "old81.mtx"
    B.lib().spmv_amd_write_stencil5_values(81, str(old).encode(), b"-4.0", b"-1.0")
    assert filecmp.cmp(old, os.path.join(GOLDEN, "example81x81.mtx"), shallow=False)


def test_reader_matches_reference_reader(B, O, tmp_path):
    files = [os.path.join(GOLDEN, "example81x81.mtx")]
    p = tmp_path / "s9.mtx"
    B.lib().write_matrix_market_stencil5(9, str(p).encode())
    files.append(str(p))
    generic = tmp_path / "generic.mtx"  # no stencil comment, exponent notation, blank-separated
    generic.write_text("%%MatrixMarket matrix coordinate real general\n% a comment\n3 4 4\n1 1 1.5e0\n3 4 -2\n2 2 1e-3\n1 4 7\n")
    files.append(str(generic))
    for f in files:
        m = B.load_matrix_market(f)
        if O.ref_io_available():
            rows, cols, nnz, grid, ent = O.ref_load_matrix_market(f)
            assert (m.c.rows, m.c.cols, m.c.nnz, m.c.grid_size) == (rows, cols, nnz, grid)
            assert np.array_equal(m.entries, ent)
    g = B.load_matrix_market(str(generic))
    assert g.c.grid_size == -1 and list(g.entries["row"]) == [0, 2, 1, 0] and list(g.entries["value"]) == [1.5, -2.0, 1e-3, 7.0]
    assert B.lib().read_matrix_type(str(generic).encode()) == 1
    with pytest.raises(IOError):
        B.load_matrix_market(str(tmp_path / "missing.mtx"))


def test_symmetric_reader_expands(B, tmp_path):
    sym = tmp_path / "sym.mtx"
    sym.write_text("%%MatrixMarket matrix coordinate real symmetric\n3 3 4\n1 1 2.0\n2 1 -1.0\n2 2 2.0\n3 2 -1.0\n")
    assert B.lib().read_matrix_type(str(sym).encode()) == 2
    m = B.load_matrix_market(str(sym))
    assert m.c.nnz == 6
    got = sorted((int(e["row"]), int(e["col"]), float(e["value"])) for e in m.entries)
    assert got == [(0, 0, 2.0), (0, 1, -1.0), (1, 0, -1.0), (1, 1, 2.0), (1, 2, -1.0), (2, 1, -1.0)]


@pytest.mark.parametrize("body,what", [("3 3 2\n1 1 2.0\n4 1 -1.0\n", "row beyond the declared size"),
                                       ("3 3 2\n1 1 2.0\n2 0 -1.0\n", "column index 0 in a 1-based file"),
                                       ("3 4 2\n1 1 2.0\n2 4 -1.0\n", "a 'symmetric' file that is not square: the mirror of (2,4) is row 4 of 3")])
def test_symmetric_reader_rejects_indices_outside_the_matrix(B, tmp_path, body, what):
    """The mirrored entry uses the column as a row and the CSR arrays are indexed by row: a malformed file must end in
    the reader's error (entries NULL, load_matrix_market -> 1), not in a write outside the arrays."""
    bad = tmp_path / "bad.mtx"
    bad.write_text("%%MatrixMarket matrix coordinate real symmetric\n" + body)
    with pytest.raises(IOError):
        B.load_matrix_market(str(bad))
    with pytest.raises(IOError):
        B.read_matrix_symtogen(str(bad))


@pytest.mark.parametrize("name", ["sym_hand3.mtx", "sym_spd40.mtx", "sym_stencil8.mtx"])
def test_symmetric_reader_matches_reference_symtogen(B, O, name):
    """read_matrix_symtogen against the reference's own reader (src/io/io.cu:189-310): against its committed output
    (tests/golden/symmetric_reader.json, made by make_symmetric_golden.py through oracle/_ref/libref_io.so) and, where
    that library exists, against a live call. Dimensions, both counts and the CSR arrays must be identical,
    including the reader's unsorted column order inside a row. load_matrix_market must hand the same expanded
    matrix on as entries (the reference leaves mat->entries unset there: documented deviation)."""
    path = os.path.join(GOLDEN, name)
    want = json.load(open(os.path.join(GOLDEN, "symmetric_reader.json")))["files"][name]
    assert B.lib().read_matrix_type(path.encode()) == 2
    rows, cols, nnz, full, rp, ci, va, ent = B.read_matrix_symtogen(path)
    assert (rows, cols, nnz, full) == (want["rows"], want["cols"], want["nnz_stored"], want["nnz_general"])
    assert rp.tolist() == want["row_ptr"] and ci.tolist() == want["col_idx"] and va.tolist() == want["values"]
    if O.ref_io_available():
        live = O.ref_read_matrix_symtogen(path)
        assert live[:4] == (rows, cols, nnz, full)
        assert np.array_equal(live[4], rp) and np.array_equal(live[5], ci) and np.array_equal(live[6], va)
    # the expanded entries, bucketed by row in list order, ARE that CSR
    m = B.load_matrix_market(path)
    assert (m.c.rows, m.c.cols, m.c.nnz) == (rows, cols, full) and np.array_equal(m.entries, ent)
    order = np.argsort(ent["row"], kind="stable")
    assert np.array_equal(ent["col"][order], ci) and np.array_equal(ent["value"][order], va)
    # and it is symmetric
    dense = np.zeros((rows, cols))
    dense[ent["row"], ent["col"]] = ent["value"]
    assert np.array_equal(dense, dense.T)
    assert m.c.grid_size == (8 if name == "sym_stencil8.mtx" else -1)


def test_plain_c_program_binds_the_boundary(B, tmp_path):
    """include/spmv_amd/api.h is valid C11 and the library links from plain C: the situation of a cgo / JNI / N-API
    binding. tests/abi/c_caller.c calls the integer helpers, the operator table and the
    Matrix Market entry points under their reference names; no GPU is needed for those."""
    import subprocess
    exe = tmp_path / "c_caller"
    lib_dir = os.path.dirname(B.LIB_PATH)
    build = subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                            os.path.join(ROOT, "tests", "abi", "c_caller.c"), "-L", lib_dir, "-lspmv_amd",
                            f"-Wl,-rpath,{lib_dir}", "-o", str(exe)], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr
    run = subprocess.run([str(exe), str(tmp_path / "s3.mtx")], capture_output=True, text=True, timeout=60)
    assert run.returncode == 0 and "c caller ok" in run.stdout, (run.returncode, run.stdout, run.stderr)


def test_host_layer_under_sanitizers(tmp_path):
    """The host layer of the boundary -- Matrix Market reader / writer, build_csr_struct (insertion and stable_sort paths), the
    ELLPACK builders -- compiled with g++ -fsanitize=address,undefined (they are plain C++ and link without HIP) and driven
    through well-formed and malformed files by tests/abi/host_sanitize.cpp: no sanitizer report, every malformed file refused.
    (GPU AddressSanitizer is not available on the test pool; this is what sanitizers can cover.)"""
    import subprocess
    csrc = os.path.join(ROOT, "cuda-spmv-benchmark_amd", "csrc")
    exe = tmp_path / "host_sanitize"
    build = subprocess.run(["g++", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-g", "-O1",
                            "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "abi", "host_sanitize.cpp"),
                            os.path.join(csrc, "matrix_market.cpp"), os.path.join(csrc, "host_matrix.cpp"), "-o", str(exe)],
                           capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    scratch = tmp_path / "files"
    scratch.mkdir()
    run = subprocess.run([str(exe), str(scratch)], capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", LD_PRELOAD=""))
    assert run.returncode == 0 and "host_sanitize: ok" in run.stdout, (run.returncode, run.stdout[-2000:], run.stderr[-3000:])
    assert "ERROR: AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr, run.stderr[-3000:]


def test_only_the_declared_api_leaves_the_library(B):
    """csrc/exports.map: exactly the declared symbols are dynamic exports; no spmv_amd::launch_* or kernel stub leaks."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", B.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if line.strip()}
    declared = set(B.DECLARED_SYMBOLS + B.DECLARED_CXX_SYMBOLS)
    assert exported == declared, (sorted(exported - declared)[:10], sorted(declared - exported)[:10])
    listed = set(re.findall(r"^\s+([A-Za-z_][A-Za-z0-9_]*);", open(os.path.join(ROOT, "cuda-spmv-benchmark_amd", "csrc", "exports.map")).read(), flags=re.M))
    assert listed == declared


def test_hooks_live_in_the_lab_build_only(B):
    """The test and measurement hooks (stand-in slabs, slab options incl. stop_at; the self-neighbour / forced-collective /
    fault-injection / placement-failure environment switches) are compiled into lib/libspmv_amd_lab.so (-DSPMV_AMD_LAB) and
    NOT into the product library: the product exports none of the lab's entry points, its binary does not even contain the
    names of the lab's environment switches or options, api.h does not declare them; the lab build = the product's exports
    + include/spmv_amd/lab.h's. The reference has no such switches (include/spmv.h:46-64)."""
    import subprocess

    def exports(path):
        out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        return {line.split()[-1] for line in out.splitlines() if line.strip()}

    product, lab = exports(B.LIB_PATH), exports(B.LAB_LIB_PATH)
    assert os.path.basename(B.LIB_PATH) == "libspmv_amd.so"
    assert lab - product == set(B.LAB_ONLY_SYMBOLS) and product - lab == set()
    api = open(os.path.join(ROOT, "include", "spmv_amd", "api.h")).read()
    lab_h = open(os.path.join(ROOT, "include", "spmv_amd", "lab.h")).read()
    for name in B.LAB_ONLY_SYMBOLS:
        assert name not in api and name in lab_h
    blob, lab_blob = open(B.LIB_PATH, "rb").read(), open(B.LAB_LIB_PATH, "rb").read()
    for word in (b"SPMV_AMD_TEST_WEDGE_OVERLAPPED_EXCHANGE", b"SPMV_AMD_SELF_NEIGHBOUR", b"SPMV_AMD_FORCE_COLLECTIVES", b"SPMV_AMD_PLACEMENT_FAIL_AFTER",
                 b"stop_at", b"spmv_event_stride", b"lead_rows"):
        assert word not in blob, word
        assert word in lab_blob, word
    for word in (b"SPMV_AMD_NO_OVERLAP", b"SPMV_AMD_P_RING", b"SPMV_AMD_WATCHDOG_S"):  # the product's documented switches
        assert word in blob, word
    assert b"LAB build" in B.use_lab().lib().spmv_amd_version() and b"LAB" not in B.lib().spmv_amd_version()


@pytest.mark.parametrize("case", ["stencil81_old", "stencil40", "random", "unbalanced", "upper", "long_rows_with_duplicate_columns"])
def test_build_csr_struct_bit_exact(B, O, fresh_host_matrices, case):
    if case == "long_rows_with_duplicate_columns":
        # rows beyond the insertion limit take std::stable_sort: equal columns must stay in input order, as insertion leaves them
        rng = np.random.default_rng(5)
        lens = [300, 2, 65, 0, 1000, 64]
        e = np.zeros(sum(lens), dtype=M.ENTRY_DTYPE)
        e["row"] = np.repeat(np.arange(len(lens)), lens)
        e["col"] = rng.integers(0, 40, size=len(e))  # 40 columns: every long row repeats columns many times
        e["value"] = rng.standard_normal(len(e))
        e = e[rng.permutation(len(e))]
        m = B.HostMatrix(e, len(lens), 40)
    elif case == "stencil81_old":
        m = B.load_matrix_market(os.path.join(GOLDEN, "example81x81.mtx"))
    elif case == "stencil40":
        m = B.HostMatrix(O.stencil5_coo(40), 1600, 1600, 40)
    elif case == "random":
        e, r, c = M.random_sparse(300, 257, 9, seed=11)
        m = B.HostMatrix(e, r, c)
    elif case == "unbalanced":
        e, r, c = M.unbalanced()
        m = B.HostMatrix(e, r, c)
    else:
        e, r, c, _ = M.upper_triangular(6)
        m = B.HostMatrix(e, r, c)
    assert B.lib().spmv_amd_build_csr_struct(m.ptr) == 0
    rp, ci, va = B.host_csr_arrays()
    orp, oci, ova = O.build_csr(m.entries, m.c.rows)
    assert np.array_equal(rp, orp) and np.array_equal(ci, oci) and np.array_equal(va, ova)
    cm = B.csr_mat()
    assert (cm.nb_rows, cm.nb_cols, cm.nb_nonzeros) == (m.c.rows, m.c.cols, m.c.nnz)
    # second call with the same (rows, nnz) reuses the arrays (spmv_cusparse_csr.cu:64-69)
    addr = C.addressof(cm.row_ptr.contents)
    assert B.lib().spmv_amd_build_csr_struct(m.ptr) == 0
    assert C.addressof(B.csr_mat().row_ptr.contents) == addr


def test_sorted_stencil_rows_are_N_W_C_E_S(B, O, fresh_host_matrices):
    n = 6
    m = B.HostMatrix(O.stencil5_coo(n), n * n, n * n, n)
    B.lib().spmv_amd_build_csr_struct(m.ptr)
    rp, ci, va = B.host_csr_arrays()
    r = 2 * n + 3
    assert list(ci[rp[r]:rp[r + 1]]) == [r - n, r - 1, r, r + 1, r + n]
    assert list(va[rp[r]:rp[r + 1]]) == [-1.0, -1.0, 5.0, -1.0, -1.0]
    assert rp[r] == B.lib().spmv_amd_interior_csr_offset(r, n) == O.interior_csr_offset(r, n)


def test_ellpack_builders(B, O, fresh_host_matrices):
    e, r, c = M.unbalanced()
    m = B.HostMatrix(e, r, c)
    assert B.lib().ensure_ellpack_structure_built(m.ptr) == 0
    w, idx, val = B.host_ell_arrays()
    orp, oci, ova = O.build_csr(e, r)
    ow, oidx, oval = O.build_ell(orp, oci, ova)
    assert w == ow == 150 and np.array_equal(idx, oidx) and np.array_equal(val, oval)
    em = B.ellpack_matrix()
    assert (em.nb_rows, em.nb_cols, em.nb_nonzeros, em.grid_size) == (r, c, len(e), -1)
    # width above MAX_WIDTH (1000) is refused
    B.lib().spmv_amd_reset_host_matrices()
    wide = M.entries([(0, j, 1.0) for j in range(1001)])
    mw = B.HostMatrix(wide, 1, 1001)
    assert B.lib().ensure_ellpack_structure_built(mw.ptr) != 0


def test_integer_helpers_match_oracle(B, O):
    L = B.lib()
    for n in (3, 81, 4096, 20000):
        for row in (n + 1, 2 * n - 2, (n - 2) * n + n - 2, (n // 2) * n + n // 2):
            assert L.spmv_amd_interior_csr_offset(row, n) == O.interior_csr_offset(row, n)
    for n, world in ((6561, 3), (400000000, 8), (100000000, 4), (10, 3), (7, 1)):
        for rank in range(world):
            assert B.partition_rows(n, world, rank) == O.partition_rows(n, world, rank)


def test_benchmark_with_stats_matches_oracle(B, O):
    rng = np.random.default_rng(5)
    for times in ([3.0, 3.1, 2.9, 3.05, 2.95, 3.0, 3.02, 9.0, 2.98, 3.01], list(rng.uniform(1, 2, 10)), [1.0, 1.0, 1.0], [5.0, 1.0, 3.0, 2.0]):
        it = iter(times)

        def run(x, y, ms):
            ms[0] = next(it)
            return 0

        st = B.BenchmarkStats()
        rc = B.lib().benchmark_with_stats(B.RUN_TIMED_FN(run), None, None, len(times), C.byref(st))
        orc, ost = O.bench_stats(times)
        assert rc == orc == 0
        for f in ("median_ms", "mean_ms", "std_dev_ms", "min_ms", "max_ms", "valid_runs", "outliers_removed"):
            assert getattr(st, f) == getattr(ost, f), f
    st = B.BenchmarkStats()
    it = iter([1.0, 2.0])
    assert B.lib().benchmark_with_stats(B.RUN_TIMED_FN(lambda x, y, ms: (ms.__setitem__(0, next(it)), 0)[1]), None, None, 2, C.byref(st)) == -1


def test_spmv_metrics_formula(B, O, fresh_host_matrices):
    class Metrics(C.Structure):
        _fields_ = [("execution_time_ms", C.c_double), ("gflops", C.c_double), ("bandwidth_gb_s", C.c_double),
                    ("matrix_rows", C.c_int), ("matrix_cols", C.c_int), ("matrix_nnz", C.c_int), ("grid_size", C.c_int),
                    ("sparsity_ratio", C.c_double), ("operator_name", C.c_char_p), ("sum_y", C.c_double), ("norm2_y", C.c_double),
                    ("gpu_info", C.c_byte * 512)]
    n = 50
    m = B.HostMatrix(O.stencil5_coo(n), n * n, n * n, n)
    B.lib().spmv_amd_build_csr_struct(m.ptr)
    out = Metrics()
    B.lib().calculate_spmv_metrics(C.c_double(0.125), m.ptr, b"stencil5-csr", C.byref(out))
    g, bw = O.spmv_metrics(0.125, n * n, n * n, m.c.nnz)
    assert out.gflops == g and out.bandwidth_gb_s == bw and out.matrix_nnz == m.c.nnz and out.grid_size == n
    # 20k figures quoted in BASELINE.md: 31,999,040,004 bytes
    nnz = 5 * 20000 * 20000 - 4 * 20000
    g, bw = O.spmv_metrics(1000.0, 400000000, 400000000, nnz)
    assert round(bw * 1e9) == 31999040004
