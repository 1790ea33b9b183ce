// cg_solver -- counterpart of reference src/main/cg_solver.cu: b = 1, x0 = 0, 3 warm-up solves,
// then 10 timed solves from x0 = 0 each.
//   cg_solver <matrix.mtx | --stencil=N> [--mode=<m1,..>] [--host|--device] [--tol=1e-6]
//             [--maxiter=1000] [--timers] [--json=<file>] [--csv=<file>]
// (The reference's own main, unmodified, also builds against this library: INTEGRATION.md. Unlike
// it, this binary restarts every timed solve from x0 = 0 instead of from the warm-up's solution.)
#include "app_common.hpp"

int main(int argc, char** argv) {
    const char *matrix = nullptr, *modes_text = "stencil5-csr", *json = nullptr, *csv = nullptr;
    int stencil = 0, maxiter = 1000, timers = 0;
    bool device = true;
    double tol = 1e-6;
    for (int i = 1; i < argc; ++i) {
        if (const char* v = app::value_of(argv[i], "--mode=")) modes_text = v;
        else if (const char* v2 = app::value_of(argv[i], "--json=")) json = v2;
        else if (const char* v3 = app::value_of(argv[i], "--csv=")) csv = v3;
        else if (const char* v4 = app::value_of(argv[i], "--stencil=")) stencil = atoi(v4);
        else if (const char* v5 = app::value_of(argv[i], "--tol=")) tol = atof(v5);
        else if (const char* v6 = app::value_of(argv[i], "--maxiter=")) maxiter = atoi(v6);
        else if (!strcmp(argv[i], "--host")) device = false;
        else if (!strcmp(argv[i], "--device")) device = true;
        else if (!strcmp(argv[i], "--timers")) timers = 1;
        else if (argv[i][0] != '-') matrix = argv[i];
    }
    if (!matrix && stencil <= 0) {
        fprintf(stderr, "Usage: %s <matrix.mtx | --stencil=N> [--mode=<modes>] [--host|--device] [--tol=] [--maxiter=] [--timers] [--json=] [--csv=]\n", argv[0]);
        return 1;
    }
    const std::vector<std::string> modes = app::split_modes(modes_text);
    for (const std::string& m : modes) {
        SpmvOperator* op = get_operator(m.c_str());
        if (!op) {
            fprintf(stderr, "Error: Unknown mode '%s'\n", m.c_str());
            return 1;
        }
        if (device && !op->run_device) {
            fprintf(stderr, "Error: mode '%s' has no device-native interface (use --host)\n", m.c_str());
            return 1;
        }
    }
    MatrixData mat;
    if (stencil > 0 ? !app::make_stencil(stencil, &mat) : load_matrix_market(matrix, &mat) != 0) {
        fprintf(stderr, "Error loading matrix\n");
        return 1;
    }
    printf("Matrix loaded: %d x %d, %d nonzeros\n", mat.rows, mat.cols, mat.nnz);
    std::vector<double> b((size_t)mat.rows, 1.0), x((size_t)mat.rows, 0.0);
    bool first_csv = true;
    for (const std::string& m : modes) {
        SpmvOperator* op = get_operator(m.c_str());
        printf("\n========================================\nCG Solver - Mode: %s\n========================================\n", m.c_str());
        printf("Interface: %s\nTolerance: %.1e, Max iterations: %d\n", device ? "Device-native (GPU)" : "Host", tol, maxiter);
        if (op->init(&mat) != 0) {
            fprintf(stderr, "Failed to initialize operator\n");
            continue;
        }
        CGConfig quiet = {maxiter, tol, 0, 0}, cfg = {maxiter, tol, 0, timers};
        CGStats st;
        printf("Warmup (3 runs)...\n");
        for (int w = 0; w < 3; ++w) {
            std::fill(x.begin(), x.end(), 0.0);
            if (device) cg_solve_device(op, &mat, b.data(), x.data(), quiet, &st);
            else cg_solve(op, &mat, b.data(), x.data(), quiet, &st);
        }
        std::fill(x.begin(), x.end(), 0.0);
        BenchmarkStats bs;
        memset(&bs, 0, sizeof bs);
        if (device) {
            printf("Running benchmark (10 runs)...\n");
            if (cg_benchmark_with_stats_device(op, &mat, b.data(), x.data(), cfg, 10, &bs, &st) != 0) {
                fprintf(stderr, "benchmark failed\n");
                op->free();
                continue;
            }
        } else {
            cg_solve(op, &mat, b.data(), x.data(), cfg, &st);
            bs.median_ms = st.time_total_ms;
            bs.valid_runs = 1;
        }
        printf("Completed: %d valid runs, %d outliers removed\n", bs.valid_runs, bs.outliers_removed);
        printf("\n--- Results for %s ---\nConverged: %s in %d iterations\nTime (median): %.3f ms\n", m.c_str(),
               st.converged ? "YES" : "NO", st.iterations, bs.median_ms);
        printf("Stats: min=%.3f ms, max=%.3f ms, std=%.3f ms\n", bs.min_ms, bs.max_ms, bs.std_dev_ms);
        printf("\n=== Output Checksum ===\nSum(x):    %.16e\nNorm2(x):  %.16e\n", st.solution_sum, st.solution_norm);
        if (json) export_cg_json(app::per_mode_path(json, m.c_str(), ".json").c_str(), m.c_str(), &mat, &bs, &st);
        if (csv) {
            export_cg_csv(csv, m.c_str(), &mat, &bs, &st, first_csv);
            first_csv = false;
        }
        op->free();
    }
    free(mat.entries);
    return 0;
}
