// spmv_bench -- counterpart of reference src/main/main.cu: x = 1, 5 warm-ups, 10 timed runs with
// the >2 sigma / median rule, checksums, metrics, one JSON/CSV file per mode.
//   spmv_bench <matrix.mtx | --stencil=N> --mode=<m1[,m2,...]> [--json=<file>] [--csv=<file>]
#include "app_common.hpp"

int main(int argc, char** argv) {
    const char *matrix = nullptr, *modes_text = nullptr, *json = nullptr, *csv = nullptr;
    int stencil = 0;
    for (int i = 1; i < argc; ++i) {
        if (const char* v = app::value_of(argv[i], "--mode=")) modes_text = v;
        else if (const char* v2 = app::value_of(argv[i], "--json=")) json = v2;
        else if (const char* v3 = app::value_of(argv[i], "--csv=")) csv = v3;
        else if (const char* v4 = app::value_of(argv[i], "--stencil=")) stencil = atoi(v4);
        else if (argv[i][0] != '-') matrix = argv[i];
    }
    if ((!matrix && stencil <= 0) || !modes_text) {
        fprintf(stderr, "Usage: %s <matrix_file.mtx | --stencil=N> --mode=<mode1[,mode2,...]> [--json=<file>] [--csv=<file>]\n", argv[0]);
        fprintf(stderr, "Available modes: cusparse-csr, stencil5-csr, ellpack, stencil5-ellpack\n");
        return EXIT_FAILURE;
    }
    const std::vector<std::string> modes = app::split_modes(modes_text);
    printf("Validating %zu mode(s): %s\n", modes.size(), modes_text);
    for (const std::string& m : modes) {
        if (get_operator(m.c_str()) == nullptr) {
            fprintf(stderr, "Error: Unknown mode '%s'\nAvailable modes: cusparse-csr, stencil5-csr, ellpack, stencil5-ellpack\n", m.c_str());
            return EXIT_FAILURE;
        }
    }
    MatrixData mat;
    if (stencil > 0) {
        if (!app::make_stencil(stencil, &mat)) {
            fprintf(stderr, "Failed to build the %dx%d stencil\n", stencil, stencil);
            return EXIT_FAILURE;
        }
    } else if (load_matrix_market(matrix, &mat) != 0) {
        fprintf(stderr, "Failed to load matrix %s\n", matrix);
        return EXIT_FAILURE;
    }
    printf("Matrix loaded: %d rows, %d cols, %d nonzeros\n", mat.rows, mat.cols, mat.nnz);

    std::vector<double> x((size_t)mat.cols, 1.0), y((size_t)mat.rows, 0.0);
    for (const std::string& m : modes) {
        printf("\n=== Testing mode: %s ===\n", m.c_str());
        SpmvOperator* op = get_operator(m.c_str());
        if (op->init(&mat) != 0) {
            fprintf(stderr, "Failed to initialize operator '%s'\n", op->name);
            continue;
        }
        std::fill(y.begin(), y.end(), 0.0);
        printf("Warmup (5 runs)...\n");
        double ms = 0.0;
        for (int w = 0; w < 5; ++w) op->run_timed(x.data(), y.data(), &ms);
        printf("Running statistical benchmark (10 iterations)...\n");
        BenchmarkStats st;
        if (benchmark_with_stats(op->run_timed, x.data(), y.data(), 10, &st) != 0) {
            fprintf(stderr, "Statistical benchmark failed for mode '%s'\n", op->name);
            op->free();
            continue;
        }
        printf("Completed: %d valid runs, %d outliers removed\n", st.valid_runs, st.outliers_removed);
        double sum = 0.0, sq = 0.0;
        for (int i = 0; i < mat.rows; ++i) {
            sum += y[i];
            sq += y[i] * y[i];
        }
        BenchmarkMetrics metrics;
        calculate_spmv_metrics(st.median_ms, &mat, op->name, &metrics);
        metrics.sum_y = sum;
        metrics.norm2_y = sqrt(sq);
        if (get_gpu_properties(&metrics) != 0) fprintf(stderr, "Warning: Could not retrieve GPU properties\n");
        print_benchmark_metrics(&metrics, stdout);
        if (json) {
            const std::string path = app::per_mode_path(json, op->name, ".json");
            if (FILE* fp = fopen(path.c_str(), "w")) {
                print_metrics_json(&metrics, fp);
                fclose(fp);
                printf("Metrics exported to JSON: %s\n", path.c_str());
            }
        }
        if (csv) {
            const std::string path = app::per_mode_path(csv, op->name, ".csv");
            if (FILE* fp = fopen(path.c_str(), "w")) {
                print_metrics_csv(&metrics, fp);
                fclose(fp);
                printf("Metrics exported to CSV: %s\n", path.c_str());
            }
        }
        printf("SpMV completed successfully using mode: %s\n", op->name);
        printf("\n=== Output Checksum ===\nSum(y):    %.16e\nNorm2(y):  %.16e\n=======================\n\n", sum, sqrt(sq));
        op->free();
    }
    free(mat.entries);
    return EXIT_SUCCESS;
}
