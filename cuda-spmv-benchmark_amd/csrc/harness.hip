// harness.hip -- the harness-side helpers the reference mains call around the hot path:
// N-run statistics, metric formulas, device/host description and the JSON/CSV exporters.
//
//   benchmark_with_stats, cg_benchmark_with_stats_device   <- reference src/spmv/benchmark_stats.cu
//   cg_benchmark_with_stats_mgpu_partitioned  <- reference src/spmv/benchmark_stats_mgpu_partitioned.cu
//   calculate_spmv_metrics, print_*           <- reference src/spmv/spmv_metrics.cu
//   get_gpu_properties                        <- reference src/spmv/gpu_detection.cu
//   export_cg_json / _mgpu_json / _csv        <- reference src/solvers/cg_metrics.cu
//
// Statistics rule (kept exactly): mean and population standard deviation over the valid runs,
// drop runs further than 2 sigma from the mean, then mean / sigma / median / min / max of the
// survivors; fewer than 3 valid runs is an error (-1). The CG wrappers restore x before every
// run and hand back the stats of the survivor at position count/2 in run order (the reference
// indexes an unsorted list there, benchmark_stats.cu:169-170).
// JSON/CSV key names are the reference's, so scripts that scrape "execution_time_ms" or
// "median_ms" keep working.
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <functional>
#include <string>
#include <vector>

#include "device_runtime.hpp"

namespace {

struct Survivors {
    std::vector<double> times;
    std::vector<int> run_index;
};

double mean_of(const std::vector<double>& t) {
    double s = 0.0;
    for (double v : t) s += v;
    return s / (double)t.size();
}

double sigma_of(const std::vector<double>& t, double mean) {
    double q = 0.0;
    for (double v : t) q += (v - mean) * (v - mean);
    return sqrt(q / (double)t.size());
}

// Returns false when fewer than 3 runs are valid.
bool summarize(const std::vector<double>& times, BenchmarkStats* out, Survivors* kept) {
    if (times.size() < 3) return false;
    const double mean = mean_of(times), sigma = sigma_of(times, mean);
    kept->times.clear();
    kept->run_index.clear();
    for (size_t i = 0; i < times.size(); ++i) {
        if (fabs(times[i] - mean) <= 2.0 * sigma) {
            kept->times.push_back(times[i]);
            kept->run_index.push_back((int)i);
        }
    }
    out->mean_ms = mean_of(kept->times);
    out->std_dev_ms = sigma_of(kept->times, out->mean_ms);
    std::vector<double> sorted = kept->times;
    std::sort(sorted.begin(), sorted.end());
    const size_t m = sorted.size();
    out->median_ms = (m % 2 == 0) ? (sorted[m / 2 - 1] + sorted[m / 2]) / 2.0 : sorted[m / 2];
    out->min_ms = sorted.front();
    out->max_ms = sorted.back();
    out->valid_runs = (int)m;
    out->outliers_removed = (int)(times.size() - m);
    return true;
}

template <class Stats, class Config>
int cg_runs_with_stats(int rows, double* x, Config config, int num_runs, BenchmarkStats* bench,
                       Stats* final_stats, const std::function<int(Config, Stats*)>& solve) {
    if (num_runs <= 0) return -1;
    std::vector<double> x_start(x, x + rows);
    std::vector<Stats> all((size_t)num_runs);
    std::vector<double> times;
    config.verbose = 0;  // silent runs
    int valid = 0;
    for (int i = 0; i < num_runs; ++i) {
        memcpy(x, x_start.data(), (size_t)rows * sizeof(double));
        if (solve(config, &all[valid]) == 0) {
            times.push_back(all[valid].time_total_ms);
            ++valid;
        }
    }
    Survivors kept;
    if (!summarize(times, bench, &kept)) return -1;
    *final_stats = all[(size_t)kept.run_index[kept.run_index.size() / 2]];
    return 0;
}

struct Traffic {
    double flops, values_bytes, index_bytes, vector_bytes, total_bytes;
};

// Byte model of calculate_spmv_metrics (spmv_metrics.cu:68-98).
Traffic traffic_model(const BenchmarkMetrics* m) {
    Traffic t;
    t.flops = 2.0 * m->matrix_nnz;
    t.vector_bytes = (double)m->matrix_cols * sizeof(double) + (double)m->matrix_rows * sizeof(double);
    const bool csr_like = m->operator_name && (!strcmp(m->operator_name, "cusparse-csr") ||
                                               !strcmp(m->operator_name, "stencil5-csr"));
    if (csr_like) {
        t.values_bytes = (double)csr_mat.nb_nonzeros * sizeof(double);
        t.index_bytes = (double)csr_mat.nb_nonzeros * sizeof(int) + ((double)csr_mat.nb_rows + 1) * sizeof(int);
    } else {
        t.values_bytes = (double)m->matrix_nnz * sizeof(double);
        t.index_bytes = (double)m->matrix_nnz * sizeof(int) * 2;
    }
    t.total_bytes = t.values_bytes + t.index_bytes + t.vector_bytes;
    return t;
}

const char* bound_label(double intensity) {
    return intensity < 0.25 ? "memory-bound" : intensity < 2.0 ? "balanced" : "compute-bound";
}

void timestamp_now(char* out, size_t cap) {
    time_t now = time(nullptr);
    strftime(out, cap, "%Y-%m-%d %H:%M:%S", localtime(&now));
}

double cg_spmv_gflops(const MatrixData* mat, int iterations, double spmv_ms) {
    return spmv_ms > 0.0 ? (2.0 * mat->nnz * iterations) / (spmv_ms * 1e6) : 0.0;
}

}  // namespace

extern "C" int benchmark_with_stats(int (*run_func)(const double*, double*, double*),
                                    const double* x, double* y, int num_runs,
                                    BenchmarkStats* stats) {
    std::vector<double> times;
    for (int i = 0; i < num_runs; ++i) {
        double ms = 0.0;
        if (run_func(x, y, &ms) == 0) times.push_back(ms);
    }
    Survivors kept;
    return summarize(times, stats, &kept) ? 0 : -1;
}

extern "C" int cg_benchmark_with_stats_device(SpmvOperator* spmv_op, MatrixData* mat, double* b,
                                              double* x, CGConfig config, int num_runs,
                                              BenchmarkStats* bench_stats, CGStats* final_stats) {
    return cg_runs_with_stats<CGStats, CGConfig>(
        mat->rows, x, config, num_runs, bench_stats, final_stats,
        [&](CGConfig c, CGStats* st) { return cg_solve_device(spmv_op, mat, b, x, c, st); });
}

extern "C" int cg_benchmark_with_stats_mgpu_partitioned(SpmvOperator* spmv_op, MatrixData* mat,
                                                        double* b, double* x,
                                                        CGConfigMultiGPU config, int num_runs,
                                                        BenchmarkStats* bench_stats,
                                                        CGStatsMultiGPU* final_stats) {
    return cg_runs_with_stats<CGStatsMultiGPU, CGConfigMultiGPU>(
        mat->rows, x, config, num_runs, bench_stats, final_stats,
        [&](CGConfigMultiGPU c, CGStatsMultiGPU* st) {
            return cg_solve_mgpu_partitioned(spmv_op, mat, b, x, c, st);
        });
}

extern "C" void calculate_spmv_metrics(double execution_time_ms, const MatrixData* mat,
                                       const char* operator_name, BenchmarkMetrics* metrics) {
    metrics->matrix_rows = mat->rows;
    metrics->matrix_cols = mat->cols;
    metrics->matrix_nnz = mat->nnz;
    metrics->grid_size = mat->grid_size;
    metrics->execution_time_ms = execution_time_ms;
    metrics->operator_name = operator_name;
    metrics->sparsity_ratio = (double)mat->nnz / ((double)mat->rows * mat->cols);
    const double secs = execution_time_ms / 1000.0;
    const Traffic t = traffic_model(metrics);
    metrics->gflops = (t.flops / secs) / 1e9;
    metrics->bandwidth_gb_s = (t.total_bytes / secs) / 1e9;
}

namespace {
// Temperature / power through ROCm SMI when the library is present (the reference shells out to
// nvidia-smi for the same fields, gpu_detection.cu:41-74). Loaded lazily with dlopen so that the
// product has no link-time dependency on it; every field stays 0 when it is unavailable.
void fill_smi_fields(int device, BenchmarkMetrics* metrics) {
    void* smi = dlopen("librocm_smi64.so", RTLD_NOW | RTLD_LOCAL);
    if (!smi) smi = dlopen("/opt/rocm/lib/librocm_smi64.so", RTLD_NOW | RTLD_LOCAL);
    if (!smi) return;
    typedef int (*init_fn)(uint64_t);
    typedef int (*temp_fn)(uint32_t, uint32_t, int, int64_t*);
    typedef int (*power_fn)(uint32_t, uint64_t*);
    typedef int (*cap_fn)(uint32_t, uint32_t, uint64_t*);
    init_fn init = (init_fn)dlsym(smi, "rsmi_init");
    temp_fn temp = (temp_fn)dlsym(smi, "rsmi_dev_temp_metric_get");
    power_fn power = (power_fn)dlsym(smi, "rsmi_dev_current_socket_power_get");
    cap_fn cap = (cap_fn)dlsym(smi, "rsmi_dev_power_cap_get");
    if (init && init(0) == 0) {
        int64_t milli_c = 0;
        uint64_t micro_w = 0;
        // edge, junction (hotspot), memory: the first sensor the board exposes
        for (uint32_t sensor = 0; temp && sensor < 3 && metrics->gpu_info.current_temp_c == 0; ++sensor)
            if (temp((uint32_t)device, sensor, /*RSMI_TEMP_CURRENT*/ 0, &milli_c) == 0)
                metrics->gpu_info.current_temp_c = (int)(milli_c / 1000);
        if (power && power((uint32_t)device, &micro_w) == 0) metrics->gpu_info.power_draw_w = (int)(micro_w / 1000000);
        if (cap && cap((uint32_t)device, 0, &micro_w) == 0) metrics->gpu_info.power_limit_w = (int)(micro_w / 1000000);
    }
}
}  // namespace

extern "C" int get_gpu_properties(BenchmarkMetrics* metrics) {
    memset(&metrics->gpu_info, 0, sizeof(metrics->gpu_info));
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 1;
    if (prop.name[0] != '\0')
        snprintf(metrics->gpu_info.name, sizeof metrics->gpu_info.name, "%s", prop.name);
    else  // some driver stacks leave the marketing name empty
        snprintf(metrics->gpu_info.name, sizeof metrics->gpu_info.name, "AMD GPU (%s)", prop.gcnArchName);
    metrics->gpu_info.memory_mb = (int)(prop.totalGlobalMem / (1024 * 1024));
    snprintf(metrics->gpu_info.compute_capability, sizeof metrics->gpu_info.compute_capability, "%s",
             prop.gcnArchName);
    metrics->gpu_info.multiprocessor_count = prop.multiProcessorCount;
    metrics->gpu_info.max_threads_per_block = prop.maxThreadsPerBlock;
    metrics->gpu_info.memory_clock_khz = prop.memoryClockRate;
    metrics->gpu_info.graphics_clock_mhz = prop.clockRate / 1000;
    int v = 0;
    if (hipRuntimeGetVersion(&v) == hipSuccess) metrics->gpu_info.cuda_runtime_version = v;
    if (hipDriverGetVersion(&v) == hipSuccess) metrics->gpu_info.cuda_driver_version = v;
    snprintf(metrics->gpu_info.persistence_mode, sizeof metrics->gpu_info.persistence_mode, "n/a");
    snprintf(metrics->gpu_info.pcie_generation, sizeof metrics->gpu_info.pcie_generation, "n/a");
    // host description from /proc, as gpu_detection.cu:9-33 does
    snprintf(metrics->gpu_info.cpu_model, sizeof metrics->gpu_info.cpu_model, "unknown");
    if (FILE* f = fopen("/proc/cpuinfo", "r")) {
        char line[512];
        while (fgets(line, sizeof line, f)) {
            if (!strncmp(line, "model name", 10)) {
                const char* colon = strchr(line, ':');
                if (colon) {
                    std::string name(colon + 1);
                    while (!name.empty() && (name.back() == '\n' || name.back() == ' ')) name.pop_back();
                    size_t start = name.find_first_not_of(' ');
                    snprintf(metrics->gpu_info.cpu_model, sizeof metrics->gpu_info.cpu_model, "%s",
                             start == std::string::npos ? "" : name.c_str() + start);
                }
                break;
            }
        }
        fclose(f);
    }
    fill_smi_fields(dev, metrics);
    if (FILE* f = fopen("/proc/meminfo", "r")) {
        long kb = 0;
        if (fscanf(f, "MemTotal: %ld kB", &kb) == 1) metrics->gpu_info.system_ram_gb = (int)(kb / (1024 * 1024));
        fclose(f);
    }
    return 0;
}

extern "C" void print_benchmark_metrics(const BenchmarkMetrics* m, FILE* fp) {
    if (!fp) fp = stdout;
    const Traffic t = traffic_model(m);
    const double intensity = t.flops / t.total_bytes;
    fprintf(fp, "\n=== SpMV Performance Metrics ===\n");
    fprintf(fp, "Operator: %s\n", m->operator_name);
    fprintf(fp, "\n--- Matrix Characteristics ---\n");
    if (m->grid_size > 0) {
        fprintf(fp, "Grid size: %d x %d (2D stencil)\n", m->grid_size, m->grid_size);
        fprintf(fp, "Matrix dimensions: %d x %d (grid^2)\n", m->matrix_rows, m->matrix_cols);
    } else {
        fprintf(fp, "Matrix dimensions: %d x %d\n", m->matrix_rows, m->matrix_cols);
    }
    fprintf(fp, "Non-zeros: %d\n", m->matrix_nnz);
    fprintf(fp, "Sparsity ratio: %.6f (%.4f%% non-zero)\n", m->sparsity_ratio, m->sparsity_ratio * 100.0);
    fprintf(fp, "\n--- Performance Metrics ---\n");
    fprintf(fp, "Execution time: %.3f ms (%.1f us)\n", m->execution_time_ms, m->execution_time_ms * 1000.0);
    fprintf(fp, "GFLOPS: %.3f\n", m->gflops);
    fprintf(fp, "Memory bandwidth: %.3f GB/s\n", m->bandwidth_gb_s);
    fprintf(fp, "\n--- Performance Analysis ---\n");
    fprintf(fp, "Arithmetic intensity: %.3f FLOP/byte\n", intensity);
    fprintf(fp, "Classification: %s\n", bound_label(intensity));
    fprintf(fp, "=============================\n\n");
}

extern "C" void print_metrics_json(const BenchmarkMetrics* m, FILE* fp) {
    if (!fp) fp = stdout;
    const Traffic t = traffic_model(m);
    const double intensity = t.flops / t.total_bytes;
    const auto& g = m->gpu_info;
    fprintf(fp, "{\n  \"gpu\": {\n");
    fprintf(fp, "    \"name\": \"%s\",\n    \"memory_mb\": %d,\n", g.name, g.memory_mb);
    fprintf(fp, "    \"compute_capability\": \"%s\",\n    \"multiprocessor_count\": %d,\n",
            g.compute_capability, g.multiprocessor_count);
    fprintf(fp, "    \"memory_clock_khz\": %d,\n    \"graphics_clock_mhz\": %d,\n", g.memory_clock_khz,
            g.graphics_clock_mhz);
    fprintf(fp, "    \"cuda_runtime_version\": %d,\n    \"cuda_driver_version\": %d,\n",
            g.cuda_runtime_version, g.cuda_driver_version);
    fprintf(fp, "    \"cusparse_version\": %d,\n    \"current_temp_c\": %d,\n", g.cusparse_version,
            g.current_temp_c);
    fprintf(fp, "    \"power_draw_w\": %d,\n    \"power_limit_w\": %d,\n", g.power_draw_w, g.power_limit_w);
    fprintf(fp, "    \"persistence_mode\": \"%s\",\n    \"pcie_generation\": \"%s\",\n", g.persistence_mode,
            g.pcie_generation);
    fprintf(fp, "    \"pcie_link_width\": %d\n  },\n", g.pcie_link_width);
    fprintf(fp, "  \"system\": {\n    \"cpu_model\": \"%s\",\n    \"system_ram_gb\": %d\n  },\n", g.cpu_model,
            g.system_ram_gb);
    fprintf(fp, "  \"benchmark\": {\n    \"operator\": \"%s\",\n    \"matrix\": {\n", m->operator_name);
    if (m->grid_size > 0)
        fprintf(fp, "      \"grid_size\": %d,\n      \"grid_dimensions\": \"%dx%d\",\n", m->grid_size,
                m->grid_size, m->grid_size);
    fprintf(fp, "      \"rows\": %d,\n      \"cols\": %d,\n      \"nnz\": %d,\n", m->matrix_rows,
            m->matrix_cols, m->matrix_nnz);
    fprintf(fp, "      \"sparsity_ratio\": %.6f,\n      \"sparsity_percent\": %.4f\n    },\n",
            m->sparsity_ratio, m->sparsity_ratio * 100.0);
    fprintf(fp, "    \"performance\": {\n      \"execution_time_ms\": %.6f,\n", m->execution_time_ms);
    fprintf(fp, "      \"execution_time_us\": %.1f,\n      \"gflops\": %.6f,\n", m->execution_time_ms * 1000.0,
            m->gflops);
    fprintf(fp, "      \"bandwidth_gb_s\": %.6f\n    },\n", m->bandwidth_gb_s);
    fprintf(fp, "    \"analysis\": {\n      \"arithmetic_intensity\": %.6f,\n", intensity);
    fprintf(fp, "      \"total_flops\": %.0f,\n      \"total_bytes\": %.0f,\n", t.flops, t.total_bytes);
    fprintf(fp, "      \"matrix_data_bytes\": %.0f,\n      \"matrix_indices_bytes\": %.0f,\n", t.values_bytes,
            t.index_bytes);
    fprintf(fp, "      \"vector_bytes\": %.0f,\n      \"performance_bound\": \"%s\"\n    },\n", t.vector_bytes,
            bound_label(intensity));
    fprintf(fp, "    \"validation\": {\n      \"sum_y\": %.16e,\n      \"norm2_y\": %.16e\n    }\n  }\n}\n",
            m->sum_y, m->norm2_y);
}

extern "C" void print_metrics_csv(const BenchmarkMetrics* m, FILE* fp) {
    if (!fp) fp = stdout;
    const Traffic t = traffic_model(m);
    const double intensity = t.flops / t.total_bytes;
    fprintf(fp,
            "operator,grid_size,matrix_rows,matrix_cols,matrix_nnz,sparsity_ratio,sparsity_percent,"
            "execution_time_ms,execution_time_us,gflops,bandwidth_gb_s,"
            "arithmetic_intensity,total_flops,performance_bound,sum_y,norm2_y\n");
    fprintf(fp, "%s,%d,%d,%d,%d,%.6f,%.4f,", m->operator_name, m->grid_size, m->matrix_rows, m->matrix_cols,
            m->matrix_nnz, m->sparsity_ratio, m->sparsity_ratio * 100.0);
    fprintf(fp, "%.6f,%.1f,%.6f,%.6f,", m->execution_time_ms, m->execution_time_ms * 1000.0, m->gflops,
            m->bandwidth_gb_s);
    fprintf(fp, "%.6f,%.0f,%s,%.16e,%.16e\n", intensity, t.flops, bound_label(intensity), m->sum_y, m->norm2_y);
}

namespace {
template <class Stats>
void cg_json_common(FILE* fp, const char* solver, const char* mode, int num_gpus, const MatrixData* mat,
                    const BenchmarkStats* b, const Stats* c, const char* extra_timing) {
    char stamp[64];
    timestamp_now(stamp, sizeof stamp);
    fprintf(fp, "{\n  \"timestamp\": \"%s\",\n  \"solver\": \"%s\",\n  \"mode\": \"%s\",\n", stamp, solver, mode);
    if (num_gpus > 0) fprintf(fp, "  \"num_gpus\": %d,\n", num_gpus);
    fprintf(fp, "  \"matrix\": {\n    \"rows\": %d,\n    \"cols\": %d,\n    \"nnz\": %d,\n    \"grid_size\": %d\n  },\n",
            mat->rows, mat->cols, mat->nnz, mat->grid_size);
    fprintf(fp, "  \"convergence\": {\n    \"converged\": %s,\n    \"iterations\": %d,\n    \"residual_norm\": %.15e\n  },\n",
            c->converged ? "true" : "false", c->iterations, c->residual_norm);
    fprintf(fp, "  \"timing\": {\n    \"median_ms\": %.3f,\n    \"mean_ms\": %.3f,\n    \"min_ms\": %.3f,\n", b->median_ms,
            b->mean_ms, b->min_ms);
    fprintf(fp, "    \"max_ms\": %.3f,\n    \"std_dev_ms\": %.3f,\n    \"spmv_ms\": %.3f,\n", b->max_ms, b->std_dev_ms,
            c->time_spmv_ms);
    fprintf(fp, "    \"blas1_ms\": %.3f,\n    \"reductions_ms\": %.3f%s\n  },\n", c->time_blas1_ms,
            c->time_reductions_ms, extra_timing);
    fprintf(fp, "  \"statistics\": {\n    \"valid_runs\": %d,\n    \"outliers_removed\": %d\n  },\n", b->valid_runs,
            b->outliers_removed);
    fprintf(fp, "  \"performance\": {\n    \"gflops_spmv\": %.3f\n  },\n",
            cg_spmv_gflops(mat, c->iterations, c->time_spmv_ms));
    fprintf(fp, "  \"validation\": {\n    \"solution_sum\": %.16e,\n    \"solution_norm\": %.16e\n  }\n}\n",
            c->solution_sum, c->solution_norm);
}
}  // namespace

extern "C" void export_cg_json(const char* filename, const char* mode, const MatrixData* mat,
                               const BenchmarkStats* bench_stats, const CGStats* cg_stats) {
    FILE* fp = fopen(filename, "w");
    if (!fp) {
        fprintf(stderr, "Error: Could not open %s for writing\n", filename);
        return;
    }
    cg_json_common(fp, "CG", mode, 0, mat, bench_stats, cg_stats, "");
    fclose(fp);
}

extern "C" void export_cg_mgpu_json(const char* filename, const char* mode, const MatrixData* mat,
                                    const BenchmarkStats* bench_stats,
                                    const CGStatsMultiGPU* cg_stats, int num_gpus) {
    FILE* fp = fopen(filename, "w");
    if (!fp) {
        fprintf(stderr, "Error: Could not open %s for writing\n", filename);
        return;
    }
    char extra[128];
    snprintf(extra, sizeof extra, ",\n    \"allreduce_ms\": %.3f,\n    \"allgather_ms\": %.3f",
             cg_stats->time_allreduce_ms, cg_stats->time_allgather_ms);
    cg_json_common(fp, "CG Multi-GPU", mode, num_gpus, mat, bench_stats, cg_stats, extra);
    fclose(fp);
}

extern "C" void export_cg_csv(const char* filename, const char* mode, const MatrixData* mat,
                              const BenchmarkStats* b, const CGStats* c, bool write_header) {
    FILE* fp = fopen(filename, write_header ? "w" : "a");
    if (!fp) {
        fprintf(stderr, "Error: Could not open %s for writing\n", filename);
        return;
    }
    if (write_header)
        fprintf(fp,
                "mode,rows,cols,nnz,grid_size,converged,iterations,residual_norm,"
                "median_ms,mean_ms,min_ms,max_ms,std_dev_ms,spmv_ms,blas1_ms,reductions_ms,"
                "valid_runs,outliers_removed,gflops_spmv,solution_sum,solution_norm\n");
    fprintf(fp, "%s,%d,%d,%d,%d,%d,%d,%.15e,", mode, mat->rows, mat->cols, mat->nnz, mat->grid_size,
            c->converged, c->iterations, c->residual_norm);
    fprintf(fp, "%.3f,%.3f,%.3f,%.3f,%.3f,", b->median_ms, b->mean_ms, b->min_ms, b->max_ms, b->std_dev_ms);
    fprintf(fp, "%.3f,%.3f,%.3f,", c->time_spmv_ms, c->time_blas1_ms, c->time_reductions_ms);
    fprintf(fp, "%d,%d,%.3f,%.16e,%.16e\n", b->valid_runs, b->outliers_removed,
            cg_spmv_gflops(mat, c->iterations, c->time_spmv_ms), c->solution_sum, c->solution_norm);
    fclose(fp);
}
