// cg_solver_mgpu_stencil -- counterpart of reference src/main/cg_solver_mgpu_stencil.cu without MPI:
// one process per GPU, ranks from the environment, RCCL over xGMI inside the library.
//   RANK=r WORLD_SIZE=P [LOCAL_RANK=d] SPMV_AMD_ID_FILE=/path/shared.id \
//       cg_solver_mgpu_stencil <matrix.mtx | --stencil=N> [--timers] [--json=<file>] [--runs=10]
// Rank 0 writes the 256-byte communicator id to SPMV_AMD_ID_FILE (atomically, via rename), the other
// ranks wait for it; tools/launch_mgpu.sh starts P ranks on one node. With WORLD_SIZE unset or 1 it
// is the single-GPU run behind the reference's "1 GPU" numbers.
// --stencil=N solves on the slab generated in HBM (no host matrix); a .mtx path goes through
// load_matrix_market + cg_benchmark_with_stats_mgpu_partitioned exactly like the reference.
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <algorithm>

#include "app_common.hpp"

static int env_int(const char* k, int d) {
    const char* v = getenv(k);
    return v && *v ? atoi(v) : d;
}

int main(int argc, char** argv) {
    const char *matrix = nullptr, *json = nullptr;
    int stencil = 0, timers = 0, runs = 10;
    for (int i = 1; i < argc; ++i) {
        if (const char* v = app::value_of(argv[i], "--json=")) json = v;
        else if (const char* v2 = app::value_of(argv[i], "--stencil=")) stencil = atoi(v2);
        else if (const char* v3 = app::value_of(argv[i], "--runs=")) runs = atoi(v3);
        else if (!strcmp(argv[i], "--timers")) timers = 1;
        else if (argv[i][0] != '-') matrix = argv[i];
    }
    const int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1);
    if (!matrix && stencil <= 0) {
        if (rank == 0) printf("Usage: %s <matrix.mtx | --stencil=N> [--timers] [--json=<file>] [--runs=10]\n", argv[0]);
        return 1;
    }
    spmv_amd_set_device(env_int("LOCAL_RANK", rank));

    SpmvAmdComm* comm = nullptr;
    if (world > 1) {
        const char* id_file = getenv("SPMV_AMD_ID_FILE");
        if (!id_file) {
            fprintf(stderr, "[Rank %d] SPMV_AMD_ID_FILE must name a path all ranks can see\n", rank);
            return 1;
        }
        char id[SPMV_AMD_COMM_ID_BYTES];
        if (rank == 0) {
            spmv_amd_comm_unique_id(id);
            const std::string tmp = std::string(id_file) + ".tmp";
            FILE* f = fopen(tmp.c_str(), "wb");
            if (!f || fwrite(id, 1, sizeof id, f) != sizeof id) return 1;
            fclose(f);
            rename(tmp.c_str(), id_file);
        } else {
            FILE* f = nullptr;
            for (int tries = 0; tries < 6000 && !(f = fopen(id_file, "rb")); ++tries) usleep(10000);
            if (!f || fread(id, 1, sizeof id, f) != sizeof id) {
                fprintf(stderr, "[Rank %d] could not read the communicator id from %s\n", rank, id_file);
                return 1;
            }
            fclose(f);
        }
        comm = spmv_amd_comm_create_rccl(rank, world, id);
        if (comm == nullptr) {
            fprintf(stderr, "[Rank %d] RCCL communicator could not be created (one rank per GPU is required)\n", rank);
            return 1;
        }
        if (spmv_amd_comm_selftest(comm) != 0) {
            fprintf(stderr, "[Rank %d] communicator self-test failed\n", rank);
            return 1;
        }
        // dot-product all-reduces: ncclAllReduce on the device scalars (what the reference's MPI_Allreduce becomes,
        // cg_solver_mgpu_partitioned.cu:531,583,645). SPMV_AMD_ALLREDUCE=mailbox opts into the peer mailbox (stores
        // between the GPUs; self-tested, all-or-nothing) -- opt-in until a run between distinct devices is on record.
        const char* ar = getenv("SPMV_AMD_ALLREDUCE");
        const int mailbox = (ar && !strcmp(ar, "mailbox")) ? spmv_amd_comm_mailbox_enable(comm) : 0;
        if (rank == 0)
            printf("Transport: %s over %d ranks; dot-product all-reduce: %s\n", spmv_amd_comm_transport(comm),
                   spmv_amd_comm_transport_ranks(comm), mailbox ? "peer mailbox (stores between the GPUs)" : "ncclAllReduce");
        spmv_amd_comm_set_world(comm);
    }

    CGConfigMultiGPU cfg = {1000, 1e-6, 0, timers};
    BenchmarkStats bs;
    CGStatsMultiGPU st;
    memset(&bs, 0, sizeof bs);
    MatrixData mat;
    memset(&mat, 0, sizeof mat);
    if (stencil > 0) {
        // resident path: slab generated on the device, 3 warm-ups + `runs` timed solves
        SpmvAmdCgSlab* slab = spmv_amd_cg_slab_create_stencil5(stencil, comm);
        if (!slab) return 1;
        mat.rows = mat.cols = stencil * stencil;
        mat.nnz = (int)(5LL * stencil * stencil - 4LL * stencil);
        mat.grid_size = stencil;
        for (int w = 0; w < 3; ++w) spmv_amd_cg_slab_solve(slab, &cfg, &st);
        std::vector<double> times;
        std::vector<CGStatsMultiGPU> all;
        for (int r = 0; r < runs; ++r) {
            spmv_amd_cg_slab_solve(slab, &cfg, &st);
            times.push_back(st.time_total_ms);
            all.push_back(st);
        }
        // same statistics rule as benchmark_with_stats: feed the recorded times through it
        static std::vector<double>* feed_times = nullptr;
        static size_t feed_pos = 0;
        feed_times = &times;
        feed_pos = 0;
        auto feed = [](const double*, double*, double* ms) -> int {
            *ms = (*feed_times)[feed_pos++];
            return 0;
        };
        if (benchmark_with_stats(feed, nullptr, nullptr, (int)times.size(), &bs) != 0) return 1;
        // stats of the survivor at position count/2 in run order (benchmark_stats.cu:169-170)
        double mean = 0.0, var = 0.0;
        for (double t : times) mean += t;
        mean /= times.size();
        for (double t : times) var += (t - mean) * (t - mean);
        const double sd = sqrt(var / times.size());
        std::vector<int> kept;
        for (size_t i = 0; i < times.size(); ++i)
            if (fabs(times[i] - mean) <= 2.0 * sd) kept.push_back((int)i);
        st = all[(size_t)kept[kept.size() / 2]];
        spmv_amd_cg_slab_destroy(slab);
    } else {
        if (rank == 0) printf("Loading matrix: %s\n", matrix);
        if (load_matrix_market(matrix, &mat) != 0) {
            fprintf(stderr, "[Rank %d] Error loading matrix: %s\n", rank, matrix);
            return 1;
        }
        std::vector<double> b((size_t)mat.rows, 1.0), x((size_t)mat.rows, 0.0);
        for (int w = 0; w < 3; ++w) {
            std::fill(x.begin(), x.end(), 0.0);
            cg_solve_mgpu_partitioned(nullptr, &mat, b.data(), x.data(), cfg, &st);
        }
        std::fill(x.begin(), x.end(), 0.0);
        if (cg_benchmark_with_stats_mgpu_partitioned(nullptr, &mat, b.data(), x.data(), cfg, runs, &bs, &st) != 0) {
            fprintf(stderr, "[Rank %d] benchmark failed\n", rank);
            return 1;
        }
        if (rank == 0) printf("Sum(x):    %.16e\nNorm2(x):  %.16e\n", st.solution_sum, st.solution_norm);
    }
    if (rank == 0) {
        printf("\n========================================\nMulti-GPU CG: %d rank(s), %d unknowns\n", world, mat.rows);
        printf("Converged: %s in %d iterations (residual %.6e)\n", st.converged ? "YES" : "NO", st.iterations, st.residual_norm);
        printf("Time (median of %d, %d outliers removed): %.3f ms  [min %.3f, max %.3f, std %.3f]\n", bs.valid_runs,
               bs.outliers_removed, bs.median_ms, bs.min_ms, bs.max_ms, bs.std_dev_ms);
        printf("Iterations/s: %.2f\n========================================\n", st.iterations / (bs.median_ms / 1e3));
        if (json) export_cg_mgpu_json(json, "stencil5-partitioned", &mat, &bs, &st, world);
    }
    if (comm) spmv_amd_comm_destroy(comm);
    free(mat.entries);
    return 0;
}
