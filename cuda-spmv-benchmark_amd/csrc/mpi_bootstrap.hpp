// mpi_bootstrap.hpp -- rank, rank count and a byte broadcast from the MPI library the PROCESS has already initialised, found at
// run time (dlsym(RTLD_DEFAULT, ...)): libspmv_amd.so has no link-time MPI dependency and works without any MPI installed.
//
// Why: the reference's multi-GPU main (src/main/cg_solver_mgpu_stencil.cu:23-27,105-131) calls MPI_Init and then
// cg_solve_mgpu_partitioned(NULL, &mat, b, x, config, &stats) -- the solver finds its rank, the rank count and its device by
// itself (src/solvers/cg_solver_mgpu_partitioned.cu:240-259: MPI_Comm_rank / MPI_Comm_size / cudaSetDevice(rank)). For that main
// to run on this library without a source change the solver must do the same: when no world communicator has been handed over
// (spmv_amd_comm_set_world) and MPI is initialised in the process, the first solve creates the RCCL communicators itself --
// rank 0 draws the 256-byte id, MPI_Bcast carries it, hipSetDevice(rank % devices) -- and keeps them for the later solves.
#pragma once

namespace spmv_amd {

struct MpiWorld {
    int rank = 0, size = 1;
    const char* flavour = "";  // "MPICH ABI" | "Open MPI"
};

// false: no MPI library in the process, an unknown one, or MPI_Init has not been called (or MPI_Finalize has)
bool mpi_world(MpiWorld* out);
// MPI_Bcast of `bytes` bytes from `root` over MPI_COMM_WORLD; false on failure
bool mpi_bcast_bytes(void* buf, int bytes, int root);

}  // namespace spmv_amd
