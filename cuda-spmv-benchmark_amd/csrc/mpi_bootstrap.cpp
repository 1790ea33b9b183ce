// mpi_bootstrap.cpp -- see mpi_bootstrap.hpp.
//
// MPI has no standard binary interface (before MPI 4.1's ABI): handles are ints with fixed values in the MPICH family (MPICH,
// Intel MPI, MVAPICH, Cray MPT: MPI_COMM_WORLD = 0x44000000, MPI_BYTE = 0x4c00010d) and addresses of global objects in Open MPI
// (&ompi_mpi_comm_world, &ompi_mpi_byte). The two families are told apart by the symbols they export; the four functions used
// take (and this file passes) handles of the family's own type. Anything else is reported as "no MPI": the caller then has to
// build its communicator itself (spmv_amd_comm_create_rccl + spmv_amd_comm_set_world).
#include "mpi_bootstrap.hpp"

#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

namespace spmv_amd {
namespace {

struct Mpi {
    bool usable = false;
    const char* flavour = "";
    // handles as machine words: an int widened (MPICH family) or a pointer (Open MPI)
    uintptr_t comm_world = 0, byte_type = 0;
    bool pointer_handles = false;
    void *initialized = nullptr, *finalized = nullptr, *comm_rank = nullptr, *comm_size = nullptr, *bcast = nullptr;
};

// MPICH, Intel MPI, MVAPICH, Cray MPT: by the library's own version string (MPI_Get_library_version may be called before MPI_Init)
bool mpich_family() {
    using VersionFn = int (*)(char*, int*);
    VersionFn get = reinterpret_cast<VersionFn>(dlsym(RTLD_DEFAULT, "MPI_Get_library_version"));
    if (get == nullptr) return dlsym(RTLD_DEFAULT, "MPII_Version_string") != nullptr || dlsym(RTLD_DEFAULT, "MPIR_Version_string") != nullptr;
    static char text[32768];  // MPI_MAX_LIBRARY_VERSION_STRING is 8192 in MPICH
    int len = 0;
    if (get(text, &len) != 0) return false;
    text[sizeof text - 1] = 0;
    return strstr(text, "MPICH") != nullptr || strstr(text, "Intel(R) MPI") != nullptr || strstr(text, "MVAPICH") != nullptr;
}

const Mpi& mpi() {
    static const Mpi m = [] {
        Mpi r;
        r.initialized = dlsym(RTLD_DEFAULT, "MPI_Initialized");
        r.finalized = dlsym(RTLD_DEFAULT, "MPI_Finalized");
        r.comm_rank = dlsym(RTLD_DEFAULT, "MPI_Comm_rank");
        r.comm_size = dlsym(RTLD_DEFAULT, "MPI_Comm_size");
        r.bcast = dlsym(RTLD_DEFAULT, "MPI_Bcast");
        if (!r.initialized || !r.comm_rank || !r.comm_size || !r.bcast) return r;  // no MPI library in this process
        void* ompi_world = dlsym(RTLD_DEFAULT, "ompi_mpi_comm_world");
        void* ompi_byte = dlsym(RTLD_DEFAULT, "ompi_mpi_byte");
        if (ompi_world && ompi_byte) {
            r.pointer_handles = true;
            r.comm_world = (uintptr_t)ompi_world;
            r.byte_type = (uintptr_t)ompi_byte;
            r.flavour = "Open MPI";
            r.usable = true;
        } else if (mpich_family()) {
            // every MPICH derivative keeps these integer handle values: they are what the MPICH ABI compatibility initiative fixes
            r.comm_world = 0x44000000u;
            r.byte_type = 0x4c00010du;
            r.flavour = "MPICH ABI";
            r.usable = true;
        }
        return r;
    }();
    return m;
}

template <class Handle>
bool world_of(const Mpi& m, MpiWorld* out) {
    using FlagFn = int (*)(int*);
    using RankFn = int (*)(Handle, int*);
    int flag = 0;
    if (reinterpret_cast<FlagFn>(m.initialized)(&flag) != 0 || !flag) return false;
    if (m.finalized && (reinterpret_cast<FlagFn>(m.finalized)(&flag) != 0 || flag)) return false;
    int rank = -1, size = 0;
    if (reinterpret_cast<RankFn>(m.comm_rank)((Handle)m.comm_world, &rank) != 0) return false;
    if (reinterpret_cast<RankFn>(m.comm_size)((Handle)m.comm_world, &size) != 0) return false;
    if (rank < 0 || size < 1 || rank >= size) return false;
    out->rank = rank;
    out->size = size;
    out->flavour = m.flavour;
    return true;
}

template <class Handle>
bool bcast_of(const Mpi& m, void* buf, int bytes, int root) {
    using BcastFn = int (*)(void*, int, Handle, int, Handle);
    return reinterpret_cast<BcastFn>(m.bcast)(buf, bytes, (Handle)m.byte_type, root, (Handle)m.comm_world) == 0;
}

}  // namespace

bool mpi_world(MpiWorld* out) {
    const Mpi& m = mpi();
    if (!m.usable) return false;
    return m.pointer_handles ? world_of<void*>(m, out) : world_of<int>(m, out);
}

bool mpi_bcast_bytes(void* buf, int bytes, int root) {
    const Mpi& m = mpi();
    if (!m.usable) return false;
    return m.pointer_handles ? bcast_of<void*>(m, buf, bytes, root) : bcast_of<int>(m, buf, bytes, root);
}

}  // namespace spmv_amd
