// mailbox.hip -- the peer-mailbox all-reduce (see comm.hpp): set-up over hipIpc, the exchange kernel, the self-test.
//
// Reference counterpart: MPI_Allreduce of one double on the host, twice per iteration
// (cg_solver_mgpu_partitioned.cu:583,645). Here the reduction never leaves the devices and costs one small launch.
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "comm.hpp"
#include "device_runtime.hpp"
#include "mailbox_device.hpp"
#include "watchdog.hpp"

struct MailboxHost {
    MailboxSlot* inbox = nullptr;             // own mailbox (uncached device memory)
    std::vector<void*> opened;                // peers' mailboxes opened through hipIpc
    unsigned long long* d_seq = nullptr;
    int* h_error = nullptr;                   // pinned, host-coherent
    hipIpcMemHandle_t handle;
    bool prepared = false;
};

namespace spmv_amd {

namespace {
__global__ __launch_bounds__(64) void mailbox_allreduce_kernel(const PeerMailbox* mb, double* value) {
    const double sum = mailbox_allreduce_wave(*mb, *value);
    if (threadIdx.x == 0) *value = sum;
}

size_t inbox_bytes(int world) { return 2 * (size_t)world * sizeof(MailboxSlot); }
}  // namespace

void launch_mailbox_allreduce(const SpmvAmdComm* comm, double* d_value, hipStream_t stream) {
    hipLaunchKernelGGL(mailbox_allreduce_kernel, dim3(1), dim3(64), 0, stream, comm->d_mailbox, d_value);
}

void mailbox_check(const SpmvAmdComm* comm) {
    if (comm->mailbox_host == nullptr || comm->mailbox_host->h_error == nullptr) return;
    const int e = __atomic_load_n(comm->mailbox_host->h_error, __ATOMIC_ACQUIRE);
    if (e != 0) {
        fprintf(stderr, "[comm/mailbox] rank %d: an all-reduce waited longer than its limit for rank %d's contribution "
                        "(peer stopped, or peer stores are not reaching this device)\n", comm->rank, e - 1);
        exit(EXIT_FAILURE);
    }
}

void mailbox_release(SpmvAmdComm* comm) {
    MailboxHost* h = comm->mailbox_host;
    if (h == nullptr) return;
    (void)hipDeviceSynchronize();
    for (void* p : h->opened) (void)hipIpcCloseMemHandle(p);
    if (comm->d_mailbox) (void)hipFree(comm->d_mailbox);
    if (h->inbox) (void)hipFree(h->inbox);
    if (h->d_seq) (void)hipFree(h->d_seq);
    if (h->h_error) (void)hipHostFree(h->h_error);
    delete h;
    comm->mailbox_host = nullptr;
    comm->d_mailbox = nullptr;
}

}  // namespace spmv_amd

using namespace spmv_amd;

// Step 1 (every rank): allocate the mailbox and produce its 64-byte hipIpc handle.
extern "C" int spmv_amd_comm_mailbox_prepare(SpmvAmdComm* comm, void* handle_out64) {
    static_assert(sizeof(hipIpcMemHandle_t) == SPMV_AMD_MAILBOX_HANDLE_BYTES, "hipIpc handle size");
    if (comm == nullptr || comm->world < 1 || comm->world > kMailboxMaxRanks) return 1;
    mailbox_release(comm);
    MailboxHost* h = new MailboxHost();
    // uncached: remote stores must be visible to this device's polling loads without any cache maintenance
    hipError_t e = hipExtMallocWithFlags((void**)&h->inbox, inbox_bytes(comm->world), hipDeviceMallocUncached);
    if (e != hipSuccess) e = hipExtMallocWithFlags((void**)&h->inbox, inbox_bytes(comm->world), hipDeviceMallocFinegrained);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        fprintf(stderr, "[comm/mailbox] rank %d: no uncached / fine-grained device memory: %s\n", comm->rank, hipGetErrorString(e));
        delete h;
        return 1;
    }
    HIP_CHECK(hipMemset(h->inbox, 0, inbox_bytes(comm->world)));
    HIP_CHECK(hipDeviceSynchronize());
    e = hipIpcGetMemHandle(&h->handle, h->inbox);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        fprintf(stderr, "[comm/mailbox] rank %d: hipIpcGetMemHandle: %s\n", comm->rank, hipGetErrorString(e));
        (void)hipFree(h->inbox);
        delete h;
        return 1;
    }
    memcpy(handle_out64, &h->handle, sizeof h->handle);
    h->prepared = true;
    comm->mailbox_host = h;
    return 0;
}

// Step 2 (every rank, after all handles have been exchanged): map the peers' mailboxes.
extern "C" int spmv_amd_comm_mailbox_connect(SpmvAmdComm* comm, const void* all_handles, int count) {
    MailboxHost* h = comm ? comm->mailbox_host : nullptr;
    if (h == nullptr || !h->prepared || count != comm->world) return 1;
    PeerMailbox mb;
    memset(&mb, 0, sizeof mb);
    mb.inbox = h->inbox;
    mb.rank = comm->rank;
    mb.world = comm->world;
    // Until the self-test has passed a wait gives up after 5 s (a mailbox that does not work must not cost more than
    // that before the communicator falls back to its transport's all-reduce); then the limit becomes
    // SPMV_AMD_MAILBOX_TIMEOUT_S (default 20 s: far above any skew between ranks inside a solve, below the host
    // watchdog's 60 s).
    mb.timeout_ticks = (long long)(5.0 * 1e8);  // wall_clock64 runs at 100 MHz
    for (int r = 0; r < comm->world; ++r) {
        if (r == comm->rank) {
            mb.peer_inbox[r] = h->inbox;
            continue;
        }
        hipIpcMemHandle_t peer;
        memcpy(&peer, (const char*)all_handles + (size_t)r * sizeof peer, sizeof peer);
        void* p = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&p, peer, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            fprintf(stderr, "[comm/mailbox] rank %d: cannot map rank %d's mailbox: %s\n", comm->rank, r, hipGetErrorString(e));
            mailbox_release(comm);
            return 1;
        }
        h->opened.push_back(p);
        mb.peer_inbox[r] = static_cast<MailboxSlot*>(p);
    }
    h->d_seq = device_alloc<unsigned long long>(1);
    HIP_CHECK(hipMemset(h->d_seq, 0, sizeof(unsigned long long)));
    HIP_CHECK(hipHostMalloc((void**)&h->h_error, sizeof(int), hipHostMallocCoherent | hipHostMallocMapped));
    *h->h_error = 0;
    mb.seq = h->d_seq;
    mb.host_error = h->h_error;
    comm->d_mailbox = device_alloc<PeerMailbox>(1);
    upload(comm->d_mailbox, &mb, 1);
    HIP_CHECK(hipDeviceSynchronize());
    return 0;
}

extern "C" void spmv_amd_comm_mailbox_disable(SpmvAmdComm* comm) {
    if (comm) mailbox_release(comm);
}

extern "C" int spmv_amd_comm_mailbox_ready(const SpmvAmdComm* comm) { return comm && comm->mailbox_ready() ? 1 : 0; }

// Collective: `rounds` all-reduces of rank-dependent values with known sums, back to back in one stream (so the
// parity double-buffering is exercised), every result compared bit for bit. 0 = every rank may trust its mailbox.
// A timeout inside is reported here as a failure, not as a process exit.
extern "C" int spmv_amd_comm_mailbox_selftest(SpmvAmdComm* comm, int rounds) {
    if (comm == nullptr || !comm->mailbox_ready()) return 1;
    if (rounds < 1) rounds = 1;
    hipStream_t st = nullptr;
    HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    std::vector<double> h((size_t)rounds);
    for (int k = 0; k < rounds; ++k) h[(size_t)k] = (double)(comm->rank + 1) * (k + 1) + 0.125 * k;
    double* d = device_alloc<double>((size_t)rounds);
    upload(d, h.data(), h.size());
    {
        WatchdogScope guard("mailbox self-test", comm->rank);
        for (int k = 0; k < rounds; ++k) launch_mailbox_allreduce(comm, d + k, st);
        HIP_CHECK(hipStreamSynchronize(st));
    }
    download(h.data(), d, h.size());
    device_release(d);
    HIP_CHECK(hipStreamDestroy(st));
    int bad = 0;
    const int P = comm->world;
    for (int k = 0; k < rounds; ++k) {
        double want = 0.0;
        for (int r = 0; r < P; ++r) want += (double)(r + 1) * (k + 1) + 0.125 * k;  // rank order, like the kernel
        if (h[(size_t)k] != want) bad = 1;
    }
    if (*comm->mailbox_host->h_error != 0) {
        fprintf(stderr, "[comm/mailbox] rank %d: self-test timed out waiting for rank %d\n", comm->rank,
                *comm->mailbox_host->h_error - 1);
        *comm->mailbox_host->h_error = 0;
        bad = 1;
    } else if (bad) {
        fprintf(stderr, "[comm/mailbox] rank %d: self-test produced a wrong sum\n", comm->rank);
    }
    if (!bad) {  // trusted from here on: the working limit replaces the probation limit
        double limit_s = 20.0;
        if (const char* v = getenv("SPMV_AMD_MAILBOX_TIMEOUT_S")) limit_s = atof(v);
        const long long ticks = (long long)(limit_s * 1e8);
        HIP_CHECK(hipMemcpy(&comm->d_mailbox->timeout_ticks, &ticks, sizeof ticks, hipMemcpyHostToDevice));
    }
    return bad;
}

// All three steps over the communicator's own transport: the handles travel as one-hot slices of an all-reduce
// (every transport has one), then connect + self-test; ranks agree on the outcome through one more all-reduce,
// so either every rank ends with a working mailbox or none does. Returns 1 if the mailbox is on.
extern "C" int spmv_amd_comm_mailbox_enable(SpmvAmdComm* comm) {
    // one rank: only worth having where the all-reduces are forced anyway (one-GPU measurements of the pipeline)
    if (comm == nullptr || !comm->collective() || comm->world > kMailboxMaxRanks) return 0;
    const int P = comm->world, HB = SPMV_AMD_MAILBOX_HANDLE_BYTES;
    hipStream_t st = nullptr;
    HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    char mine[SPMV_AMD_MAILBOX_HANDLE_BYTES];
    memset(mine, 0, sizeof mine);
    double ok = spmv_amd_comm_mailbox_prepare(comm, mine) == 0 ? 1.0 : 0.0;
    // [P x HB bytes as doubles | P "prepared" flags]
    std::vector<double> box((size_t)P * HB + P, 0.0);
    for (int b = 0; b < HB; ++b) box[(size_t)comm->rank * HB + b] = (double)(unsigned char)mine[b];
    box[(size_t)P * HB + comm->rank] = ok;
    double* d_box = device_alloc<double>(box.size());
    auto exchange = [&](const char* what) {
        WatchdogScope guard(what, comm->rank);
        upload(d_box, box.data(), box.size());
        comm->allreduce_sum(d_box, (int)box.size(), st);
        HIP_CHECK(hipStreamSynchronize(st));
        download(box.data(), d_box, box.size());
    };
    exchange("mailbox set-up: exchanging the hipIpc handles");
    bool all = true;
    for (int r = 0; r < P; ++r) all = all && box[(size_t)P * HB + r] == 1.0;
    int on = 0;
    if (all) {
        std::vector<char> handles((size_t)P * HB);
        for (size_t i = 0; i < handles.size(); ++i) handles[i] = (char)(unsigned char)box[i];
        on = spmv_amd_comm_mailbox_connect(comm, handles.data(), P) == 0 ? 1 : 0;
    }
    // agree on "connected everywhere", then on "self-test passed everywhere"
    for (int phase = 0; phase < 2; ++phase) {
        std::fill(box.begin(), box.end(), 0.0);
        box[0] = on ? 1.0 : 0.0;
        exchange(phase == 0 ? "mailbox set-up: agreeing on the mapping" : "mailbox set-up: agreeing on the self-test");
        const bool everyone = box[0] == (double)P;
        if (!everyone) {
            on = 0;
            break;
        }
        // 2048 all-reduces back to back with known sums (~20 ms): more than a whole benchmark run issues, every result
        // compared bit for bit, before a solver may rely on the mailbox
        if (phase == 0) on = spmv_amd_comm_mailbox_selftest(comm, 2048) == 0 ? 1 : 0;
    }
    if (!on) mailbox_release(comm);
    device_release(d_box);
    HIP_CHECK(hipStreamDestroy(st));
    return on;
}
