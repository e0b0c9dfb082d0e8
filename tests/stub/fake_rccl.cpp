// fake_rccl.cpp -- TEST INFRASTRUCTURE ONLY: a stand-in for the eight RCCL entry points pt_comm.cpp resolves (csrc/pt_comm.cpp, load_rccl),
// so that the library's N-rank plumbing - unique id, ncclCommInitRank per process, ONE reduce per frame and rank whatever buffers the
// caller passed, RGBA8 pack on the root, communicator reuse and teardown - runs with N > 1 processes on a box that has ONE GPU
// (tests/test_multi_rank_gpu.py::test_library_reduce_with_stub_collective; selected with PT_RCCL_PATH).  It is not RCCL and proves
// nothing about RCCL: ranks meet in a directory under /tmp named by the unique id, a reduce is a blocking sum through files.  The real
// library is exercised by the same test on boxes with N GPUs (test_library_reduce_across_processes) and by bench.py --gpus N.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

struct ncclComm {
    int rank, world;
    std::string dir;
    unsigned long seq;
};

namespace {
bool wait_for(const std::string& path, double seconds)
{
    const auto t0 = std::chrono::steady_clock::now();
    struct stat st;
    while (stat(path.c_str(), &st) != 0) {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) return false;
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
    }
    return true;
}
bool write_file(const std::string& path, const void* data, size_t bytes)
{
    const std::string tmp = path + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return false;
    const bool ok = fwrite(data, 1, bytes, f) == bytes;
    fclose(f);
    return ok && rename(tmp.c_str(), path.c_str()) == 0;
}
} // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    std::memset(id->internal, 0, sizeof(id->internal));
    snprintf(id->internal, sizeof(id->internal), "fake_rccl_%d_%lld", (int)getpid(), (long long)std::chrono::steady_clock::now().time_since_epoch().count());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    ncclComm* c = new ncclComm{rank, nranks, std::string("/tmp/") + std::string(id.internal, strnlen(id.internal, sizeof(id.internal))), 0};
    mkdir(c->dir.c_str(), 0700); // every rank may be first
    const char one = 1;
    if (!write_file(c->dir + "/init_" + std::to_string(rank), &one, 1)) { delete c; return ncclSystemError; }
    for (int r = 0; r < nranks; ++r)
        if (!wait_for(c->dir + "/init_" + std::to_string(r), 120.0)) { delete c; return ncclSystemError; }
    *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t*, int, const int*) { return ncclInvalidUsage; } // one process per rank only

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    delete comm;
    return ncclSuccess;
}

// blocking: drains `stream`, then non-roots publish their buffer, the root adds them to its own in rank order
ncclResult_t ncclReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, int root, ncclComm_t comm, hipStream_t stream)
{
    if (!comm || datatype != ncclFloat32 || op != ncclSum || root < 0 || root >= comm->world) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    const unsigned long seq = comm->seq++;
    std::vector<float> mine(count);
    if (hipMemcpy(mine.data(), sendbuff, count * 4, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    const std::string base = comm->dir + "/red_" + std::to_string(seq) + "_";
    if (comm->rank != root) {
        if (!write_file(base + std::to_string(comm->rank), mine.data(), count * 4)) return ncclSystemError;
        // (like the real call, a non-root may return before the root is done; its receive buffer is unspecified)
        return ncclSuccess;
    }
    std::vector<float> other(count);
    for (int r = 0; r < comm->world; ++r) {
        if (r == root) continue;
        const std::string path = base + std::to_string(r);
        if (!wait_for(path, 300.0)) return ncclSystemError;
        FILE* f = fopen(path.c_str(), "rb");
        if (!f || fread(other.data(), 1, count * 4, f) != count * 4) { if (f) fclose(f); return ncclSystemError; }
        fclose(f);
        remove(path.c_str());
        for (size_t i = 0; i < count; ++i) mine[i] += other[i];
    }
    if (hipMemcpy(recvbuff, mine.data(), count * 4, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() { return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return ncclSuccess; }
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error (fake_rccl)" : "fake_rccl error"; }

} // extern "C"
