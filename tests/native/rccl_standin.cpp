// rccl_standin.cpp -- a STAND-IN for librccl.so on boxes with fewer GPUs than ranks.  TEST INFRASTRUCTURE, never shipped, never
// loaded unless BEVW_RCCL_LIB points at it (csrc/bevw_comm.h).
//
// Why: real RCCL refuses two ranks of one communicator on the same device, and the GPU pool hands out 1-GPU boxes, so the rank > 0
// branches of the product's native exchange (bevw_shard_gather_parts, bevw_shard_allgather_vsums, k_vsums_interleave in
// csrc/bevwarp.hip / bevw_comm.h) had never executed anywhere.  This library exports the nine entry points the
// product dlsym()s, with the same signatures (<rccl/rccl.h>) and the same stream semantics as far as a consumer on the same stream
// can tell, for several PROCESSES that share device 0:
//   * ncclGetUniqueId: 128 random bytes (they name the abstract unix sockets of the communicator);
//   * ncclCommInitRank: rank r listens on "\0bevw_rccl_<id>_<r>", connects to every lower rank, accepts every higher one: a full mesh
//     of stream sockets;
//   * ncclSend / ncclRecv / ncclAllGather: wait for the stream (everything enqueued before the call is complete), move the bytes
//     device -> host -> socket -> host -> device with synchronous copies (so that everything enqueued after the call sees them);
//     inside ncclGroupStart / ncclGroupEnd the operations are queued and executed at the end, the sends on a helper thread so that a
//     group with sends and receives cannot deadlock; a send and a receive of a rank to itself are matched by a device copy.
// Byte counts only (the product passes ncclUint8); any other type is refused.
//
// Build (tests/test_camera_shard.py does it): hipcc -O2 -std=c++17 -fPIC -shared tests/native/rccl_standin.cpp -o <dir>/librccl_standin.so
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/un.h>
#include <time.h>
#include <unistd.h>

#include <string>
#include <thread>
#include <vector>

namespace {

struct Comm {
    int rank = 0, world = 1;
    std::vector<int> fd;    // socket to every other rank (-1 for itself)
    int listener = -1;
};

struct Op {
    int kind;               // 0 send, 1 recv
    const void *src;
    void *dst;
    size_t bytes;
    int peer;
    Comm *comm;
    hipStream_t stream;
};

// BEVW_RCCL_STANDIN_HOST=1: the buffers are HOST memory (the CPU suite checks the mesh / group logic of this file without a GPU)
const bool g_host = [] { const char *e = getenv("BEVW_RCCL_STANDIN_HOST"); return e && e[0] == '1'; }();
hipError_t copy(void *dst, const void *src, size_t n, hipMemcpyKind kind)
{
    if (g_host) { memmove(dst, src, n); return hipSuccess; }
    return hipMemcpy(dst, src, n, kind);
}

thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

std::string sock_name(const ncclUniqueId &id, int rank)
{
    char hex[40];
    const unsigned char *b = reinterpret_cast<const unsigned char *>(id.internal);
    for (int i = 0; i < 16; ++i) snprintf(hex + 2 * i, 3, "%02x", b[i]);
    return std::string("bevw_rccl_") + hex + "_" + std::to_string(rank);
}

socklen_t fill_addr(sockaddr_un &a, const std::string &name)
{
    memset(&a, 0, sizeof(a));
    a.sun_family = AF_UNIX;
    a.sun_path[0] = 0;      // abstract namespace: nothing to unlink, gone with the process
    memcpy(a.sun_path + 1, name.data(), name.size());
    return (socklen_t)(offsetof(sockaddr_un, sun_path) + 1 + name.size());
}

bool write_all(int fd, const void *p, size_t n)
{
    const char *c = static_cast<const char *>(p);
    while (n) {
        const ssize_t k = ::send(fd, c, n, MSG_NOSIGNAL);
        if (k < 0) { if (errno == EINTR) continue; return false; }
        c += k; n -= (size_t)k;
    }
    return true;
}

bool read_all(int fd, void *p, size_t n)
{
    char *c = static_cast<char *>(p);
    while (n) {
        const ssize_t k = ::recv(fd, c, n, 0);
        if (k < 0) { if (errno == EINTR) continue; return false; }
        if (k == 0) return false;
        c += k; n -= (size_t)k;
    }
    return true;
}

// every queued operation of the calling thread, in order; sends on a helper thread
ncclResult_t run_ops(std::vector<Op> &ops)
{
    if (ops.empty()) return ncclSuccess;
    for (const Op &o : ops)
        if (!g_host && hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
    // a rank's sends to itself pair with its receives from itself, in order
    std::vector<size_t> self_send, self_recv;
    for (size_t i = 0; i < ops.size(); ++i)
        if (ops[i].peer == ops[i].comm->rank) (ops[i].kind == 0 ? self_send : self_recv).push_back(i);
    if (self_send.size() != self_recv.size()) return ncclInvalidUsage;
    for (size_t i = 0; i < self_send.size(); ++i) {
        const Op &s = ops[self_send[i]], &r = ops[self_recv[i]];
        if (s.bytes != r.bytes) return ncclInvalidUsage;
        if (s.bytes && copy(r.dst, s.src, s.bytes, hipMemcpyDeviceToDevice) != hipSuccess) return ncclUnhandledCudaError;
    }
    bool send_ok = true;
    int device = 0;
    if (!g_host) (void)hipGetDevice(&device);
    std::thread sender([&] {
        if (!g_host) (void)hipSetDevice(device);
        std::vector<char> host;
        for (const Op &o : ops) {
            if (o.kind != 0 || o.peer == o.comm->rank) continue;
            host.resize(o.bytes);
            if (o.bytes && copy(host.data(), o.src, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) { send_ok = false; return; }
            const uint64_t n = o.bytes;
            if (!write_all(o.comm->fd[(size_t)o.peer], &n, 8) || !write_all(o.comm->fd[(size_t)o.peer], host.data(), o.bytes)) { send_ok = false; return; }
        }
    });
    bool recv_ok = true;
    std::vector<char> host;
    for (const Op &o : ops) {
        if (o.kind != 1 || o.peer == o.comm->rank) continue;
        uint64_t n = 0;
        if (!read_all(o.comm->fd[(size_t)o.peer], &n, 8) || n != o.bytes) { recv_ok = false; break; }   // sizes must match, as in NCCL
        host.resize(o.bytes);
        if (!read_all(o.comm->fd[(size_t)o.peer], host.data(), o.bytes)) { recv_ok = false; break; }
        if (o.bytes && copy(o.dst, host.data(), o.bytes, hipMemcpyHostToDevice) != hipSuccess) { recv_ok = false; break; }
    }
    sender.join();
    return send_ok && recv_ok ? ncclSuccess : ncclSystemError;
}

ncclResult_t enqueue(const Op &o)
{
    if (!o.comm || o.peer < 0 || o.peer >= o.comm->world) return ncclInvalidArgument;
    g_ops.push_back(o);
    if (g_depth > 0) return ncclSuccess;
    const ncclResult_t r = run_ops(g_ops);
    g_ops.clear();
    return r;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof(*id));
    const int fd = open("/dev/urandom", O_RDONLY);
    if (fd < 0 || read(fd, id->internal, 16) != 16) {
        if (fd >= 0) close(fd);
        return ncclSystemError;
    }
    close(fd);
    memcpy(id->internal + 16, "bevw-rccl-standin", 17);   // (lets a test tell which library answered)
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int world, ncclUniqueId id, int rank)
{
    if (!out || world < 1 || rank < 0 || rank >= world) return ncclInvalidArgument;
    Comm *c = new Comm;
    c->rank = rank; c->world = world;
    c->fd.assign((size_t)world, -1);
    auto fail = [&](ncclResult_t r) {
        for (int f : c->fd) if (f >= 0) close(f);
        if (c->listener >= 0) close(c->listener);
        delete c;
        return r;
    };
    if (world > 1) {
        sockaddr_un a;
        c->listener = socket(AF_UNIX, SOCK_STREAM, 0);
        if (c->listener < 0) return fail(ncclSystemError);
        socklen_t len = fill_addr(a, sock_name(id, rank));
        if (bind(c->listener, reinterpret_cast<sockaddr *>(&a), len) != 0 || listen(c->listener, world) != 0) return fail(ncclSystemError);
        for (int p = 0; p < rank; ++p) {            // the lower ranks are listening, or will be within the time-out
            len = fill_addr(a, sock_name(id, p));
            int fd = -1;
            for (int attempt = 0; attempt < 6000; ++attempt) {   // 60 s
                fd = socket(AF_UNIX, SOCK_STREAM, 0);
                if (fd < 0) return fail(ncclSystemError);
                if (connect(fd, reinterpret_cast<sockaddr *>(&a), len) == 0) break;
                close(fd);
                fd = -1;
                const timespec ts = {0, 10 * 1000 * 1000};
                nanosleep(&ts, nullptr);
            }
            if (fd < 0) return fail(ncclSystemError);
            const int32_t me = rank;
            if (!write_all(fd, &me, 4)) { close(fd); return fail(ncclSystemError); }
            c->fd[(size_t)p] = fd;
        }
        for (int k = rank + 1; k < world; ++k) {
            const int fd = accept(c->listener, nullptr, nullptr);
            int32_t who = -1;
            if (fd < 0 || !read_all(fd, &who, 4) || who <= rank || who >= world || c->fd[(size_t)who] >= 0) {
                if (fd >= 0) close(fd);
                return fail(ncclSystemError);
            }
            c->fd[(size_t)who] = fd;
        }
    }
    *out = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c) return ncclSuccess;
    for (int f : c->fd) if (f >= 0) close(f);
    if (c->listener >= 0) close(c->listener);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart(void) { ++g_depth; return ncclSuccess; }

ncclResult_t ncclGroupEnd(void)
{
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    const ncclResult_t r = run_ops(g_ops);
    g_ops.clear();
    return r;
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
    if (type != ncclUint8 && type != ncclInt8) return ncclInvalidArgument;
    return enqueue(Op{0, buf, nullptr, count, peer, reinterpret_cast<Comm *>(comm), stream});
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
    if (type != ncclUint8 && type != ncclInt8) return ncclInvalidArgument;
    return enqueue(Op{1, nullptr, buf, count, peer, reinterpret_cast<Comm *>(comm), stream});
}

// every rank's `count` bytes into every rank's recv buffer, rank-major: a grouped exchange of the pieces
ncclResult_t ncclAllGather(const void *sendbuf, void *recvbuf, size_t count, ncclDataType_t type, ncclComm_t comm, hipStream_t stream)
{
    if (type != ncclUint8 && type != ncclInt8) return ncclInvalidArgument;
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c) return ncclInvalidArgument;
    ncclGroupStart();
    ncclResult_t r = ncclSuccess;
    for (int p = 0; p < c->world && r == ncclSuccess; ++p) {
        r = ncclSend(sendbuf, count, type, p, comm, stream);
        if (r == ncclSuccess) r = ncclRecv(static_cast<char *>(recvbuf) + (size_t)p * count, count, type, p, comm, stream);
    }
    const ncclResult_t e = ncclGroupEnd();
    return r != ncclSuccess ? r : e;
}

const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
        case ncclSuccess: return "no error (stand-in)";
        case ncclUnhandledCudaError: return "HIP call failed (stand-in)";
        case ncclSystemError: return "socket error (stand-in)";
        case ncclInvalidArgument: return "invalid argument (stand-in)";
        case ncclInvalidUsage: return "invalid usage (stand-in)";
        default: return "error (stand-in)";
    }
}

}  // extern "C"
