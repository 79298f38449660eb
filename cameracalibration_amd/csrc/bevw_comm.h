// bevw_comm.h -- the ONE exchange step of the camera-per-GPU mode over RCCL (xGMI), native and behind the C-ABI.
//
// The reference has no multi-GPU code; this is the scale-out form of BevGenerator.__call__ (surroundBEV.py:312-325) for
// BASELINE config 5: every rank stitches the cameras it owns, then
//   * balance only: ncclAllGather of the per-frame V sums (8 B x cameras x batch) so that every rank forms the same
//     4-camera mean (luminance_balance, surroundBEV.py:60-66);
//   * grouped ncclSend / ncclRecv of the packed mask boxes to the stitch rank (boxes differ in size, and on xGMI every
//     peer -> root transfer rides its own link).  Never a summing collective: cv2.add saturates (surroundBEV.py:318-320),
//     a wrapping u8 reduce would be wrong on every double-covered seam pixel.
// Everything is enqueued on the handle's own HIP stream: no host synchronisation between the rank-local stitch, the
// exchange and the combine.  librccl.so is dlopen'ed on first use, so the library still loads (and the single-GPU paths
// still run) on machines without RCCL; the types come from <rccl/rccl.h>, nothing is linked.
#pragma once
#include <dlfcn.h>
#include <stdlib.h>
#include <rccl/rccl.h>

namespace bevw {

struct RcclApi {
    void *so = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
    const char *why = "librccl.so not loaded";
};

static inline const RcclApi &rccl()
{
    static const RcclApi api = [] {
        RcclApi a;
        // BEVW_RCCL_LIB (read once per process): the library to use instead -- another RCCL build, or the stand-in of tests/native/rccl_standin.cpp
        // with which the test-suite runs the rank > 0 branches below on a box with fewer GPUs than ranks (real RCCL refuses two ranks on one device)
        const char *over = getenv("BEVW_RCCL_LIB");
        if (over && over[0]) {
            a.so = dlopen(over, RTLD_NOW | RTLD_LOCAL);
            if (!a.so) { a.why = "the library BEVW_RCCL_LIB names could not be dlopen'ed"; return a; }
        }
        const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            if (a.so) break;
            a.so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        }
        if (!a.so) { a.why = "librccl.so could not be dlopen'ed"; return a; }
#define BEVW_SYM(field, name)                                                           \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.so, name));                   \
    if (!a.field) { a.why = "librccl.so lacks " name; return a; }
        BEVW_SYM(GetUniqueId, "ncclGetUniqueId")
        BEVW_SYM(CommInitRank, "ncclCommInitRank")
        BEVW_SYM(CommDestroy, "ncclCommDestroy")
        BEVW_SYM(AllGather, "ncclAllGather")
        BEVW_SYM(Send, "ncclSend")
        BEVW_SYM(Recv, "ncclRecv")
        BEVW_SYM(GroupStart, "ncclGroupStart")
        BEVW_SYM(GroupEnd, "ncclGroupEnd")
        BEVW_SYM(GetErrorString, "ncclGetErrorString")
#undef BEVW_SYM
        a.ok = true;
        a.why = "";
        return a;
    }();
    return api;
}

// per-rank V sums gathered rank-major ([rank][batch][ncams_r], ranks of one group own equally many cameras) -> the
// [batch][4] layout bevw_shard_run_device expects (group order == ascending camera order)
static __global__ void k_vsums_interleave(const unsigned long long *__restrict__ gathered, int nranks, int per_rank, int batch,
                                   unsigned long long *__restrict__ all4)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = batch * nranks * per_rank;
    if (i >= total) return;
    const int b = i / (nranks * per_rank), c = i % (nranks * per_rank);   // c = camera slot 0..3 of frame set b
    const int r = c / per_rank, k = c % per_rank;
    all4[(size_t)b * (nranks * per_rank) + c] = gathered[((size_t)r * batch + b) * per_rank + k];
}

}  // namespace bevw
