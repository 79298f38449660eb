// bevw_host.h -- host-side plumbing shared by the translation units of libbevwarp.so (bevwarp.hip: handles, tables, tools, the
// camera-per-GPU exchange; bevwarp_plan.hip: the tile plan and its kernels; bevwarp_jpeg.hip: the JPEG codec): the thread-local
// error string behind bevw_last_error(), the HIP_TRY / BEVW_TRY early-return macros, device / pinned buffers, the lap timer.
#pragma once
#include "../../include/bevwarp.h"

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

// one string per thread and library (C++17 inline variable: the same object in every translation unit)
inline thread_local char g_bevw_err[512] = "";

static inline int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_bevw_err, sizeof g_bevw_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                                      \
    do {                                                                                                   \
        hipError_t _e = (expr);                                                                            \
        if (_e != hipSuccess)                                                                              \
            return fail(BEVW_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

#define BEVW_TRY(expr)             \
    do {                           \
        int _s = (expr);           \
        if (_s != BEVW_OK) return _s; \
    } while (0)

static inline int use_device(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(BEVW_E_NO_DEVICE, "no HIP device is visible: libbevwarp has no CPU path");
    }
    if (device < 0 || device >= n) return fail(BEVW_E_NO_DEVICE, "device %d requested, %d visible", device, n);
    HIP_TRY(hipSetDevice(device));
    return BEVW_OK;
}

static inline int launch_check(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(BEVW_E_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
    return BEVW_OK;
}

// Lap timer: HIP events recorded on an engine's own stream WITHOUT synchronising, read back after the caller's final sync
// (bench.py: one mark in front of every step -> per-step durations, median instead of one mean over a 13 ms region).
struct LapTimer {
    std::vector<hipEvent_t> ev;
    int mark(int slot, hipStream_t st)
    {
        if (slot < 0 || slot >= 65536) return fail(BEVW_E_INVALID, "timer slot %d out of range", slot);
        if ((size_t)slot >= ev.size()) ev.resize((size_t)slot + 1, nullptr);
        if (!ev[slot]) HIP_TRY(hipEventCreate(&ev[slot]));
        HIP_TRY(hipEventRecord(ev[slot], st));
        return BEVW_OK;
    }
    int between(int a, int b, float *ms)
    {
        if (!ms || a < 0 || b < 0 || (size_t)a >= ev.size() || (size_t)b >= ev.size() || !ev[a] || !ev[b])
            return fail(BEVW_E_INVALID, "timer slots %d / %d were not marked", a, b);
        HIP_TRY(hipEventSynchronize(ev[b]));
        HIP_TRY(hipEventElapsedTime(ms, ev[a], ev[b]));
        return BEVW_OK;
    }
    void release() { for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e); ev.clear(); }
};

// owns one device allocation: released on every exit path (early HIP_TRY / BEVW_TRY returns included)
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    int reserve(size_t n)
    {
        if (n <= cap) return BEVW_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) {
            p = nullptr;
            (void)hipGetLastError();   // a caller may carry on without this buffer: the failed allocation must not show up as its next launch's error
            return fail(BEVW_E_NOMEM, "hipMalloc(%zu) failed: %s", n, hipGetErrorString(e));
        }
        cap = n;
        return BEVW_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <typename T> T *as() const { return static_cast<T *>(p); }
};

struct PinnedBuf {
    void *p = nullptr;
    size_t cap = 0;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf &) = delete;
    PinnedBuf &operator=(const PinnedBuf &) = delete;
    ~PinnedBuf() { release(); }
    int reserve(size_t n)
    {
        if (n <= cap) return BEVW_OK;
        release();
        hipError_t e = hipHostMalloc(&p, n, hipHostMallocDefault);
        if (e != hipSuccess) { p = nullptr; (void)hipGetLastError(); return fail(BEVW_E_NOMEM, "hipHostMalloc(%zu) failed: %s", n, hipGetErrorString(e)); }
        cap = n;
        return BEVW_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

// (internal, C++ linkage) what other translation units may know of an engine handle: its stream and device
hipStream_t bevw_internal_handle_stream(bevw_handle *h);
int bevw_internal_handle_device(bevw_handle *h);
