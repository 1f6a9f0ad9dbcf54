// Shared host-side helpers for libwrs_mi355x (gfx950 only; no CPU fallback anywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/wrs.h"

namespace wrs {

typedef _Float16 half_t;

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

void set_last_error(const std::string& m);

#define WRS_THROW(code, ...)                                     \
    do {                                                         \
        char _b[512];                                            \
        snprintf(_b, sizeof(_b), __VA_ARGS__);                   \
        throw ::wrs::Error((code), std::string(_b));             \
    } while (0)

#define WRS_REQUIRE(cond, ...)                                   \
    do {                                                         \
        if (!(cond)) WRS_THROW(WRS_EINVAL, __VA_ARGS__);         \
    } while (0)

#define HIP_CHECK(expr)                                                                              \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess)                                                                        \
            WRS_THROW(WRS_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

// translate exceptions into status codes at the C boundary
template <class F>
static inline int guarded(F&& f) {
    try {
        f();
        return WRS_OK;
    } catch (const Error& e) {
        set_last_error(e.what());
        return e.code;
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return WRS_EINVAL;
    } catch (...) {
        set_last_error("unknown error");
        return WRS_EINVAL;
    }
}

hipStream_t utility_stream();  // a non-blocking stream of the calling thread's current device (created on first use)

// device allocation, zero-initialised (GEMM tiles may over-read into padding: it must be finite)
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void alloc(size_t n) {
        release();
        if (n == 0) n = 256;
        n = (n + 255) & ~size_t(255);
        HIP_CHECK(hipMalloc(&p, n));
        // Zero fill on a utility stream of this device (never the legacy NULL stream: that one orders against other
        // threads' streams and is refused while any of them captures a graph — window-pool workers allocate while their
        // neighbours replay/capture decode steps), and wait for it: the library's work streams are non-blocking, so a
        // late-landing zero fill could otherwise overwrite a later copy/kernel.  Allocation is never on the hot path.
        hipStream_t us = utility_stream();
        HIP_CHECK(hipMemsetAsync(p, 0, n, us));
        HIP_CHECK(hipStreamSynchronize(us));
        bytes = n;
    }
    void ensure(size_t n) {
        if (n > bytes) alloc(n);
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// pinned, device-mapped host memory
struct PinnedBuf {
    void* p = nullptr;
    void* dev = nullptr;
    size_t bytes = 0;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf&) = delete;
    PinnedBuf& operator=(const PinnedBuf&) = delete;
    ~PinnedBuf() { release(); }
    void alloc(size_t n) {
        release();
        if (n == 0) n = 256;
        HIP_CHECK(hipHostMalloc(&p, n, hipHostMallocMapped));
        memset(p, 0, n);
        HIP_CHECK(hipHostGetDevicePointer(&dev, p, 0));
        bytes = n;
    }
    void ensure(size_t n) {
        if (n > bytes) alloc(n);
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = dev = nullptr;
        bytes = 0;
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
    template <class T> T* dev_as() const { return reinterpret_cast<T*>(dev); }
};

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t round_up(int64_t a, int64_t b) { return cdiv(a, b) * b; }

void require_device(int device_id);  // throws WRS_ENODEVICE unless device_id is a gfx950 GPU

// Experiment switches (environment variables that select a measured-and-not-kept variant, or switch a default off for an A/B run)
// are honoured only by the TEST build of the library (libwrs_mi355x_debug.so: debug_api.cpp sets the flag when it is loaded).
// In the product library exp_env() returns nullptr for every name: no environment variable changes what it runs.
extern bool g_experiment_switches;
const char* exp_env(const char* name);

}  // namespace wrs
