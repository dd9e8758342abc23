// common.h -- shared host-side plumbing of libannembed_hip: error reporting, device buffers, the
// per-device stream, launch helpers.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/annembed_hip.h"

namespace ae {

// PROBA_MIN, src/embedder.rs:50
constexpr float kProbaMin = 1.0e-4f;
// thresholds of src/graphlaplace.rs:13-15
constexpr uint64_t kFullMatRepr = 5000;
constexpr uint64_t kFullSvdSizeLimit = 5000;
// Philox stream tags (third counter word) for the non-CE random draws
constexpr uint32_t kTagOmega = 0xFFFF0001u;
constexpr uint32_t kTagProj = 0xFFFF0002u;
constexpr uint32_t kTagRandInit = 0xFFFF0003u;
constexpr uint64_t kDefaultSeed = 4664397ull;  // src/tools/svdapprox.rs:70

struct Error : std::exception {
    int32_t code;
    std::string msg;
    Error(int32_t c, std::string m) : code(c), msg(std::move(m)) {}
    const char* what() const noexcept override { return msg.c_str(); }
};

[[noreturn]] inline void fail(int32_t code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    throw Error(code, buf);
}

void set_last_error(const std::string& s);
void set_last_warning(const std::string& s);   // ae_last_warning_message: a call that succeeded but has something to say
int& api_depth();                               // nesting of entry points on this thread (an outermost call clears the warning)

#define AE_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            ::ae::fail(e_ == hipErrorOutOfMemory ? AE_ERR_OOM : AE_ERR_NO_DEVICE, "%s failed: %s (%s:%d)", #expr, \
                       hipGetErrorString(e_), __FILE__, __LINE__);                                \
    } while (0)

// The library keeps per-process scratch (the stream, pooled buffers, the alternating Gram accumulators ...): entry points
// are serialised by one recursive lock, so handles may be used from several host threads (one at a time runs).
inline std::recursive_mutex& api_mutex() {
    static std::recursive_mutex m;
    return m;
}

// every C-ABI entry point body is wrapped in this: exceptions never cross the ABI
template <class F>
inline int32_t guard(F&& f) {
    std::lock_guard<std::recursive_mutex> lock(api_mutex());
    struct Depth {
        Depth() { if (api_depth()++ == 0) set_last_warning(std::string()); }
        ~Depth() { --api_depth(); }
    } depth;
    try {
        f();
        return AE_OK;
    } catch (const Error& e) {
        set_last_error(e.msg);
        return e.code;
    } catch (const std::bad_alloc&) {
        set_last_error("host allocation failed");
        return AE_ERR_OOM;
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return AE_ERR_INVALID_ARG;
    }
}

// the stream every kernel of the library is launched on (one per process / current device)
hipStream_t stream();
// while it lives, stream() of THIS thread returns `s` (work that runs beside the library stream: ce.hip's batch preparation)
struct StreamScope {
    explicit StreamScope(hipStream_t s);
    ~StreamScope();
    StreamScope(const StreamScope&) = delete;
    StreamScope& operator=(const StreamScope&) = delete;
    hipStream_t prev;
};
void require_device();
// stream-ordered allocations (hipMallocAsync on the library stream, default pool kept warm): for the temporaries
// of the per-call hot functions -- a hipMalloc / hipFree pair costs a device synchronisation
void* pool_alloc(size_t bytes);
void pool_free(void* p);

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    bool pooled = false;
    DevBuf() = default;
    explicit DevBuf(size_t count) { alloc(count); }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n), pooled(o.pooled) { o.p = nullptr; o.n = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { release(); p = o.p; n = o.n; pooled = o.pooled; o.p = nullptr; o.n = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void release() {
        if (p) { if (pooled) pool_free(p); else (void)hipFree(p); }
        p = nullptr; n = 0;
    }
    void alloc(size_t count) {
        release();
        n = count;
        pooled = false;
        if (count) AE_HIP(hipMalloc((void**)&p, count * sizeof(T)));
    }
    // scratch that never leaves the library's stream (not for buffers handed to RCCL / other streams)
    void alloc_pooled(size_t count) {
        release();
        n = count;
        pooled = true;
        if (count) p = static_cast<T*>(pool_alloc(count * sizeof(T)));
    }
    void upload(const T* host, size_t count) {
        if (count > n) alloc(count);
        if (count) AE_HIP(hipMemcpyAsync(p, host, count * sizeof(T), hipMemcpyHostToDevice, stream()));
    }
    void from_host(const T* host, size_t count) { alloc(count); upload(host, count); AE_HIP(hipStreamSynchronize(stream())); }
    void download(T* host, size_t count) const {
        if (count) AE_HIP(hipMemcpyAsync(host, p, count * sizeof(T), hipMemcpyDeviceToHost, stream()));
        AE_HIP(hipStreamSynchronize(stream()));
    }
    void zero() { if (n) AE_HIP(hipMemsetAsync(p, 0, n * sizeof(T), stream())); }
    std::vector<T> to_host() const { std::vector<T> v(n); download(v.data(), n); return v; }
};

// Tuning / A-B switches and the profiling output are read from the environment ONLY when AE_DEBUG_KNOBS is set: a release run
// of the library cannot be altered (numerically or otherwise) from the environment.
inline const char* debug_knob(const char* name) { return getenv("AE_DEBUG_KNOBS") ? getenv(name) : nullptr; }

// one thread per work item: a dispatch carries at most 2^32 - 1 work items (the AQL grid size is 32 bits) -- beyond that the
// kernel must be a grid-stride loop launched through grid_cap()
inline unsigned blocks_for(uint64_t work, unsigned block) {
    if (work + block > 0xFFFFFFFFull) fail(AE_ERR_INVALID_ARG, "internal: a launch of %llu work items exceeds one dispatch", (unsigned long long)work);
    return (unsigned)((work + block - 1) / block);
}
// grid-stride launches: cap the grid at 256 CUs x 8 workgroups (guide G11)
inline unsigned grid_cap(uint64_t work, unsigned block, unsigned cap = 2048 * 4) {
    uint64_t b = (work + block - 1) / block;
    if (b < 1) b = 1;
    return (unsigned)(b > cap ? cap : b);
}

inline void check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) fail(AE_ERR_NO_DEVICE, "kernel launch %s failed: %s", what, hipGetErrorString(e));
}
inline void sync() { AE_HIP(hipStreamSynchronize(stream())); }

}  // namespace ae
