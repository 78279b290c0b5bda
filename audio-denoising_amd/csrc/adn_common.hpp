// Shared host/device helpers for libadn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string>
#include <vector>

#include "../../include/adn.h"

namespace adn {

// ---------------------------------------------------------------- errors
void set_error(const char *fmt, ...);

#define ADN_HIP(call)                                                              \
    do {                                                                           \
        hipError_t e_ = (call);                                                    \
        if (e_ != hipSuccess) {                                                    \
            adn::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),  \
                           __FILE__, __LINE__);                                    \
            return ADN_ERR_HIP;                                                    \
        }                                                                          \
    } while (0)

#define ADN_REQUIRE(cond, ...)              \
    do {                                    \
        if (!(cond)) {                      \
            adn::set_error(__VA_ARGS__);    \
            return ADN_ERR_ARG;             \
        }                                   \
    } while (0)

#define ADN_TRY(expr)                \
    do {                             \
        int rc_ = (expr);            \
        if (rc_ != ADN_OK) return rc_; \
    } while (0)

// --------------------------------------------------------------- context
struct StftTable {      // window / twiddle tables of one (n_fft, precision) pair
    int n_fft = 0;
    int elem = 0;       // sizeof(T)
    void *dev = nullptr;
    double wsum = 0.0;  // sum of the window
};

struct KernelRec {      // one timed launch: events recorded on the context's stream
    const char *name;
    hipEvent_t start, stop;
};

struct Ctx {
    StftTable tables[16];
    int n_tables = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    char *ws = nullptr;        // device workspace
    size_t ws_bytes = 0;
    size_t ws_used = 0;        // bump pointer, reset per call
    bool ws_fixed = false;     // the workspace is a caller-independent fixed buffer: never grow, never allocate
    char *scratch = nullptr;   // this device's copy of the library's static scratch array (set by adn_create)
    char *aux = nullptr;       // second growable buffer (planar copies of the time-axis multi-channel mode)
    size_t aux_bytes = 0;
    int num_cus = 256;
    bool force_generic = false;   // test hook: skip the specialised kernels
    bool force_fallback = false;  // test hook: the single-pass threshold statistics send every block to their exact fallback
    int ablate = 0;               // diagnostics: stages the fused STFT kernel skips (timing only)
    bool ktime = false;           // ADN_OPT_KERNEL_TIMING: event pairs around the launches
    std::vector<KernelRec> krecs; // launches timed since the option was set / the last report
    std::string ktime_only;       // adn_kernel_timing_only: the one launch site that is timed ("" = all)
};

void drop_kernel_records(Ctx *c);

// Event pair around one launch when kernel timing is on (costs nothing otherwise).
struct KScope {
    Ctx *c;
    int idx = -1;
    KScope(Ctx *ctx, const char *name) : c(ctx) {
        if (!c->ktime || c->ws_fixed) return;      // creating events allocates: never on the path that allocates nothing
        if (!c->ktime_only.empty() && c->ktime_only != name) return;
        KernelRec r{name, nullptr, nullptr};
        if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) return;
        (void)hipEventRecord(r.start, c->stream);
        c->krecs.push_back(r);
        idx = (int)c->krecs.size() - 1;
    }
    ~KScope() {
        if (idx >= 0) (void)hipEventRecord(c->krecs[idx].stop, c->stream);
    }
};

// Every kernel launch of the library goes through this macro (`c` is the context in scope).
#define ADN_LAUNCH_NAMED(name, kern, grid, block, lds, stream, ...)            \
    do {                                                                       \
        adn::KScope ks_(c, name);                                              \
        hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);       \
    } while (0)
#define ADN_LAUNCH(kern, grid, block, lds, stream, ...) \
    ADN_LAUNCH_NAMED(#kern, kern, grid, block, lds, stream, __VA_ARGS__)

// Ensure the workspace holds at least `bytes`; growing synchronises the stream
// (the old buffer may still be in use) and reallocates.
int ws_reserve(Ctx *c, size_t bytes);
// Same for the second buffer (whole-buffer use, no bump allocator).
int aux_reserve(Ctx *c, size_t bytes);
// Reset the bump allocator (start of an entry point).
inline void ws_reset(Ctx *c) { c->ws_used = 0; }
// Carve `bytes` (256-B aligned) out of the reserved workspace.
template <typename T>
inline T *ws_take(Ctx *c, size_t count) {
    size_t bytes = (count * sizeof(T) + 255) & ~size_t(255);
    T *p = reinterpret_cast<T *>(c->ws + c->ws_used);
    c->ws_used += bytes;
    return p;
}
inline size_t ws_need(size_t count, size_t elem) { return (count * elem + 255) & ~size_t(255); }

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ------------------------------------------------- order-preserving keys
// Monotone map from IEEE floats to unsigned integers (NaNs sort last).
__device__ __forceinline__ uint32_t key_of(float v) {
    uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float val_of(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __uint_as_float(u);
}
__device__ __forceinline__ uint64_t key_of(double v) {
    uint64_t u = (uint64_t)__double_as_longlong(v);
    return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double val_of(uint64_t k) {
    uint64_t u = (k & 0x8000000000000000ull) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    return __longlong_as_double((long long)u);
}

template <typename T> struct KeyT;
template <> struct KeyT<float> { using type = uint32_t; static constexpr int bits = 32; };
template <> struct KeyT<double> { using type = uint64_t; static constexpr int bits = 64; };

// ------------------------------------------------------- PCM file samples
// Integer value of a normalised sample in a 16/24/32-bit PCM file, the way libsndfile converts
// with clipping enabled: scale by 2^31 (exact), round to nearest even, saturate to int32, keep
// the top `bits` bits with an arithmetic shift.  NaN becomes 0.  The float version is exact too:
// the scaling is a power of two and the rounding is that of the same real number.
__device__ __forceinline__ int pcm_int32(float v) {
    const float s = rintf(v * 2147483648.0f);
    const int k = (int)fminf(fmaxf(s, -2147483648.0f), 2147483520.0f);   // largest float below 2^31
    return v != v ? 0 : (s >= 2147483648.0f ? 0x7FFFFFFF : k);
}
__device__ __forceinline__ int pcm_int32(double v) {
    double s = rint(v * 2147483648.0);
    s = fmin(fmax(s, -2147483648.0), 2147483647.0);
    return v != v ? 0 : (int)s;
}

// --------------------------------------------------------- filter banks
// Decomposition low-pass taps; the other three filters follow from them.
struct FilterBank {
    int F;             // taps
    double h[16];      // dec_lo
    double g[16];      // dec_hi
    double rl[16];     // rec_lo
    double rh[16];     // rec_hi
};
const FilterBank &filter_bank(int wavelet);

}  // namespace adn

struct adn_ctx : adn::Ctx {};
