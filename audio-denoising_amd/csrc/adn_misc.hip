// Small kernels: RMS-window energies for the window profiler and the
// bit-reproducible synthetic test-signal generator.
#include "adn_kernels.hpp"

namespace adn {

// ------------------------------------------------------------ window RMS
// One wave per window: sum of squares over every channel of the window's rows
// (lib/windowBundle.py:67-74), divided by the row count, square root.
template <typename T>
__global__ __launch_bounds__(256) void k_window_rms(const T *__restrict__ x, int64_t n_rows,
                                                    int n_ch, int64_t win, int64_t n_win,
                                                    double *__restrict__ rms) {
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (w >= n_win) return;
    const int64_t r0 = w * win, r1 = (r0 + win < n_rows) ? r0 + win : n_rows;
    const T *p = x + r0 * n_ch;
    const int64_t cnt = (r1 - r0) * n_ch;
    double s = 0.0;
    for (int64_t i = lane; i < cnt; i += 64) {
        const double v = (double)p[i];
        s += v * v;
    }
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    if (lane == 0) rms[w] = sqrt(s / (double)(r1 - r0));
}

template <typename T>
int window_rms(Ctx *c, const T *d_x, int64_t n_rows, int n_ch, int64_t win, double *d_rms) {
    ADN_REQUIRE(c && d_x && d_rms, "null pointer argument");
    ADN_REQUIRE(n_rows >= 1 && n_ch >= 1 && win >= 1, "n_rows, n_ch and win must be >= 1");
    DeviceGuard dg(c->device);
    const int64_t n_win = ceil_div(n_rows, win);
    ADN_LAUNCH((k_window_rms<T>), dim3((unsigned)ceil_div(n_win, 4)), dim3(256), 0,
                       c->stream, d_x, n_rows, n_ch, win, n_win, d_rms);
    ADN_HIP(hipGetLastError());
    return ADN_OK;
}

// ------------------------------------------------------------ PCM samples
// q[i] = PCM value of x[i]; xq[i] (optional, may alias x) = the value a reader of the file
// gets back, q / 2^(bits-1).  16-bit values are stored as int16, 24- and 32-bit ones as int32.
template <typename T, typename Q>
__global__ __launch_bounds__(256) void k_pcm_encode(const T *x, int64_t n, int bits,
                                                    Q *__restrict__ q, T *xq) {
    constexpr int V = 16 / (int)sizeof(T);                 // 16-byte loads
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * V;
    if (i0 >= n) return;
    const int shift = 32 - bits;
    const T inv = (T)1 / (T)(1u << (bits - 1));
    if (i0 + V <= n) {
        T v[V];
        Q o[V];
        *reinterpret_cast<float4 *>(v) = *reinterpret_cast<const float4 *>(x + i0);
#pragma unroll
        for (int u = 0; u < V; ++u) {
            const int k = pcm_int32(v[u]) >> shift;
            o[u] = (Q)k;
            v[u] = (T)k * inv;
        }
        if (sizeof(Q) * V == 8) *reinterpret_cast<float2 *>(q + i0) = *reinterpret_cast<float2 *>(o);
        else if (sizeof(Q) * V == 4) *reinterpret_cast<float *>(q + i0) = *reinterpret_cast<float *>(o);
        else *reinterpret_cast<float4 *>(q + i0) = *reinterpret_cast<float4 *>(o);
        if (xq) *reinterpret_cast<float4 *>(xq + i0) = *reinterpret_cast<float4 *>(v);
    } else {
        for (int64_t i = i0; i < n; ++i) {
            const int k = pcm_int32(x[i]) >> shift;
            q[i] = (Q)k;
            if (xq) xq[i] = (T)k * inv;
        }
    }
}

template <typename T>
int pcm_encode(Ctx *c, const T *d_x, int64_t n, int bits, void *d_q, T *d_xq) {
    ADN_REQUIRE(c, "null context");
    ADN_REQUIRE(n >= 0, "n must be >= 0");
    ADN_REQUIRE(bits == 16 || bits == 24 || bits == 32, "bits must be 16, 24 or 32");
    if (n == 0) return ADN_OK;
    ADN_REQUIRE(d_x && d_q, "null pointer argument");
    ADN_REQUIRE(((uintptr_t)d_x & 15) == 0 && ((uintptr_t)d_q & 15) == 0 && ((uintptr_t)d_xq & 15) == 0,
                "sample buffers must be 16-byte aligned");
    DeviceGuard dg(c->device);
    constexpr int V = 16 / (int)sizeof(T);
    const unsigned nb = (unsigned)ceil_div(n, (int64_t)256 * V);
    if (bits == 16)
        ADN_LAUNCH((k_pcm_encode<T, int16_t>), dim3(nb), dim3(256), 0, c->stream, d_x, n, bits,
                           static_cast<int16_t *>(d_q), d_xq);
    else
        ADN_LAUNCH((k_pcm_encode<T, int32_t>), dim3(nb), dim3(256), 0, c->stream, d_x, n, bits,
                           static_cast<int32_t *>(d_q), d_xq);
    ADN_HIP(hipGetLastError());
    return ADN_OK;
}

// ------------------------------------------------------ synthetic signal
// Every operation below is an exactly rounded IEEE double operation and FP
// contraction is off, so the host generator (same formulas) produces the same bits.
#pragma clang fp contract(off)

__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z ^= z >> 30;
    z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27;
    z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}

__device__ __forceinline__ double psin(double p) {
#pragma clang fp contract(off)
    const double q = p - floor(p + 0.5);
    return q * (8.0 - 16.0 * fabs(q));
}

template <typename T>
__global__ __launch_bounds__(256) void k_synth(T *__restrict__ x, int64_t n, int64_t start,
                                               double sr, uint64_t base, double sigma,
                                               double speech_gain) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t idx = (uint64_t)(start + i);
    const uint64_t h = splitmix64(idx * 0xA0761D6478BD642Full + base);
    const double s4 = (double)((h & 0xFFFF) + ((h >> 16) & 0xFFFF) + ((h >> 32) & 0xFFFF) + ((h >> 48) & 0xFFFF));
    const double noise = (s4 - 131070.0) * (1.7320508075688772 / 65536.0);
    const double t = (double)idx / sr;
    const double phase = 140.0 * t + (20.0 / 3.0) * psin(0.5 * t) * 0.5;
    double r = 3.0 * t;
    r = r - floor(r);
    const double ramp = fmin(r, 0.5 - r) / 0.03;
    const double env = fmin(fmax(ramp, 0.0), 1.0);
    const double s = 0.25 * psin(phase) + 0.12 * psin(2.0 * phase) + 0.06 * psin(3.0 * phase);
    const double sp = env * s;
    const double v = speech_gain * sp + sigma * noise;
    x[i] = (T)v;
}

template <typename T>
int synth_noisy_speech(Ctx *c, T *d_x, int64_t n, int64_t start, int sample_rate, uint64_t seed,
                       uint64_t stream, double sigma, double speech_gain) {
    ADN_REQUIRE(c && d_x, "null pointer argument");
    ADN_REQUIRE(n >= 0 && start >= 0 && sample_rate > 0, "n, start must be >= 0 and sample_rate > 0");
    if (n == 0) return ADN_OK;
    DeviceGuard dg(c->device);
    const uint64_t base = seed * 0x9E3779B97F4A7C15ull + stream * 0xD1B54A32D192ED03ull;
    ADN_LAUNCH((k_synth<T>), dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, c->stream, d_x,
                       n, start, (double)sample_rate, base, sigma, speech_gain);
    ADN_HIP(hipGetLastError());
    return ADN_OK;
}

template int pcm_encode<float>(Ctx *, const float *, int64_t, int, void *, float *);
template int pcm_encode<double>(Ctx *, const double *, int64_t, int, void *, double *);
template int window_rms<float>(Ctx *, const float *, int64_t, int, int64_t, double *);
template int window_rms<double>(Ctx *, const double *, int64_t, int, int64_t, double *);
template int synth_noisy_speech<float>(Ctx *, float *, int64_t, int64_t, int, uint64_t, uint64_t, double, double);
template int synth_noisy_speech<double>(Ctx *, double *, int64_t, int64_t, int, uint64_t, uint64_t, double, double);

}  // namespace adn
