// Declarations shared by the generic (adn_stft.hip) and register-resident
// (adn_stft_fast.hip) STFT kernels.
#pragma once
#include "adn_kernels.hpp"

namespace adn {

template <typename T> struct Vec2;
template <> struct Vec2<float> { using type = float2; };
template <> struct Vec2<double> { using type = double2; };

template <typename V> __device__ __forceinline__ V cadd(V a, V b) { return V{a.x + b.x, a.y + b.y}; }
template <typename V> __device__ __forceinline__ V csub(V a, V b) { return V{a.x - b.x, a.y - b.y}; }
template <typename V> __device__ __forceinline__ V cmul(V a, V b) {
    return V{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
template <typename V> __device__ __forceinline__ V cmulc(V a, V b) {   // a * conj(b)
    return V{a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y};
}
template <typename V> __device__ __forceinline__ V conj(V a) { return V{a.x, -a.y}; }

// Orders LDS traffic between lanes of one wave (the hardware issues a wave's LDS
// instructions in order; this only stops the compiler from moving them).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every
// outstanding global store to be acknowledged; the sliding kernels store finished output
// samples right before a barrier and never read them back, so they wait for LDS alone and
// let the stores drain in the background.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ int brev(int k, int logn) { return (int)(__brev((unsigned)k) >> (32 - logn)); }

template <typename T>
struct Tables {
    const T *window;
    const typename Vec2<T>::type *tw;
    const typename Vec2<T>::type *rtw;
    const typename Vec2<T>::type *fast_tw;   // n_fft 2048 only: LDS image of the register-resident transform's twiddles
    double wsum;
    int n_fft, logc;   // logc = log2(N/2)
};

// ---------------------------------------------------------- signal access
struct Signals {
    const int64_t *off;
    const int64_t *len;
    int n_ch;
    int n_fft, hop, padded;
};

__host__ __device__ inline int64_t stft_frames(int64_t n, int n_fft, int hop, int padded) {
    if (n < n_fft) return 0;
    int64_t nadd = 0;
    if (padded) {
        const int64_t r = (n - n_fft) % hop;
        nadd = ((hop - r) % hop) % n_fft;
    }
    return 1 + (n + nadd - n_fft) / hop;
}

// Sample p of the zero-padded, mono-folded signal (mean over channels like np.mean(axis=1)).
template <typename T>
__device__ __forceinline__ T fold_sample(const T *__restrict__ x, int64_t p, int64_t n, int n_ch) {
    if (p >= n) return (T)0;
    if (n_ch == 1) return x[p];
    T s = 0;
    for (int c = 0; c < n_ch; ++c) s += x[p * n_ch + c];
    return s / (T)n_ch;
}

enum { APPLY_SUBTRACT = 0, APPLY_GAIN = 1 };

template <typename T>
struct ApplyArgs {
    const T *x;
    T *y;
    const int64_t *y_off;
    Signals sg;
    Tables<T> tb;
    const T *row;        // noise magnitude (SUBTRACT) or gain (GAIN), F per signal
    int64_t row_stride;  // elements between signals (0 = shared)
    int64_t row_t_stride;  // GAIN mode: elements between the gain rows of consecutive frames (0 = one row for all)
    int16_t *pcm;        // optional: 16-bit PCM values of the output (same offsets as y); y then holds pcm / 32768
    int win_lds;         // shape-generic kernel: the window table is staged in LDS (set by the launcher)
    int no_clip;         // 1: the output is not clipped to [-1, 1] (HPSS writes the ISTFT as it is)
    T alpha, beta;
    int st, sf;          // box sizes along time / frequency
    int64_t segs_per_sig;
    int64_t n_sig;       // signals of the batch (kernels whose grid is not a multiple of it)
    int seg_hops;        // output hops per workgroup (or per wave)
    int ablate;          // diagnostics: bit 0 skips the analysis, 1 the synthesis, 2 the overlap-add
    T *dummy;            // 512 writable elements nobody reads (target of stores a kernel issues unconditionally)
    unsigned long long *dbg;   // diagnostics (ablate bit 4): per-phase shader-clock totals of wave 0
    const int *wg_start;       // one-wave-per-run kernels, optional: first workgroup of every signal, their total, hops per run (n_sig + 2):
                               // signals then take the workgroups their own length needs instead of segs_per_sig / W each
};


// Workspace the fused apply kernels take beyond what their callers carve out themselves; callers
// add it to their ws_reserve request.
static constexpr int APPLY_MAP_MAX = 4096;                 // signals of a batch whose workgroups are mapped by their own lengths
static constexpr size_t APPLY_WS_BYTES = 4096 + (((size_t)(APPLY_MAP_MAX + 2) * sizeof(int) + 255) & ~(size_t)255);

// register-resident path (binary32, n_fft = 2048)
bool apply_fast_ok(int n_fft, int hop, int st, int sf);
// the same kernels at binary64 (hop 512 only); the caller reserves APPLY_WS_BYTES beyond what it took
bool apply_wave64_ok(int n_fft, int hop, int st, int sf, int mode);
template <int MODE>
int apply_wave64(Ctx *c, ApplyArgs<double> a, int64_t n_sig, int64_t Tmax);
// binary32 rows in groups of four frames (see k_rows_wave); the caller reserves APPLY_WS_BYTES beyond what it took
// 16-byte slots per group of four frames: the 1025 bins padded to a multiple of eight, so that the eight rows
// whose pieces share a 128-byte line are the same eight rows in every group
constexpr int64_t ROWS_GROUP_STRIDE = 1032;
int64_t rows_wave_sig_floats(int64_t Tmax);                // floats of one signal's rows
bool rows_wave_ok(const Signals &sg, int n_fft);
// vad_e (may be null): n_sig x nv_max binary64 sums, zero on entry, receive the energy of every voice-activity frame
// of frame_len samples (512 <= frame_len <= rows_wave_vad_max_frame())
int64_t rows_wave_vad_max_frame();
template <typename R>
int rows_wave(Ctx *c, const R *d_x, const Signals &sg, const Tables<R> &tb, int64_t n_sig, int64_t Tmax,
              int power, R *S, int64_t sig_stride, double *vad_e, int64_t frame_len, int64_t nv_max,
              unsigned long long *sig_max = nullptr);    // sig_max (may be null): per signal, atomic maximum of the keys of the values
template <int MODE>
int apply_fast(Ctx *c, ApplyArgs<float> a, int64_t n_sig, int64_t Tmax);
int rows_fast(Ctx *c, const float *d_x, const Signals &sg, const Tables<float> &tb, int64_t n_sig,
              int64_t Tmax, int power, float *S, unsigned long long *sig_max);

}  // namespace adn
