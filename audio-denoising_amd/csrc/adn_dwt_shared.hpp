// Definitions shared by the wavelet translation units (adn_dwt.hip and adn_dwt_wres.hip): block geometry,
// pywt.threshold, filter taps in the operand forms of the two-wide multiply-adds.
#pragma once
#include <cstddef>
#include "adn_kernels.hpp"

namespace adn {

struct BlockGeom {
    const int64_t *in_off;
    const int64_t *frames;
    const int64_t *out_off;
    int64_t n_blocks;
    int64_t a_stride;   // elements per block in each approximation ping-pong buffer
    int64_t d_stride;   // elements per block in the detail region
};

template <typename T>
__device__ __forceinline__ T shrink(T d, T t, int thr_mode) {
    if (thr_mode == ADN_THR_SOFT) {
        T f = (T)1 - t / fabs(d);
        f = f < (T)0 ? (T)0 : f;   // keeps NaN (0/0), maps -inf (t/0) to 0
        return d * f;
    }
    return fabs(d) < t ? (T)0 : d;
}

// How the wave-resident tiles (adn_dwt_wres.hip) cover a block of n >= window samples: windows start at
// 0, advance, 2 advance, ... while they fit the block; the i-th stores samples [i advance + lead, (i + 1) advance + lead)
// and owns the coefficients above them.  window == 0: no such tiles.
// A block whose length is a multiple of `run` (the samples a lane holds) and at least one window is covered by the
// wave-resident tiles alone: their windows go round it.
struct WresPlan {
    int window, advance, lead, run;
};
__host__ __device__ inline bool wres_ring(int64_t n, int window, int run) { return n >= window && n % run == 0; }
// Any other block of n >= window samples: windows at 0, advance, ... while they fit, and one more that ends within a
// run of the block's end (at the largest multiple of `run` that leaves it inside the block; it owns what the others have
// not).  The LDS tiles take samples [0, lead) and [wres_tail(n, p), n).
__host__ __device__ inline int64_t wres_last_window(int64_t n, const WresPlan &p) { return (n - p.window) / p.run * p.run; }
__host__ __device__ inline int64_t wres_tail(int64_t n, const WresPlan &p) { return wres_last_window(n, p) + p.lead + p.advance; }

struct M1State {                  // per block
    unsigned long long lo_m, hi_m;    // key bracket of the median (lo > hi: unusable)
    double mhat;                      // centre estimate, a value of the block's dtype
    double alo, ahi;                  // value bracket of |d - mhat| (alo < 0: open below, ahi = inf: open above)
    unsigned long long below_m, below_a;
    unsigned int n_m, n_a;            // keys gathered (may exceed the capacity: overflow)
    int has_nan, usable;
    int spill;                        // a tile held more bracket keys than its LDS stage: lists incomplete
    int tie_m, tie_a;                 // lo_m == hi_m (alo == ahi): the keys equal to it are counted, not gathered
};

static_assert(offsetof(M1State, n_m) % 8 == 0 && offsetof(M1State, n_a) == offsetof(M1State, n_m) + 4 && sizeof(M1State) % 8 == 0,
              "n_m and n_a form one aligned 64-bit word");
// the two list lengths as one word: n_m in the low half, n_a in the high half
__device__ __forceinline__ unsigned long long *m1_counts(M1State *st) { return reinterpret_cast<unsigned long long *>(&st->n_m); }

template <typename T>
__device__ __forceinline__ typename KeyT<T>::type m1_key(T x) {
    using K = typename KeyT<T>::type;
    return x != x ? ~(K)0 : key_of(x);
}

// Where the single-pass statistics keep a batch's states and candidate lists (see adn_dwt_mad1p.hpp); the lists hold
// values of the batch's precision
struct WresGather {
    M1State *st;
    void *cand_m;
    int64_t cap_m;
    void *cand_a;
    int64_t cap_a;
};

namespace tiles {

enum { MODE_STATS = 0, MODE_APPLY = 1 };

template <typename T> struct Pk;
template <> struct Pk<float> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct Pk<double> { typedef double type __attribute__((ext_vector_type(2))); };

template <typename T, int F>
struct TapsF {
    T h[F], g[F], rl[F], rh[F];
    // the same taps as operand pairs of the two-wide multiply-adds:
    // hg[j] = (h[j], g[j]); rlp[q] = (rl[2q+1], rl[2q]); rhp[q] = (rh[2q+1], rh[2q])
    typename Pk<T>::type hg[F], rlp[F / 2], rhp[F / 2];
};

template <typename T, int F>
static TapsF<T, F> make_taps_f(int wavelet) {
    const FilterBank &b = filter_bank(wavelet);
    TapsF<T, F> t{};
    for (int j = 0; j < F; ++j) {
        t.h[j] = (T)b.h[j];
        t.g[j] = (T)b.g[j];
        t.rl[j] = (T)b.rl[j];
        t.rh[j] = (T)b.rh[j];
        t.hg[j].x = (T)b.h[j];
        t.hg[j].y = (T)b.g[j];
    }
    for (int q = 0; q < F / 2; ++q) {
        t.rlp[q].x = (T)b.rl[2 * q + 1];
        t.rlp[q].y = (T)b.rl[2 * q];
        t.rhp[q].x = (T)b.rh[2 * q + 1];
        t.rhp[q].y = (T)b.rh[2 * q];
    }
    return t;
}

// pywt.threshold with the fp32 division replaced by the hardware reciprocal (1 ulp); keeps the
// NaN / inf behaviour of the exact form (0/0 -> NaN, t/0 -> inf -> factor 0)
__device__ __forceinline__ float shrink_fast(float d, float t, int thr_mode) {
    if (thr_mode == ADN_THR_SOFT) {
        float f = 1.0f - t * __builtin_amdgcn_rcpf(fabsf(d));
        f = f < 0.0f ? 0.0f : f;
        return d * f;
    }
    return fabsf(d) < t ? 0.0f : d;
}
__device__ __forceinline__ double shrink_fast(double d, double t, int thr_mode) { return shrink(d, t, thr_mode); }

// acc + coef * (q.x, q.x) (HALF = 0) or coef * (q.y, q.y) (HALF = 1): the two-wide fused multiply-add with one half
// of its second operand selected for both results
// (half is a constant after unrolling: the branch folds away)
__device__ __forceinline__ Pk<float>::type pk_fma_half(int half, Pk<float>::type coef, Pk<float>::type q, Pk<float>::type acc) {
    Pk<float>::type r;
    if (half == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "s"(coef), "v"(q), "v"(acc));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(r) : "s"(coef), "v"(q), "v"(acc));
    return r;
}
__device__ __forceinline__ Pk<double>::type pk_fma_half(int half, Pk<double>::type coef, Pk<double>::type q, Pk<double>::type acc) {
    const double x = half ? q.y : q.x;
    return __builtin_elementwise_fma(coef, Pk<double>::type{x, x}, acc);
}

}  // namespace tiles

int wres_tiles(Ctx *c, int F, int TL, int mode, const float *x, float *y, const BlockGeom &g, int64_t max_frames, int wavelet,
               const double *thresh, int thr_mode, int64_t *zeroed, float *d1, int64_t d1_stride, float *top, int64_t top_stride,
               int use_top, WresPlan *plan);
int wres_tiles(Ctx *c, int F, int TL, int mode, const double *x, double *y, const BlockGeom &g, int64_t max_frames, int wavelet,
               const double *thresh, int thr_mode, int64_t *zeroed, double *d1, int64_t d1_stride, double *top, int64_t top_stride,
               int use_top, WresPlan *plan);

// The gather pass of the single-pass statistics (finest-band coefficients against the blocks' brackets) for the
// coefficients the windows of TL levels own (`*plan` says which: all of a block with wres_ring(frames, plan->window,
// plan->run), else those above samples [lead, n_windows advance + lead) of a block that holds a window); when
// top != null the level-TL approximation of every window is written as by the statistics mode of wres_tiles.
int wres_gather(Ctx *c, int F, int TL, const float *x, const BlockGeom &g, int64_t max_frames, int wavelet, WresGather gat,
                float *top, int64_t top_stride, WresPlan *plan);
int wres_gather(Ctx *c, int F, int TL, const double *x, const BlockGeom &g, int64_t max_frames, int wavelet, WresGather gat,
                double *top, int64_t top_stride, WresPlan *plan);

}  // namespace adn
