// Periodized / symmetric Daubechies filter banks and the VisuShrink block denoiser.
//
// Reference behaviour restated here (not its code): pywt.wavedec / pywt.threshold /
// pywt.waverec as called from denoise.py:77-89, and mad() of denoise.py:16-23.
//
// Two execution paths for mono blocks (transform along time):
//   * level-by-level kernels: any block length, any level, both extension modes;
//     every level round-trips through the workspace.
//   * fused tile kernels (adn_dwt_tiles.hpp): periodization, any block length; up to
//     four levels stay in LDS per tile (deeper levels run on the level-4 approximation
//     in one workgroup per block): one read of the block for the statistics and one
//     read + one write for analysis, shrink and synthesis.  Used by wavelet_denoise.
// Multi-channel blocks are transformed across the channel axis of every frame, as
// the reference does (pywt's default axis is the last one); that per-frame
// transform is a tiny linear map, built on the host and applied from registers.
#include <math.h>

#include <vector>

#include "adn_kernels.hpp"
#include "adn_dwt_shared.hpp"

namespace adn {

// ------------------------------------------------------------------ taps
static const double DB4_LO[8] = {-0.010597401785069032, 0.0328830116668852, 0.030841381835560764,
                                 -0.18703481171909309,  -0.027983769416859854,
                                 0.6308807679298589,    0.7148465705529157,
                                 0.2303778133088965};
static const double DB8_LO[16] = {
    -0.00011747678412476953, 0.0006754494064505693, -0.00039174037337694705,
    -0.004870352993451574,   0.008746094047405777,  0.013981027917398282,
    -0.044088253930794755,   -0.017369301001807547, 0.12874742662047847,
    0.0004724845739132828,   -0.2840155429615469,   -0.015829105256349306,
    0.5853546836542067,      0.6756307362972898,    0.31287159091429995,
    0.05441584224310401};

static FilterBank make_bank(const double *lo, int F) {
    FilterBank b{};
    b.F = F;
    for (int j = 0; j < F; ++j) {
        b.h[j] = lo[j];
        b.g[j] = ((j % 2 == 0) ? -1.0 : 1.0) * lo[F - 1 - j];
    }
    for (int j = 0; j < F; ++j) {
        b.rl[j] = b.h[F - 1 - j];
        b.rh[j] = b.g[F - 1 - j];
    }
    return b;
}

const FilterBank &filter_bank(int wavelet) {
    static const FilterBank db4 = make_bank(DB4_LO, 8);
    static const FilterBank db8 = make_bank(DB8_LO, 16);
    return wavelet == ADN_WAVELET_DB8 ? db8 : db4;
}

template <typename T>
struct Taps {
    int F;
    T h[16], g[16], rl[16], rh[16];
};

template <typename T>
static Taps<T> make_taps(int wavelet) {
    const FilterBank &b = filter_bank(wavelet);
    Taps<T> t{};
    t.F = b.F;
    for (int j = 0; j < b.F; ++j) {
        t.h[j] = (T)b.h[j];
        t.g[j] = (T)b.g[j];
        t.rl[j] = (T)b.rl[j];
        t.rh[j] = (T)b.rh[j];
    }
    return t;
}

// -------------------------------------------------- per-block geometry
// Length of the level-l approximation/detail band of an n-sample block.
__host__ __device__ inline int64_t band_len(int64_t n, int l, int F, int ext) {
    for (int i = 0; i < l; ++i) n = ext == ADN_EXT_PERIODIZATION ? (n + 1) / 2 : (n + F - 1) / 2;
    return n;
}
// Offset of detail band l (1 = finest) inside a block's detail region.
__host__ __device__ inline int64_t band_off(int64_t n, int l, int F, int ext) {
    int64_t off = 0;
    for (int i = 1; i < l; ++i) off += band_len(n, i, F, ext);
    return off;
}


// ------------------------------------------------ level-by-level kernels
// Analysis of level l for every block.  in: x (l == 1) or A[(l-1)&1]; out: A[l&1], D band l.
template <typename T>
__global__ __launch_bounds__(256) void k_fwd_level(const T *__restrict__ x, T *__restrict__ A0,
                                                   T *__restrict__ A1, T *__restrict__ D,
                                                   BlockGeom g, int l, int ext, Taps<T> tp,
                                                   int64_t tiles_per_block) {
    const int64_t b = blockIdx.x / tiles_per_block;
    const int64_t k = (blockIdx.x % tiles_per_block) * 256 + threadIdx.x;
    const int64_t n0 = g.frames[b];
    const int F = tp.F;
    const int64_t N = band_len(n0, l - 1, F, ext);   // input length
    const int64_t K = band_len(n0, l, F, ext);       // output length
    if (k >= K || N <= 0) return;
    const T *in = l == 1 ? x + g.in_off[b] : ((l - 1) & 1 ? A1 : A0) + b * g.a_stride;
    T *cA = (l & 1 ? A1 : A0) + b * g.a_stride;
    T *cD = D + b * g.d_stride + band_off(n0, l, F, ext);
    T sa = 0, sd = 0;
    if (ext == ADN_EXT_PERIODIZATION) {
        // block lengths fit 32 bits: one 32-bit modulo, then the window walks backwards with wrap
        const int Mi = (int)(N + (N & 1)), Ni = (int)N;
        int i = (int)((2 * k + F / 2) % Mi);
        for (int j = 0; j < F; ++j) {
            const T v = in[i < Ni ? i : Ni - 1];
            sa = fma(tp.h[j], v, sa);
            sd = fma(tp.g[j], v, sd);
            i = i == 0 ? Mi - 1 : i - 1;
        }
    } else {
        for (int j = 0; j < F; ++j) {
            int64_t i = (2 * k + 1 - j) % (2 * N);
            if (i < 0) i += 2 * N;
            if (i >= N) i = 2 * N - 1 - i;
            const T v = in[i];
            sa = fma(tp.h[j], v, sa);
            sd = fma(tp.g[j], v, sd);
        }
    }
    cA[k] = sa;
    cD[k] = sd;
}


// Synthesis of level l for every block, with the detail band shrunk on the fly.
// a: A[l&1] (cA_L when l == L, else the reconstruction), d: D band l.
// out: y (l == 1) or A[(l-1)&1].  thresh may be null (plain waverec).
template <typename T>
__global__ __launch_bounds__(256) void k_inv_level(T *__restrict__ y, T *__restrict__ A0,
                                                   T *__restrict__ A1, const T *__restrict__ D,
                                                   BlockGeom g, int l, int ext, Taps<T> tp,
                                                   int64_t tiles_per_block,
                                                   const double *__restrict__ thresh,
                                                   int thr_mode, int64_t *__restrict__ zeroed) {
    const int64_t b = blockIdx.x / tiles_per_block;
    const int64_t n = (blockIdx.x % tiles_per_block) * 256 + threadIdx.x;
    const int64_t n0 = g.frames[b];
    const int F = tp.F;
    const int64_t K = band_len(n0, l, F, ext);
    const int64_t n_out = ext == ADN_EXT_PERIODIZATION ? 2 * K : 2 * K - F + 2;
    const bool live = K > 0 && n < n_out;
    const T *a = (l & 1 ? A1 : A0) + b * g.a_stride;
    const T *d = D + b * g.d_stride + band_off(n0, l, F, ext);
    T *out = l == 1 ? y + g.out_off[b] : ((l - 1) & 1 ? A1 : A0) + b * g.a_stride;
    const bool do_thr = thresh != nullptr;
    const T t = do_thr ? (T)thresh[b] : (T)0;
    T s = 0;
    if (live) {
        if (ext == ADN_EXT_PERIODIZATION) {
            // y[n] = sum over the taps j of n's parity class; k walks backwards with wrap
            const int Ki = (int)K;
            const int par = (int)((n + F / 2 - 1) & 1);
            int k = (int)(((n + F / 2 - 1 - par) >> 1) % Ki);
            for (int j = par; j < F; j += 2) {
                T dv = d[k];
                if (do_thr) dv = shrink(dv, t, thr_mode);
                s = fma(tp.rl[j], a[k], s);
                s = fma(tp.rh[j], dv, s);
                k = k == 0 ? Ki - 1 : k - 1;
            }
        } else {
            const int64_t m = n + F - 2;
            for (int j = 0; j < F; ++j) {
                const int64_t u = m - j;
                if (u < 0 || (u & 1) || (u >> 1) >= K) continue;
                T dv = d[u >> 1];
                if (do_thr) dv = shrink(dv, t, thr_mode);
                s = fma(tp.rl[j], a[u >> 1], s);
                s = fma(tp.rh[j], dv, s);
            }
        }
        out[n] = s;
    }
    if (zeroed != nullptr && do_thr) {
        // every detail coefficient is counted once: by the thread whose index equals k
        const bool z = K > 0 && n < K && fabs(d[n]) < t;
        const unsigned long long m = __ballot(z);
        if ((threadIdx.x & 63) == 0 && m) atomicAdd((unsigned long long *)&zeroed[b], (unsigned long long)__popcll(m));
    }
}

// Segment descriptors of the finest detail band and the universal threshold.
// per_frame == 0: the band of a mono block, ceil(frames/2) values (periodization);
// per_frame > 0: per_frame finest-band values for every frame (channel-axis transform).
__global__ void k_cd1_segments(BlockGeom g, int F, int ext, int64_t per_frame, int64_t *seg_off,
                               int64_t *seg_len) {
    const int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (b >= g.n_blocks) return;
    seg_off[b] = b * g.d_stride;
    seg_len[b] = per_frame > 0 ? g.frames[b] * per_frame : band_len(g.frames[b], 1, F, ext);
}

__global__ void k_universal_threshold(const int64_t *frames, int64_t n_blocks, const double *sigma,
                                      double *thresh, double *sigma_out, double *thresh_out) {
    const int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    // thresh = sigma * sqrt(2 ln(frames)); frames == 1 gives 0, frames == 0 cannot occur
    const double t = sigma[b] * sqrt(2.0 * log((double)frames[b]));
    thresh[b] = t;
    if (sigma_out) sigma_out[b] = sigma[b];
    if (thresh_out) thresh_out[b] = t;
}

// ------------------------------------------------ channel-axis transform
// For a (frames, n_ch) block the reference transforms each frame's n_ch samples.
// That map is linear: coefficients = W x (n_coef x n_ch), frame = V coefficients.
static constexpr int CH_MAXC = 16;
template <typename T>
struct ChannelMap {
    int n_ch, n_coef, n_approx, n_d1;   // detail coefficients are [n_approx, n_coef); cD1 the last n_d1
    T W[CH_MAXC][8];
    T V[8][CH_MAXC];
};

static std::vector<double> host_dwt_per(const std::vector<double> &x, const FilterBank &fb,
                                        std::vector<double> *cD) {
    const int64_t N = (int64_t)x.size(), M = N + (N & 1), K = M / 2;
    std::vector<double> cA(K, 0.0);
    cD->assign(K, 0.0);
    for (int64_t k = 0; k < K; ++k)
        for (int j = 0; j < fb.F; ++j) {
            int64_t i = ((2 * k + fb.F / 2 - j) % M + M) % M;
            const double v = x[i < N ? i : N - 1];
            cA[k] += fb.h[j] * v;
            (*cD)[k] += fb.g[j] * v;
        }
    return cA;
}

static std::vector<double> host_idwt_per(const std::vector<double> &a, const std::vector<double> &d,
                                         const FilterBank &fb) {
    const int64_t K = (int64_t)d.size();
    std::vector<double> y(2 * K, 0.0);
    for (int64_t k = 0; k < K; ++k)
        for (int j = 0; j < fb.F; ++j) {
            int64_t n = ((j + 2 * k - (fb.F / 2 - 1)) % (2 * K) + 2 * K) % (2 * K);
            y[n] += fb.rl[j] * a[k] + fb.rh[j] * d[k];
        }
    return y;
}

template <typename T>
static int build_channel_map(int n_ch, int wavelet, int level, ChannelMap<T> *out) {
    const FilterBank &fb = filter_bank(wavelet);
    std::vector<int64_t> lens;   // detail lengths, level 1 first
    int64_t n = n_ch;
    for (int l = 0; l < level; ++l) {
        n = (n + 1) / 2;
        lens.push_back(n);
    }
    int64_t n_coef = lens.back();
    for (auto v : lens) n_coef += v;
    if (n_coef > CH_MAXC) {
        set_error("channel-axis transform with %d channels and level %d needs %lld coefficients (max %d)",
                  n_ch, level, (long long)n_coef, CH_MAXC);
        return ADN_ERR_UNSUPPORTED;
    }
    ChannelMap<T> m{};
    m.n_ch = n_ch;
    m.n_coef = (int)n_coef;
    m.n_approx = (int)lens.back();
    m.n_d1 = (int)lens[0];
    // coefficient order: [cA_L, cD_L, ..., cD_1]
    auto analyse = [&](const std::vector<double> &x) {
        std::vector<std::vector<double>> det;
        std::vector<double> a = x;
        for (int l = 0; l < level; ++l) {
            std::vector<double> d;
            a = host_dwt_per(a, fb, &d);
            det.push_back(d);
        }
        std::vector<double> c = a;
        for (int l = level - 1; l >= 0; --l) c.insert(c.end(), det[l].begin(), det[l].end());
        return c;
    };
    auto synthesise = [&](const std::vector<double> &c) {
        size_t pos = 0;
        std::vector<double> a(c.begin(), c.begin() + lens.back());
        pos = lens.back();
        for (int l = level - 1; l >= 0; --l) {
            std::vector<double> d(c.begin() + pos, c.begin() + pos + lens[l]);
            pos += lens[l];
            if (a.size() == d.size() + 1) a.pop_back();
            a = host_idwt_per(a, d, fb);
        }
        return a;
    };
    for (int i = 0; i < n_ch; ++i) {
        std::vector<double> e(n_ch, 0.0);
        e[i] = 1.0;
        std::vector<double> c = analyse(e);
        for (int k = 0; k < m.n_coef; ++k) m.W[k][i] = (T)c[k];
    }
    for (int k = 0; k < m.n_coef; ++k) {
        std::vector<double> e(m.n_coef, 0.0);
        e[k] = 1.0;
        std::vector<double> yv = synthesise(e);
        if ((int)yv.size() != n_ch) {
            set_error("channel-axis transform of %d channels returns %d channels; the reference cannot write that block",
                      n_ch, (int)yv.size());
            return ADN_ERR_UNSUPPORTED;
        }
        for (int i = 0; i < n_ch; ++i) m.V[i][k] = (T)yv[i];
    }
    *out = m;
    return ADN_OK;
}

// Frames per thread of the channel-axis kernels: as many as one 16-byte access covers.
template <typename T, int NCH>
struct ChVec {
    static constexpr int FPT = (16 / (int)sizeof(T)) / NCH > 0 ? (16 / (int)sizeof(T)) / NCH : 1;
    static constexpr bool one_access = FPT * NCH * (int)sizeof(T) == 16;
};

// Loads FPT consecutive frames (FPT * NCH values).  When they are exactly 16 bytes and the
// address is 16-byte aligned (uniform per block) that is ONE memory instruction.
template <typename T, int NCH>
__device__ __forceinline__ void ch_load(const T *__restrict__ p, T (&v)[ChVec<T, NCH>::FPT * NCH], bool vec, int nf) {
    constexpr int FPT = ChVec<T, NCH>::FPT, VE = FPT * NCH;
    if (ChVec<T, NCH>::one_access && vec && nf == FPT) {
        const float4 q = *reinterpret_cast<const float4 *>(p);
        const T *e = reinterpret_cast<const T *>(&q);
#pragma unroll
        for (int i = 0; i < VE; ++i) v[i] = e[i];
    } else {
#pragma unroll
        for (int i = 0; i < VE; ++i) v[i] = i < nf * NCH ? p[i] : (T)0;
    }
}
template <typename T, int NCH>
__device__ __forceinline__ void ch_store(T *__restrict__ p, const T (&v)[ChVec<T, NCH>::FPT * NCH], bool vec, int nf) {
    constexpr int FPT = ChVec<T, NCH>::FPT, VE = FPT * NCH;
    if (ChVec<T, NCH>::one_access && vec && nf == FPT) {
        float4 q;
        T *e = reinterpret_cast<T *>(&q);
#pragma unroll
        for (int i = 0; i < VE; ++i) e[i] = v[i];
        *reinterpret_cast<float4 *>(p) = q;
    } else {
#pragma unroll
        for (int i = 0; i < VE; ++i)
            if (i < nf * NCH) p[i] = v[i];
    }
}

// Chunks (16-byte accesses) per thread of the channel-axis kernels: the per-wave set-up (block geometry,
// threshold, map entries: a chain of dependent scalar loads) is paid once per CH_CPT chunks, and a tile
// that lies inside its block issues all its loads before the first use.
constexpr int CH_CPT = 1;

// Pass 1: finest-band coefficients of every frame -> D (frames * n_d1 values per block).
// NC: number of coefficients per frame when known at compile time (0: m.n_coef, up to CH_MAXC);
// NC != 0 also states n_approx == 1 and n_d1 == 1.
template <typename T, int NCH, int NC>
__global__ __launch_bounds__(256) void k_ch_stats(const T *__restrict__ x, T *__restrict__ D,
                                                  BlockGeom g, ChannelMap<T> m,
                                                  int64_t tiles_per_block) {
#pragma clang fp contract(off)     // products and sums rounded one by one, like the single-pass statistics
    constexpr int FPT = ChVec<T, NCH>::FPT, KMAX = NC ? NC : CH_MAXC;
    const int n_coef = NC ? NC : m.n_coef, n_d1 = NC ? 1 : m.n_d1;
    const int64_t b = blockIdx.x / tiles_per_block;
    const int64_t c0 = (blockIdx.x % tiles_per_block) * (256 * CH_CPT) + threadIdx.x;
    const int64_t frames = g.frames[b];
    const T *in = x + g.in_off[b];
    const bool vec = (reinterpret_cast<uintptr_t>(in) & 15) == 0;
    T *dblk = D + b * g.d_stride;
    T v[CH_CPT][FPT * NCH];
    int nf[CH_CPT];
#pragma unroll
    for (int j = 0; j < CH_CPT; ++j) {
        const int64_t f0 = (c0 + j * 256) * FPT;
        nf[j] = frames - f0 >= FPT ? FPT : (frames > f0 ? (int)(frames - f0) : 0);
        if (nf[j] > 0) ch_load<T, NCH>(in + f0 * NCH, v[j], vec, nf[j]);
    }
#pragma unroll
    for (int j = 0; j < CH_CPT; ++j) {
        T *dst = dblk + (c0 + j * 256) * FPT * n_d1;
#pragma unroll
        for (int f = 0; f < FPT; ++f) {
            if (f >= nf[j]) break;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                if (k >= n_coef - n_d1 && k < n_coef) {
                    T s = 0;
#pragma unroll
                    for (int i = 0; i < NCH; ++i) s += m.W[k][i] * v[j][f * NCH + i];
                    dst[f * n_d1 + k - (n_coef - n_d1)] = s;
                }
            }
        }
    }
}

// Soft / hard shrink of the channel-axis kernels: the same IEEE division as the mono path.
template <typename T>
__device__ __forceinline__ T shrink_ch(T d, T t, int thr_mode) { return shrink(d, t, thr_mode); }

// Pass 2: analysis, shrink of every detail coefficient, synthesis, per frame.
template <typename T, int NCH, int NC>
__global__ __launch_bounds__(256) void k_ch_apply(const T *__restrict__ x, T *__restrict__ y,
                                                  BlockGeom g, ChannelMap<T> m,
                                                  int64_t tiles_per_block,
                                                  const double *__restrict__ thresh, int thr_mode,
                                                  int64_t *__restrict__ zeroed) {
    constexpr int FPT = ChVec<T, NCH>::FPT, KMAX = NC ? NC : CH_MAXC;
    const int n_coef = NC ? NC : m.n_coef, n_approx = NC ? 1 : m.n_approx;
    const int64_t b = blockIdx.x / tiles_per_block;
    const int64_t c0 = (blockIdx.x % tiles_per_block) * (256 * CH_CPT) + threadIdx.x;
    const int64_t frames = g.frames[b];
    const T t = (T)thresh[b];
    const T *in = x + g.in_off[b];
    T *out = y + g.out_off[b];
    const bool vec_in = (reinterpret_cast<uintptr_t>(in) & 15) == 0;
    const bool vec_out = (reinterpret_cast<uintptr_t>(out) & 15) == 0;
    T v[CH_CPT][FPT * NCH];
    int nf[CH_CPT];
#pragma unroll
    for (int j = 0; j < CH_CPT; ++j) {
        const int64_t f0 = (c0 + j * 256) * FPT;
        nf[j] = frames - f0 >= FPT ? FPT : (frames > f0 ? (int)(frames - f0) : 0);
        if (nf[j] > 0) ch_load<T, NCH>(in + f0 * NCH, v[j], vec_in, nf[j]);
    }
    int nz = 0;
#pragma unroll
    for (int j = 0; j < CH_CPT; ++j) {
        if (nf[j] == 0) continue;
        T o[FPT * NCH];
#pragma unroll
        for (int f = 0; f < FPT; ++f) {
            T c[KMAX];
#pragma unroll
            for (int i = 0; i < NCH; ++i) o[f * NCH + i] = 0;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                T s = 0;
                if (k < n_coef) {
#pragma unroll
                    for (int i = 0; i < NCH; ++i) s += m.W[k][i] * v[j][f * NCH + i];
                    if (k >= n_approx) {
                        nz += (f < nf[j] && fabs(s) < t) ? 1 : 0;
                        s = shrink_ch(s, t, thr_mode);
                    }
                }
                c[k] = s;
            }
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < n_coef) {
#pragma unroll
                    for (int i = 0; i < NCH; ++i) o[f * NCH + i] += m.V[i][k] * c[k];
                }
        }
        ch_store<T, NCH>(out + (c0 + j * 256) * FPT * NCH, o, vec_out, nf[j]);
    }
    if (zeroed != nullptr) {
        for (int d = 32; d >= 1; d >>= 1) nz += __shfl_xor(nz, d);
        if ((threadIdx.x & 63) == 0 && nz) atomicAdd((unsigned long long *)&zeroed[b], (unsigned long long)nz);
    }
}

// ---------------------------------------------------------- host drivers
template <typename T>
static int mad_threshold(Ctx *c, const T *D, const BlockGeom &g, int F, int ext,
                         int64_t per_frame, int64_t max_seg, double *d_sigma_out,
                         double *d_thresh_out, double **thresh_ws) {
    int64_t *seg_off = ws_take<int64_t>(c, g.n_blocks);
    int64_t *seg_len = ws_take<int64_t>(c, g.n_blocks);
    double *med = ws_take<double>(c, g.n_blocks);
    double *sigma = ws_take<double>(c, g.n_blocks);
    double *thresh = ws_take<double>(c, g.n_blocks);
    const unsigned nb = (unsigned)ceil_div(g.n_blocks, 256);
    ADN_LAUNCH(k_cd1_segments, dim3(nb), dim3(256), 0, c->stream, g, F, ext, per_frame,
                       seg_off, seg_len);
    SelectArgs a;
    a.n_seg = g.n_blocks;
    a.d_seg_off = seg_off;
    a.d_seg_len = seg_len;
    a.max_len = max_seg;
    a.mode = SEL_MEDIAN;
    a.d_out = med;
    ADN_TRY(launch_select<T>(c, D, a));
    a.xform = XF_ABSDEV;
    a.d_center = med;
    a.d_out = sigma;
    ADN_TRY(launch_select<T>(c, D, a));
    ADN_LAUNCH(k_universal_threshold, dim3(nb), dim3(256), 0, c->stream, g.frames,
                       g.n_blocks, sigma, thresh, d_sigma_out, d_thresh_out);
    ADN_HIP(hipGetLastError());
    *thresh_ws = thresh;
    return ADN_OK;
}

static size_t mad_ws_bytes(int64_t n_blocks, int64_t max_seg) {
    return 2 * ws_need(n_blocks, sizeof(int64_t)) + 3 * ws_need(n_blocks, sizeof(double)) +
           2 * select_ws_bytes(n_blocks, max_seg);
}

#include "adn_dwt_mad1p.hpp"
#include "adn_dwt_tiles.hpp"

template <typename T, int NCH>
static int channel_denoise(Ctx *c, const T *d_in, T *d_out, BlockGeom g, int64_t max_frames,
                           const ChannelMap<T> &m, int thr_mode, double *d_sigma,
                           double *d_thresh, int64_t *d_zeroed) {
    g.d_stride = max_frames * m.n_d1;
    const int64_t tiles = ceil_div(max_frames, 256 * CH_CPT * ChVec<T, NCH>::FPT);
    dim3 grid((unsigned)(tiles * g.n_blocks)), block(256);
    // stereo at the reference's level 2 has three coefficients per frame: that shape is compiled in
    const bool three = NCH == 2 && m.n_coef == 3 && m.n_approx == 1 && m.n_d1 == 1;
    double *thresh = nullptr;
    if (three && m1_applicable<T>(c, g.n_blocks, max_frames)) {
        // threshold statistics in one pass over the input, no coefficient array
        ADN_TRY(ws_reserve(c, m1_ws_bytes<T>(g.n_blocks, max_frames)));
        ADN_TRY(stereo_mad_single_pass<T>(c, d_in, g, max_frames, m.W[2][0], m.W[2][1], d_sigma, d_thresh, &thresh));
    } else {
        ADN_TRY(ws_reserve(c, ws_need((size_t)g.n_blocks * g.d_stride, sizeof(T)) +
                              mad_ws_bytes(g.n_blocks, max_frames * m.n_d1)));
        T *D = ws_take<T>(c, (size_t)g.n_blocks * g.d_stride);
        if (three) ADN_LAUNCH((k_ch_stats<T, NCH, 3>), grid, block, 0, c->stream, d_in, D, g, m, tiles);
        else ADN_LAUNCH((k_ch_stats<T, NCH, 0>), grid, block, 0, c->stream, d_in, D, g, m, tiles);
        ADN_TRY(mad_threshold<T>(c, D, g, 8, ADN_EXT_PERIODIZATION, m.n_d1, max_frames * m.n_d1,
                                 d_sigma, d_thresh, &thresh));
    }
    if (three)
        ADN_LAUNCH((k_ch_apply<T, NCH, 3>), grid, block, 0, c->stream, d_in, d_out, g, m, tiles, thresh,
                           thr_mode, d_zeroed);
    else
        ADN_LAUNCH((k_ch_apply<T, NCH, 0>), grid, block, 0, c->stream, d_in, d_out, g, m, tiles, thresh,
                           thr_mode, d_zeroed);
    ADN_HIP(hipGetLastError());
    return ADN_OK;
}

template <typename T>
static int general_denoise(Ctx *c, const T *d_in, T *d_out, BlockGeom g, int64_t max_frames,
                           int wavelet, int level, int ext, int thr_mode, double *d_sigma,
                           double *d_thresh, int64_t *d_zeroed, const double *d_thresh_given) {
    const Taps<T> tp = make_taps<T>(wavelet);
    const int F = tp.F;
    // symmetric synthesis of level l writes 2 K - F + 2 values, one more than the band above
    // keeps when that band's length is odd (pywt.waverec drops it): one spare slot per block
    g.a_stride = band_len(max_frames, 1, F, ext) + 2;
    g.d_stride = level + 1;
    for (int l = 1; l <= level; ++l) g.d_stride += band_len(max_frames, l, F, ext);
    const int64_t max_seg = band_len(max_frames, 1, F, ext);
    const size_t nA = (size_t)g.n_blocks * g.a_stride, nD = (size_t)g.n_blocks * g.d_stride;
    ADN_TRY(ws_reserve(c, 2 * ws_need(nA, sizeof(T)) + ws_need(nD, sizeof(T)) +
                          mad_ws_bytes(g.n_blocks, max_seg)));
    T *A0 = ws_take<T>(c, nA), *A1 = ws_take<T>(c, nA), *D = ws_take<T>(c, nD);
    for (int l = 1; l <= level; ++l) {
        const int64_t tiles = ceil_div(band_len(max_frames, l, F, ext), 256);
        ADN_LAUNCH((k_fwd_level<T>), dim3((unsigned)(tiles * g.n_blocks)), dim3(256), 0,
                           c->stream, d_in, A0, A1, D, g, l, ext, tp, tiles);
    }
    double *thresh = const_cast<double *>(d_thresh_given);
    if (!d_thresh_given) ADN_TRY(mad_threshold<T>(c, D, g, F, ext, 0, max_seg, d_sigma, d_thresh, &thresh));
    for (int l = level; l >= 1; --l) {
        const int64_t tiles = ceil_div(2 * band_len(max_frames, l, F, ext), 256);
        ADN_LAUNCH((k_inv_level<T>), dim3((unsigned)(tiles * g.n_blocks)), dim3(256), 0,
                           c->stream, d_out, A0, A1, D, g, l, ext, tp, tiles, thresh, thr_mode,
                           d_zeroed);
    }
    ADN_HIP(hipGetLastError());
    return ADN_OK;
}

// Mono blocks (transform along time): fused tiles where they apply, level kernels otherwise.
template <typename T>
static int mono_denoise(Ctx *c, const T *d_in, T *d_out, const BlockGeom &g, int64_t max_frames,
                        int wavelet, int level, int ext, int thr_mode, double *d_sigma,
                        double *d_thresh, int64_t *d_zeroed, const double *d_thresh_given = nullptr) {
    if (ext == ADN_EXT_PERIODIZATION) {
        bool handled = false;
        ADN_TRY(wavelet_denoise_fused<T>(c, d_in, d_out, g, max_frames, wavelet, level, thr_mode,
                                         d_sigma, d_thresh, d_zeroed, d_thresh_given, &handled));
        if (handled) return ADN_OK;
    }
    return general_denoise<T>(c, d_in, d_out, g, max_frames, wavelet, level, ext, thr_mode, d_sigma,
                              d_thresh, d_zeroed, d_thresh_given);
}

// ADN_AXIS_TIME on interleaved multi-channel blocks: channel ch of block b becomes the mono block
// b * n_ch + ch of a planar copy (plane stride pstride), and back.
template <typename T>
__global__ __launch_bounds__(256) void k_planar_split(const T *__restrict__ x, T *__restrict__ P, BlockGeom g,
                                                      int n_ch, int64_t pstride, int64_t tiles_per_block) {
    const int64_t b = blockIdx.x / tiles_per_block;
    const int64_t e = (blockIdx.x % tiles_per_block) * 256 + threadIdx.x;    // element of the interleaved block
    if (e >= g.frames[b] * n_ch) return;
    const int64_t f = e / n_ch;
    const int ch = (int)(e % n_ch);
    P[(b * n_ch + ch) * pstride + f] = x[g.in_off[b] + e];
}

template <typename T>
__global__ __launch_bounds__(256) void k_planar_join(const T *__restrict__ Q, T *__restrict__ y, BlockGeom g,
                                                     int n_ch, int64_t pstride, int64_t tiles_per_block) {
    const int64_t b = blockIdx.x / tiles_per_block;
    const int64_t e = (blockIdx.x % tiles_per_block) * 256 + threadIdx.x;
    const int64_t out_frames = g.frames[b] + (g.frames[b] & 1);              // every synthesis returns an even length
    if (e >= out_frames * n_ch) return;
    const int64_t f = e / n_ch;
    const int ch = (int)(e % n_ch);
    y[g.out_off[b] + e] = Q[(b * n_ch + ch) * pstride + f];
}

__global__ void k_planar_desc(const int64_t *__restrict__ frames, int64_t n_blocks, int n_ch, int64_t pstride,
                              int64_t *__restrict__ off2, int64_t *__restrict__ frames2) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n_blocks * n_ch) return;
    off2[i] = i * pstride;
    frames2[i] = frames[i / n_ch];
}

template <typename T>
int wavelet_denoise(Ctx *c, const T *d_in, T *d_out, int64_t n_blocks, const int64_t *d_in_off,
                    const int64_t *d_frames, const int64_t *d_out_off, int64_t max_frames,
                    int n_ch, int axis_mode, int wavelet, int level, int ext_mode, int thr_mode,
                    double *d_sigma, double *d_thresh, int64_t *d_zeroed) {
    ADN_REQUIRE(c && d_in && d_out && d_in_off && d_frames && d_out_off, "null pointer argument");
    ADN_REQUIRE(n_blocks >= 0 && max_frames >= 1, "n_blocks must be >= 0 and max_frames >= 1");
    ADN_REQUIRE(wavelet == ADN_WAVELET_DB4 || wavelet == ADN_WAVELET_DB8, "unknown wavelet id %d", wavelet);
    ADN_REQUIRE(level >= 1 && level <= 16, "level must be in [1, 16], got %d", level);
    ADN_REQUIRE(thr_mode == ADN_THR_SOFT || thr_mode == ADN_THR_HARD, "unknown threshold mode %d", thr_mode);
    ADN_REQUIRE(ext_mode == ADN_EXT_PERIODIZATION || ext_mode == ADN_EXT_SYMMETRIC, "unknown extension mode %d", ext_mode);
    ADN_REQUIRE(axis_mode == ADN_AXIS_REFERENCE || axis_mode == ADN_AXIS_TIME, "unknown axis mode %d", axis_mode);
    ADN_REQUIRE(n_ch >= 1, "n_ch must be >= 1");
    if (n_blocks == 0) return ADN_OK;
    ADN_REQUIRE(n_blocks * n_ch * ceil_div(max_frames, 256) < (int64_t)0x7FFFFFFF, "batch too large for one launch");
    DeviceGuard dg(c->device);
    ws_reset(c);
    const int64_t n_units = (n_ch > 1 && axis_mode == ADN_AXIS_TIME) ? n_blocks * n_ch : n_blocks;
    if (d_zeroed) ADN_HIP(hipMemsetAsync(d_zeroed, 0, sizeof(int64_t) * n_units, c->stream));
    BlockGeom g{d_in_off, d_frames, d_out_off, n_blocks, 0, 0};
    if (n_ch == 1)
        return mono_denoise<T>(c, d_in, d_out, g, max_frames, wavelet, level, ext_mode, thr_mode, d_sigma,
                               d_thresh, d_zeroed);
    if (axis_mode == ADN_AXIS_TIME) {
        // planar copies live in the context's second buffer: the mono path owns the workspace
        const int64_t pstride = (max_frames + 2 + 63) & ~int64_t(63);
        const size_t plane = ws_need((size_t)n_units * pstride, sizeof(T));
        ADN_TRY(aux_reserve(c, 2 * plane + 2 * ws_need(n_units, sizeof(int64_t))));
        T *P = reinterpret_cast<T *>(c->aux), *Q = reinterpret_cast<T *>(c->aux + plane);
        int64_t *off2 = reinterpret_cast<int64_t *>(c->aux + 2 * plane);
        int64_t *frames2 = reinterpret_cast<int64_t *>(c->aux + 2 * plane + ws_need(n_units, sizeof(int64_t)));
        const int64_t tiles = ceil_div((max_frames + 1) * n_ch, 256);
        ADN_LAUNCH(k_planar_desc, dim3((unsigned)ceil_div(n_units, 256)), dim3(256), 0, c->stream, d_frames, n_blocks,
                   n_ch, pstride, off2, frames2);
        ADN_LAUNCH((k_planar_split<T>), dim3((unsigned)(tiles * n_blocks)), dim3(256), 0, c->stream, d_in, P, g, n_ch,
                   pstride, tiles);
        BlockGeom g2{off2, frames2, off2, n_units, 0, 0};
        ADN_TRY(mono_denoise<T>(c, P, Q, g2, max_frames, wavelet, level, ext_mode, thr_mode, d_sigma, d_thresh,
                                d_zeroed));
        ADN_LAUNCH((k_planar_join<T>), dim3((unsigned)(tiles * n_blocks)), dim3(256), 0, c->stream, Q, d_out, g, n_ch,
                   pstride, tiles);
        ADN_HIP(hipGetLastError());
        return ADN_OK;
    }
    if (ext_mode != ADN_EXT_PERIODIZATION) {
        set_error("the channel-axis transform of multi-channel blocks is built for periodization only");
        return ADN_ERR_UNSUPPORTED;
    }
    if (n_ch % 2 != 0 || n_ch > 8) {
        set_error("n_ch = %d: the channel-axis transform needs an even channel count <= 8 "
                  "(the reference fails on odd counts: waverec returns n_ch+1 channels)", n_ch);
        return ADN_ERR_UNSUPPORTED;
    }
    ChannelMap<T> m;
    ADN_TRY(build_channel_map<T>(n_ch, wavelet, level, &m));
    switch (n_ch) {
        case 2: return channel_denoise<T, 2>(c, d_in, d_out, g, max_frames, m, thr_mode, d_sigma, d_thresh, d_zeroed);
        case 4: return channel_denoise<T, 4>(c, d_in, d_out, g, max_frames, m, thr_mode, d_sigma, d_thresh, d_zeroed);
        case 6: return channel_denoise<T, 6>(c, d_in, d_out, g, max_frames, m, thr_mode, d_sigma, d_thresh, d_zeroed);
        default: return channel_denoise<T, 8>(c, d_in, d_out, g, max_frames, m, thr_mode, d_sigma, d_thresh, d_zeroed);
    }
}

// Mono blocks shrunk with thresholds the caller hands in (one per block): the statistics of a block
// that is shared between devices are reduced outside, every device then shrinks its part of the block.
template <typename T>
int wavelet_shrink(Ctx *c, const T *d_in, T *d_out, int64_t n_blocks, const int64_t *d_in_off,
                   const int64_t *d_frames, const int64_t *d_out_off, int64_t max_frames, int wavelet, int level,
                   int ext_mode, int thr_mode, const double *d_thresh, int64_t *d_zeroed) {
    ADN_REQUIRE(c && d_in && d_out && d_in_off && d_frames && d_out_off && d_thresh, "null pointer argument");
    ADN_REQUIRE(n_blocks >= 0 && max_frames >= 1, "n_blocks must be >= 0 and max_frames >= 1");
    ADN_REQUIRE(wavelet == ADN_WAVELET_DB4 || wavelet == ADN_WAVELET_DB8, "unknown wavelet id %d", wavelet);
    ADN_REQUIRE(level >= 1 && level <= 16, "level must be in [1, 16], got %d", level);
    ADN_REQUIRE(thr_mode == ADN_THR_SOFT || thr_mode == ADN_THR_HARD, "unknown threshold mode %d", thr_mode);
    ADN_REQUIRE(ext_mode == ADN_EXT_PERIODIZATION || ext_mode == ADN_EXT_SYMMETRIC, "unknown extension mode %d", ext_mode);
    if (n_blocks == 0) return ADN_OK;
    ADN_REQUIRE(n_blocks * ceil_div(max_frames, 256) < (int64_t)0x7FFFFFFF, "batch too large for one launch");
    DeviceGuard dg(c->device);
    ws_reset(c);
    if (d_zeroed) ADN_HIP(hipMemsetAsync(d_zeroed, 0, sizeof(int64_t) * n_blocks, c->stream));
    BlockGeom g{d_in_off, d_frames, d_out_off, n_blocks, 0, 0};
    return mono_denoise<T>(c, d_in, d_out, g, max_frames, wavelet, level, ext_mode, thr_mode, nullptr, nullptr, d_zeroed,
                           d_thresh);
}

// ------------------------------------------- single-signal wavedec / waverec
template <typename T>
int wavedec(Ctx *c, const T *d_x, int64_t n, int wavelet, int level, int ext_mode, T *d_coeffs,
            int64_t cap, int64_t *h_band_len) {
    ADN_REQUIRE(c && d_x && d_coeffs && h_band_len, "null pointer argument");
    ADN_REQUIRE(n >= 1 && level >= 1 && level <= 32, "n and level must be >= 1");
    ADN_REQUIRE(ext_mode == ADN_EXT_PERIODIZATION || ext_mode == ADN_EXT_SYMMETRIC, "unknown extension mode");
    ADN_REQUIRE(wavelet == ADN_WAVELET_DB4 || wavelet == ADN_WAVELET_DB8, "unknown wavelet id %d", wavelet);
    DeviceGuard dg(c->device);
    ws_reset(c);
    const Taps<T> tp = make_taps<T>(wavelet);
    const int F = tp.F;
    int64_t total = band_len(n, level, F, ext_mode);
    h_band_len[0] = total;
    for (int l = level; l >= 1; --l) {
        h_band_len[level - l + 1] = band_len(n, l, F, ext_mode);
        total += band_len(n, l, F, ext_mode);
    }
    ADN_REQUIRE(total <= cap, "coefficient buffer too small: need %lld", (long long)total);
    const int64_t a_stride = band_len(n, 1, F, ext_mode) + 1, d_stride = total + 1;
    ADN_TRY(ws_reserve(c, 2 * ws_need(a_stride, sizeof(T)) + ws_need(d_stride, sizeof(T)) + 3 * 256));
    T *A0 = ws_take<T>(c, a_stride), *A1 = ws_take<T>(c, a_stride), *D = ws_take<T>(c, d_stride);
    int64_t *desc = ws_take<int64_t>(c, 3);
    const int64_t hdesc[3] = {0, n, 0};
    ADN_HIP(hipMemcpyAsync(desc, hdesc, sizeof(hdesc), hipMemcpyHostToDevice, c->stream));
    ADN_HIP(hipStreamSynchronize(c->stream));
    BlockGeom g{desc, desc + 1, desc + 2, 1, a_stride, d_stride};
    for (int l = 1; l <= level; ++l) {
        const int64_t tiles = ceil_div(band_len(n, l, F, ext_mode), 256);
        ADN_LAUNCH((k_fwd_level<T>), dim3((unsigned)tiles), dim3(256), 0, c->stream, d_x,
                           A0, A1, D, g, l, ext_mode, tp, tiles);
    }
    // output order [cA_L, cD_L, ..., cD_1]
    int64_t pos = 0;
    ADN_HIP(hipMemcpyAsync(d_coeffs, (level & 1) ? A1 : A0, sizeof(T) * h_band_len[0],
                           hipMemcpyDeviceToDevice, c->stream));
    pos += h_band_len[0];
    for (int l = level; l >= 1; --l) {
        const int64_t len = band_len(n, l, F, ext_mode);
        ADN_HIP(hipMemcpyAsync(d_coeffs + pos, D + band_off(n, l, F, ext_mode), sizeof(T) * len,
                               hipMemcpyDeviceToDevice, c->stream));
        pos += len;
    }
    ADN_HIP(hipStreamSynchronize(c->stream));
    return ADN_OK;
}

template <typename T>
int waverec(Ctx *c, const T *d_coeffs, const int64_t *h_band_len, int wavelet, int level,
            int ext_mode, T *d_y, int64_t cap, int64_t *h_n_out) {
    ADN_REQUIRE(c && d_coeffs && h_band_len && d_y && h_n_out, "null pointer argument");
    ADN_REQUIRE(level >= 1 && level <= 32, "level must be >= 1");
    ADN_REQUIRE(ext_mode == ADN_EXT_PERIODIZATION || ext_mode == ADN_EXT_SYMMETRIC, "unknown extension mode");
    ADN_REQUIRE(wavelet == ADN_WAVELET_DB4 || wavelet == ADN_WAVELET_DB8, "unknown wavelet id %d", wavelet);
    DeviceGuard dg(c->device);
    ws_reset(c);
    const Taps<T> tp = make_taps<T>(wavelet);
    const int F = tp.F;
    // pywt.waverec: a = cA; for each d: drop a's last sample when len(a) == len(d)+1, then idwt
    int64_t max_len = 0;
    for (int i = 0; i <= level; ++i) {
        ADN_REQUIRE(h_band_len[i] >= 1, "band %d is empty", i);
        max_len = h_band_len[i] > max_len ? h_band_len[i] : max_len;
    }
    const int64_t buf = 2 * max_len + 2;
    ADN_TRY(ws_reserve(c, 2 * ws_need(buf, sizeof(T)) + 4 * 256));
    T *B0 = ws_take<T>(c, buf), *B1 = ws_take<T>(c, buf);
    int64_t *desc = ws_take<int64_t>(c, 3);
    ADN_HIP(hipMemcpyAsync(B0, d_coeffs, sizeof(T) * h_band_len[0], hipMemcpyDeviceToDevice, c->stream));
    int64_t a_len = h_band_len[0], pos = h_band_len[0];
    T *cur = B0, *nxt = B1;
    for (int i = 1; i <= level; ++i) {
        const int64_t K = h_band_len[i];
        if (a_len == K + 1) a_len = K;
        ADN_REQUIRE(a_len == K, "coefficient shape mismatch at band %d", i);
        // one synthesis level as a one-block, one-level problem: frames such that band_len(.,1) == K
        const int64_t n_out = ext_mode == ADN_EXT_PERIODIZATION ? 2 * K : 2 * K - F + 2;
        ADN_REQUIRE(n_out >= 1, "band %d too short for symmetric synthesis", i);
        const int64_t frames1 = ext_mode == ADN_EXT_PERIODIZATION ? 2 * K : 2 * K - F + 2;
        ADN_REQUIRE(band_len(frames1, 1, F, ext_mode) == K, "internal: synthesis geometry");
        const int64_t hdesc[3] = {0, frames1, 0};
        ADN_HIP(hipMemcpyAsync(desc, hdesc, sizeof(hdesc), hipMemcpyHostToDevice, c->stream));
        ADN_HIP(hipStreamSynchronize(c->stream));
        BlockGeom g{desc, desc + 1, desc + 2, 1, 0, 0};
        const int64_t tiles = ceil_div(n_out, 256);
        T *dst = i == level ? d_y : nxt;
        if (i == level) ADN_REQUIRE(n_out <= cap, "output buffer too small: need %lld", (long long)n_out);
        // level index 1 => a from A1 (l & 1), out to y; pass cur as A1 and detail pointer offset 0
        ADN_LAUNCH((k_inv_level<T>), dim3((unsigned)tiles), dim3(256), 0, c->stream, dst,
                           (T *)nullptr, cur, d_coeffs + pos, g, 1, ext_mode, tp, tiles,
                           (const double *)nullptr, 0, (int64_t *)nullptr);
        ADN_HIP(hipGetLastError());
        ADN_HIP(hipStreamSynchronize(c->stream));
        pos += K;
        a_len = n_out;
        T *t = cur;
        cur = nxt;
        nxt = t;
        *h_n_out = n_out;
    }
    return ADN_OK;
}

#define ADN_INST(T)                                                                               \
    template int wavelet_denoise<T>(Ctx *, const T *, T *, int64_t, const int64_t *,              \
                                    const int64_t *, const int64_t *, int64_t, int, int, int, int, \
                                    int, int, double *, double *, int64_t *);                     \
    template int wavelet_shrink<T>(Ctx *, const T *, T *, int64_t, const int64_t *, const int64_t *,  \
                                   const int64_t *, int64_t, int, int, int, int, const double *, int64_t *); \
    template int wavedec<T>(Ctx *, const T *, int64_t, int, int, int, T *, int64_t, int64_t *);   \
    template int waverec<T>(Ctx *, const T *, const int64_t *, int, int, int, T *, int64_t, int64_t *);
ADN_INST(float)
ADN_INST(double)

}  // namespace adn
