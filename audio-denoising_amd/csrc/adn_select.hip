// Exact segmented order statistics: wavefront-ballot radix select.
//
// One 256-thread workgroup per segment.  Values are mapped to order-preserving
// unsigned keys; the k-th smallest key is found most-significant digit first
// (8 bits per pass).  Lanes of a wave that hold the same digit are matched with
// ballots so that one lane adds the whole group's population count to the LDS
// histogram (no same-address atomic serialisation when most values share an
// exponent).  Segments up to SEL_LDS_KEYS keys are staged in LDS once and every
// later pass reads LDS; longer segments are re-read from memory each pass.
//
// Results follow numpy: median of an even count is the mean of the two middle
// order statistics; percentiles interpolate linearly at (n-1)*q/100; a NaN
// anywhere in the segment gives NaN.
#include <type_traits>

#include <algorithm>

#include "adn_kernels.hpp"

namespace adn {

static constexpr int SEL_THREADS = 256;
static constexpr int SEL_LDS_BYTES = 48 * 1024;
static constexpr int64_t SEL_ONE_WAVE_MAX = 1024;       // segments up to this length are selected by one wave
static constexpr size_t SEL_LDS_MAX_BYTES = 96 * 1024;  // longest segment the lean LDS kernel stages
static constexpr size_t SUBSET_LDS_BYTES = 156 * 1024;  // ... when it selects from gathered candidate lists (one workgroup per CU)

__device__ __forceinline__ void hist_add_matched(uint32_t *hist, uint32_t digit, bool active) {
    unsigned long long m = __ballot(active);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const bool bit = (digit >> b) & 1u;
        const unsigned long long bm = __ballot(bit);
        m &= bit ? bm : ~bm;
    }
    if (active) {
        const int lane = threadIdx.x & 63;
        if (lane == __ffsll((long long)m) - 1) atomicAdd(&hist[digit], (uint32_t)__popcll(m));
    }
}

// Value of a selection from its two order statistics a <= b (numpy rules: the median of an even
// count is the mean of the middle pair; percentiles interpolate linearly with numpy's t >= 0.5
// branch, in the array's dtype).  SEL_PERCENTILE_DB20 interpolates 20*log10(v + 1e-12) of the two
// order statistics instead (the percentile of a dB-converted array, in binary64: the map is monotone).
// The two ranks that bracket the statistic of n values and the interpolation weight between them:
// numpy.median (ranks (n-1)/2 and n/2) or numpy.percentile's 'linear' method (virtual index (n-1) q/100,
// rounded before its fraction is taken, as numpy does).
__host__ __device__ inline void stat_ranks(int64_t n, int mode, double q, int64_t *r0, int64_t *r1, double *frac) {
#pragma clang fp contract(off)
    if (mode == SEL_MEDIAN) {
        *r0 = (n - 1) / 2;
        *r1 = n / 2;
        *frac = 0.5;
        return;
    }
    if (mode == SEL_BRACKET_LO || mode == SEL_BRACKET_HI) {
        const double p = q / 100.0, d = 4.5 * sqrt((double)n * p * (1.0 - p)) + 2.0;
        double at = (double)(n - 1) * p + (mode == SEL_BRACKET_LO ? -d : d);
        at = mode == SEL_BRACKET_LO ? floor(at) : ceil(at);
        at = at < 0.0 ? 0.0 : (at > (double)(n - 1) ? (double)(n - 1) : at);
        *r0 = *r1 = (int64_t)at;
        *frac = 0.0;
        return;
    }
    const double pos = (double)(n - 1) * (q / 100.0);
    int64_t lo = (int64_t)floor(pos);
    if (lo > n - 1) lo = n - 1;
    *r0 = lo;
    *frac = pos - (double)lo;
    *r1 = lo + 1 < n ? lo + 1 : n - 1;
}

template <typename T>
__device__ __forceinline__ double select_result(T a, T b, bool same, double frac, int mode) {
#pragma clang fp contract(off)   // numpy rounds the product and the sum separately
    if (mode == SEL_MEDIAN) return same ? (double)a : (double)(T)((a + b) / (T)2);
    double da = (double)a, db = (double)b;
    if (mode == SEL_PERCENTILE_DB20) {
        da = 20.0 * log10(da + 1e-12);
        db = 20.0 * log10(db + 1e-12);
    }
    const double diff = db - da;
    double res = frac >= 0.5 ? db - diff * (1.0 - frac) : da + diff * frac;
    if (same) res = da;
    return mode == SEL_PERCENTILE_DB20 ? res : (double)(T)res;
}

// Exclusive prefix sum of one value per thread over the 256-thread workgroup.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *wsum) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t y = __shfl_up(x, d);
        if (lane >= d) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    __syncthreads();
    return base + x - v;
}

// Optional fused statistic of the first pass: masked mean of the segment,
// out[seg] = sum(vals[i] where mask[(seg / mask_div) * mask_stride + i] != 0) / max(count, 1)
// (the VAD-guided speech PSD of src/wiener_filter.py:99-102 shares the row with the percentile).
struct MaskedMean {
    const unsigned char *mask;
    int64_t mask_div, mask_stride;
    double *out;
};

// Optional: the segment handed to k_select is a gathered SUBSET (the keys inside a value
// bracket) of a longer segment.  Ranks are then those of the original segment minus the
// number of keys below the bracket; segments flagged invalid are left untouched.
struct SubsetRanks {
    const int64_t *orig_len;              // per (seg / orig_div), or null: orig_uniform
    int64_t orig_div, orig_uniform;
    const unsigned long long *below;      // keys of the original segment below the bracket
    const int *valid;                     // 0: skip this segment
    const unsigned long long *lohi;       // bracket keys [2*seg], [2*seg+1]: every key lies between them
    double *pair = nullptr;               // optional: the two order statistics, [2*seg], [2*seg+1]
};

template <typename T, int XF>
__global__ __launch_bounds__(SEL_THREADS) void k_select(const T *__restrict__ vals,
                                                        const int64_t *__restrict__ seg_off,
                                                        int64_t seg_stride,
                                                        const int64_t *__restrict__ seg_len,
                                                        int64_t len_div, int64_t uniform_len,
                                                        int mode, double q,
                                                        const double *__restrict__ center,
                                                        double *__restrict__ out, int lds_keys,
                                                        MaskedMean mm, SubsetRanks sub) {
    using K = typename KeyT<T>::type;
    constexpr int BITS = KeyT<T>::bits;
    constexpr int PASSES = BITS / 8;
    extern __shared__ __align__(16) unsigned char smem[];
    K *cur = reinterpret_cast<K *>(smem);        // list the current pass walks (LDS-resident segments)
    K *oth = cur + lds_keys;                     // the next, shorter list is gathered here
    __shared__ uint32_t hist[256];
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t s_digit, s_before, s_count, s_nan, s_fill, s_next;
    __shared__ K s_lo_max, s_hi_min, s_kmin, s_kmax;
    __shared__ double s_msum[4];
    __shared__ int s_mcnt[4];

    const int64_t seg = blockIdx.x;
    const int64_t off = seg_off ? seg_off[seg] : seg * seg_stride;
    const int64_t n = seg_len ? seg_len[seg / len_div] : uniform_len;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (sub.valid && !sub.valid[seg]) return;
    if (n <= 0) {
        if (tid == 0) {
            out[seg] = __longlong_as_double(0x7FF8000000000000ll);
            if (mm.out) mm.out[seg] = 0.0;
        }
        return;
    }
    int64_t r0, r1;
    double frac;
    {
        // ranks come from the length of the ORIGINAL segment
        const int64_t nr = sub.below ? (sub.orig_len ? sub.orig_len[seg / sub.orig_div] : sub.orig_uniform) : n;
        stat_ranks(nr, mode, q, &r0, &r1, &frac);
        if (sub.below) {
            r0 -= (int64_t)sub.below[seg];
            r1 -= (int64_t)sub.below[seg];
        }
    }
    const T c = XF == XF_ABSDEV ? (T)center[seg] : (T)0;
    // n <= lds_keys: keys live in LDS and every pass gathers the chosen bin into a shorter list;
    // n <= 2*lds_keys: keys live in LDS (both halves), passes re-scan them; else re-read memory
    const bool cached = n <= (int64_t)lds_keys;
    const bool resident = !cached && n <= 2 * (int64_t)lds_keys;
    const T *v = vals + off;
    const unsigned char *mk = mm.out ? mm.mask + (seg / mm.mask_div) * mm.mask_stride : nullptr;
    auto load_key = [&](int64_t i) -> K {
        T x = v[i];
        if (XF == XF_ABSDEV) x = fabs(x - c);
        return x != x ? ~(K)0 : key_of(x);
    };

    if (tid == 0) {
        s_nan = 0;
        s_kmin = ~(K)0;
        s_kmax = 0;
    }
    K prefix = 0;                   // digits fixed so far, right aligned
    uint64_t r = (uint64_t)r0;      // rank of s[r0] inside the current list
    int64_t n_act = n;              // length of the current list (cached) or of the segment
    K k0 = 0, k1 = 0;
    bool done = false;
    T msum = 0;
    int mcnt = 0;
    const bool in_lds = cached || resident;
    // LDS-resident segments: one load phase stages the keys and finds their range.  The digits
    // are then taken from (key - min): every pass spreads over up to 256 bins even when all
    // values share sign and exponent (no same-address LDS atomics), and only the
    // ceil(bits(max - min) / 8) passes that can differ are run.
    K kmin = 0;
    int passes = PASSES;
    __syncthreads();
    if (in_lds) {
        K tmin = ~(K)0, tmax = 0;
        bool saw_nan = false;
        const int64_t its = (n + SEL_THREADS - 1) / SEL_THREADS;
        for (int64_t it0 = 0; it0 < its; it0 += 8) {
            T xv[8];
            unsigned char mv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t i = (it0 + u) * SEL_THREADS + tid;
                xv[u] = i < n ? v[i] : (T)0;
                mv[u] = (mk && i < n) ? mk[i] : (unsigned char)0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t i = (it0 + u) * SEL_THREADS + tid;
                if (i < n) {
                    const T x = xv[u];
                    if (mv[u]) {
                        msum += x;
                        ++mcnt;
                    }
                    T xx = x;
                    if (XF == XF_ABSDEV) xx = fabs(x - c);
                    const K k = xx != xx ? ~(K)0 : key_of(xx);
                    cur[i] = k;
                    saw_nan |= (k == ~(K)0);
                    tmin = k < tmin ? k : tmin;
                    tmax = k > tmax ? k : tmax;
                }
            }
        }
        for (int d = 32; d >= 1; d >>= 1) {
            const K a = __shfl_xor(tmin, d), b = __shfl_xor(tmax, d);
            tmin = a < tmin ? a : tmin;
            tmax = b > tmax ? b : tmax;
        }
        if (lane == 0) {
            atomicMin(&s_kmin, tmin);
            atomicMax(&s_kmax, tmax);
        }
        if (saw_nan) s_nan = 1;
        __syncthreads();
        kmin = s_kmin;
        const K width = s_kmax - kmin;
        const int bits = width ? BITS - (BITS == 64 ? __clzll((long long)width) : __clz((int)width)) : 0;
        passes = (bits + 7) / 8;
        if (s_nan) {
            done = true;                  // the result is NaN whatever the keys are
        } else if (passes == 0) {
            k0 = k1 = 0;                  // all keys equal: both order statistics are the minimum
            done = true;
        }
    }
    for (int pass = 0; pass < passes && !done; ++pass) {
        const int shift = 8 * (passes - 1 - pass);
        hist[tid] = 0;
        if (tid == 0) {
            s_fill = 0;
            s_next = 256;
            s_lo_max = 0;
            s_hi_min = ~(K)0;
        }
        __syncthreads();
        // ---- histogram of this pass's digit
        bool saw_nan = false;
        const int64_t its = (n_act + SEL_THREADS - 1) / SEL_THREADS;
        if (!in_lds && pass == 0) {
            // segments too long for LDS: every pass reads memory; the first one also looks for
            // NaNs and accumulates the masked mean.  Loads go out in batches of 8 per thread.
            for (int64_t it0 = 0; it0 < its; it0 += 8) {
                T xv[8];
                unsigned char mv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int64_t i = (it0 + u) * SEL_THREADS + tid;
                    xv[u] = i < n_act ? v[i] : (T)0;
                    mv[u] = (mk && i < n_act) ? mk[i] : (unsigned char)0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int64_t i = (it0 + u) * SEL_THREADS + tid;
                    if (i < n_act) {
                        const T x = xv[u];
                        if (mv[u]) {
                            msum += x;
                            ++mcnt;
                        }
                        T xx = x;
                        if (XF == XF_ABSDEV) xx = fabs(x - c);
                        const K k = xx != xx ? ~(K)0 : key_of(xx);
                        saw_nan |= (k == ~(K)0);
                        atomicAdd(&hist[(uint32_t)((k >> shift) & 0xFF)], 1u);
                    }
                }
            }
            if (saw_nan) s_nan = 1;
        } else {
            for (int64_t it = 0; it < its; ++it) {
                const int64_t i = it * SEL_THREADS + tid;
                if (i < n_act) {
                    K k;
                    bool match = true;
                    if (cached) {
                        k = cur[i] - kmin;           // the list holds only keys with this prefix
                    } else {
                        k = resident ? cur[i] - kmin : load_key(i);
                        match = pass == 0 || (k >> (shift + 8)) == prefix;
                    }
                    if (match) atomicAdd(&hist[(uint32_t)((k >> shift) & 0xFF)], 1u);
                }
            }
        }
        __syncthreads();
        // ---- bin of s[r0]; is s[r1] in the same bin?
        const uint32_t h = hist[tid];
        const uint32_t before = block_excl_scan(h, wsum);
        if (r >= before && r < (uint64_t)before + h) {
            s_digit = tid;
            s_before = before;
            s_count = h;
        }
        __syncthreads();
        const uint32_t digit = s_digit, in_bin = s_count;
        const uint64_t within = r - s_before;
        const bool together = (r1 == r0) || (within + 1 < in_bin);
        const K new_prefix = (prefix << 8) | (K)digit;
        if (!together) {
            // s[r0] is the largest key of its bin and s[r1] the smallest key of the next non-empty
            // bin: one sweep over the current list settles both
            if (h && (uint32_t)tid > digit) atomicMin(&s_next, (uint32_t)tid);
            __syncthreads();
            const K pre_hi = (prefix << 8) | (K)s_next;
            K lo_max = 0, hi_min = ~(K)0;
            for (int64_t i = tid; i < n_act; i += SEL_THREADS) {
                const K k = in_lds ? cur[i] - kmin : load_key(i);
                const K top = k >> shift;
                if (top == new_prefix && k > lo_max) lo_max = k;
                if (top == pre_hi && k < hi_min) hi_min = k;
            }
            for (int d = 32; d >= 1; d >>= 1) {
                const K a = __shfl_xor(lo_max, d), b = __shfl_xor(hi_min, d);
                lo_max = a > lo_max ? a : lo_max;
                hi_min = b < hi_min ? b : hi_min;
            }
            if (lane == 0) {
                atomicMax(&s_lo_max, lo_max);
                atomicMin(&s_hi_min, hi_min);
            }
            __syncthreads();
            k0 = s_lo_max;
            k1 = s_hi_min;
            done = true;
        } else if (pass + 1 == passes) {
            k0 = k1 = new_prefix;             // both order statistics are copies of one key
            done = true;
        } else {
            if (cached) {
                // gather the keys of the chosen bin: later passes walk this short list only
                for (int64_t it = 0; it < its; ++it) {
                    const int64_t i = it * SEL_THREADS + tid;
                    const bool in = i < n_act;
                    const K k = in ? cur[i] : 0;
                    const bool keep = in && ((K)(k - kmin) >> shift) == new_prefix;
                    const unsigned long long m = __ballot(keep);
                    uint32_t base = 0;
                    if (lane == 0 && m) base = atomicAdd(&s_fill, (uint32_t)__popcll(m));
                    base = __shfl(base, 0);
                    if (keep) oth[base + __popcll(m & ((1ull << lane) - 1ull))] = k;
                }
                n_act = in_bin;
                K *t = cur;
                cur = oth;
                oth = t;
            }
            prefix = new_prefix;
            r = within;
        }
        __syncthreads();
    }
    k0 += kmin;                               // digits were taken from (key - kmin)
    k1 += kmin;
    if (mm.out) {
        for (int d = 32; d >= 1; d >>= 1) {
            msum += __shfl_xor(msum, d);
            mcnt += __shfl_xor(mcnt, d);
        }
        if (lane == 0) {
            s_msum[wave] = (double)msum;
            s_mcnt[wave] = mcnt;
        }
        __syncthreads();
    }
    if (tid == 0) {
        double res;
        if (s_nan) {
            res = __longlong_as_double(0x7FF8000000000000ll);
        } else {
            const T a = val_of(k0), b = val_of(k1);
            res = select_result<T>(a, b, r0 == r1, frac, mode);
            if (sub.pair) {
                sub.pair[2 * seg] = (double)a;
                sub.pair[2 * seg + 1] = (double)b;
            }
        }
        out[seg] = res;
        if (mm.out) {
            const T tot = (T)s_msum[0] + (T)s_msum[1] + (T)s_msum[2] + (T)s_msum[3];
            const int cn = s_mcnt[0] + s_mcnt[1] + s_mcnt[2] + s_mcnt[3];
            mm.out[seg] = (double)(tot / (T)(cn > 1 ? cn : 1));
        }
    }
}

// ----------------------------------------------------------------------------
// Lean variant for segments whose keys fit LDS (the common case: rows of the transposed
// spectra, bracket candidates): 32-bit indexing, keys staged once, every pass re-scans them
// (only keys that match the digits fixed so far touch the histogram), LDS = n keys exactly so
// that several workgroups share a CU.  Same results as k_select.
// NT threads per workgroup: 64 (one wave, no cross-wave waits) for short segments, 256, or more
// for rows of several thousand keys (fewer keys per lane in every sweep).
template <typename T, int XF, bool MASKED, int NT = SEL_THREADS>
__global__ __launch_bounds__(NT) void k_select_lds(const T *__restrict__ vals,
                                                            const int64_t *__restrict__ seg_off,
                                                            int64_t seg_stride,
                                                            const int64_t *__restrict__ seg_len,
                                                            int64_t len_div, int64_t uniform_len,
                                                            int mode, double q,
                                                            const double *__restrict__ center,
                                                            double *__restrict__ out,
                                                            MaskedMean mm, SubsetRanks sub) {
    using K = typename KeyT<T>::type;
    constexpr int BITS = KeyT<T>::bits;
    extern __shared__ __align__(16) unsigned char smem[];
    K *keys = reinterpret_cast<K *>(smem);
    __shared__ uint32_t hist[256];
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t s_digit, s_before, s_count, s_nan, s_next;
    __shared__ K s_lo_max, s_hi_min, s_kmin, s_kmax;
    constexpr int NW = NT / 64 < 4 ? 4 : NT / 64;          // per-wave partial sums (at least the four of 256 threads)
    __shared__ double s_msum[NW];
    __shared__ int s_mcnt[NW];

    const int64_t seg = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (sub.valid && !sub.valid[seg]) return;
    const int n = (int)(seg_len ? seg_len[seg / len_div] : uniform_len);
    if (n <= 0) {
        if (tid == 0) {
            out[seg] = __longlong_as_double(0x7FF8000000000000ll);
            if (MASKED) mm.out[seg] = 0.0;
        }
        return;
    }
    int64_t r0l, r1l;
    double frac;
    {
        const int64_t nr = sub.below ? (sub.orig_len ? sub.orig_len[seg / sub.orig_div] : sub.orig_uniform) : n;
        stat_ranks(nr, mode, q, &r0l, &r1l, &frac);
        if (sub.below) {
            r0l -= (int64_t)sub.below[seg];
            r1l -= (int64_t)sub.below[seg];
        }
    }
    const uint32_t r0 = (uint32_t)r0l, r1 = (uint32_t)r1l;
    const T c = XF == XF_ABSDEV ? (T)center[seg] : (T)0;
    const T *vrow = vals + (seg_off ? seg_off[seg] : seg * seg_stride);
    const unsigned char *mrow = MASKED ? mm.mask + (seg / mm.mask_div) * mm.mask_stride : nullptr;
    if (tid == 0) {
        s_nan = 0;
        s_kmin = ~(K)0;
        s_kmax = 0;
    }
    __syncthreads();
    // ---- load phase: keys to LDS, their range, NaN flag, masked mean
    K tmin = ~(K)0, tmax = 0;
    bool saw_nan = false;
    T msum = 0;
    int mcnt = 0;
    auto take = [&](int i, T x, unsigned char m) {
        if (MASKED && m) {
            msum += x;
            ++mcnt;
        }
        if (XF == XF_ABSDEV) x = fabs(x - c);
        const K k = x != x ? ~(K)0 : key_of(x);
        keys[i] = k;
        saw_nan |= (k == ~(K)0);
        tmin = k < tmin ? k : tmin;
        tmax = k > tmax ? k : tmax;
    };
    constexpr int VEC = 16 / (int)sizeof(T);            // elements per 16-byte load
    const bool vec_ok = (reinterpret_cast<uintptr_t>(vrow) & 15) == 0 &&
                        (!MASKED || (reinterpret_cast<uintptr_t>(mrow) & (VEC - 1)) == 0);
    if (vec_ok) {
        // 16-byte loads of the values, VEC-byte loads of the mask: a quarter (half) of the memory
        // instructions of the scalar form; the tail of the row is taken element by element
        using VT = typename std::conditional<sizeof(T) == 4, float4, double2>::type;
        using MT = typename std::conditional<sizeof(T) == 4, uint32_t, uint16_t>::type;
        const int nv = n / VEC;
        const VT *vv = reinterpret_cast<const VT *>(vrow);
        const MT *mv4 = reinterpret_cast<const MT *>(mrow);
        constexpr int LBV = 6;
        for (int j0 = 0; j0 < nv; j0 += LBV * NT) {
            VT xq[LBV];
            MT mq[LBV];
#pragma unroll
            for (int u = 0; u < LBV; ++u) {
                const int j = j0 + u * NT + tid;
                if (j < nv) {
                    xq[u] = vv[j];
                    if (MASKED) mq[u] = mv4[j];
                }
            }
#pragma unroll
            for (int u = 0; u < LBV; ++u) {
                const int j = j0 + u * NT + tid;
                if (j < nv) {
                    const T *xe = reinterpret_cast<const T *>(&xq[u]);
#pragma unroll
                    for (int e = 0; e < VEC; ++e)
                        take(j * VEC + e, xe[e], MASKED ? (unsigned char)((mq[u] >> (8 * e)) & 0xFF) : (unsigned char)0);
                }
            }
        }
        for (int i = nv * VEC + tid; i < n; i += NT) take(i, vrow[i], MASKED ? mrow[i] : (unsigned char)0);
    } else {
        constexpr int LB = 8;
        for (int i0 = 0; i0 < n; i0 += LB * NT) {
            T xv[LB];
            unsigned char mv[LB];
#pragma unroll
            for (int u = 0; u < LB; ++u) {
                const int i = i0 + u * NT + tid;
                const bool ok = i < n;
                xv[u] = ok ? vrow[i] : (T)0;
                if (MASKED) mv[u] = ok ? mrow[i] : (unsigned char)0;
            }
#pragma unroll
            for (int u = 0; u < LB; ++u) {
                const int i = i0 + u * NT + tid;
                if (i < n) take(i, xv[u], MASKED ? mv[u] : (unsigned char)0);
            }
        }
    }
    for (int d = 32; d >= 1; d >>= 1) {
        const K a = __shfl_xor(tmin, d), b = __shfl_xor(tmax, d);
        tmin = a < tmin ? a : tmin;
        tmax = b > tmax ? b : tmax;
        if (MASKED) {
            msum += __shfl_xor(msum, d);
            mcnt += __shfl_xor(mcnt, d);
        }
    }
    if (MASKED && NT < 256) {                       // partial sums of the waves that do not exist
        if (tid < NW) {
            s_msum[tid] = 0.0;
            s_mcnt[tid] = 0;
        }
        __syncthreads();
    }
    if (lane == 0) {
        atomicMin(&s_kmin, tmin);
        atomicMax(&s_kmax, tmax);
        if (MASKED) {
            s_msum[wave] = (double)msum;
            s_mcnt[wave] = mcnt;
        }
    }
    if (saw_nan) s_nan = 1;
    __syncthreads();
    const K kmin = s_kmin;
    const K width = s_kmax - kmin;
    const int bits = width ? BITS - (BITS == 64 ? __clzll((long long)width) : __clz((int)width)) : 0;
    const int passes = (bits + 7) / 8;
    K k0 = 0, k1 = 0, prefix = 0;
    uint32_t r = r0;
    bool done = s_nan != 0 || passes == 0;
    for (int pass = 0; pass < passes && !done; ++pass) {
        const int shift = 8 * (passes - 1 - pass);
        for (int b = tid; b < 256; b += NT) hist[b] = 0;
        if (tid == 0) {
            s_next = 256;
            s_lo_max = 0;
            s_hi_min = ~(K)0;
        }
        __syncthreads();
        if (pass == 0) {
#pragma unroll 4
            for (int i = tid; i < n; i += NT)
                atomicAdd(&hist[(uint32_t)((keys[i] - kmin) >> shift) & 0xFFu], 1u);
        } else {
#pragma unroll 4
            for (int i = tid; i < n; i += NT) {
                const K u = keys[i] - kmin;
                if ((u >> (shift + 8)) == prefix) atomicAdd(&hist[(uint32_t)(u >> shift) & 0xFFu], 1u);
            }
        }
        __syncthreads();
        // bin of s[r0]: one wave scans the 256 counters (4 per lane), no further barriers
        if (tid < 64) {
            const uint32_t c0 = hist[4 * tid], c1 = hist[4 * tid + 1], c2v = hist[4 * tid + 2], c3 = hist[4 * tid + 3];
            const uint32_t sum = c0 + c1 + c2v + c3;
            uint32_t incl = sum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(incl, d);
                if (tid >= d) incl += o;
            }
            const unsigned long long m = __ballot(incl > r);
            if (m && tid == (int)__ffsll(m) - 1) {
                uint32_t bef = incl - sum, cnt = c0;
                int dg = 4 * tid;
                if (r >= bef + cnt) { bef += cnt; cnt = c1; ++dg;
                    if (r >= bef + cnt) { bef += cnt; cnt = c2v; ++dg;
                        if (r >= bef + cnt) { bef += cnt; cnt = c3; ++dg; } } }
                s_digit = (uint32_t)dg;
                s_before = bef;
                s_count = cnt;
            }
        }
        __syncthreads();
        const uint32_t digit = s_digit, in_bin = s_count;
        const uint32_t within = r - s_before;
        const bool together = (r1 == r0) || (within + 1 < in_bin);
        const K new_prefix = (prefix << 8) | (K)digit;
        if (!together) {
            // s[r0] is the largest key of its bin and s[r1] the smallest key of the next
            // non-empty bin: one sweep settles both
            for (int b = tid; b < 256; b += NT)
                if (hist[b] && (uint32_t)b > digit) atomicMin(&s_next, (uint32_t)b);
            __syncthreads();
            const K pre_hi = (prefix << 8) | (K)s_next;
            K lo_max = 0, hi_min = ~(K)0;
            for (int i = tid; i < n; i += NT) {
                const K u = keys[i] - kmin;
                const K top = u >> shift;
                if (top == new_prefix && u > lo_max) lo_max = u;
                if (top == pre_hi && u < hi_min) hi_min = u;
            }
            for (int d = 32; d >= 1; d >>= 1) {
                const K a = __shfl_xor(lo_max, d), b = __shfl_xor(hi_min, d);
                lo_max = a > lo_max ? a : lo_max;
                hi_min = b < hi_min ? b : hi_min;
            }
            if (lane == 0) {
                atomicMax(&s_lo_max, lo_max);
                atomicMin(&s_hi_min, hi_min);
            }
            __syncthreads();
            k0 = s_lo_max;
            k1 = s_hi_min;
            done = true;
        } else if (pass + 1 == passes) {
            k0 = k1 = new_prefix;
            done = true;
        } else {
            prefix = new_prefix;
            r = within;
        }
        __syncthreads();
    }
    if (tid == 0) {
        double res;
        if (s_nan) {
            res = __longlong_as_double(0x7FF8000000000000ll);
        } else {
            const T a = val_of((K)(k0 + kmin)), b = val_of((K)(k1 + kmin));
            res = select_result<T>(a, b, r0 == r1, frac, mode);
            if (sub.pair) {
                sub.pair[2 * seg] = (double)a;
                sub.pair[2 * seg + 1] = (double)b;
            }
        }
        out[seg] = res;
        if (MASKED) {
            T tot = (T)s_msum[0] + (T)s_msum[1] + (T)s_msum[2] + (T)s_msum[3];
            int cn = s_mcnt[0] + s_mcnt[1] + s_mcnt[2] + s_mcnt[3];
#pragma unroll
            for (int w = 4; w < NW; ++w) {
                tot += (T)s_msum[w];
                cn += s_mcnt[w];
            }
            mm.out[seg] = (double)(tot / (T)(cn > 1 ? cn : 1));
        }
    }
}

// ----------------------------------------------------------------------------
// Long segments: every segment is split over many workgroups.  Each pass histograms
// one 11-bit digit of the keys that still match the prefix (LDS histogram per
// workgroup, non-empty bins merged into a global histogram per segment); a one-
// workgroup-per-segment scan picks the bin holding the wanted rank and clears the
// histogram for the next pass.  32-bit keys take 3 passes, 64-bit keys 6.
static constexpr int BIG_BINS = 2048;
static constexpr int BIG_TILE = 8192;          // elements per workgroup
static constexpr int GATHER_TILE = 8192;       // ... of the bracket gather
static constexpr int64_t BIG_THRESHOLD = 32768;

struct BigState {            // per segment
    unsigned long long prefix;    // key bits fixed so far (right aligned)
    unsigned long long rank;      // rank of s[r0] among the keys matching the prefix
    unsigned long long k0, k1;    // keys of s[r0], s[r1]
    unsigned long long cnt_le;    // extra pass: keys <= k0
    unsigned long long min_gt;    // extra pass: smallest key > k0
    int need_extra;               // s[r1] lies beyond the last histogram of s[r0]
    int has_nan;
    // value bracket from a sorted sample: keys in [lo, hi] are gathered, keys below lo counted
    unsigned long long lo, hi, below;
    unsigned int n_cand;
    int overflow;                 // more candidates than the buffer holds
    int resolved;                 // both order statistics lie inside the bracket
    int tie;                      // the bracket is ONE key and the counts prove that both order statistics equal it
};

__device__ __forceinline__ void big_ranks(int64_t n, int mode, double q, int64_t *r0, int64_t *r1, double *frac) {
    stat_ranks(n, mode, q, r0, r1, frac);
}

__global__ void k_big_init(BigState *st, const int64_t *seg_len, int64_t len_div, int64_t uniform_len,
                           int64_t n_seg, int mode, double q) {
    const int64_t seg = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (seg >= n_seg) return;
    const int64_t n = seg_len ? seg_len[seg / len_div] : uniform_len;
    int64_t r0 = 0, r1 = 0;
    double frac;
    if (n > 0) big_ranks(n, mode, q, &r0, &r1, &frac);
    BigState s{};
    s.rank = (unsigned long long)r0;
    s.min_gt = ~0ull;
    st[seg] = s;
}

__device__ __forceinline__ void hist_add_matched11(uint32_t *hist, uint32_t digit, bool active) {
    unsigned long long m = __ballot(active);
#pragma unroll
    for (int b = 0; b < 11; ++b) {
        const bool bit = (digit >> b) & 1u;
        const unsigned long long bm = __ballot(bit);
        m &= bit ? bm : ~bm;
    }
    if (active) {
        const int lane = threadIdx.x & 63;
        if (lane == __ffsll((long long)m) - 1) atomicAdd(&hist[digit], (uint32_t)__popcll(m));
    }
}

// pass p histograms bits [shift, shift + width) of the keys whose higher bits equal the prefix
template <typename T, int XF>
__global__ __launch_bounds__(256) void k_big_hist(const T *__restrict__ vals, const int64_t *__restrict__ seg_off,
                                                  int64_t seg_stride, const int64_t *__restrict__ seg_len,
                                                  int64_t len_div, int64_t uniform_len,
                                                  const double *__restrict__ center, BigState *__restrict__ st,
                                                  uint32_t *__restrict__ ghist, int64_t tiles_per_seg, int shift,
                                                  int width, int first) {
    using K = typename KeyT<T>::type;
    __shared__ uint32_t hist[BIG_BINS];
    __shared__ int s_nan;
    const int64_t seg = blockIdx.x / tiles_per_seg, tile = blockIdx.x % tiles_per_seg;
    if (st[seg].resolved) return;
    const int64_t n = seg_len ? seg_len[seg / len_div] : uniform_len;
    const int64_t i0 = tile * BIG_TILE;
    if (i0 >= n) return;
    const int tid = threadIdx.x;
    for (int i = tid; i < BIG_BINS; i += 256) hist[i] = 0;
    if (tid == 0) s_nan = 0;
    __syncthreads();
    const T *v = vals + (seg_off ? seg_off[seg] : seg * seg_stride);
    const T c = XF == XF_ABSDEV ? (T)center[seg] : (T)0;
    const K prefix = (K)st[seg].prefix;
    const uint32_t mask = (1u << width) - 1u;
    bool saw_nan = false;
#pragma unroll 4
    for (int it = 0; it < BIG_TILE / 256; ++it) {
        const int64_t i = i0 + it * 256 + tid;
        const bool in = i < n;
        K k = 0;
        if (in) {
            T x = v[i];
            if (XF == XF_ABSDEV) x = fabs(x - c);
            saw_nan |= (x != x);
            k = x != x ? ~(K)0 : key_of(x);
        }
        const bool match = in && (first || (k >> (shift + width)) == prefix);
        hist_add_matched11(hist, (uint32_t)(k >> shift) & mask, match);
    }
    if (first && saw_nan) s_nan = 1;
    __syncthreads();
    uint32_t *gh = ghist + seg * BIG_BINS;
    for (int i = tid; i < BIG_BINS; i += 256) {
        const uint32_t h = hist[i];
        if (h) atomicAdd(&gh[i], h);
    }
    if (first && tid == 0 && s_nan) st[seg].has_nan = 1;
}

// one workgroup per segment: find the bin of the wanted rank, extend the prefix, clear the histogram
template <typename T>
__global__ __launch_bounds__(256) void k_big_scan(BigState *__restrict__ st, uint32_t *__restrict__ ghist,
                                                  const int64_t *__restrict__ seg_len, int64_t len_div,
                                                  int64_t uniform_len, int width, int last, int mode, double q) {
    __shared__ uint32_t wsum[4];
    __shared__ unsigned long long s_part[256];
    __shared__ int s_bin;
    __shared__ unsigned long long s_before, s_count;
    const int64_t seg = blockIdx.x;
    const int tid = threadIdx.x;
    if (st[seg].resolved) return;
    const int64_t n = seg_len ? seg_len[seg / len_div] : uniform_len;
    uint32_t *gh = ghist + seg * BIG_BINS;
    if (n <= 0) {
        for (int i = tid; i < BIG_BINS; i += 256) gh[i] = 0;
        return;
    }
    // each thread owns 8 consecutive bins
    uint32_t h[8];
    unsigned long long tot = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        h[i] = gh[tid * 8 + i];
        gh[tid * 8 + i] = 0;
        tot += h[i];
    }
    s_part[tid] = tot;
    __syncthreads();
    unsigned long long before = 0;
    for (int i = 0; i < tid; ++i) before += s_part[i];      // 256 short serial sums: negligible
    const unsigned long long r = st[seg].rank;
    if (r >= before && r < before + tot) {
        unsigned long long b = before;
        for (int i = 0; i < 8; ++i) {
            if (r < b + h[i]) {
                s_bin = tid * 8 + i;
                s_before = b;
                s_count = h[i];
                break;
            }
            b += h[i];
        }
    }
    __syncthreads();
    if (last) {
        // the bins of the last pass are whole keys: resolve s[r1] here when it shares the prefix
        int64_t r0, r1;
        double frac;
        big_ranks(n, mode, q, &r0, &r1, &frac);
        __shared__ int s_next;
        if (tid == 0) s_next = BIG_BINS;
        __syncthreads();
        int mine = BIG_BINS;
        for (int i = 7; i >= 0; --i)
            if (h[i] && tid * 8 + i > s_bin) mine = tid * 8 + i;
        if (mine < BIG_BINS) atomicMin(&s_next, mine);
        __syncthreads();
        if (tid == 0) {
            BigState s = st[seg];
            const unsigned long long pre = s.prefix << width;
            s.k0 = pre | (unsigned long long)s_bin;
            const unsigned long long within = s.rank - s_before;       // index of s[r0] inside its bin
            if (r1 == r0 || within + 1 < s_count) {
                s.k1 = s.k0;
            } else if (s_next < BIG_BINS) {
                s.k1 = pre | (unsigned long long)s_next;
            } else {
                s.need_extra = 1;                                      // next key has another prefix
                s.k1 = s.k0;
            }
            st[seg] = s;
        }
    } else if (tid == 0) {
        BigState s = st[seg];
        s.prefix = (s.prefix << width) | (unsigned long long)s_bin;
        s.rank -= s_before;
        st[seg] = s;
    }
}

// rare: s[r1] is the smallest key above k0 and lives under another prefix
template <typename T, int XF>
__global__ __launch_bounds__(256) void k_big_next(const T *__restrict__ vals, const int64_t *__restrict__ seg_off,
                                                  int64_t seg_stride, const int64_t *__restrict__ seg_len,
                                                  int64_t len_div, int64_t uniform_len,
                                                  const double *__restrict__ center, BigState *__restrict__ st,
                                                  int64_t tiles_per_seg) {
    using K = typename KeyT<T>::type;
    const int64_t seg = blockIdx.x / tiles_per_seg, tile = blockIdx.x % tiles_per_seg;
    if (st[seg].resolved || !st[seg].need_extra) return;
    const int64_t n = seg_len ? seg_len[seg / len_div] : uniform_len;
    const int64_t i0 = tile * BIG_TILE;
    if (i0 >= n) return;
    const T *v = vals + (seg_off ? seg_off[seg] : seg * seg_stride);
    const T c = XF == XF_ABSDEV ? (T)center[seg] : (T)0;
    const K k0 = (K)st[seg].k0;
    unsigned long long mn = ~0ull;
    for (int it = 0; it < BIG_TILE / 256; ++it) {
        const int64_t i = i0 + it * 256 + threadIdx.x;
        if (i < n) {
            T x = v[i];
            if (XF == XF_ABSDEV) x = fabs(x - c);
            const K k = x != x ? ~(K)0 : key_of(x);
            if (k > k0 && (unsigned long long)k < mn) mn = k;
        }
    }
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = __shfl_xor(mn, d);
        mn = o < mn ? o : mn;
    }
    if ((threadIdx.x & 63) == 0 && mn != ~0ull) atomicMin(&st[seg].min_gt, mn);
}

template <typename T>
__global__ void k_big_finish(const BigState *__restrict__ st, const int64_t *__restrict__ seg_len, int64_t len_div,
                             int64_t uniform_len, int64_t n_seg, int mode, double q, double *__restrict__ out) {
    using K = typename KeyT<T>::type;
    const int64_t seg = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (seg >= n_seg) return;
    const int64_t n = seg_len ? seg_len[seg / len_div] : uniform_len;
    const BigState s = st[seg];
    if (n <= 0 || s.has_nan) {
        out[seg] = __longlong_as_double(0x7FF8000000000000ll);
        return;
    }
    if (s.resolved && !s.tie) return;    // written by the selection over the gathered keys
    int64_t r0, r1;
    double frac;
    big_ranks(n, mode, q, &r0, &r1, &frac);
    const K k0 = (K)s.k0;
    const K k1 = s.need_extra ? (K)s.min_gt : (K)s.k1;
    const T a = val_of(k0), b = val_of(k1);
    out[seg] = select_result<T>(a, b, r0 == r1, frac, mode);
}

// ---- bracket fast path ---------------------------------------------------------------
// A sorted sample of BIG_SAMPLE keys brackets the wanted order statistics; one pass over the
// segment counts the keys below the bracket and gathers those inside it; the exact selection
// then runs on the few gathered keys.  If the statistics fall outside the bracket (or the
// bracket holds too many keys, e.g. a constant segment) the radix passes above run instead.
static constexpr int BIG_SAMPLE = 4096;
static constexpr int SAMPLE_THREADS = 256;     // threads of the sampling workgroup (one per segment)

// Where the wanted ranks of a list come from.  Level 1 works on the segment itself (all null);
// deeper levels work on the keys gathered by the previous level: the ranks are those of the
// ORIGINAL segment minus the keys already known to lie below the list.
struct RankSrc {
    const int64_t *orig_len;              // per (seg / orig_div), or null: orig_uniform; 0 / null / 0: the list itself
    int64_t orig_div, orig_uniform;
    const unsigned long long *below;      // null: 0
    const int *valid;                     // null, or 0: an earlier level failed, skip the segment
};
__device__ __forceinline__ bool list_ranks(const RankSrc &rs, int64_t seg, int64_t n, int mode, double q,
                                           int64_t *r0, int64_t *r1, double *frac) {
    if (rs.valid && !rs.valid[seg]) return false;
    const int64_t nr = rs.below ? (rs.orig_len ? rs.orig_len[seg / rs.orig_div] : rs.orig_uniform) : n;
    big_ranks(nr, mode, q, r0, r1, frac);
    if (rs.below) {
        *r0 -= (int64_t)rs.below[seg];
        *r1 -= (int64_t)rs.below[seg];
    }
    return true;
}

template <typename T, int XF>
__global__ __launch_bounds__(SAMPLE_THREADS) void k_big_sample(const T *__restrict__ vals, const int64_t *__restrict__ seg_off,
                                                    int64_t seg_stride, const int64_t *__restrict__ seg_len,
                                                    int64_t len_div, int64_t uniform_len,
                                                    const double *__restrict__ center, BigState *__restrict__ st,
                                                    int mode, double q, RankSrc rsrc) {
    using K = typename KeyT<T>::type;
    __shared__ K sk[BIG_SAMPLE];
    __shared__ unsigned int hist[256];
    __shared__ unsigned int s_digit, s_rank;
    const int64_t seg = blockIdx.x;
    const int64_t n = seg_len ? seg_len[seg / len_div] : uniform_len;
    const int tid = threadIdx.x;
    int64_t r0, r1;
    double frac;
    const bool live = n > 0 && list_ranks(rsrc, seg, n, mode, q, &r0, &r1, &frac) && r0 >= 0 && r1 < n;
    if (!live || n < 2 * BIG_SAMPLE) {
        // an earlier level failed: empty bracket (nothing gathered, segment unresolved).
        // A short list of a deeper level passes through whole: bracket = every key.
        if (tid == 0) {
            const bool pass = live && rsrc.below != nullptr;
            st[seg].lo = pass ? 0ull : 1ull;
            st[seg].hi = pass ? ~0ull : 0ull;
        }
        return;
    }
    const T *v = vals + (seg_off ? seg_off[seg] : seg * seg_stride);
    const T c = XF == XF_ABSDEV ? (T)center[seg] : (T)0;
    // stratified sample: BIG_SAMPLE / RUN evenly spaced runs of RUN consecutive elements (one or
    // two cache lines per run instead of one line per sample)
    constexpr int RUN = 16, NRUN = BIG_SAMPLE / RUN;
    {
        // all loads of a thread are issued before the first key is formed
        constexpr int PER = BIG_SAMPLE / SAMPLE_THREADS;
        T xs[PER];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int j = tid + SAMPLE_THREADS * u;
            const int run = j / RUN, w = j % RUN;
            int64_t i = sample_run_start(n, NRUN, RUN, run) + w;
            i = i < 0 ? 0 : (i < n ? i : n - 1);
            xs[u] = v[i];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            T x = xs[u];
            if (XF == XF_ABSDEV) x = fabs(x - c);
            sk[tid + SAMPLE_THREADS * u] = x != x ? ~(K)0 : key_of(x);
        }
    }
    const double p = (double)r0 / (double)n;
    const int delta = (int)(4.5 * sqrt((double)BIG_SAMPLE * p * (1.0 - p))) + 3;
    const int rs = (int)((double)r0 * (double)BIG_SAMPLE / (double)n);
    const int want[2] = {rs - delta, rs + delta + 1};
    K found[2] = {(K)0, ~(K)0};
    // two order statistics of the sample by 8-bit MSD radix selection (no full sort)
    constexpr int BITS = KeyT<T>::bits;
    for (int which = 0; which < 2; ++which) {
        int rank = want[which];
        if (rank < 0 || rank >= BIG_SAMPLE) continue;   // open end of the bracket
        K prefix = 0;
        for (int shift = BITS - 8; shift >= 0; shift -= 8) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            for (int j = tid; j < BIG_SAMPLE; j += SAMPLE_THREADS) {
                const K k = sk[j];
                const bool match = shift == BITS - 8 || ((k ^ prefix) >> (shift + 8 < BITS ? shift + 8 : 0)) == 0;
                if (match) atomicAdd(&hist[(unsigned)(k >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (tid < 64) {
                const unsigned c0 = hist[4 * tid], c1 = hist[4 * tid + 1], c2 = hist[4 * tid + 2], c3 = hist[4 * tid + 3];
                const unsigned sum = c0 + c1 + c2 + c3;
                unsigned incl = sum;
                for (int d = 1; d < 64; d <<= 1) {
                    const unsigned o = __shfl_up(incl, d);
                    if (tid >= d) incl += o;
                }
                const unsigned long long m = __ballot(incl > (unsigned)rank);
                if (m && tid == (int)__ffsll((unsigned long long)m) - 1) {
                    unsigned r = (unsigned)rank - (incl - sum);
                    int dgt = 4 * tid;
                    if (r >= c0) { r -= c0; ++dgt;
                        if (r >= c1) { r -= c1; ++dgt;
                            if (r >= c2) { r -= c2; ++dgt; } } }
                    s_digit = (unsigned)dgt;
                    s_rank = r;
                }
            }
            __syncthreads();
            prefix |= (K)s_digit << shift;
            rank = (int)s_rank;
        }
        found[which] = prefix;
    }
    if (tid == 0) {
        st[seg].lo = want[0] < 0 ? 0ull : (unsigned long long)found[0];
        st[seg].hi = want[1] >= BIG_SAMPLE ? ~0ull : (unsigned long long)found[1];
    }
}

template <typename T, int XF>
__global__ __launch_bounds__(256) void k_big_gather(const T *__restrict__ vals, const int64_t *__restrict__ seg_off,
                                                    int64_t seg_stride, const int64_t *__restrict__ seg_len,
                                                    int64_t len_div, int64_t uniform_len,
                                                    const double *__restrict__ center, BigState *__restrict__ st,
                                                    T *__restrict__ cand, int64_t cap, int64_t tiles_per_seg,
                                                    int count_ties) {
    using K = typename KeyT<T>::type;
    constexpr int STAGE = 2048;                    // candidates staged per workgroup before one append
    __shared__ T stage[STAGE];
    __shared__ unsigned long long s_below;
    __shared__ unsigned int s_fill, s_base;
    __shared__ int s_nan;
    const int64_t seg = blockIdx.x / tiles_per_seg, tile = blockIdx.x % tiles_per_seg;
    const int64_t n = seg_len ? seg_len[seg / len_div] : uniform_len;
    const int64_t i0 = tile * GATHER_TILE;
    if (i0 >= n) return;
    const K lo = (K)st[seg].lo, hi = (K)st[seg].hi;
    if (lo > hi) return;
    // a bracket of ONE key (constant or heavily tied segment): its keys are counted, not gathered
    const bool tie = count_ties && lo == hi;
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid == 0) {
        s_below = 0;
        s_nan = 0;
        s_fill = 0;
    }
    __syncthreads();
    const T *v = vals + (seg_off ? seg_off[seg] : seg * seg_stride);
    const T c = XF == XF_ABSDEV ? (T)center[seg] : (T)0;
    T *dst = cand + seg * cap;
    if (lo == 0 && hi == ~(K)0) {
        // a short list that passes through this level whole (k_big_sample): copied in place, no staging (a tile of
        // such a list holds more keys than the stage)
        const int64_t i1 = i0 + GATHER_TILE < n ? i0 + GATHER_TILE : n;
        bool nan_here = false, over = false;
        for (int64_t i = i0 + tid; i < i1; i += 256) {
            T x = v[i];
            if (XF == XF_ABSDEV) x = fabs(x - c);
            nan_here |= x != x;
            if (i < cap) dst[i] = x;
            else over = true;
        }
        if (nan_here) st[seg].has_nan = 1;
        if (over) st[seg].overflow = 1;
        if (tid == 0) atomicAdd(&st[seg].n_cand, (unsigned int)(i1 - i0));
        return;
    }
    unsigned int below = 0;
    bool saw_nan = false, spilled = false;
    // batches of BATCH elements per thread: all loads of a batch are in flight together, the
    // keep decisions are one bit each, and a wave appends with ONE LDS atomic per batch.
    // A tile that lies inside its segment and starts on a 16-byte boundary takes 16-byte loads
    // and no bounds checks (the order of the gathered keys does not matter).
    constexpr int BATCH = 32, VEC = 16 / (int)sizeof(T);
    const bool full = i0 + GATHER_TILE <= n && (reinterpret_cast<uintptr_t>(v + i0) & 15) == 0;
    for (int b0 = 0; b0 < GATHER_TILE / 256; b0 += BATCH) {
        if (i0 + (int64_t)b0 * 256 >= n) break;
        T xs[BATCH];
        bool keep[BATCH];
        if (full) {
            const float4 *v4 = reinterpret_cast<const float4 *>(v + i0 + (int64_t)b0 * 256);
            float4 q[BATCH / VEC];
#pragma unroll
            for (int u = 0; u < BATCH / VEC; ++u) q[u] = v4[u * 256 + tid];
#pragma unroll
            for (int u = 0; u < BATCH / VEC; ++u) {
                const T *e = reinterpret_cast<const T *>(&q[u]);
#pragma unroll
                for (int w = 0; w < VEC; ++w) xs[u * VEC + w] = e[w];
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                T x = xs[u];
                if (XF == XF_ABSDEV) x = fabs(x - c);
                xs[u] = x;
                saw_nan |= (x != x);
                const K k = x != x ? ~(K)0 : key_of(x);
                below += (k < lo);
                keep[u] = (K)(k - lo) <= (K)(hi - lo);
            }
        } else {
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int64_t i = i0 + (int64_t)(b0 + u) * 256 + tid;
                xs[u] = i < n ? v[i] : (T)0;
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int64_t i = i0 + (int64_t)(b0 + u) * 256 + tid;
                const bool in = i < n;
                T x = xs[u];
                if (XF == XF_ABSDEV) x = fabs(x - c);
                xs[u] = x;
                saw_nan |= in && (x != x);
                const K k = x != x ? ~(K)0 : key_of(x);
                below += (in && k < lo);
                keep[u] = in && k >= lo && k <= hi;
            }
        }
        // positions from wave ballots (no cross-lane round trips): key u of lane l goes after the
        // kept keys u' < u of the whole wave and the kept keys u of the lanes below l
        unsigned long long bal[BATCH];
        unsigned int total = 0;
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            bal[u] = __ballot(keep[u]);
            total += (unsigned int)__popcll(bal[u]);
        }
        if (total == 0) continue;
        unsigned int base = 0;
        if (lane == 0) base = atomicAdd(&s_fill, total);
        base = __builtin_amdgcn_readfirstlane(base);
        if (tie) continue;
        if (base + total <= STAGE) {
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                if (keep[u])
                    stage[base + __builtin_amdgcn_mbcnt_hi((unsigned int)(bal[u] >> 32),
                                                           __builtin_amdgcn_mbcnt_lo((unsigned int)bal[u], 0u))] = xs[u];
                base += (unsigned int)__popcll(bal[u]);
            }
        } else {
            spilled = true;                          // far more candidates than a bracket should hold
        }
    }
    for (int d = 32; d >= 1; d >>= 1) below += __shfl_xor(below, d);
    if (lane == 0 && below) atomicAdd(&s_below, (unsigned long long)below);
    if (saw_nan) s_nan = 1;
    if (spilled) st[seg].overflow = 1;
    __syncthreads();
    const unsigned int cnt = tie ? 0u : (s_fill < STAGE ? s_fill : STAGE);
    if (tid == 0) {
        s_base = s_fill ? atomicAdd(&st[seg].n_cand, s_fill) : 0;   // s_fill (not cnt): overflow stays visible
        if (s_below) atomicAdd(&st[seg].below, s_below);
        if (s_nan) st[seg].has_nan = 1;
    }
    __syncthreads();
    const int64_t base = s_base;
    for (unsigned int p = tid; p < cnt; p += 256) {
        if (base + p < cap) dst[base + p] = stage[p];
        else st[seg].overflow = 1;
    }
}

// per segment: did the bracket catch both order statistics?
__global__ void k_big_check(BigState *st, const int64_t *seg_len, int64_t len_div, int64_t uniform_len,
                            int64_t n_seg, int mode, double q, int64_t cap, RankSrc rsrc, int64_t *cand_len,
                            int *valid, unsigned long long *below, unsigned long long *lohi) {
    const int64_t seg = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (seg >= n_seg) return;
    const int64_t n = seg_len ? seg_len[seg / len_div] : uniform_len;
    BigState s = st[seg];
    int ok = 0, tie = 0;
    const bool one_key = rsrc.below == nullptr && s.lo == s.hi;     // first level only: counted, nothing gathered
    if (n > 0 && s.lo <= s.hi && !s.has_nan && (one_key || (!s.overflow && (int64_t)s.n_cand <= cap))) {
        int64_t r0, r1;
        double frac;
        if (list_ranks(rsrc, seg, n, mode, q, &r0, &r1, &frac))
            ok = r0 >= 0 && (unsigned long long)r0 >= s.below && (unsigned long long)r1 < s.below + s.n_cand;
    }
    if (one_key) {
        // both order statistics are that key (or the radix passes take the segment): no list to select from
        tie = ok;
        ok = 0;
        if (tie) {
            st[seg].k0 = s.lo;
            st[seg].k1 = s.lo;
            st[seg].need_extra = 0;
            st[seg].tie = 1;
        }
    }
    st[seg].resolved = ok || tie;
    valid[seg] = ok;
    cand_len[seg] = ok ? (int64_t)s.n_cand : 0;
    below[seg] = s.below + (rsrc.below ? rsrc.below[seg] : 0ull);   // keys of the ORIGINAL segment below this list
    lohi[2 * seg] = s.lo;
    lohi[2 * seg + 1] = s.hi;
}

// level >= 2 state: empty counters
__global__ void k_big_init_level(BigState *st, int64_t n_seg) {
    const int64_t seg = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (seg >= n_seg) return;
    BigState s{};
    s.min_gt = ~0ull;
    st[seg] = s;
}
// the radix passes (on the original segment) must run for everything the last level left open
__global__ void k_big_resolve(BigState *st, const int *valid, int64_t n_seg) {
    const int64_t seg = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (seg < n_seg) st[seg].resolved = valid[seg] || st[seg].tie;
}

int64_t select_short_limit() { return BIG_THRESHOLD; }

// Capacity (keys) of a segment's candidate list.  The 4096-key sample brackets about 7 % of the
// segment (2 * 4.5 sigma of a binomial rank); 30 % head-room on top.  While that fits the lean LDS
// kernel the capacity is kept that small (its LDS is sized by it: occupancy); beyond, the list is
// an eighth of the segment and the general kernel reads it from memory.  A segment that gathers
// more (many equal keys, e.g. silence) is flagged and takes the radix passes instead.
static int64_t big_cand_cap(int64_t max_len, size_t key_bytes) {
    const int64_t lds_cap = (int64_t)(SEL_LDS_MAX_BYTES / key_bytes);
    const int64_t expect = (int64_t)(0.0703 * 1.3 * (double)max_len) + 1024;
    if ((int64_t)(0.0703 * 1.05 * (double)max_len) + 256 <= lds_cap) return expect < lds_cap ? (expect < 4096 ? 4096 : expect) : lds_cap;
    const int64_t c = max_len / 8 + 1024;
    return c < 16384 ? 16384 : c;
}

// Candidate capacities of the bracket levels: a level whose list would not fit the lean LDS
// kernel is bracketed again (sample -> gather) on the gathered keys, about 14x shorter each time.
static constexpr int BIG_MAX_LEVELS = 7;
static constexpr size_t SEL_LEVEL_STOP_BYTES = 32 * 1024;
static int big_level_caps(int64_t max_len, size_t key_bytes, int64_t *caps) {
    int n = 0;
    int64_t len = max_len;
    while (n < BIG_MAX_LEVELS) {
        int64_t cap = big_cand_cap(len, key_bytes);
        if (cap < 2 * BIG_SAMPLE) cap = 2 * BIG_SAMPLE;      // a short list passes through a level whole
        caps[n++] = cap;
        // stop once the list is small enough for several selecting workgroups to share a CU
        if ((size_t)cap * key_bytes <= SEL_LEVEL_STOP_BYTES) break;
        len = cap;
    }
    return n;
}

size_t select_ws_bytes(int64_t n_seg, int64_t max_len) {
    if (max_len <= BIG_THRESHOLD) return 0;
    int64_t c4[BIG_MAX_LEVELS], c8[BIG_MAX_LEVELS];
    const int n4 = big_level_caps(max_len, 4, c4), n8 = big_level_caps(max_len, 8, c8);
    size_t cand = 0;
    const int nl = std::max(n4, n8);
    for (int l = 0; l < nl; ++l)
        cand += ws_need((size_t)n_seg * std::max(l < n4 ? c4[l] : 0, l < n8 ? c8[l] : 0), 8);
    return ws_need((size_t)n_seg * BIG_BINS, sizeof(uint32_t)) + cand +
           (size_t)nl * (ws_need((size_t)n_seg, sizeof(BigState)) + 4 * ws_need((size_t)n_seg, 8) + ws_need((size_t)n_seg, 4));
}

template <typename T, int XF>
static int launch_select_big(Ctx *c, const T *d_vals, const SelectArgs &a) {
    constexpr int BITS = KeyT<T>::bits;
    using K = typename KeyT<T>::type;
    int64_t caps[BIG_MAX_LEVELS];
    const int n_levels = big_level_caps(a.max_len, sizeof(K), caps);
    uint32_t *ghist = ws_take<uint32_t>(c, (size_t)a.n_seg * BIG_BINS);
    BigState *st_l[BIG_MAX_LEVELS];
    T *cand_l[BIG_MAX_LEVELS];
    int64_t *len_l[BIG_MAX_LEVELS];
    unsigned long long *below_l[BIG_MAX_LEVELS], *lohi_l[BIG_MAX_LEVELS];
    int *valid_l[BIG_MAX_LEVELS];
    for (int l = 0; l < n_levels; ++l) {
        st_l[l] = ws_take<BigState>(c, (size_t)a.n_seg);
        cand_l[l] = reinterpret_cast<T *>(ws_take<double>(c, (size_t)a.n_seg * caps[l]));
        len_l[l] = ws_take<int64_t>(c, (size_t)a.n_seg);
        below_l[l] = ws_take<unsigned long long>(c, (size_t)a.n_seg);
        valid_l[l] = ws_take<int>(c, (size_t)a.n_seg);
        lohi_l[l] = ws_take<unsigned long long>(c, 2 * (size_t)a.n_seg);
    }
    BigState *st = st_l[0];
    if (c->ws_used > c->ws_bytes) {
        set_error("internal: workspace not reserved for the long-segment select");
        return ADN_ERR_NOMEM;
    }
    const int64_t tiles = ceil_div(a.max_len, BIG_TILE);
    ADN_REQUIRE(a.n_seg * tiles < (int64_t)0x7FFFFFFF, "select batch too large for one launch");
    const unsigned nb = (unsigned)ceil_div(a.n_seg, 256);
    ADN_HIP(hipMemsetAsync(ghist, 0, sizeof(uint32_t) * a.n_seg * BIG_BINS, c->stream));
    ADN_LAUNCH(k_big_init, dim3(nb), dim3(256), 0, c->stream, st, a.d_seg_len, a.len_div, a.uniform_len,
                       a.n_seg, a.mode, a.q);
    // ---- bracket fast path: sample, count + gather (repeated on the gathered keys while they do
    // not fit LDS), then the exact selection over the last list
    {
        const RankSrc none{nullptr, 1, 0, nullptr, nullptr};
        ADN_LAUNCH((k_big_sample<T, XF>), dim3((unsigned)a.n_seg), dim3(SAMPLE_THREADS), 0, c->stream, d_vals, a.d_seg_off,
                           a.seg_stride, a.d_seg_len, a.len_div, a.uniform_len, a.d_center, st, a.mode, a.q, none);
        const int64_t gtiles = ceil_div(a.max_len, GATHER_TILE);
        ADN_REQUIRE(a.n_seg * gtiles < (int64_t)0x7FFFFFFF, "select batch too large for one launch");
        ADN_LAUNCH((k_big_gather<T, XF>), dim3((unsigned)(a.n_seg * gtiles)), dim3(256), 0, c->stream, d_vals,
                           a.d_seg_off, a.seg_stride, a.d_seg_len, a.len_div, a.uniform_len, a.d_center, st, cand_l[0],
                           caps[0], gtiles, 1);
        ADN_LAUNCH(k_big_check, dim3(nb), dim3(256), 0, c->stream, st, a.d_seg_len, a.len_div, a.uniform_len,
                           a.n_seg, a.mode, a.q, caps[0], none, len_l[0], valid_l[0], below_l[0], lohi_l[0]);
    }
    for (int l = 1; l < n_levels; ++l) {
        const RankSrc rs{a.d_seg_len, a.len_div, a.uniform_len, below_l[l - 1], valid_l[l - 1]};
        const int64_t ltiles = ceil_div(caps[l - 1], GATHER_TILE);
        ADN_REQUIRE(a.n_seg * ltiles < (int64_t)0x7FFFFFFF, "select batch too large for one launch");
        ADN_LAUNCH(k_big_init_level, dim3(nb), dim3(256), 0, c->stream, st_l[l], a.n_seg);
        ADN_LAUNCH((k_big_sample<T, XF_NONE>), dim3((unsigned)a.n_seg), dim3(SAMPLE_THREADS), 0, c->stream, cand_l[l - 1],
                           (const int64_t *)nullptr, caps[l - 1], len_l[l - 1], (int64_t)1, (int64_t)0,
                           (const double *)nullptr, st_l[l], a.mode, a.q, rs);
        ADN_LAUNCH((k_big_gather<T, XF_NONE>), dim3((unsigned)(a.n_seg * ltiles)), dim3(256), 0, c->stream,
                           cand_l[l - 1], (const int64_t *)nullptr, caps[l - 1], len_l[l - 1], (int64_t)1, (int64_t)0,
                           (const double *)nullptr, st_l[l], cand_l[l], caps[l], ltiles, 0);
        ADN_LAUNCH(k_big_check, dim3(nb), dim3(256), 0, c->stream, st_l[l], len_l[l - 1], (int64_t)1, (int64_t)0,
                           a.n_seg, a.mode, a.q, caps[l], rs, len_l[l], valid_l[l], below_l[l], lohi_l[l]);
    }
    const int L = n_levels - 1;
    if (L > 0) ADN_LAUNCH(k_big_resolve, dim3(nb), dim3(256), 0, c->stream, st, valid_l[L], a.n_seg);
    {
        const int64_t cap = caps[L];
        T *cand = cand_l[L];
        MaskedMean mm{nullptr, 1, 0, nullptr};
        SubsetRanks sub{a.d_seg_len, a.len_div, a.uniform_len, below_l[L], valid_l[L], lohi_l[L]};
        if ((size_t)cap * sizeof(K) <= SEL_LDS_MAX_BYTES) {
            const size_t dyn = (size_t)cap * sizeof(K);
            auto kern = k_select_lds<T, XF_NONE, false>;
            ADN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
            ADN_LAUNCH_NAMED("k_select_lds<T, XF_NONE, false>", kern, dim3((unsigned)a.n_seg), dim3(SEL_THREADS), dyn, c->stream, cand,
                               (const int64_t *)nullptr, cap, len_l[L], (int64_t)1, (int64_t)0, a.mode, a.q,
                               (const double *)nullptr, a.d_out, mm, sub);
        } else {
            int lds_keys = (int)(SEL_LDS_BYTES / sizeof(K));
            const size_t dyn = 2 * (size_t)lds_keys * sizeof(K);
            auto kern = k_select<T, XF_NONE>;
            ADN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
            ADN_LAUNCH_NAMED("k_select<T, XF_NONE>", kern, dim3((unsigned)a.n_seg), dim3(SEL_THREADS), dyn, c->stream, cand,
                               (const int64_t *)nullptr, cap, len_l[L], (int64_t)1, (int64_t)0, a.mode, a.q,
                               (const double *)nullptr, a.d_out, lds_keys, mm, sub);
        }
    }
    // ---- exact radix passes for whatever the bracket did not resolve (early exit otherwise)
    int remaining = BITS;
    bool first = true;
    while (remaining > 0) {
        // widths 11, 11, 10 (32-bit keys) / 11 x 4, 10 x 2 (64-bit keys)
        const int passes_left = (remaining + 10) / 11;
        const int width = (remaining + passes_left - 1) / passes_left;
        remaining -= width;
        ADN_LAUNCH((k_big_hist<T, XF>), dim3((unsigned)(a.n_seg * tiles)), dim3(256), 0, c->stream, d_vals,
                           a.d_seg_off, a.seg_stride, a.d_seg_len, a.len_div, a.uniform_len, a.d_center, st, ghist,
                           tiles, remaining, width, first ? 1 : 0);
        ADN_LAUNCH((k_big_scan<T>), dim3((unsigned)a.n_seg), dim3(256), 0, c->stream, st, ghist,
                           a.d_seg_len, a.len_div, a.uniform_len, width, remaining == 0 ? 1 : 0, a.mode, a.q);
        first = false;
    }
    ADN_LAUNCH((k_big_next<T, XF>), dim3((unsigned)(a.n_seg * tiles)), dim3(256), 0, c->stream, d_vals,
                       a.d_seg_off, a.seg_stride, a.d_seg_len, a.len_div, a.uniform_len, a.d_center, st, tiles);
    ADN_LAUNCH((k_big_finish<T>), dim3(nb), dim3(256), 0, c->stream, st, a.d_seg_len, a.len_div,
                       a.uniform_len, a.n_seg, a.mode, a.q, a.d_out);
    ADN_HIP(hipGetLastError());
    return ADN_OK;
}

// ----------------------------------------------------------------------------
// Order statistics of a segment that is SPLIT over several devices (or over several calls): the
// selection is an 8-bit MSD radix descent whose only shared state is, per segment, the key prefix
// decided so far and the rank that remains below it.  Every holder of a part histograms the next
// digit of its keys under the prefix (select_hist); the histograms are summed outside (an
// all-reduce of n_seg x 256 counters); every holder then takes the same decision from the sum
// (select_resolve).  After key-bits / 8 rounds the prefix is the key of the wanted order statistic.
template <typename T, int XF>
__global__ __launch_bounds__(256) void k_part_hist(const T *__restrict__ vals, const int64_t *__restrict__ seg_off,
                                                   const int64_t *__restrict__ seg_len, const double *__restrict__ center,
                                                   const unsigned long long *__restrict__ prefix, int done_bits,
                                                   unsigned int *__restrict__ hist, int64_t tiles_per_seg) {
    using K = typename KeyT<T>::type;
    constexpr int BITS = KeyT<T>::bits;
    __shared__ unsigned int lh[256];
    const int64_t seg = blockIdx.x / tiles_per_seg, tile = blockIdx.x % tiles_per_seg;
    const int64_t n = seg_len[seg];
    lh[threadIdx.x] = 0;
    __syncthreads();
    const T c = XF == XF_ABSDEV ? (T)center[seg] : (T)0;
    const T *v = vals + seg_off[seg];
    const K pre = (K)prefix[seg];
    const int shift = BITS - done_bits - 8;                 // position of the digit to histogram
    const int64_t i0 = tile * 4096, i1 = i0 + 4096 < n ? i0 + 4096 : n;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
        T x = v[i];
        if (XF == XF_ABSDEV) x = fabs(x - c);
        const K k = x != x ? ~(K)0 : key_of(x);
        const bool match = done_bits == 0 || (k >> (BITS - done_bits)) == (pre >> (BITS - done_bits));
        if (match) atomicAdd(&lh[(unsigned)(k >> shift) & 255u], 1u);
    }
    __syncthreads();
    const unsigned int h = lh[threadIdx.x];
    if (h) atomicAdd(&hist[seg * 256 + threadIdx.x], h);
}

__global__ void k_part_resolve(const unsigned int *__restrict__ hist, int64_t n_seg, int key_bits, int done_bits,
                               long long *__restrict__ rank, unsigned long long *__restrict__ prefix) {
    const int64_t seg = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (seg >= n_seg) return;
    long long r = rank[seg];
    const unsigned int *h = hist + seg * 256;
    int d = 0;
    long long below = 0;
    for (; d < 255; ++d) {
        if (r < below + (long long)h[d]) break;
        below += h[d];
    }
    rank[seg] = r - below;
    prefix[seg] |= (unsigned long long)d << (key_bits - done_bits - 8);
}

template <typename T>
__global__ void k_part_value(const unsigned long long *__restrict__ prefix, int64_t n_seg, double *__restrict__ out) {
    using K = typename KeyT<T>::type;
    const int64_t seg = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (seg >= n_seg) return;
    const K k = (K)prefix[seg];
    out[seg] = k == ~(K)0 ? __longlong_as_double(0x7FF8000000000000ll) : (double)val_of(k);
}

template <typename T>
int select_hist(Ctx *c, const T *d_vals, int64_t n_seg, const int64_t *d_off, const int64_t *d_len, int64_t max_len,
                int absdev, const double *d_center, const unsigned long long *d_prefix, int done_bits,
                unsigned int *d_hist) {
    ADN_REQUIRE(c && d_vals && d_off && d_len && d_prefix && d_hist, "null pointer argument");
    ADN_REQUIRE(!absdev || d_center, "the absolute-deviation transform needs the centres");
    ADN_REQUIRE(done_bits >= 0 && done_bits % 8 == 0 && done_bits < KeyT<T>::bits, "done_bits must be a multiple of 8 below %d",
                KeyT<T>::bits);
    ADN_REQUIRE(n_seg >= 0 && max_len >= 0, "negative size");
    if (n_seg == 0) return ADN_OK;
    DeviceGuard dg(c->device);
    ADN_HIP(hipMemsetAsync(d_hist, 0, sizeof(unsigned int) * 256 * n_seg, c->stream));
    const int64_t tiles = ceil_div(max_len > 0 ? max_len : 1, (int64_t)4096);
    ADN_REQUIRE(n_seg * tiles < (int64_t)0x7FFFFFFF, "batch too large for one launch");
    if (absdev)
        ADN_LAUNCH((k_part_hist<T, XF_ABSDEV>), dim3((unsigned)(n_seg * tiles)), dim3(256), 0, c->stream, d_vals, d_off, d_len,
                   d_center, d_prefix, done_bits, d_hist, tiles);
    else
        ADN_LAUNCH((k_part_hist<T, XF_NONE>), dim3((unsigned)(n_seg * tiles)), dim3(256), 0, c->stream, d_vals, d_off, d_len,
                   d_center, d_prefix, done_bits, d_hist, tiles);
    ADN_HIP(hipGetLastError());
    return ADN_OK;
}

int select_resolve(Ctx *c, const unsigned int *d_hist, int64_t n_seg, int key_bits, int done_bits, long long *d_rank,
                   unsigned long long *d_prefix) {
    ADN_REQUIRE(c && d_hist && d_rank && d_prefix, "null pointer argument");
    ADN_REQUIRE((key_bits == 32 || key_bits == 64) && done_bits >= 0 && done_bits % 8 == 0 && done_bits < key_bits,
                "key_bits must be 32 or 64 and done_bits a multiple of 8 below it");
    if (n_seg <= 0) return ADN_OK;
    DeviceGuard dg(c->device);
    ADN_LAUNCH(k_part_resolve, dim3((unsigned)ceil_div(n_seg, 256)), dim3(256), 0, c->stream, d_hist, n_seg, key_bits,
               done_bits, d_rank, d_prefix);
    ADN_HIP(hipGetLastError());
    return ADN_OK;
}

template <typename T>
int select_value(Ctx *c, const unsigned long long *d_prefix, int64_t n_seg, double *d_out) {
    ADN_REQUIRE(c && d_prefix && d_out, "null pointer argument");
    if (n_seg <= 0) return ADN_OK;
    DeviceGuard dg(c->device);
    ADN_LAUNCH((k_part_value<T>), dim3((unsigned)ceil_div(n_seg, 256)), dim3(256), 0, c->stream, d_prefix, n_seg, d_out);
    ADN_HIP(hipGetLastError());
    return ADN_OK;
}

// The statistic from its two neighbouring order statistics, with the very arithmetic of the one-call kernels:
// the median of an even count (q < 0) or numpy's linear interpolation at percentile q of n values.
template <typename T>
__global__ void k_part_combine(const double *__restrict__ a, const double *__restrict__ b, int64_t n_seg, bool same,
                               double frac, int mode, double *__restrict__ out) {
    const int64_t seg = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (seg >= n_seg) return;
    out[seg] = select_result<T>((T)a[seg], (T)b[seg], same, frac, mode);
}

template <typename T>
int select_combine(Ctx *c, const double *d_lo, const double *d_hi, int64_t n_seg, int64_t n_values, double q,
                   double *d_out) {
    ADN_REQUIRE(c && d_lo && d_hi && d_out, "null pointer argument");
    ADN_REQUIRE(n_values >= 1 && q <= 100.0, "a statistic of %lld values at percentile %g", (long long)n_values, q);
    if (n_seg <= 0) return ADN_OK;
    const int mode = q < 0.0 ? SEL_MEDIAN : SEL_PERCENTILE;
    int64_t r0, r1;
    double frac;
    stat_ranks(n_values, mode, q, &r0, &r1, &frac);
    DeviceGuard dg(c->device);
    ADN_LAUNCH((k_part_combine<T>), dim3((unsigned)ceil_div(n_seg, 256)), dim3(256), 0, c->stream, d_lo, d_hi, n_seg,
               r0 == r1, frac, mode, d_out);
    ADN_HIP(hipGetLastError());
    return ADN_OK;
}
template int select_combine<float>(Ctx *, const double *, const double *, int64_t, int64_t, double, double *);
template int select_combine<double>(Ctx *, const double *, const double *, int64_t, int64_t, double, double *);

template int select_hist<float>(Ctx *, const float *, int64_t, const int64_t *, const int64_t *, int64_t, int, const double *,
                                const unsigned long long *, int, unsigned int *);
template int select_hist<double>(Ctx *, const double *, int64_t, const int64_t *, const int64_t *, int64_t, int,
                                 const double *, const unsigned long long *, int, unsigned int *);
template int select_value<float>(Ctx *, const unsigned long long *, int64_t, double *);
template int select_value<double>(Ctx *, const unsigned long long *, int64_t, double *);

// ----------------------------------------------------------------------------
// Rows of up to a few thousand binary32 values: one wave per segment, the keys held in
// registers (KPL per lane).  The wanted order statistic is found bit by bit on the
// range-normalised keys: for each bit, from the top, the number of keys below the trial
// value is one vector compare per held key (the compare's lane mask goes to the scalar
// unit, which counts and adds while the vector unit runs the next compare).  No LDS, no
// barrier, no atomics; rows of one workgroup are independent of each other.
static constexpr int SEL_CK = 8;                        // candidate keys per lane after compaction

// Number of keys of the wave (N per lane) below `trial`.  Two ways of counting share the work, because each is
// bound by another unit: (a) vector compare -> lane mask in scalar registers -> scalar bit count and addition (one
// vector, two scalar instructions per key; a SIMD gets one scalar issue slot in four cycles); (b) vector compare ->
// add-with-carry into a per-lane counter (two vector instructions per key), summed over the lanes once per call
// from the ballots of the counter's bits.  Per twelve keys: eight by (a), four by (b), which loads both units
// alike.  (a) runs in groups of eight compares followed by their eight counts: a scalar instruction that reads
// a register the vector unit has just written waits for that instruction to leave the vector pipeline, and with
// the compares grouped the wait is paid once per group.
__device__ __forceinline__ void count_add_vector(uint32_t &vc, uint32_t key, uint32_t trial) {
    asm volatile("v_cmp_gt_u32_e32 vcc, %2, %1\n\tv_addc_co_u32_e32 %0, vcc, 0, %0, vcc" : "+v"(vc) : "v"(key), "s"(trial) : "vcc");
}
__device__ __forceinline__ void count_add_vector(uint32_t &vc, uint64_t key, uint64_t trial) {
    asm volatile("v_cmp_gt_u64_e32 vcc, %2, %1\n\tv_addc_co_u32_e32 %0, vcc, 0, %0, vcc" : "+v"(vc) : "v"(key), "s"(trial) : "vcc");
}

template <typename K, int N>
__device__ __forceinline__ uint32_t count_below(const K (&key)[N], K trial) {
    uint32_t cnt = 0, vc = 0;
#pragma unroll
    for (int g = 0; g < N; g += 12) {
        unsigned long long m[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) m[i] = g + i < N ? __ballot(key[g + i] < trial) : 0ull;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 8; i < 12; ++i)
            if (g + i < N) count_add_vector(vc, key[g + i], trial);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (g + i < N) cnt += (uint32_t)__builtin_popcountll(m[i]);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (N > 8) {
        constexpr int VB = N >= 12 * 16 ? 7 : (N >= 12 * 8 ? 6 : (N >= 12 * 4 ? 5 : 4));      // bits of a lane's counter
#pragma unroll
        for (int k = 0; k < VB; ++k) cnt += (uint32_t)__builtin_popcountll(__ballot((vc >> k) & 1u)) << k;
    }
    return cnt;
}

// four consecutive values of a row as one load
template <typename T> struct Quad4;
template <> struct Quad4<float> { typedef float4 type; };
template <> struct Quad4<double> { typedef double4 type; };

// T: binary32 (keys of 32 bits, three waves per SIMD up to 96 keys per lane) or binary64 (64-bit keys: twice the registers)
//
// NW > 1: a row longer than one wave's registers is shared by the NW waves of a workgroup (one row per workgroup).
// Wave w holds quads w * KPL / 4 * 64 ... of the row; every count of the search is the sum of the waves' counts,
// exchanged through LDS with one barrier per decided bit (all waves take the same decisions); once the candidates
// fit SEL_CK registers per lane every wave holds all of them and finishes the search alone.
template <typename T, int XF, bool MASKED, int KPL, int NW = 1>
__global__ __launch_bounds__(NW <= 4 ? 256 : NW * 64)
__attribute__((amdgpu_waves_per_eu(sizeof(T) == 8 ? (KPL <= 96 ? 2 : 1) : (KPL <= 88 ? 4 : (KPL <= 96 ? 3 : 2)))))
void k_select_wave(const T *__restrict__ vals, const int64_t *__restrict__ seg_off,
                                                     int64_t seg_stride, const int64_t *__restrict__ seg_len,
                                                     int64_t len_div, int64_t uniform_len, int64_t n_seg, int mode,
                                                     double q, const double *__restrict__ center,
                                                     double *__restrict__ out, MaskedMean mm, int max_bits,
                                                     int64_t quad_stride, int64_t blk_rows, int64_t blk_sig_stride,
                                                     const int *__restrict__ done) {
    typedef typename KeyT<T>::type K;
    typedef typename Quad4<T>::type Q4;
    constexpr K KEY_MAX = ~(K)0;
    const int lane = threadIdx.x & 63;
    // the wave index as a scalar: the row's addresses then live in scalar registers and every load of the row
    // takes one 32-bit lane offset instead of a 64-bit address pair
    // Rows in groups of four values: the two workgroups that share the 128-byte lines of eight neighbouring rows
    // must meet in one L2, and consecutive workgroups go to different XCDs: workgroup b (XCD b mod 8) takes the
    // rows of logical workgroup ((b / 16) * 8 + b mod 8) * 2 + (b / 8) mod 2.
    int64_t wg = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int64_t seg;
    if (NW == 1) {
        if (quad_stride != 1 && (int64_t)gridDim.x % 16 == 0) wg = ((wg >> 4) * 8 + (wg & 7)) * 2 + ((wg >> 3) & 1);
        seg = wg * 4 + wave;
    } else {
        // one row per workgroup: the eight rows that share 128-byte lines go to workgroups b, b + 8, ..., b + 56
        // (one XCD) of a block of 64 workgroups
        if (quad_stride != 1 && (int64_t)gridDim.x % 64 == 0) wg = ((wg >> 6) << 6) + ((wg & 7) << 3) + ((wg >> 3) & 7);
        seg = wg;
    }
    const int wv = NW == 1 ? 0 : wave;                      // this wave's slice of the row
    if (seg >= n_seg) return;
    if (done != nullptr && done[seg]) return;               // the row has its result already (rows_stream below)
    // exchange between the waves of a row (NW > 1).  Sums and minima: every wave adds to / lowers one LDS word, a
    // barrier, every wave reads the word; three words used in turn, the next one cleared by the call before, so that
    // one barrier per exchange is enough (a wave can be at most one exchange ahead of the slowest).
    __shared__ unsigned long long sh_min[3], sh_a[NW], sh_b[NW], sh_m[NW];
    __shared__ uint32_t sh_sum[3], sh_c[NW], sh_d[NW];
    __shared__ uint32_t sh_ctr;
    __shared__ K cbuf_all[NW == 1 ? 4 : 1][SEL_CK * 64];
    int turn = 0;
    if (NW > 1 && threadIdx.x == 0) {
        sh_ctr = 0;
        sh_sum[0] = 0;
        sh_min[0] = ~0ull;
    }
    auto next_turn = [&]() {
        const int nxt = turn == 2 ? 0 : turn + 1;
        if (threadIdx.x == 0) {
            sh_sum[nxt] = 0;
            sh_min[nxt] = ~0ull;
        }
        return nxt;
    };
    auto wg_sum = [&](uint32_t v) -> uint32_t {             // v: one value per wave
        const int nxt = next_turn();
        if (lane == 0) atomicAdd(&sh_sum[turn], v);
        __syncthreads();
        const uint32_t sum = sh_sum[turn];
        turn = nxt;
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)sum);
    };
    auto wg_min = [&](K v) -> K {
        const int nxt = next_turn();
        if (lane == 0) atomicMin(&sh_min[turn], (unsigned long long)v);
        __syncthreads();
        const K m = (K)sh_min[turn];
        turn = nxt;
        return m;
    };
    const int n = (int)(seg_len ? seg_len[seg / len_div] : uniform_len);
    if (n <= 0) {
        if (lane == 0) {
            out[seg] = __longlong_as_double(0x7FF8000000000000ll);
            if (MASKED) mm.out[seg] = 0.0;
        }
        return;
    }
    int64_t r0l, r1l;
    double frac;
    big_ranks(n, mode, q, &r0l, &r1l, &frac);
    const uint32_t r0 = (uint32_t)r0l, r1 = (uint32_t)r1l;
    const T c = XF == XF_ABSDEV ? (T)center[seg] : (T)0;
    const T *vrow = quad_stride != 1 ? vals + (seg / blk_rows) * blk_sig_stride + (seg % blk_rows) * 4
                                         : vals + (seg_off ? seg_off[seg] : seg * seg_stride);
    const uint32_t qs = (uint32_t)quad_stride;
    const unsigned char *mrow = MASKED ? mm.mask + (seg / mm.mask_div) * mm.mask_stride : nullptr;

    // ---- load: all loads of the row are issued back to back, with no branch between them.  Whole quads:
    // element 4 (lane + 64 i) + e is key[4 i + e] (a quad index beyond the row re-reads the row's last whole quad and
    // is ignored); the up to three elements after the last whole quad go to the extra slot key[KPL] of lanes 0..2.
    // Positions that hold no element of the row keep the largest key.
    K key[KPL + 1];
    T msum = 0;
    int mcnt = 0;
    bool saw_nan = false;
    K kmin = KEY_MAX, kmax = 0;
    auto take = [&](int slot, T x, unsigned m, bool valid) {
        if (MASKED && valid && m) {
            msum += x;
            ++mcnt;
        }
        if (XF == XF_ABSDEV) x = (T)fabs(x - c);
        const bool is_nan = x != x;
        const K k = (!valid || is_nan) ? KEY_MAX : key_of(x);
        saw_nan |= valid && is_nan;
        key[slot] = k;
        if (valid) {
            kmin = k < kmin ? k : kmin;
            kmax = k > kmax ? k : kmax;
        }
    };
    const bool vec_ok = quad_stride != 1 ||
                        ((reinterpret_cast<uintptr_t>(vrow) & (sizeof(Q4) - 1)) == 0 && (!MASKED || (reinterpret_cast<uintptr_t>(mrow) & 3) == 0));
    // The loads go out in batches (all loads of a batch back to back, no branch between them) and a batch is
    // turned into keys before the next one is requested: the staging registers of ONE batch beside the keys keep
    // the kernel at three waves per SIMD, and it is the other waves that cover a wave's memory round trips.
    constexpr int SB = KPL < 16 ? KPL : 16;                 // single elements per batch (rows that are not 16-byte aligned)
    constexpr int QBMAX = sizeof(T) == 8 ? 12 : 24;
    constexpr int QB = KPL / 4 <= QBMAX ? KPL / 4 : QBMAX;    // quads per batch
    if (vec_ok && n >= 4) {
        const int nq = n >> 2;                              // whole quads
        const Q4 *vv = reinterpret_cast<const Q4 *>(vrow);
        const uint32_t *mv = reinterpret_cast<const uint32_t *>(mrow);
        const int jt = 4 * nq + lane;                       // tail element of lanes 0..2
        const bool tail = wv == 0 && lane < 3 && jt < n;
        const T xt = vrow[tail ? (size_t)nq * 4 * qs + lane : 0];
        const unsigned mt = MASKED ? mrow[tail ? jt : 0] : 0u;
#pragma unroll
        for (int b0 = 0; b0 < KPL / 4; b0 += QB) {
            Q4 xq[QB];
            uint32_t mq[QB];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < QB; ++i) {
                const uint32_t j = (uint32_t)lane + 64u * (uint32_t)(b0 + i + wv * (KPL / 4));
                const uint32_t js = j < (uint32_t)nq ? j : (uint32_t)nq - 1u;
                xq[i] = vv[js * qs];
                mq[i] = MASKED ? mv[js] : 0u;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < QB; ++i) {
                const bool valid = lane + 64 * (b0 + i + wv * (KPL / 4)) < nq;
                take(4 * (b0 + i) + 0, xq[i].x, mq[i] & 0xFF, valid);
                take(4 * (b0 + i) + 1, xq[i].y, (mq[i] >> 8) & 0xFF, valid);
                take(4 * (b0 + i) + 2, xq[i].z, (mq[i] >> 16) & 0xFF, valid);
                take(4 * (b0 + i) + 3, xq[i].w, mq[i] >> 24, valid);
            }
        }
        take(KPL, xt, mt, tail);
    } else {
#pragma unroll
        for (int b0 = 0; b0 < KPL; b0 += SB) {
            T xs[SB];
            unsigned ms[SB];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < SB; ++i) {
                const uint32_t j = (uint32_t)lane + 64u * (uint32_t)(b0 + i + wv * KPL);
                const uint32_t js = j < (uint32_t)n ? j : (uint32_t)n - 1u;
                xs[i] = vrow[js];
                ms[i] = MASKED ? mrow[js] : 0u;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < SB; ++i) take(b0 + i, xs[i], ms[i], lane + 64 * (b0 + i + wv * KPL) < n);
        }
        key[KPL] = KEY_MAX;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const K a = __shfl_xor(kmin, d), b = __shfl_xor(kmax, d);
        kmin = a < kmin ? a : kmin;
        kmax = b > kmax ? b : kmax;
        if (MASKED) {
            msum += __shfl_xor(msum, d);
            mcnt += __shfl_xor(mcnt, d);
        }
    }
    bool any_nan = __ballot(saw_nan) != 0;
    if (NW > 1) {
        // every wave leaves its partial results in its own slot; lane w of every wave then takes wave w's and the
        // lanes combine them in a fixed order (the masked sum must not depend on which wave arrives first)
        if (lane == 0) {
            sh_a[wv] = (unsigned long long)kmin;
            sh_b[wv] = (unsigned long long)kmax;
            sh_c[wv] = any_nan ? 1u : 0u;
            if (MASKED) {
                sh_d[wv] = (uint32_t)mcnt;
                sh_m[wv] = sizeof(T) == 8 ? (unsigned long long)__double_as_longlong((double)msum)
                                          : (unsigned long long)__float_as_uint((float)msum);
            }
        }
        __syncthreads();
        const bool has = lane < NW;
        kmin = has ? (K)sh_a[has ? lane : 0] : KEY_MAX;
        kmax = has ? (K)sh_b[has ? lane : 0] : (K)0;
        uint32_t nan_w = has ? sh_c[has ? lane : 0] : 0u;
        if (MASKED) {
            mcnt = has ? (int)sh_d[has ? lane : 0] : 0;
            const unsigned long long mb = sh_m[has ? lane : 0];
            msum = has ? (sizeof(T) == 8 ? (T)__longlong_as_double((long long)mb) : (T)__uint_as_float((uint32_t)mb)) : (T)0;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const K a = __shfl_xor(kmin, d), b = __shfl_xor(kmax, d);
            kmin = a < kmin ? a : kmin;
            kmax = b > kmax ? b : kmax;
            nan_w |= __shfl_xor(nan_w, d);
            if (MASKED) {
                msum += __shfl_xor(msum, d);
                mcnt += __shfl_xor(mcnt, d);
            }
        }
        any_nan = nan_w != 0;
    }
    if (MASKED && lane == 0 && wv == 0) mm.out[seg] = (double)(msum / (T)(mcnt > 1 ? mcnt : 1));
    if (any_nan) {
        if (lane == 0 && wv == 0) out[seg] = __longlong_as_double(0x7FF8000000000000ll);
        return;
    }
    // ---- selection on key - kmin (the padding keys stay above every trial value that can be accepted)
    int bits = kmax == kmin ? 0 : (int)(8 * sizeof(K)) - (sizeof(K) == 8 ? __clzll((long long)(kmax - kmin)) : __clz((int)(kmax - kmin)));
    bits = bits < max_bits ? bits : max_bits;              // diagnostics (ADN_OPT_ABLATE): results become wrong
#pragma unroll
    for (int i = 0; i <= KPL; ++i) key[i] -= kmin;
    // lo_cnt = keys below res, hi_cnt = keys below res + 2^(undecided bits): their difference is the number of
    // keys that still share the decided prefix.  Once those candidates fit SEL_CK registers per lane they are
    // compacted (through the wave's LDS slot) and the remaining bits are decided on them alone: a row of some
    // thousand spread-out keys needs a handful of full sweeps, not one per bit.
    K res = 0;
    // (the padding keys, KEY_MAX - kmin now, lie below 2^bits only when the row spans half the key space: they
    // then count as keys of the top interval, so that a compaction never meets more candidates than it counted)
    const bool pad_low = bits >= (int)(8 * sizeof(K)) || ((K)(KEY_MAX - kmin) >> bits) == 0;
    uint32_t lo_cnt = 0, hi_cnt = pad_low ? (uint32_t)(NW * (KPL + 1) * 64) : (uint32_t)n;
    K ck[SEL_CK];
    bool compact = false, unique = false;
    int b = bits - 1;
    for (; b >= 0; --b) {
        const K trial = res | ((K)1 << b);
        uint32_t cnt = count_below<K, KPL + 1>(key, trial);
        if (NW > 1) cnt = wg_sum(cnt);
        if (cnt <= r0) {                                    // the r0-th key (0-based) is >= trial
            res = trial;
            lo_cnt = cnt;
        } else {
            hi_cnt = cnt;
        }
        if (b > 0 && hi_cnt - lo_cnt == 1) {                // ONE key shares the prefix: it is the answer
            unique = true;
            break;
        }
        if (KPL > SEL_CK && b >= 3 && hi_cnt - lo_cnt <= SEL_CK * 64) {
            compact = true;
            break;
        }
    }
    if (unique) {
        const K span = (K)1 << b;                           // the key in [res, res + span)
        K mn = KEY_MAX;
#pragma unroll
        for (int i = 0; i <= KPL; ++i) mn = (key[i] - res < span && key[i] < mn) ? key[i] : mn;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const K o = __shfl_xor(mn, d);
            mn = o < mn ? o : mn;
        }
        if (NW > 1) mn = wg_min(mn);
        res = mn;
    }
    if (compact) {
        K *cbuf = cbuf_all[NW == 1 ? wave : 0];
        const K span = (K)1 << b;                      // candidates: res <= key < res + span
        uint32_t base = 0;
        if (NW == 1) {
#pragma unroll
            for (int i = 0; i <= KPL; ++i) {
                const bool in = key[i] - res < span;
                const unsigned long long m = __ballot(in);
                const uint32_t pos = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (in) cbuf[pos] = key[i];
                base += (uint32_t)__builtin_popcountll(m);
                if ((i & 7) == 7) asm volatile("" : "+s"(base));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
            // the waves append to one list: a wave counts its candidates, reserves their places with ONE LDS atomic
            // and writes them there
            uint32_t mine = 0;
#pragma unroll
            for (int i = 0; i <= KPL; ++i) {
                mine += (uint32_t)__builtin_popcountll(__ballot(key[i] - res < span));
                if ((i & 7) == 7) asm volatile("" : "+s"(mine));
            }
            uint32_t at = 0;
            if (lane == 0) at = atomicAdd(&sh_ctr, mine);
            at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
#pragma unroll
            for (int i = 0; i <= KPL; ++i) {
                const bool in = key[i] - res < span;
                const unsigned long long m = __ballot(in);
                const uint32_t pos = at + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (in) cbuf[pos] = key[i];
                at += (uint32_t)__builtin_popcountll(m);
                if ((i & 7) == 7) asm volatile("" : "+s"(at));
            }
            __syncthreads();
            base = sh_ctr;
        }
#pragma unroll
        for (int i = 0; i < SEL_CK; ++i) ck[i] = lane + 64 * i < (int)base ? cbuf[lane + 64 * i] : KEY_MAX;
        uint32_t lo2 = lo_cnt;
        for (--b; b >= 0; --b) {
            const K trial = res | ((K)1 << b);
            const uint32_t cnt = lo_cnt + count_below<K, SEL_CK>(ck, trial);
            if (cnt <= r0) {
                res = trial;
                lo2 = cnt;
            } else {
                hi_cnt = cnt;
            }
            if (b > 0 && hi_cnt - lo2 == 1) {               // ONE candidate left
                const K span = (K)1 << b;
                K mn = KEY_MAX;
#pragma unroll
                for (int i = 0; i < SEL_CK; ++i) mn = (ck[i] - res < span && ck[i] < mn) ? ck[i] : mn;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) {
                    const K o = __shfl_xor(mn, d);
                    mn = o < mn ? o : mn;
                }
                res = mn;
                break;
            }
        }
    }
    K k0 = res, k1 = res;
    if (r1 != r0) {
        // the next order statistic: the same value if it occurs often enough, else the smallest key above it
        uint32_t cnt_le = 0;
        K above = KEY_MAX;
#pragma unroll
        for (int i = 0; i <= KPL; ++i) {
            cnt_le += (uint32_t)__builtin_popcountll(__ballot(key[i] <= res));
            if ((i & 7) == 7) asm volatile("" : "+s"(cnt_le));
            above = (key[i] > res && key[i] < above) ? key[i] : above;
        }
        if (NW > 1) cnt_le = wg_sum(cnt_le);
        if (cnt_le <= r1) {
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                const K o = __shfl_xor(above, d);
                above = o < above ? o : above;
            }
            if (NW > 1) above = wg_min(above);
            k1 = above;
        }
    }
    if (lane == 0 && wv == 0) out[seg] = select_result<T>(val_of((K)(k0 + kmin)), val_of((K)(k1 + kmin)), r0 == r1, frac, mode);
}

// ---------------------------------------------------------------- long rows of the grouped layout
// A row of several ten thousand values fills the registers of all the waves a CU holds (k_select_wave with sixteen
// waves per row: one row in flight per CU, so that no other row's search covers a row's load).  Such rows are
// selected in three steps that stream instead:
//  (1) two order statistics of every RS_STEP-th group of four values of the row (the grouped one-wave select on an
//      RS_STEP times longer stride; modes SEL_BRACKET_LO / _HI) bracket the wanted order statistic;
//  (2) one pass over the rows (a lane per row, a wave per 64 neighbouring rows and a chunk of their groups: every
//      load is 1 KB of consecutive bytes) counts the values below the bracket, appends those inside it to the row's
//      candidate list and adds up the masked values of its chunk;
//  (3) the exact order statistics of the candidates at the ranks that remain, and the masked mean from the chunks'
//      sums in chunk order.
// The bracket only has to be SOME pair of values: a row is done when both wanted ranks lie among its candidates and
// the list did not overflow; any other row (bracket missed, NaN in the row, no values) is left to k_select_wave,
// which skips the rows marked done.
static constexpr int RS_STEP = 16;
// candidates kept per row: the bracket of a median holds about 9 sqrt(RS_STEP n / 4) values
// (8192 up to 131 072 values, 16 384 up to 524 288, 32 768 beyond; binary64 lists fill LDS at 16 384)
__host__ __device__ static inline int64_t rows_stream_cap(int64_t max_len, size_t value_bytes) {
    const int64_t short_rows = value_bytes == 8 ? 98304 : 131072;
    return max_len <= short_rows ? 8192 : (max_len <= 524288 ? 16384 : 32768);
}
// longest row of the grouped layout with the streaming route: the candidate lists of longer rows outgrow LDS, and
// the register-resident select of the sample addresses a row's groups with 32-bit byte offsets
static constexpr int64_t RS_LIMIT_F32 = 786432, RS_LIMIT_F64 = 393216;
static constexpr int RS_BATCH = 4;                      // groups a lane has in flight
static constexpr int RS_MAX_CHUNKS = 256;
template <typename T> struct RowStreamMin;              // rows longer than this take the streaming route by themselves
template <> struct RowStreamMin<float> { static constexpr int64_t value = 4 * 88 * 64; };
template <> struct RowStreamMin<double> { static constexpr int64_t value = 4 * 96 * 64; };

struct RowStream {
    unsigned long long *below;   // [row] values below the bracket
    unsigned long long *count;   // [row] candidates appended (beyond cap: not stored, the row is not done)
    int *nan;                    // [row]
    int *done;                   // [row]
    double *psum;                // [chunk][row] masked sum of the chunk
    int *pcnt;                   // [chunk][row]
    int64_t n_rows;
    int64_t cap;                 // places of a row's candidate list
    int chunks;
};

static bool rows_stream_wanted(const Ctx *c, int64_t max_len, size_t value_bytes) {
    if (c->force_generic || (c->ablate & 32768)) return false;
    if (c->ablate & 16384) return max_len >= 4 * RS_STEP;
    return max_len > (value_bytes == 8 ? RowStreamMin<double>::value : RowStreamMin<float>::value);
}
static int rows_stream_chunks(int64_t n_sig, int64_t rows_per_sig, int64_t max_len) {
    const int64_t cols = n_sig * ceil_div(rows_per_sig, (int64_t)64);
    int64_t ch = ceil_div((int64_t)4096, cols);
    const int64_t most = ceil_div(ceil_div(max_len, (int64_t)4), (int64_t)(8 * RS_BATCH));    // at least 8 batches per chunk
    ch = ch > most ? most : ch;
    return (int)(ch < 1 ? 1 : (ch > RS_MAX_CHUNKS ? RS_MAX_CHUNKS : ch));
}
size_t select_rows_stream_ws_bytes(const Ctx *c, int64_t n_sig, int64_t rows_per_sig, int64_t max_len, size_t value_bytes) {
    if (!rows_stream_wanted(c, max_len, value_bytes)) return 0;
    const size_t n_rows = (size_t)(n_sig * rows_per_sig);
    const size_t ch = (size_t)rows_stream_chunks(n_sig, rows_per_sig, max_len);
    return ws_need(n_rows * (size_t)rows_stream_cap(max_len, value_bytes), value_bytes) + 2 * ws_need(n_rows, 8) + ws_need((size_t)n_sig, 8) +
           ws_need(2 * n_rows + (n_rows + 1) / 2, 8) + ws_need(n_rows, 4) + ws_need(ch * n_rows, 8) + ws_need(ch * n_rows, 4);
}

__global__ void k_rowq_sample_len(const int64_t *__restrict__ seg_len, int64_t n_sig, int64_t *__restrict__ out) {
    const int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (s >= n_sig) return;
    const int64_t nq = seg_len[s] >> 2;                    // whole groups
    out[s] = 4 * ((nq + RS_STEP - 1) / RS_STEP);
}

template <typename T, bool MASKED>
__global__ __launch_bounds__(64) void k_rowq_classify(const T *__restrict__ vals, int64_t qs, int64_t rows_per_sig,
                                                      int64_t sig_stride, const int64_t *__restrict__ seg_len,
                                                      const double *__restrict__ lo_v, const double *__restrict__ hi_v,
                                                      const unsigned char *__restrict__ mask, int64_t mask_stride,
                                                      T *__restrict__ cand, RowStream rs, int64_t groups_per_chunk) {
    typedef typename KeyT<T>::type K;
    typedef typename Quad4<T>::type Q4;
    const int lane = threadIdx.x;
    const int64_t sig = blockIdx.z, chunk = blockIdx.x;
    const int64_t bin = (int64_t)blockIdx.y * 64 + lane;
    const bool act = bin < rows_per_sig;
    const int64_t row = sig * rows_per_sig + (act ? bin : 0);
    const int64_t n = seg_len[sig];
    const int64_t groups = (n + 3) >> 2;                   // the last one may hold fewer than four values
    const int64_t j0 = chunk * groups_per_chunk;
    const int64_t j1 = j0 + groups_per_chunk < groups ? j0 + groups_per_chunk : groups;
    if (j0 >= j1) {                                        // no frames in this chunk
        if (MASKED && act) {
            rs.psum[chunk * rs.n_rows + row] = 0.0;
            rs.pcnt[chunk * rs.n_rows + row] = 0;
        }
        return;
    }
    const K klo = key_of((T)lo_v[row]), khi = key_of((T)hi_v[row]);
    const bool ordered = klo <= khi;
    const K base = ordered ? klo : ~(K)0, span = ordered ? (K)(khi - klo) : (K)0;
    const Q4 *vv = reinterpret_cast<const Q4 *>(vals + sig * sig_stride) + (act ? bin : 0);
    const uint32_t *mv = MASKED ? reinterpret_cast<const uint32_t *>(mask + sig * mask_stride) : nullptr;
    T *list = cand + row * rs.cap;
    unsigned long long below = 0;
    double msum = 0.0;
    int mcnt = 0;
    bool nan = false;
    for (int64_t j = j0; j < j1; j += RS_BATCH) {
        Q4 xq[RS_BATCH];
        uint32_t mq[RS_BATCH];
#pragma unroll
        for (int i = 0; i < RS_BATCH; ++i) {
            const int64_t jj = j + i < j1 ? j + i : j1 - 1;
            xq[i] = vv[jj * qs];
            mq[i] = MASKED ? mv[jj] : 0u;
        }
        T x[4 * RS_BATCH];
        uint32_t in = 0;
#pragma unroll
        for (int i = 0; i < RS_BATCH; ++i) {
            x[4 * i + 0] = xq[i].x;
            x[4 * i + 1] = xq[i].y;
            x[4 * i + 2] = xq[i].z;
            x[4 * i + 3] = xq[i].w;
            const int64_t left = n - 4 * (j + i);          // values of the group (<= 0: the batch ran past the chunk)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool valid = j + i < j1 && e < left;
                const T v = x[4 * i + e];
                if (MASKED && valid && ((mq[i] >> (8 * e)) & 0xFFu)) {
                    msum += (double)v;
                    ++mcnt;
                }
                const bool is_nan = v != v;
                nan |= valid && is_nan;
                const K k = is_nan ? ~(K)0 : key_of(v);
                below += valid && k < klo ? 1u : 0u;
                in |= valid && (K)(k - base) <= span ? 1u << (4 * i + e) : 0u;
            }
        }
        if (act && in) {
            const unsigned long long at = atomicAdd(&rs.count[row], (unsigned long long)__popc(in));
#pragma unroll
            for (int i = 0; i < 4 * RS_BATCH; ++i) {
                const unsigned long long p = at + (unsigned long long)__popc(in & ((1u << i) - 1u));
                if ((in >> i & 1u) && p < (unsigned long long)rs.cap) list[p] = x[i];
            }
        }
    }
    if (!act) return;
    if (below) atomicAdd(&rs.below[row], below);
    if (nan) rs.nan[row] = 1;
    if (MASKED) {
        rs.psum[chunk * rs.n_rows + row] = msum;
        rs.pcnt[chunk * rs.n_rows + row] = mcnt;
    }
}

template <bool MASKED>
__global__ void k_rowq_check(const int64_t *__restrict__ seg_len, int64_t rows_per_sig, int mode, double q, RowStream rs,
                             double *__restrict__ mean_out) {
    const int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (row >= rs.n_rows) return;
    const int64_t n = seg_len[row / rows_per_sig];
    int ok = 0;
    if (n > 0 && !rs.nan[row]) {
        int64_t r0, r1;
        double frac;
        stat_ranks(n, mode, q, &r0, &r1, &frac);
        const unsigned long long b = rs.below[row], cnt = rs.count[row];
        ok = cnt <= (unsigned long long)rs.cap && (unsigned long long)r0 >= b && (unsigned long long)r1 < b + cnt;
    }
    rs.done[row] = ok;
    if (MASKED && ok) {
        double sum = 0.0;
        long long cnt = 0;
        for (int ch = 0; ch < rs.chunks; ++ch) {
            sum += rs.psum[ch * rs.n_rows + row];
            cnt += rs.pcnt[ch * rs.n_rows + row];
        }
        mean_out[row] = sum / (double)(cnt > 1 ? cnt : 1);
    }
}

// The rows the streaming route left, whatever their length (rows beyond the register-resident select's reach have no
// other exact route): radix selection, eight key bits per pass over the row, one workgroup per row, early exit for the
// rows that are done.  The masked mean adds per-thread sums in thread order.  Few rows come here (a bracket misses
// about one row in 10^5), unless the data has NaN or the sample misleads: then this is slow and exact.
template <typename T, bool MASKED>
__global__ __launch_bounds__(256) void k_rowq_exact(const T *__restrict__ vals, int64_t qs, int64_t rows_per_sig,
                                                    int64_t sig_stride, const int64_t *__restrict__ seg_len,
                                                    const unsigned char *__restrict__ mask, int64_t mask_stride,
                                                    const int *__restrict__ done, int mode, double q,
                                                    double *__restrict__ out, double *__restrict__ mean_out) {
    typedef typename KeyT<T>::type K;
    typedef typename Quad4<T>::type Q4;
    constexpr int BITS = KeyT<T>::bits;
    const int64_t row = blockIdx.x;
    if (done[row]) return;
    const int tid = threadIdx.x;
    const int64_t sig = row / rows_per_sig, bin = row % rows_per_sig;
    const int64_t n = seg_len[sig];
    if (n <= 0) {
        if (tid == 0) {
            out[row] = __longlong_as_double(0x7FF8000000000000ll);
            if (MASKED) mean_out[row] = 0.0;
        }
        return;
    }
    int64_t r0, r1;
    double frac;
    stat_ranks(n, mode, q, &r0, &r1, &frac);
    const Q4 *vv = reinterpret_cast<const Q4 *>(vals + sig * sig_stride) + bin;
    const uint32_t *mv = MASKED ? reinterpret_cast<const uint32_t *>(mask + sig * mask_stride) : nullptr;
    const int64_t groups = (n + 3) >> 2;
    __shared__ uint32_t hist[256];
    __shared__ unsigned long long s_rank;
    __shared__ K s_prefix;
    __shared__ int s_nan;
    __shared__ double s_sum[256];
    __shared__ int s_cnt[256];
    __shared__ K s_above[256];
    __shared__ uint32_t s_le[256];
    // one sweep over the row: f(value, key, mask byte) for every value of the row, four groups per thread in flight
    auto sweep = [&](bool with_mask, auto &&f) {
        for (int64_t j = tid; j < groups; j += 4 * 256) {
            Q4 xq[4];
            uint32_t mq[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t jj = j + i * 256 < groups ? j + i * 256 : j;
                xq[i] = vv[jj * qs];
                mq[i] = MASKED && with_mask ? mv[jj] : 0u;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t jj = j + i * 256;
                if (jj >= groups) break;
                const int64_t left = n - 4 * jj;
                const T x[4] = {xq[i].x, xq[i].y, xq[i].z, xq[i].w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (e < left) {
                        const bool is_nan = x[e] != x[e];
                        f(x[e], is_nan ? ~(K)0 : key_of(x[e]), (mq[i] >> (8 * e)) & 0xFFu);
                    }
            }
        }
    };
    K prefix = 0;
    unsigned long long rank = (unsigned long long)r0;
    double msum = 0.0;
    int mcnt = 0;
    bool nan = false;
    for (int pass = 0; pass < BITS / 8; ++pass) {
        const int shift = BITS - 8 * (pass + 1);
        hist[tid] = 0;
        if (tid == 0 && pass == 0) s_nan = 0;
        __syncthreads();
        if (pass == 0) {
            sweep(true, [&](T v, K k, uint32_t m) {
                if (MASKED && m) {
                    msum += (double)v;
                    ++mcnt;
                }
                nan |= k == ~(K)0;
                atomicAdd(&hist[(uint32_t)(k >> shift) & 255u], 1u);
            });
            if (nan) s_nan = 1;
            if (MASKED) {
                s_sum[tid] = msum;
                s_cnt[tid] = mcnt;
            }
        } else {
            sweep(false, [&](T, K k, uint32_t) {
                if ((K)(k >> (shift + 8)) == prefix) atomicAdd(&hist[(uint32_t)(k >> shift) & 255u], 1u);
            });
        }
        __syncthreads();
        if (tid == 0) {
            if (pass == 0 && MASKED) {
                double sum = 0.0;
                long long cnt = 0;
                for (int i = 0; i < 256; ++i) {
                    sum += s_sum[i];
                    cnt += s_cnt[i];
                }
                mean_out[row] = sum / (double)(cnt > 1 ? cnt : 1);
            }
            unsigned long long before = 0;
            int d = 0;
            for (; d < 255; ++d) {
                if (before + hist[d] > rank) break;
                before += hist[d];
            }
            s_rank = rank - before;
            s_prefix = (K)((prefix << 8) | (K)d);
        }
        __syncthreads();
        if (s_nan) {                                           // a NaN in the row: the statistic is NaN
            if (tid == 0) out[row] = __longlong_as_double(0x7FF8000000000000ll);
            return;
        }
        rank = s_rank;
        prefix = s_prefix;
    }
    const K k0 = prefix;
    K k1 = k0;
    if (r1 != r0) {
        // the next order statistic: k0 again if it occurs often enough, else the smallest key above it
        uint32_t le = 0;
        K above = ~(K)0;
        sweep(false, [&](T, K k, uint32_t) {
            le += k <= k0 ? 1u : 0u;
            above = (k > k0 && k < above) ? k : above;
        });
        s_le[tid] = le;
        s_above[tid] = above;
        __syncthreads();
        if (tid == 0) {
            unsigned long long tot = 0;
            K mn = ~(K)0;
            for (int i = 0; i < 256; ++i) {
                tot += s_le[i];
                mn = s_above[i] < mn ? s_above[i] : mn;
            }
            k1 = tot > (unsigned long long)r1 ? k0 : mn;
        }
    }
    if (tid == 0) out[row] = select_result<T>(val_of(k0), val_of(k1), r0 == r1, frac, mode);
}

template <typename T, int XF, bool MASKED>
static int launch_select_wave(Ctx *c, const T *d_vals, const SelectArgs &a);

// Steps (1) to (3) for every row of `a` (grouped layout, one length per signal); *done_out: the rows that have their
// result (the caller runs k_select_wave on the others).
template <typename T, bool MASKED>
static int rows_stream(Ctx *c, const T *d_vals, const SelectArgs &a, const int **done_out) {
    typedef typename KeyT<T>::type K;
    const int64_t n_sig = a.n_seg / a.blk_rows, n_rows = a.n_seg;
    const int chunks = rows_stream_chunks(n_sig, a.blk_rows, a.max_len);
    const int64_t cap = rows_stream_cap(a.max_len, sizeof(T));
    T *cand = ws_take<T>(c, (size_t)n_rows * cap);
    double *lo_v = ws_take<double>(c, n_rows), *hi_v = ws_take<double>(c, n_rows);
    int64_t *slen = ws_take<int64_t>(c, n_sig);
    RowStream rs;
    // (the three arrays that start at zero lie side by side: one fill)
    unsigned long long *zeroed = ws_take<unsigned long long>(c, 2 * (size_t)n_rows + ((size_t)n_rows + 1) / 2);
    rs.below = zeroed;
    rs.count = zeroed + n_rows;
    rs.nan = reinterpret_cast<int *>(zeroed + 2 * n_rows);
    rs.done = ws_take<int>(c, n_rows);
    rs.psum = ws_take<double>(c, (size_t)chunks * n_rows);
    rs.pcnt = ws_take<int>(c, (size_t)chunks * n_rows);
    rs.n_rows = n_rows;
    rs.cap = cap;
    rs.chunks = chunks;
    ADN_HIP(hipMemsetAsync(zeroed, 0, (size_t)n_rows * 20, c->stream));
    // (1) the bracket: the sample's order statistics 4.5 standard deviations of the rank (and two ranks) to either side
    // of the wanted one, for every row by the length of its own sample
    ADN_LAUNCH(k_rowq_sample_len, dim3((unsigned)ceil_div(n_sig, (int64_t)256)), dim3(256), 0, c->stream, a.d_seg_len, n_sig, slen);
    const int64_t m_max = 4 * ceil_div(a.max_len >> 2, (int64_t)RS_STEP);
    SelectArgs sa = a;
    sa.rows_stream = false;
    sa.quad_stride = a.quad_stride * RS_STEP;
    sa.d_seg_len = slen;
    sa.max_len = m_max;
    sa.q = a.mode == SEL_MEDIAN ? 50.0 : a.q;
    sa.d_mask = nullptr;
    sa.d_masked_mean = nullptr;
    sa.mode = SEL_BRACKET_LO;
    sa.d_out = lo_v;
    ADN_TRY((launch_select_wave<T, XF_NONE, false>(c, d_vals, sa)));
    sa.mode = SEL_BRACKET_HI;
    sa.d_out = hi_v;
    ADN_TRY((launch_select_wave<T, XF_NONE, false>(c, d_vals, sa)));
    // (2)
    int64_t gpc = ceil_div(ceil_div(a.max_len, (int64_t)4), (int64_t)chunks);
    gpc = ceil_div(gpc, (int64_t)RS_BATCH) * RS_BATCH;
    ADN_LAUNCH_NAMED("k_rowq_classify", (k_rowq_classify<T, MASKED>), dim3((unsigned)chunks, (unsigned)ceil_div(a.blk_rows, (int64_t)64), (unsigned)n_sig),
                     dim3(64), 0, c->stream, d_vals, a.quad_stride, a.blk_rows, a.blk_sig_stride, a.d_seg_len, lo_v, hi_v,
                     a.d_mask, a.mask_stride, cand, rs, gpc);
    ADN_LAUNCH(k_rowq_check<MASKED>, dim3((unsigned)ceil_div(n_rows, (int64_t)256)), dim3(256), 0, c->stream, a.d_seg_len, a.blk_rows,
               a.mode, a.q, rs, a.d_masked_mean);
    // (3)
    const size_t dyn = (size_t)cap * sizeof(K);
    auto kern = k_select_lds<T, XF_NONE, false>;
    ADN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    SubsetRanks sub{a.d_seg_len, a.blk_rows, 0, rs.below, rs.done, nullptr};
    ADN_LAUNCH_NAMED("k_select_lds<rows, candidates>", kern, dim3((unsigned)n_rows), dim3(SEL_THREADS), dyn, c->stream, cand,
                     (const int64_t *)nullptr, cap, reinterpret_cast<const int64_t *>(rs.count), (int64_t)1, (int64_t)0, a.mode, a.q,
                     (const double *)nullptr, a.d_out, MaskedMean{nullptr, 1, 0, nullptr}, sub);
    ADN_HIP(hipGetLastError());
    *done_out = rs.done;
    return ADN_OK;
}

static constexpr int64_t SEL_WAVE_MAX = 128 * 64;       // longest row ONE wave holds in registers

// Rows of the grouped layout beyond that (binary32: beyond 88 keys per lane, the longest row four waves per SIMD hold):
// NW waves per row, KPL keys per lane each.  binary32: 64 or 88 keys, four waves per SIMD, so that a workgroup of
// sixteen waves fits a CU; binary64: 96 keys of two registers, two waves per SIMD, eight waves per workgroup.
template <typename T> struct SelWg;
template <> struct SelWg<float> { static constexpr int64_t ONE_WAVE = 88 * 64, LIMIT = 16 * 88 * 64; };
template <> struct SelWg<double> { static constexpr int64_t ONE_WAVE = 128 * 64, LIMIT = 8 * 96 * 64; };

template <typename T, int XF, bool MASKED>
static int launch_select_wave(Ctx *c, const T *d_vals, const SelectArgs &a) {
    MaskedMean mm{a.d_mask, a.mask_div, a.mask_stride, a.d_masked_mean};
    // (grouped rows: a multiple of 16 workgroups, so that the kernel's workgroup permutation is one)
    const int64_t wgs = a.quad_stride != 1 ? ceil_div(ceil_div(a.n_seg, (int64_t)4), (int64_t)16) * 16 : ceil_div(a.n_seg, (int64_t)4);
    dim3 grid((unsigned)wgs), block(256);
#define ADN_SEL_WAVE(KPLV)                                                                                          \
    ADN_LAUNCH_NAMED("k_select_wave<" #KPLV ">", (k_select_wave<T, XF, MASKED, KPLV>), grid, block, 0, c->stream, d_vals,  \
                     a.d_seg_off, a.seg_stride, a.d_seg_len, a.len_div, a.uniform_len, a.n_seg, a.mode, a.q, a.d_center, \
                     a.d_out, mm, (c->ablate & 32) ? 0 : 64, a.quad_stride, a.blk_rows, a.blk_sig_stride, (const int *)nullptr)
#define ADN_SEL_WG(NWV, KPLV)                                                                                        \
    ADN_LAUNCH_NAMED("k_select_wave<" #KPLV ", " #NWV " waves per row>", (k_select_wave<T, XF, MASKED, KPLV, NWV>),   \
                     dim3((unsigned)(ceil_div(a.n_seg, (int64_t)64) * 64)), dim3(NWV * 64), 0, c->stream, d_vals,   \
                     a.d_seg_off, a.seg_stride, a.d_seg_len, a.len_div, a.uniform_len, a.n_seg, a.mode, a.q, a.d_center, \
                     a.d_out, mm, (c->ablate & 32) ? 0 : 64, a.quad_stride, a.blk_rows, a.blk_sig_stride, done)
    // (diagnostic option bit 1024: rows that one wave could hold are shared by two waves all the same)
    const bool stream = XF == XF_NONE && a.quad_stride != 1 && a.rows_stream && rows_stream_wanted(c, a.max_len, sizeof(T)) &&
                        a.d_seg_len != nullptr && a.len_div == a.blk_rows && a.n_seg % a.blk_rows == 0 &&
                        (!MASKED || (a.mask_div == a.blk_rows && a.mask_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(a.d_mask) & 3) == 0));
    const bool share = a.quad_stride != 1 && (a.max_len > SelWg<T>::ONE_WAVE || (c->ablate & 1024) || stream);
    const int *done = nullptr;
    if constexpr (XF == XF_NONE) {
        const bool beyond = a.quad_stride != 1 && a.max_len > SelWg<T>::LIMIT;     // no register kernel holds such a row
        if (beyond && !stream) {
            set_error("internal: grouped rows of %lld values need the streaming row select", (long long)a.max_len);
            return ADN_ERR_UNSUPPORTED;
        }
        if (stream) ADN_TRY((rows_stream<T, MASKED>(c, d_vals, a, &done)));
        if (stream && (beyond || (c->ablate & 65536))) {
            ADN_LAUNCH_NAMED("k_rowq_exact", (k_rowq_exact<T, MASKED>), dim3((unsigned)a.n_seg), dim3(256), 0, c->stream, d_vals,
                             a.quad_stride, a.blk_rows, a.blk_sig_stride, a.d_seg_len, a.d_mask, a.mask_stride, done, a.mode,
                             a.q, a.d_out, a.d_masked_mean);
            ADN_HIP(hipGetLastError());
            return ADN_OK;
        }
        if (share) {
            const int64_t n = a.max_len;
            if constexpr (sizeof(T) == 4) {
                if (n <= 2 * 64 * 64) ADN_SEL_WG(2, 64);
                else if (n <= 2 * 88 * 64) ADN_SEL_WG(2, 88);
                else if (n <= 4 * 64 * 64) ADN_SEL_WG(4, 64);
                else if (n <= 4 * 88 * 64) ADN_SEL_WG(4, 88);
                else if (n <= 8 * 64 * 64) ADN_SEL_WG(8, 64);
                else if (n <= 8 * 88 * 64) ADN_SEL_WG(8, 88);
                else if (n <= 16 * 64 * 64) ADN_SEL_WG(16, 64);
                else ADN_SEL_WG(16, 88);
            } else {
                if (n <= 2 * 96 * 64) ADN_SEL_WG(2, 96);
                else if (n <= 4 * 96 * 64) ADN_SEL_WG(4, 96);
                else ADN_SEL_WG(8, 96);
            }
            ADN_HIP(hipGetLastError());
            return ADN_OK;
        }
    }
    if (a.max_len <= 8 * 64) ADN_SEL_WAVE(8);
    else if (a.max_len <= 16 * 64) ADN_SEL_WAVE(16);       // (clips of 10 s: 938 frames)
    else if (a.max_len <= 24 * 64) ADN_SEL_WAVE(24);
    else if (a.max_len <= 32 * 64) ADN_SEL_WAVE(32);       // (20 s: 1875 frames)
    else if (a.max_len <= 48 * 64) ADN_SEL_WAVE(48);
    else if (a.max_len <= 88 * 64 && sizeof(T) == 4) ADN_SEL_WAVE(88);     // binary32: four waves per SIMD still
    else if (a.max_len <= 96 * 64) ADN_SEL_WAVE(96);
    else ADN_SEL_WAVE(128);
#undef ADN_SEL_WAVE
#undef ADN_SEL_WG
    ADN_HIP(hipGetLastError());
    return ADN_OK;
}

int64_t select_grouped_limit(size_t value_bytes) { return value_bytes == 8 ? SelWg<double>::LIMIT : SelWg<float>::LIMIT; }
int64_t select_grouped_limit_stream(const Ctx *c, size_t value_bytes) {
    const int64_t lim = value_bytes == 8 ? RS_LIMIT_F64 : RS_LIMIT_F32;
    return rows_stream_wanted(c, lim, value_bytes) ? lim : select_grouped_limit(value_bytes);
}

template <typename T>
int launch_select(Ctx *c, const T *d_vals, const SelectArgs &a) {
    if (a.n_seg <= 0) return ADN_OK;
    using K = typename KeyT<T>::type;
    if (a.quad_stride != 1) {
        const int64_t longest = a.rows_stream ? select_grouped_limit_stream(c, sizeof(T)) : select_grouped_limit(sizeof(T));
        ADN_REQUIRE(a.max_len > 0 && a.max_len <= longest && a.xform == XF_NONE &&
                        a.mode != SEL_PERCENTILE_DB20 && (reinterpret_cast<uintptr_t>(d_vals) & 31) == 0 &&
                        a.blk_sig_stride % 4 == 0 && (a.quad_stride * (a.max_len / 4 + 1)) < (int64_t(1) << 28),
                    "internal: grouped rows are rows of at most %lld values", (long long)longest);
    }
    if (a.max_len > BIG_THRESHOLD && a.quad_stride == 1) {
        if (a.d_masked_mean) {
            set_error("internal: the fused masked mean is only available for LDS-resident segments");
            return ADN_ERR_UNSUPPORTED;
        }
        return a.xform == XF_ABSDEV ? launch_select_big<T, XF_ABSDEV>(c, d_vals, a)
                                    : launch_select_big<T, XF_NONE>(c, d_vals, a);
    }
    // rows of up to a few thousand values: one wave per row, keys in registers (binary64: rows in the grouped
    // layout only, the other binary64 rows keep the LDS kernels)
    if ((!c->force_generic || a.quad_stride != 1) && a.max_len > 0 && (a.max_len <= SEL_WAVE_MAX || a.quad_stride != 1) &&
        a.mode != SEL_PERCENTILE_DB20 &&
        (std::is_same<T, float>::value || a.quad_stride != 1)) {
        const bool masked = a.d_masked_mean != nullptr;
        if (a.xform == XF_ABSDEV) return masked ? launch_select_wave<T, XF_ABSDEV, true>(c, d_vals, a)
                                                : launch_select_wave<T, XF_ABSDEV, false>(c, d_vals, a);
        return masked ? launch_select_wave<T, XF_NONE, true>(c, d_vals, a) : launch_select_wave<T, XF_NONE, false>(c, d_vals, a);
    }
    // segments whose keys fit LDS take the lean kernel (LDS = the keys, several workgroups per CU)
    if (a.max_len > 0 && (size_t)a.max_len * sizeof(K) <= SEL_LDS_MAX_BYTES) {
        const size_t dyn = (size_t)a.max_len * sizeof(K);
        MaskedMean mm{a.d_mask, a.mask_div, a.mask_stride, a.d_masked_mean};
        const SubsetRanks none{nullptr, 1, 0, nullptr, nullptr, nullptr};
        dim3 grid((unsigned)a.n_seg), block(SEL_THREADS);
#define ADN_LAUNCH_LDS_NT(XFV, MK, NTV)                                                                      \
    do {                                                                                                     \
        auto kern = k_select_lds<T, XFV, MK, NTV>;                                                           \
        ADN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                                    \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));                  \
        ADN_LAUNCH_NAMED("k_select_lds<" #XFV ", masked=" #MK ", " #NTV ">", kern, grid, dim3(NTV), dyn, c->stream, d_vals, a.d_seg_off, a.seg_stride,         \
                           a.d_seg_len, a.len_div, a.uniform_len, a.mode, a.q, a.d_center, a.d_out, mm, none); \
    } while (0)
        /* short segments: one wave per segment, more segments resident per CU and no cross-wave waits */   \
#define ADN_LAUNCH_LDS(XFV, MK)                                                                              \
    do {                                                                                                     \
        if (a.max_len <= SEL_ONE_WAVE_MAX) ADN_LAUNCH_LDS_NT(XFV, MK, 64);                                   \
        else ADN_LAUNCH_LDS_NT(XFV, MK, SEL_THREADS);                                                        \
    } while (0)
        const bool masked = a.d_masked_mean != nullptr;
        if (a.xform == XF_ABSDEV) {
            if (masked) ADN_LAUNCH_LDS(XF_ABSDEV, true);
            else ADN_LAUNCH_LDS(XF_ABSDEV, false);
        } else {
            if (masked) ADN_LAUNCH_LDS(XF_NONE, true);
            else ADN_LAUNCH_LDS(XF_NONE, false);
        }
#undef ADN_LAUNCH_LDS
#undef ADN_LAUNCH_LDS_NT
        ADN_HIP(hipGetLastError());
        return ADN_OK;
    }
    // longer segments (up to the multi-workgroup threshold): keys are re-read from memory each pass
    int lds_keys = (int)(SEL_LDS_BYTES / sizeof(K));
    const size_t dyn = 2 * (size_t)lds_keys * sizeof(K);
    MaskedMean mm{a.d_mask, a.mask_div, a.mask_stride, a.d_masked_mean};
    if (a.xform == XF_ABSDEV) {
        auto kern = k_select<T, XF_ABSDEV>;
        ADN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    } else {
        auto kern = k_select<T, XF_NONE>;
        ADN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    }
    dim3 grid((unsigned)a.n_seg), block(SEL_THREADS);
    if (a.xform == XF_ABSDEV) {
        ADN_LAUNCH((k_select<T, XF_ABSDEV>), grid, block, dyn, c->stream, d_vals,
                           a.d_seg_off, a.seg_stride, a.d_seg_len, a.len_div, a.uniform_len,
                           a.mode, a.q, a.d_center, a.d_out, lds_keys, mm, SubsetRanks{nullptr, 1, 0, nullptr, nullptr, nullptr});
    } else {
        ADN_LAUNCH((k_select<T, XF_NONE>), grid, block, dyn, c->stream, d_vals,
                           a.d_seg_off, a.seg_stride, a.d_seg_len, a.len_div, a.uniform_len,
                           a.mode, a.q, a.d_center, a.d_out, lds_keys, mm, SubsetRanks{nullptr, 1, 0, nullptr, nullptr, nullptr});
    }
    ADN_HIP(hipGetLastError());
    return ADN_OK;
}

template int launch_select<float>(Ctx *, const float *, const SelectArgs &);
template int launch_select<double>(Ctx *, const double *, const SelectArgs &);

// Lists up to this many keys are staged whole in LDS (one workgroup per CU).  Longer ones are bracketed again
// (sample -> count + gather, on the list, with the rank offset) until the keys left fit LDS: the levels of the
// long-segment select, started on a list instead of a segment.
static size_t subset_lds_keys(size_t key_bytes) { return SUBSET_LDS_BYTES / key_bytes; }
size_t select_subset_max_keys(size_t key_bytes) { return (size_t)1 << 27; }
size_t select_subset_lds_keys(size_t key_bytes) { return subset_lds_keys(key_bytes); }

size_t select_subset_ws_bytes(int64_t n_seg, int64_t cap, size_t key_bytes) {
    if ((size_t)cap <= subset_lds_keys(key_bytes)) return 0;
    int64_t caps[BIG_MAX_LEVELS];
    const int nl = big_level_caps(cap, key_bytes, caps);
    size_t total = 0;
    for (int l = 0; l < nl; ++l)
        total += ws_need((size_t)n_seg * caps[l], 8) + ws_need((size_t)n_seg, sizeof(BigState)) + 4 * ws_need((size_t)n_seg, 8) +
                 ws_need((size_t)n_seg, 4);
    return total;
}

// lists that an earlier step had declared usable but a bracket level lost: their results must not look valid
__global__ void k_subset_invalidate(const int *valid_in, const int *valid_out, int64_t n_seg, double *out, double *pair) {
    const int64_t seg = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (seg >= n_seg || (valid_in && !valid_in[seg]) || valid_out[seg]) return;
    const double nan = __longlong_as_double(0x7FF8000000000000ll);
    out[seg] = nan;
    if (pair) pair[2 * seg] = pair[2 * seg + 1] = nan;
}

template <typename T>
static int launch_select_subset_levels(Ctx *c, const T *d_cand, const SubsetSelect &a) {
    using K = typename KeyT<T>::type;
    int64_t caps[BIG_MAX_LEVELS];
    const int n_levels = big_level_caps(a.cap, sizeof(K), caps);
    BigState *st_l[BIG_MAX_LEVELS];
    T *cand_l[BIG_MAX_LEVELS];
    int64_t *len_l[BIG_MAX_LEVELS];
    unsigned long long *below_l[BIG_MAX_LEVELS], *lohi_l[BIG_MAX_LEVELS];
    int *valid_l[BIG_MAX_LEVELS];
    for (int l = 0; l < n_levels; ++l) {
        st_l[l] = ws_take<BigState>(c, (size_t)a.n_seg);
        cand_l[l] = reinterpret_cast<T *>(ws_take<double>(c, (size_t)a.n_seg * caps[l]));
        len_l[l] = ws_take<int64_t>(c, (size_t)a.n_seg);
        below_l[l] = ws_take<unsigned long long>(c, (size_t)a.n_seg);
        valid_l[l] = ws_take<int>(c, (size_t)a.n_seg);
        lohi_l[l] = ws_take<unsigned long long>(c, 2 * (size_t)a.n_seg);
    }
    if (c->ws_used > c->ws_bytes) {
        set_error("internal: workspace not reserved for the bracket levels of long candidate lists");
        return ADN_ERR_NOMEM;
    }
    const unsigned nb = (unsigned)ceil_div(a.n_seg, 256);
    const T *src = d_cand;
    const int64_t *src_len = a.d_len;
    int64_t src_cap = a.cap;
    const unsigned long long *src_below = a.d_below;
    const int *src_valid = a.d_valid;
    for (int l = 0; l < n_levels; ++l) {
        const RankSrc rs{a.d_orig_len, 1, 0, src_below, src_valid};
        const int64_t ltiles = ceil_div(src_cap, GATHER_TILE);
        ADN_REQUIRE(a.n_seg * ltiles < (int64_t)0x7FFFFFFF, "select batch too large for one launch");
        ADN_LAUNCH(k_big_init_level, dim3(nb), dim3(256), 0, c->stream, st_l[l], a.n_seg);
        // the first level sees the list as it was gathered (raw coefficients: |x - centre| is formed on the way and
        // it is those values that the level gathers); deeper levels see transformed keys
        if (l == 0 && a.xform == XF_ABSDEV) {
            ADN_LAUNCH((k_big_sample<T, XF_ABSDEV>), dim3((unsigned)a.n_seg), dim3(SAMPLE_THREADS), 0, c->stream, src,
                       (const int64_t *)nullptr, src_cap, src_len, (int64_t)1, (int64_t)0, a.d_center, st_l[l], (int)SEL_MEDIAN, 50.0, rs);
            ADN_LAUNCH((k_big_gather<T, XF_ABSDEV>), dim3((unsigned)(a.n_seg * ltiles)), dim3(256), 0, c->stream, src,
                       (const int64_t *)nullptr, src_cap, src_len, (int64_t)1, (int64_t)0, a.d_center, st_l[l], cand_l[l], caps[l], ltiles, 0);
        } else {
            ADN_LAUNCH((k_big_sample<T, XF_NONE>), dim3((unsigned)a.n_seg), dim3(SAMPLE_THREADS), 0, c->stream, src,
                       (const int64_t *)nullptr, src_cap, src_len, (int64_t)1, (int64_t)0, (const double *)nullptr, st_l[l], (int)SEL_MEDIAN, 50.0, rs);
            ADN_LAUNCH((k_big_gather<T, XF_NONE>), dim3((unsigned)(a.n_seg * ltiles)), dim3(256), 0, c->stream, src,
                       (const int64_t *)nullptr, src_cap, src_len, (int64_t)1, (int64_t)0, (const double *)nullptr, st_l[l], cand_l[l], caps[l], ltiles, 0);
        }
        ADN_LAUNCH(k_big_check, dim3(nb), dim3(256), 0, c->stream, st_l[l], src_len, (int64_t)1, (int64_t)0, a.n_seg,
                   (int)SEL_MEDIAN, 50.0, caps[l], rs, len_l[l], valid_l[l], below_l[l], lohi_l[l]);
        src = cand_l[l];
        src_len = len_l[l];
        src_cap = caps[l];
        src_below = below_l[l];
        src_valid = valid_l[l];
    }
    const int L = n_levels - 1;
    ADN_LAUNCH(k_subset_invalidate, dim3(nb), dim3(256), 0, c->stream, a.d_valid, valid_l[L], a.n_seg, a.d_out, a.d_pair);
    const MaskedMean mm{nullptr, 1, 0, nullptr};
    SubsetRanks sub{a.d_orig_len, 1, 0, below_l[L], valid_l[L], lohi_l[L]};
    sub.pair = a.d_pair;
    const size_t dyn = (size_t)caps[L] * sizeof(K);
    ADN_REQUIRE(dyn <= SEL_LDS_MAX_BYTES, "internal: the last bracket level left lists of %lld keys", (long long)caps[L]);
    auto kern = k_select_lds<T, XF_NONE, false>;
    ADN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    ADN_LAUNCH_NAMED("k_select_lds<XF_NONE, subset levels>", kern, dim3((unsigned)a.n_seg), dim3(SEL_THREADS), dyn, c->stream,
                     cand_l[L], (const int64_t *)nullptr, caps[L], len_l[L], (int64_t)1, (int64_t)0, (int)SEL_MEDIAN, 50.0,
                     (const double *)nullptr, a.d_out, mm, sub);
    ADN_HIP(hipGetLastError());
    return ADN_OK;
}

template <typename T>
int launch_select_subset(Ctx *c, const T *d_cand, const SubsetSelect &a) {
    using K = typename KeyT<T>::type;
    if (a.n_seg <= 0) return ADN_OK;
    ADN_REQUIRE((size_t)a.cap <= select_subset_max_keys(sizeof(K)), "internal: candidate lists too long for the subset select");
    const MaskedMean mm{nullptr, 1, 0, nullptr};
    SubsetRanks sub{a.d_orig_len, 1, 0, a.d_below, a.d_valid, nullptr};
    sub.pair = a.d_pair;
    const double *center = a.xform == XF_ABSDEV ? a.d_center : nullptr;
    if ((size_t)a.cap <= subset_lds_keys(sizeof(K))) {
        const size_t dyn = (size_t)a.cap * sizeof(K);
        constexpr int NT = 1024;                   // lists of tens of thousands of keys: few keys per lane in every sweep
        if (a.xform == XF_ABSDEV) {
            auto kern = k_select_lds<T, XF_ABSDEV, false, NT>;
            ADN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
            ADN_LAUNCH_NAMED("k_select_lds<XF_ABSDEV, subset, 1024>", kern, dim3((unsigned)a.n_seg), dim3(NT), dyn, c->stream, d_cand, (const int64_t *)nullptr, a.cap,
                               a.d_len, (int64_t)1, (int64_t)0, (int)SEL_MEDIAN, 50.0, center, a.d_out, mm, sub);
        } else {
            auto kern = k_select_lds<T, XF_NONE, false, NT>;
            ADN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
            ADN_LAUNCH_NAMED("k_select_lds<XF_NONE, subset, 1024>", kern, dim3((unsigned)a.n_seg), dim3(NT), dyn, c->stream, d_cand, (const int64_t *)nullptr, a.cap,
                               a.d_len, (int64_t)1, (int64_t)0, (int)SEL_MEDIAN, 50.0, center, a.d_out, mm, sub);
        }
    } else {
        return launch_select_subset_levels<T>(c, d_cand, a);
    }
    ADN_HIP(hipGetLastError());
    return ADN_OK;
}
template int launch_select_subset<float>(Ctx *, const float *, const SubsetSelect &);
template int launch_select_subset<double>(Ctx *, const double *, const SubsetSelect &);

}  // namespace adn
