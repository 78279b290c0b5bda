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
#include "adn_kernels.hpp"

namespace adn {

static constexpr int SEL_THREADS = 256;
static constexpr int SEL_LDS_BYTES = 48 * 1024;

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

template <typename T, int XF>
__global__ __launch_bounds__(SEL_THREADS) void k_select(const T *__restrict__ vals,
                                                        const int64_t *__restrict__ seg_off,
                                                        int64_t seg_stride,
                                                        const int64_t *__restrict__ seg_len,
                                                        int64_t len_div, int64_t uniform_len,
                                                        int mode, double q,
                                                        const double *__restrict__ center,
                                                        double *__restrict__ out, int lds_keys,
                                                        MaskedMean mm) {
    using K = typename KeyT<T>::type;
    constexpr int BITS = KeyT<T>::bits;
    constexpr int PASSES = BITS / 8;
    extern __shared__ __align__(16) unsigned char smem[];
    K *cur = reinterpret_cast<K *>(smem);        // list the current pass walks (LDS-resident segments)
    K *oth = cur + lds_keys;                     // the next, shorter list is gathered here
    __shared__ uint32_t hist[256];
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t s_digit, s_before, s_count, s_nan, s_fill, s_next;
    __shared__ K s_lo_max, s_hi_min;
    __shared__ double s_msum[4];
    __shared__ int s_mcnt[4];

    const int64_t seg = blockIdx.x;
    const int64_t off = seg_off ? seg_off[seg] : seg * seg_stride;
    const int64_t n = seg_len ? seg_len[seg / len_div] : uniform_len;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (n <= 0) {
        if (tid == 0) {
            out[seg] = __longlong_as_double(0x7FF8000000000000ll);
            if (mm.out) mm.out[seg] = 0.0;
        }
        return;
    }
    int64_t r0, r1;
    double frac;
    if (mode == SEL_MEDIAN) {
        r0 = (n - 1) / 2;
        r1 = n / 2;
        frac = 0.5;
    } else {
        const double pos = (double)(n - 1) * (q / 100.0);
        r0 = (int64_t)floor(pos);
        if (r0 > n - 1) r0 = n - 1;
        frac = pos - (double)r0;
        r1 = r0 + 1 < n ? r0 + 1 : n - 1;
    }
    const T c = XF == XF_ABSDEV ? (T)center[seg] : (T)0;
    const bool cached = n <= (int64_t)lds_keys;
    const T *v = vals + off;
    const unsigned char *mk = mm.out ? mm.mask + (seg / mm.mask_div) * mm.mask_stride : nullptr;
    auto load_key = [&](int64_t i) -> K {
        T x = v[i];
        if (XF == XF_ABSDEV) x = fabs(x - c);
        return x != x ? ~(K)0 : key_of(x);
    };

    if (tid == 0) s_nan = 0;
    K prefix = 0;                   // digits fixed so far, right aligned
    uint64_t r = (uint64_t)r0;      // rank of s[r0] inside the current list
    int64_t n_act = n;              // length of the current list (cached) or of the segment
    K k0 = 0, k1 = 0;
    bool done = false;
    T msum = 0;
    int mcnt = 0;
    for (int pass = 0; pass < PASSES && !done; ++pass) {
        const int shift = BITS - 8 * (pass + 1);
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
        for (int64_t it = 0; it < its; ++it) {
            const int64_t i = it * SEL_THREADS + tid;
            const bool in = i < n_act;
            K k = 0;
            bool match = in;
            if (in) {
                if (pass == 0) {
                    const T x = v[i];
                    if (mk && mk[i]) {
                        msum += x;
                        ++mcnt;
                    }
                    T xx = x;
                    if (XF == XF_ABSDEV) xx = fabs(x - c);
                    k = xx != xx ? ~(K)0 : key_of(xx);
                    if (cached) cur[i] = k;
                    saw_nan |= (k == ~(K)0);
                } else if (cached) {
                    k = cur[i];                      // the list holds only keys with this prefix
                } else {
                    k = load_key(i);
                    match = (k >> (shift + 8)) == prefix;
                }
            }
            hist_add_matched(hist, (uint32_t)((k >> shift) & 0xFF), match);
        }
        if (pass == 0 && saw_nan) s_nan = 1;
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
                const K k = (cached ) ? cur[i] : load_key(i);
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
        } else if (pass + 1 == PASSES) {
            k0 = k1 = new_prefix;             // both order statistics are copies of one key
            done = true;
        } else {
            if (cached) {
                // gather the keys of the chosen bin: later passes walk this short list only
                for (int64_t it = 0; it < its; ++it) {
                    const int64_t i = it * SEL_THREADS + tid;
                    const bool in = i < n_act;
                    const K k = in ? cur[i] : 0;
                    const bool keep = in && (k >> shift) == new_prefix;
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
            if (mode == SEL_MEDIAN) {
                res = (r0 == r1) ? (double)a : (double)(T)((a + b) / (T)2);
            } else {
                const double da = (double)a, db = (double)b, diff = db - da;
                res = frac >= 0.5 ? db - diff * (1.0 - frac) : da + diff * frac;
                if (r0 == r1) res = da;
                res = (double)(T)res;
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
// Long segments: every segment is split over many workgroups.  Each pass histograms
// one 11-bit digit of the keys that still match the prefix (LDS histogram per
// workgroup, non-empty bins merged into a global histogram per segment); a one-
// workgroup-per-segment scan picks the bin holding the wanted rank and clears the
// histogram for the next pass.  32-bit keys take 3 passes, 64-bit keys 6.
static constexpr int BIG_BINS = 2048;
static constexpr int BIG_TILE = 8192;          // elements per workgroup
static constexpr int64_t BIG_THRESHOLD = 32768;

struct BigState {            // per segment
    unsigned long long prefix;    // key bits fixed so far (right aligned)
    unsigned long long rank;      // rank of s[r0] among the keys matching the prefix
    unsigned long long k0, k1;    // keys of s[r0], s[r1]
    unsigned long long cnt_le;    // extra pass: keys <= k0
    unsigned long long min_gt;    // extra pass: smallest key > k0
    int need_extra;               // s[r1] lies beyond the last histogram of s[r0]
    int has_nan;
};

__device__ __forceinline__ void big_ranks(int64_t n, int mode, double q, int64_t *r0, int64_t *r1, double *frac) {
    if (mode == SEL_MEDIAN) {
        *r0 = (n - 1) / 2;
        *r1 = n / 2;
        *frac = 0.5;
    } else {
        const double pos = (double)(n - 1) * (q / 100.0);
        int64_t lo = (int64_t)floor(pos);
        if (lo > n - 1) lo = n - 1;
        *r0 = lo;
        *frac = pos - (double)lo;
        *r1 = lo + 1 < n ? lo + 1 : n - 1;
    }
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
    if (!st[seg].need_extra) return;
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
    int64_t r0, r1;
    double frac;
    big_ranks(n, mode, q, &r0, &r1, &frac);
    const K k0 = (K)s.k0;
    const K k1 = s.need_extra ? (K)s.min_gt : (K)s.k1;
    const T a = val_of(k0), b = val_of(k1);
    double res;
    if (mode == SEL_MEDIAN) {
        res = (r0 == r1) ? (double)a : (double)(T)((a + b) / (T)2);
    } else {
        const double da = (double)a, db = (double)b, diff = db - da;
        res = frac >= 0.5 ? db - diff * (1.0 - frac) : da + diff * frac;
        if (r0 == r1) res = da;
        res = (double)(T)res;
    }
    out[seg] = res;
}

int64_t select_short_limit() { return BIG_THRESHOLD; }

size_t select_ws_bytes(int64_t n_seg, int64_t max_len) {
    if (max_len <= BIG_THRESHOLD) return 0;
    return ws_need((size_t)n_seg, sizeof(BigState)) + ws_need((size_t)n_seg * BIG_BINS, sizeof(uint32_t));
}

template <typename T, int XF>
static int launch_select_big(Ctx *c, const T *d_vals, const SelectArgs &a) {
    constexpr int BITS = KeyT<T>::bits;
    BigState *st = ws_take<BigState>(c, (size_t)a.n_seg);
    uint32_t *ghist = ws_take<uint32_t>(c, (size_t)a.n_seg * BIG_BINS);
    if (c->ws_used > c->ws_bytes) {
        set_error("internal: workspace not reserved for the long-segment select");
        return ADN_ERR_NOMEM;
    }
    const int64_t tiles = ceil_div(a.max_len, BIG_TILE);
    ADN_REQUIRE(a.n_seg * tiles < (int64_t)0x7FFFFFFF, "select batch too large for one launch");
    const unsigned nb = (unsigned)ceil_div(a.n_seg, 256);
    ADN_HIP(hipMemsetAsync(ghist, 0, sizeof(uint32_t) * a.n_seg * BIG_BINS, c->stream));
    hipLaunchKernelGGL(k_big_init, dim3(nb), dim3(256), 0, c->stream, st, a.d_seg_len, a.len_div, a.uniform_len,
                       a.n_seg, a.mode, a.q);
    int remaining = BITS;
    bool first = true;
    while (remaining > 0) {
        // widths 11, 11, 10 (32-bit keys) / 11 x 4, 10 x 2 (64-bit keys)
        const int passes_left = (remaining + 10) / 11;
        const int width = (remaining + passes_left - 1) / passes_left;
        remaining -= width;
        hipLaunchKernelGGL((k_big_hist<T, XF>), dim3((unsigned)(a.n_seg * tiles)), dim3(256), 0, c->stream, d_vals,
                           a.d_seg_off, a.seg_stride, a.d_seg_len, a.len_div, a.uniform_len, a.d_center, st, ghist,
                           tiles, remaining, width, first ? 1 : 0);
        hipLaunchKernelGGL((k_big_scan<T>), dim3((unsigned)a.n_seg), dim3(256), 0, c->stream, st, ghist,
                           a.d_seg_len, a.len_div, a.uniform_len, width, remaining == 0 ? 1 : 0, a.mode, a.q);
        first = false;
    }
    hipLaunchKernelGGL((k_big_next<T, XF>), dim3((unsigned)(a.n_seg * tiles)), dim3(256), 0, c->stream, d_vals,
                       a.d_seg_off, a.seg_stride, a.d_seg_len, a.len_div, a.uniform_len, a.d_center, st, tiles);
    hipLaunchKernelGGL((k_big_finish<T>), dim3(nb), dim3(256), 0, c->stream, st, a.d_seg_len, a.len_div,
                       a.uniform_len, a.n_seg, a.mode, a.q, a.d_out);
    ADN_HIP(hipGetLastError());
    return ADN_OK;
}

template <typename T>
int launch_select(Ctx *c, const T *d_vals, const SelectArgs &a) {
    if (a.n_seg <= 0) return ADN_OK;
    using K = typename KeyT<T>::type;
    if (a.max_len > BIG_THRESHOLD) {
        if (a.d_masked_mean) {
            set_error("internal: the fused masked mean is only available for LDS-resident segments");
            return ADN_ERR_UNSUPPORTED;
        }
        return a.xform == XF_ABSDEV ? launch_select_big<T, XF_ABSDEV>(c, d_vals, a)
                                    : launch_select_big<T, XF_NONE>(c, d_vals, a);
    }
    int lds_keys = (int)(SEL_LDS_BYTES / sizeof(K));
    if (a.max_len > 0 && a.max_len < lds_keys) lds_keys = (int)a.max_len;
    const size_t dyn = 2 * (size_t)lds_keys * sizeof(K);      // keys + candidate list
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
        hipLaunchKernelGGL((k_select<T, XF_ABSDEV>), grid, block, dyn, c->stream, d_vals,
                           a.d_seg_off, a.seg_stride, a.d_seg_len, a.len_div, a.uniform_len,
                           a.mode, a.q, a.d_center, a.d_out, lds_keys, mm);
    } else {
        hipLaunchKernelGGL((k_select<T, XF_NONE>), grid, block, dyn, c->stream, d_vals,
                           a.d_seg_off, a.seg_stride, a.d_seg_len, a.len_div, a.uniform_len,
                           a.mode, a.q, a.d_center, a.d_out, lds_keys, mm);
    }
    ADN_HIP(hipGetLastError());
    return ADN_OK;
}

template int launch_select<float>(Ctx *, const float *, const SelectArgs &);
template int launch_select<double>(Ctx *, const double *, const SelectArgs &);

}  // namespace adn
