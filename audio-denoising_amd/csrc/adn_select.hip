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

template <typename T, int XF>
__global__ __launch_bounds__(SEL_THREADS) void k_select(const T *__restrict__ vals,
                                                        const int64_t *__restrict__ seg_off,
                                                        int64_t seg_stride,
                                                        const int64_t *__restrict__ seg_len,
                                                        int64_t len_div, int64_t uniform_len,
                                                        int mode, double q,
                                                        const double *__restrict__ center,
                                                        double *__restrict__ out, int lds_keys) {
    using K = typename KeyT<T>::type;
    constexpr int BITS = KeyT<T>::bits;
    extern __shared__ __align__(16) unsigned char smem[];
    K *keys = reinterpret_cast<K *>(smem);
    __shared__ uint32_t hist[256];
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t s_digit, s_before, s_nan;
    __shared__ unsigned long long s_cnt_le;
    __shared__ K s_min_gt;

    const int64_t seg = blockIdx.x;
    const int64_t off = seg_off ? seg_off[seg] : seg * seg_stride;
    const int64_t n = seg_len ? seg_len[seg / len_div] : uniform_len;
    const int tid = threadIdx.x;
    if (n <= 0) {
        if (tid == 0) out[seg] = __longlong_as_double(0x7FF8000000000000ll);
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
    auto load_key = [&](int64_t i) -> K {
        T x = v[i];
        if (XF == XF_ABSDEV) x = fabs(x - c);
        return x != x ? ~(K)0 : key_of(x);
    };

    if (tid == 0) s_nan = 0;
    K prefix = 0;
    uint64_t r = (uint64_t)r0;
    const int64_t iters = (n + SEL_THREADS - 1) / SEL_THREADS;
    for (int pass = 0; pass < BITS / 8; ++pass) {
        const int shift = BITS - 8 * (pass + 1);
        hist[tid] = 0;
        __syncthreads();
        bool saw_nan = false;
        for (int64_t it = 0; it < iters; ++it) {
            const int64_t i = it * SEL_THREADS + tid;
            const bool in = i < n;
            K k = 0;
            if (in) {
                if (pass == 0) {
                    k = load_key(i);
                    if (cached) keys[i] = k;
                    saw_nan |= (k == ~(K)0);
                } else {
                    k = cached ? keys[i] : load_key(i);
                }
            }
            const bool match = in && (pass == 0 || (k >> (shift + 8)) == prefix);
            hist_add_matched(hist, (uint32_t)((k >> shift) & 0xFF), match);
        }
        if (pass == 0 && saw_nan) s_nan = 1;
        __syncthreads();
        const uint32_t h = hist[tid];
        const uint32_t before = block_excl_scan(h, wsum);
        if (r >= before && r < (uint64_t)before + h) {
            s_digit = tid;
            s_before = before;
        }
        __syncthreads();
        prefix = (prefix << 8) | (K)s_digit;
        r -= s_before;
        __syncthreads();
    }
    const K k0 = prefix;
    K k1 = k0;
    if (r1 != r0) {
        // s[r0+1] equals s[r0] when at least r0+2 keys are <= k0, else the smallest key above k0
        if (tid == 0) {
            s_cnt_le = 0;
            s_min_gt = ~(K)0;
        }
        __syncthreads();
        unsigned long long cnt = 0;
        K mn = ~(K)0;
        for (int64_t i = tid; i < n; i += SEL_THREADS) {
            const K k = cached ? keys[i] : load_key(i);
            cnt += (k <= k0);
            if (k > k0 && k < mn) mn = k;
        }
        for (int d = 32; d >= 1; d >>= 1) {
            cnt += __shfl_xor(cnt, d);
            const K o = __shfl_xor(mn, d);
            mn = o < mn ? o : mn;
        }
        if ((tid & 63) == 0) {
            atomicAdd(&s_cnt_le, cnt);
            atomicMin(&s_min_gt, mn);
        }
        __syncthreads();
        k1 = s_cnt_le >= (unsigned long long)r0 + 2 ? k0 : s_min_gt;
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
    }
}

template <typename T>
int launch_select(Ctx *c, const T *d_vals, const SelectArgs &a) {
    if (a.n_seg <= 0) return ADN_OK;
    using K = typename KeyT<T>::type;
    int lds_keys = (int)(SEL_LDS_BYTES / sizeof(K));
    if (a.max_len > 0 && a.max_len < lds_keys) lds_keys = (int)a.max_len;
    const size_t dyn = (size_t)lds_keys * sizeof(K);
    dim3 grid((unsigned)a.n_seg), block(SEL_THREADS);
    if (a.xform == XF_ABSDEV) {
        hipLaunchKernelGGL((k_select<T, XF_ABSDEV>), grid, block, dyn, c->stream, d_vals,
                           a.d_seg_off, a.seg_stride, a.d_seg_len, a.len_div, a.uniform_len,
                           a.mode, a.q, a.d_center, a.d_out, lds_keys);
    } else {
        hipLaunchKernelGGL((k_select<T, XF_NONE>), grid, block, dyn, c->stream, d_vals,
                           a.d_seg_off, a.seg_stride, a.d_seg_len, a.len_div, a.uniform_len,
                           a.mode, a.q, a.d_center, a.d_out, lds_keys);
    }
    ADN_HIP(hipGetLastError());
    return ADN_OK;
}

template int launch_select<float>(Ctx *, const float *, const SelectArgs &);
template int launch_select<double>(Ctx *, const double *, const SelectArgs &);

}  // namespace adn
