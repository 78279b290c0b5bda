// Threshold statistics of stereo blocks in ONE pass over the input (included by adn_dwt.hip).
//
// The reference's threshold is sigma * sqrt(2 ln n) with sigma = median(|d - median(d)|) over the
// finest detail coefficients d of a block (denoise.py:16-23, 80-83).  For a stereo block at level 2
// every frame has one such coefficient, d = w0 * left + w1 * right.  The two medians are nested, so
// the plain route materialises d and selects twice (three passes over it).  Here:
//
//   1. k_m1_sample: a stratified sample of S coefficients per block (runs of 8 consecutive coefficients)
//      gives (a) a key bracket [lo_m, hi_m] that holds the block's median ranks with probability
//      ~ 1 - 1e-5 and (b), with m^ = the bracket's midpoint, a value bracket [alo, ahi] for
//      |d - m^| that holds the ranks of the absolute deviation, widened by the largest possible
//      |median - m^| so that it still holds them once the true median is known.
//   2. k_m1_gather: one pass over the block: coefficients on the fly, keys below each bracket
//      counted, keys inside gathered (~3.5 % and ~9 % of the block).
//   3. Exact selection among the gathered keys (LDS radix select with rank offsets): the median,
//      then |d - median| of the second list.
//   4. m1_verified (first step of the last kernel) proves the second selection: both order statistics must lie inside
//      (alo + e, ahi - e) with e = |median - m^| + rounding slack; then every key counted as
//      "below" really is below them and every key left out above really is above.
//   5. A bracket that is ONE value (silence, quantised quiet passages: so many coefficients tie
//      that sample ranks lo and hi hold the same key) gathers nothing: the keys equal to it are
//      counted, and the counts alone prove the median (and, with the median pinned, the deviation).
//   6. k_m1_fallback: blocks that fail any check (bracket missed, list or stage overflow, non-finite
//      data, strongly autocorrelated coefficients for which a run of consecutive coefficients is worth few samples)
//      are selected exactly by one workgroup streaming the block: one sweep histograms the leading
//      11 bits of the on-the-fly keys, a second gathers the rank's bin into the block's (now useless)
//      candidate list, the remaining digits are found there.  Resolved blocks cost it one early exit;
//      the same kernel writes the thresholds.
//
// Every result is an exact order statistic of the same multiset the plain route selects from.
#pragma once

template <typename T>
struct M1 {
    using K = typename KeyT<T>::type;
    // sample keys per block, LDS resident.  binary32: 15 per thread = 60 KB, so that with the two histograms two
    // sampling workgroups share a CU; binary64: 128 KB
    static constexpr int S = sizeof(T) == 4 ? 15360 : 16384;
    static constexpr double FRAC_M = 0.05;         // list capacities as fractions of the block: the brackets hold
    static constexpr double FRAC_A = 0.13;         // ~3.5 % and ~9 % of a Gaussian block
};
static constexpr int M1_THREADS = 1024;     // sample / fallback workgroups
static constexpr int M1_RUN = 8;            // consecutive coefficients per sample run (a 128-byte line holds 16 stereo binary32 frames: shorter runs cost
                                            // more lines, longer ones are worth fewer samples when the coefficients are autocorrelated)
static constexpr int M1_BINS = 2048;        // 11-bit digits
static constexpr int M1_TILE = 4096;        // frames per gather workgroup (256 threads x 16)
static constexpr int M1_STAGE_M = 1024;     // bracket keys a tile can stage: median list (~3.5 % expected),
static constexpr int M1_STAGE_A = 2048;     // absolute-deviation list (~9 %)

template <typename T>
__device__ __forceinline__ T m1_d1(T w0, T w1, T v0, T v1) {
#pragma clang fp contract(off)     // two rounded products and one rounded sum everywhere the coefficient is formed
    T s = 0;
    s += w0 * v0;
    s += w1 * v1;
    return s;
}
// Where the coefficients come from.  A source knows how many coefficients a block of `frames` frames
// has and forms any one of them from the block's samples; S is the size of the bracket sample and
// FRAC_M / FRAC_A the candidate-list capacities (fractions of a block) that go with it: the brackets
// of a Gaussian block hold ~3.5 % and ~9 % of it for a 16 K sample, ~7 % and ~19 % for a 4 K sample.
template <typename T>
struct SrcStereo {                 // d = w0 * left + w1 * right, one coefficient per frame
    static constexpr bool STEREO = true;
    static constexpr int S = M1<T>::S, SAMPLE_THREADS = M1_THREADS;
    static constexpr double FRAC_M = M1<T>::FRAC_M, FRAC_A = M1<T>::FRAC_A;
    T w0, w1;
    __host__ __device__ static int64_t count(int64_t frames) { return frames; }
    __device__ __forceinline__ T at(const T *__restrict__ in, int64_t frames, int64_t i) const {
        return m1_d1(w0, w1, in[2 * i], in[2 * i + 1]);
    }
};
// Finest detail band of a mono block, periodization: d[k] = sum_j g[j] xe[(2 k + F/2 - j) mod M], M = frames
// rounded up to even, xe = the block with its last sample repeated when frames is odd.  Fused
// multiply-adds in tap order from zero: the operation sequence of the tile kernels, so that every
// kernel forms the same bits for a coefficient.
template <typename T, int F_, int S_>
struct SrcMono {
    static constexpr bool STEREO = false;
    static constexpr int F = F_, S = S_;
    static constexpr int SAMPLE_THREADS = S_ <= 4096 ? 256 : M1_THREADS;     // a small sample: small workgroups, cheap barriers
    static constexpr double FRAC_M = S_ <= 4096 ? 0.10 : (S_ <= 8192 ? 0.075 : 0.05);
    static constexpr double FRAC_A = S_ <= 4096 ? 0.25 : (S_ <= 8192 ? 0.19 : 0.13);
    T g[F_];
    __host__ __device__ static int64_t count(int64_t frames) { return (frames + 1) >> 1; }
    // the F samples under the taps of coefficient k (v[j] meets tap j): independent loads
    __device__ __forceinline__ void window(const T *__restrict__ in, int64_t frames, int64_t k, T (&v)[F_]) const {
        const int n = (int)frames, M = n + (n & 1);
        int idx0 = 2 * (int)k + F / 2;               // k < M / 2: one subtraction wraps it unless the block is tiny
        while (idx0 >= M) idx0 -= M;
        if (idx0 >= F - 1 && idx0 < n) {             // the window neither wraps nor touches the repeated sample
            const T *w = in + idx0;
#pragma unroll
            for (int j = 0; j < F; ++j) v[j] = w[-j];
        } else {
#pragma unroll
            for (int j = 0; j < F; ++j) {
                int idx = idx0 - j;
                while (idx < 0) idx += M;            // F / M + 1 rounds at most
                v[j] = in[idx < n ? idx : n - 1];
            }
        }
    }
    __device__ __forceinline__ T eval(const T (&v)[F_]) const {
        T sd = 0;
#pragma unroll
        for (int j = 0; j < F; ++j) sd = fma(g[j], v[j], sd);
        return sd;
    }
    __device__ __forceinline__ T at(const T *__restrict__ in, int64_t frames, int64_t k) const {
        T v[F_];
        window(in, frames, k, v);
        return eval(v);
    }
};

// Histogram update of a wave: when every active lane holds the same digit (constant data, or the
// leading digits of narrow data) one lane adds the whole count, otherwise every lane adds its own.
__device__ __forceinline__ void m1_hist_add(uint32_t *hist, uint32_t digit, bool active) {
    const unsigned long long act = __ballot(active);
    if (act == 0) return;
    const int first = (int)__ffsll((long long)act) - 1;
    const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)digit, first);
    const unsigned long long same = __ballot(active && digit == d0);
    if (same == act) {
        if ((int)(threadIdx.x & 63) == first) atomicAdd(&hist[d0], (uint32_t)__popcll(act));
    } else if (active) {
        atomicAdd(&hist[digit], 1u);
    }
}

// Bin of `rank` in a histogram of M1_BINS counters: all NT threads, M1_BINS / NT consecutive bins each.
// Returns through dig / bef (LDS words) the bin and the count of keys in lower bins.
template <int NT>
__device__ __forceinline__ uint32_t m1_scan_bins(const uint32_t *hist, uint32_t (&cbin)[M1_BINS / NT], uint32_t *wtot) {
    constexpr int BPT = M1_BINS / NT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t sum = 0;
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
        cbin[q] = hist[BPT * tid + q];
        sum += cbin[q];
    }
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; ++w) base += wtot[w];
    return incl + base - sum;                       // keys in the bins below this thread's
}
template <int NT>
__device__ __forceinline__ void m1_pick_bin(const uint32_t (&cbin)[M1_BINS / NT], uint32_t excl, uint32_t rank, uint32_t *dig,
                                            uint32_t *bef) {
    constexpr int BPT = M1_BINS / NT;
    uint32_t lo = excl;
#pragma unroll
    for (int q = 0; q < BPT; ++q) {
        if (lo <= rank && rank < lo + cbin[q]) {
            *dig = (uint32_t)(BPT * threadIdx.x + q);
            *bef = lo;
        }
        lo += cbin[q];
    }
}
template <int NT = M1_THREADS>
__device__ __forceinline__ void m1_find_bin(const uint32_t *hist, uint32_t rank, uint32_t *wtot, uint32_t *dig,
                                            uint32_t *bef) {
    uint32_t cbin[M1_BINS / NT];
    const uint32_t excl = m1_scan_bins<NT>(hist, cbin, wtot);
    m1_pick_bin<NT>(cbin, excl, rank, dig, bef);
    __syncthreads();
}
// The bins of two ranks of the SAME histogram from one scan: results in dig[0..1], bef[0..1].
template <int NT = M1_THREADS>
__device__ __forceinline__ void m1_find_bin2(const uint32_t *hist, uint32_t ra, uint32_t rb, uint32_t *wtot,
                                             uint32_t *dig, uint32_t *bef) {
    uint32_t cbin[M1_BINS / NT];
    const uint32_t excl = m1_scan_bins<NT>(hist, cbin, wtot);
    m1_pick_bin<NT>(cbin, excl, ra, &dig[0], &bef[0]);
    m1_pick_bin<NT>(cbin, excl, rb, &dig[1], &bef[1]);
    __syncthreads();
}

// Keys that bracket the ranks ra <= rb of the S sample keys in LDS: 11-bit MSD radix selection on
// key - kmin, both ranks in the same sweeps (a second histogram once their prefixes part).
// M1_PASSES digits are resolved (all of them while the two ranks share their prefix: tied samples
// give ka == kb exactly); the bits below them are rounded outwards (*ka <= key of rank ra,
// *kb >= key of rank rb), which widens a bracket by less than 2^-10 of an octave.
static constexpr int M1_PASSES = 2;
template <typename K, int S, int NT>
__device__ void m1_select2(const K *sk, K kmin, K kmax, uint32_t ra, uint32_t rb, uint32_t (*hist)[M1_BINS],
                           uint32_t (*wtot)[NT / 64], uint32_t *s_dig, uint32_t *s_bef, K *ka, K *kb) {
    constexpr int BITS = (int)sizeof(K) * 8;
    const int tid = threadIdx.x;
    const K width = kmax - kmin;
    const int bits = width ? BITS - (BITS == 64 ? __clzll((long long)width) : __clz((int)width)) : 0;
    const int passes = (bits + 10) / 11;
    int done = 0;
    K pa = 0, pb = 0;
    // beyond M1_PASSES digits only while both ranks still share their prefix (ties need the exact key)
    for (int p = 0; p < passes && (p < M1_PASSES || pa == pb); ++p) {
        done = p + 1;
        const int shift = 11 * (passes - 1 - p);
        const bool same = pa == pb;
        for (int i = tid; i < 2 * M1_BINS; i += NT) (&hist[0][0])[i] = 0;
        __syncthreads();
        for (int j = tid; j < S; j += NT) {
            const K u = sk[j] - kmin;
            const K top = p == 0 ? (K)0 : (K)(u >> (shift + 11));
            const uint32_t digit = (uint32_t)(u >> shift) & (M1_BINS - 1);
            if (top == pa) atomicAdd(&hist[0][digit], 1u);
            if (!same && top == pb) atomicAdd(&hist[1][digit], 1u);
        }
        __syncthreads();
        if (same) {
            m1_find_bin2<NT>(hist[0], ra, rb, wtot[0], s_dig, s_bef);
        } else {
            m1_find_bin<NT>(hist[0], ra, wtot[0], &s_dig[0], &s_bef[0]);
            m1_find_bin<NT>(hist[1], rb, wtot[1], &s_dig[1], &s_bef[1]);
        }
        pa = (pa << 11) | (K)s_dig[0];
        pb = (pb << 11) | (K)s_dig[1];
        ra -= s_bef[0];
        rb -= s_bef[1];
        __syncthreads();
    }
    const int rest = 11 * (passes - done);            // unresolved low bits
    *ka = kmin + (pa << rest);
    const K top_b = (pb << rest) | (rest ? (((K)1 << rest) - 1) : (K)0);
    *kb = top_b > width ? kmax : kmin + top_b;
}

template <typename T, typename Src>
__global__ __launch_bounds__(Src::SAMPLE_THREADS) void k_m1_sample(const T *__restrict__ x, BlockGeom g, Src src,
                                                          M1State *__restrict__ st, int64_t *__restrict__ counts) {
    using K = typename KeyT<T>::type;
    constexpr int S = Src::S, NT = Src::SAMPLE_THREADS, PER = S / NT, NRUN = S / M1_RUN;
    extern __shared__ __align__(16) unsigned char smem[];
    K *sk = reinterpret_cast<K *>(smem);
    __shared__ uint32_t hist[2][M1_BINS];
    __shared__ uint32_t wtot[2][NT / 64];
    __shared__ uint32_t s_dig[2], s_bef[2];
    __shared__ K s_kmin, s_kmax;
    const int64_t b = blockIdx.x;
    const int64_t frames = g.frames[b];
    const int64_t n = Src::count(frames);              // coefficients of the block
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid == 0 && counts) counts[b] = n;
    M1State s{};
    if (n <= S) {
        // short block: every coefficient is a candidate of both selections
        if (tid == 0) {
            s.lo_m = 0;
            s.hi_m = ~0ull;
            s.mhat = 0.0;
            s.alo = -1.0;
            s.ahi = __longlong_as_double(0x7FF0000000000000ll);
            s.usable = 1;
            st[b] = s;
        }
        return;
    }
    const T *in = x + g.in_off[b];
    if (tid == 0) {
        s_kmin = ~(K)0;
        s_kmax = 0;
    }
    __syncthreads();
    K tmin = ~(K)0, tmax = 0;
    if constexpr (Src::STEREO) {
        T v0[PER], v1[PER];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int j = tid + NT * u;
            const int run = j / M1_RUN, w = j % M1_RUN;
            int64_t i = sample_run_start(n, NRUN, M1_RUN, run) + w;       // (n < 2^31: the products fit 64 bits)
            i = i < 0 ? 0 : (i < n ? i : n - 1);
            v0[u] = in[2 * i];
            v1[u] = in[2 * i + 1];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const K k = m1_key(m1_d1(src.w0, src.w1, v0[u], v1[u]));
            sk[tid + NT * u] = k;
            tmin = k < tmin ? k : tmin;
            tmax = k > tmax ? k : tmax;
        }
    } else {
        // four coefficients at a time: their 4 F loads are in flight together
        constexpr int G = PER < 4 ? PER : 4;
        static_assert(PER % G == 0, "sample coefficients per thread");
#pragma unroll
        for (int u0 = 0; u0 < PER; u0 += G) {
            T v[G][Src::F];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < G; ++q) {
                const int j = tid + NT * (u0 + q);
                const int run = j / M1_RUN, w = j % M1_RUN;
                int64_t i = sample_run_start(n, NRUN, M1_RUN, run) + w;
                i = i < 0 ? 0 : (i < n ? i : n - 1);
                src.window(in, frames, i, v[q]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < G; ++q) {
                const K k = m1_key(src.eval(v[q]));
                sk[tid + NT * (u0 + q)] = k;
                tmin = k < tmin ? k : tmin;
                tmax = k > tmax ? k : tmax;
            }
        }
    }
    for (int d = 32; d >= 1; d >>= 1) {
        const K a = __shfl_xor(tmin, d), c = __shfl_xor(tmax, d);
        tmin = a < tmin ? a : tmin;
        tmax = c > tmax ? c : tmax;
    }
    if (lane == 0) {
        atomicMin(&s_kmin, tmin);
        atomicMax(&s_kmax, tmax);
    }
    __syncthreads();
    // sample ranks that bracket the block's median ranks (4.5 sigma of the binomial + 3)
    const int64_t r0 = (n - 1) / 2;
    const int delta = (int)(4.5 * sqrt((double)S * 0.25)) + 3;
    const int rs = (int)((double)r0 * (double)S / (double)n);
    int want_lo = rs - delta, want_hi = rs + delta + 1;
    const bool open_lo = want_lo < 0, open_hi = want_hi >= S;
    want_lo = open_lo ? 0 : want_lo;
    want_hi = open_hi ? S - 1 : want_hi;
    K klo = 0, khi = 0;
    m1_select2<K, S, NT>(sk, s_kmin, s_kmax, (uint32_t)want_lo, (uint32_t)want_hi, hist, wtot, s_dig, s_bef, &klo, &khi);
    const T lo_v = val_of(klo), hi_v = val_of(khi);
    const bool finite = klo != ~(K)0 && khi != ~(K)0 && fabs(lo_v) <= (T)1e30 && fabs(hi_v) <= (T)1e30;
    if (!finite || open_lo || open_hi) {
        // non-finite data or a degenerate sample: the exact fallback takes the block
        if (tid == 0) {
            s.lo_m = 1;
            s.hi_m = 0;
            s.alo = 1.0;
            s.ahi = 0.0;
            s.usable = 0;
            st[b] = s;
        }
        return;
    }
    // klo == khi: so many coefficients tie around the median (silence, quantised quiet passages) that the
    // bracket is ONE value; it is then the median itself if the counts of the pass confirm it
    const bool tie_m = klo == khi;
    const T mhat = tie_m ? lo_v : lo_v + (hi_v - lo_v) / (T)2;
    const double eps_v = tie_m ? 0.0 : fmax((double)hi_v - (double)mhat, (double)mhat - (double)lo_v);
    // the same sample as |d - mhat|
    __syncthreads();
    if (tid == 0) {
        s_kmin = ~(K)0;
        s_kmax = 0;
    }
    __syncthreads();
    tmin = ~(K)0;
    tmax = 0;
    for (int j = tid; j < S; j += NT) {
        const K kk = sk[j];
        const K k = kk == ~(K)0 ? kk : m1_key((T)fabs(val_of(kk) - mhat));
        sk[j] = k;
        tmin = k < tmin ? k : tmin;
        tmax = k > tmax ? k : tmax;
    }
    for (int d = 32; d >= 1; d >>= 1) {
        const K a = __shfl_xor(tmin, d), c = __shfl_xor(tmax, d);
        tmin = a < tmin ? a : tmin;
        tmax = c > tmax ? c : tmax;
    }
    if (lane == 0) {
        atomicMin(&s_kmin, tmin);
        atomicMax(&s_kmax, tmax);
    }
    __syncthreads();
    K kalo = 0, kahi = 0;
    m1_select2<K, S, NT>(sk, s_kmin, s_kmax, (uint32_t)want_lo, (uint32_t)want_hi, hist, wtot, s_dig, s_bef, &kalo, &kahi);
    if (tid == 0) {
        const double eps_t = sizeof(T) == 4 ? 1.1920929e-7 : 2.220446049250313e-16;
        const bool open_above = kahi == ~(K)0 || !(fabs(val_of(kahi)) <= (T)1e30);
        const double a_lo = (double)val_of(kalo), a_hi = open_above ? 0.0 : (double)val_of(kahi);
        const double slop = 16.0 * eps_t * (a_hi + fabs((double)mhat) + eps_v) + 1e-300;
        const double alo = a_lo - eps_v - slop, ahi = a_hi + eps_v + slop;
        // with the median pinned, tied absolute deviations are counted the same way
        const bool tie_a = tie_m && kalo == kahi && !open_above;
        s.lo_m = (unsigned long long)klo;
        s.hi_m = (unsigned long long)khi;
        s.mhat = (double)mhat;
        s.alo = tie_a ? a_lo : (alo <= 0.0 ? -1.0 : (double)(T)alo);
        s.ahi = tie_a ? a_lo : (open_above ? __longlong_as_double(0x7FF0000000000000ll) : (double)(T)ahi);
        s.tie_m = tie_m ? 1 : 0;
        s.tie_a = tie_a ? 1 : 0;
        s.usable = 1;
        st[b] = s;
    }
}

// The sixteen coefficients of a thread against both brackets: counts of the keys below them and the
// keys inside appended to the workgroup's LDS stages (positions from wave ballots, eight keys at a
// time so that the ballots stay in scalar registers).  FULL: every coefficient exists (a tile inside
// its block), no per-key bounds.
template <typename T, bool FULL, int BATCH = 16>
__device__ __forceinline__ void m1_classify(const T (&d)[BATCH], const bool (&live_in)[BATCH], typename KeyT<T>::type lo,
                                            typename KeyT<T>::type hi, T mhat, T alo, T ahi, T *stage_m, T *stage_a,
                                            unsigned int *s_fill_m, unsigned int *s_fill_a, unsigned int &below_m,
                                            unsigned int &below_a, bool &saw_nan, bool tie_m, bool tie_a) {
    using K = typename KeyT<T>::type;
    constexpr int SUB = 8;
    const int lane = threadIdx.x & 63;
    const K span = hi - lo;
#pragma unroll
    for (int h = 0; h < BATCH; h += SUB) {
        // ---- median bracket (keys)
        {
            bool keep[SUB];
            unsigned long long bal[SUB];
            unsigned int total = 0;
#pragma unroll
            for (int u = 0; u < SUB; ++u) {
                const bool live = FULL || live_in[h + u];
                const T xv = d[h + u];
                saw_nan |= live && (xv != xv);
                const K k = m1_key(xv);
                below_m += (live && k < lo);
                keep[u] = live && (K)(k - lo) <= span;
                bal[u] = __ballot(keep[u]);
                total += (unsigned int)__popcll(bal[u]);
            }
            if (total) {
                unsigned int base = 0;
                if (lane == 0) base = atomicAdd(s_fill_m, total);
                base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
                for (int u = 0; u < SUB; ++u) {
                    const unsigned int pos = base + __builtin_amdgcn_mbcnt_hi((unsigned int)(bal[u] >> 32),
                                                                              __builtin_amdgcn_mbcnt_lo((unsigned int)bal[u], 0u));
                    if (keep[u] && pos < M1_STAGE_M && !tie_m) stage_m[pos] = d[h + u];
                    base += (unsigned int)__popcll(bal[u]);
                }
            }
        }
        // ---- absolute-deviation bracket (values of |d - mhat|); the coefficient itself is gathered
        {
            bool keep[SUB];
            unsigned long long bal[SUB];
            unsigned int total = 0;
#pragma unroll
            for (int u = 0; u < SUB; ++u) {
                const bool live = FULL || live_in[h + u];
                const T a = fabs(d[h + u] - mhat);
                below_a += (live && a < alo);
                keep[u] = live && a >= alo && a <= ahi;
                bal[u] = __ballot(keep[u]);
                total += (unsigned int)__popcll(bal[u]);
            }
            if (total) {
                unsigned int base = 0;
                if (lane == 0) base = atomicAdd(s_fill_a, total);
                base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
                for (int u = 0; u < SUB; ++u) {
                    const unsigned int pos = base + __builtin_amdgcn_mbcnt_hi((unsigned int)(bal[u] >> 32),
                                                                              __builtin_amdgcn_mbcnt_lo((unsigned int)bal[u], 0u));
                    if (keep[u] && pos < M1_STAGE_A && !tie_a) stage_a[pos] = d[h + u];
                    base += (unsigned int)__popcll(bal[u]);
                }
            }
        }
    }
}

// Second half of a gather workgroup: the BATCH coefficients of every thread (runs of ADJ neighbouring ones:
// coefficient ADJ ((u / ADJ) NT + tid) + u % ADJ of the
// tile that starts at coefficient f0 of block b, which has n of them) are classified, the counts and
// the staged keys go to the block's state and lists.
template <typename T, bool FULL, int BATCH = 16, int NT = 256, int ADJ = 1>
__device__ __forceinline__ void m1_gather_finish(const T (&d)[BATCH], const bool (&live)[BATCH], int64_t b, int64_t f0, int64_t n,
                                                 typename KeyT<T>::type lo, typename KeyT<T>::type hi, T mhat, T alo, T ahi,
                                                 M1State *__restrict__ st, T *__restrict__ cand_m, int64_t cap_m,
                                                 T *__restrict__ cand_a, int64_t cap_a, T *stage_m, T *stage_a,
                                                 unsigned long long *s_below_m, unsigned long long *s_below_a,
                                                 unsigned int *s_fill_m, unsigned int *s_fill_a, unsigned int *s_base_m,
                                                 unsigned int *s_base_a, int *s_nan) {
    using K = typename KeyT<T>::type;
    static_assert(BATCH * NT == M1_TILE, "a workgroup takes one tile");
    const int tid = threadIdx.x, lane = tid & 63;
    unsigned int below_m = 0, below_a = 0;
    bool saw_nan = false;
    if (lo == 0 && hi == ~(K)0) {
        // short block, every coefficient is a candidate of both lists: stored in place, no staging
        T *dm = cand_m + b * cap_m + f0, *da = cand_a + b * cap_a + f0;
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            // a thread holds ADJ neighbouring coefficients in d[ADJ q .. ADJ q + ADJ - 1]
            const int64_t f = ADJ * ((int64_t)(u / ADJ) * NT + tid) + u % ADJ;
            if (live[u]) {
                dm[f] = d[u];
                da[f] = d[u];
                saw_nan |= d[u] != d[u];
            }
        }
        if (saw_nan) st[b].has_nan = 1;
        if (tid == 0) {
            const unsigned int cnt = (unsigned int)(n - f0 < M1_TILE ? n - f0 : M1_TILE);
            atomicAdd(m1_counts(&st[b]), (unsigned long long)cnt | ((unsigned long long)cnt << 32));
        }
        return;
    }
    const bool tie_m = st[b].tie_m != 0, tie_a = st[b].tie_a != 0;
    m1_classify<T, FULL, BATCH>(d, live, lo, hi, mhat, alo, ahi, stage_m, stage_a, s_fill_m, s_fill_a, below_m, below_a, saw_nan,
                         tie_m, tie_a);
    for (int dd = 32; dd >= 1; dd >>= 1) {
        below_m += __shfl_xor(below_m, dd);
        below_a += __shfl_xor(below_a, dd);
    }
    if (lane == 0) {
        if (below_m) atomicAdd(s_below_m, (unsigned long long)below_m);
        if (below_a) atomicAdd(s_below_a, (unsigned long long)below_a);
    }
    if (saw_nan) *s_nan = 1;
    __syncthreads();
    // tied brackets: the keys are only counted (n_m / n_a), nothing is written
    const unsigned int cm = tie_m ? 0u : (*s_fill_m < M1_STAGE_M ? *s_fill_m : M1_STAGE_M);
    const unsigned int ca = tie_a ? 0u : (*s_fill_a < M1_STAGE_A ? *s_fill_a : M1_STAGE_A);
    if (tid == 0) {
        if ((!tie_m && *s_fill_m > M1_STAGE_M) || (!tie_a && *s_fill_a > M1_STAGE_A)) st[b].spill = 1;
        const unsigned int add_m = tie_m ? *s_fill_m : cm, add_a = tie_a ? *s_fill_a : ca;
        // both list lengths in one returning atomic (n_m and n_a share an aligned 64-bit word; neither reaches 2^32):
        // the tiles of a block queue up on the block's state, and it is the returning atomics they wait for
        unsigned long long was = 0;
        if (add_m | add_a) was = atomicAdd(m1_counts(&st[b]), (unsigned long long)add_m | ((unsigned long long)add_a << 32));
        *s_base_m = (unsigned int)was;
        *s_base_a = (unsigned int)(was >> 32);
        if (*s_below_m) atomicAdd(&st[b].below_m, *s_below_m);
        if (*s_below_a) atomicAdd(&st[b].below_a, *s_below_a);
        if (*s_nan) st[b].has_nan = 1;
    }
    __syncthreads();
    {
        const int64_t base = *s_base_m;
        T *dst = cand_m + b * cap_m;
        for (unsigned int p = tid; p < cm; p += NT)
            if (base + p < cap_m) dst[base + p] = stage_m[p];      // beyond the capacity: seen by n_m > cap_m
    }
    {
        const int64_t base = *s_base_a;
        T *dst = cand_a + b * cap_a;
        for (unsigned int p = tid; p < ca; p += NT)
            if (base + p < cap_a) dst[base + p] = stage_a[p];
    }
}

// One pass over the blocks: keys below each bracket counted, keys inside gathered.
// FULL: workgroup (block b, tile t) of a grid of n_blocks x tiles_per_block takes the tile if it lies
// inside the block (one frame per load instruction, no bounds); the other instantiation runs one
// workgroup per block on the block's last, partial tile.
template <typename T, bool FULL>
__global__ __launch_bounds__(256) void k_m1_gather(const T *__restrict__ x, BlockGeom g, T w0, T w1,
                                                   M1State *__restrict__ st, T *__restrict__ cand_m, int64_t cap_m,
                                                   T *__restrict__ cand_a, int64_t cap_a, int64_t tiles_per_block) {
    using K = typename KeyT<T>::type;
    using V2 = T __attribute__((ext_vector_type(2)));
    constexpr int BATCH = 16;                               // frames per thread
    __shared__ T stage_m[M1_STAGE_M], stage_a[M1_STAGE_A];
    __shared__ unsigned long long s_below_m, s_below_a;
    __shared__ unsigned int s_fill_m, s_fill_a, s_base_m, s_base_a;
    __shared__ int s_nan;
    const int64_t b = FULL ? blockIdx.x / tiles_per_block : blockIdx.x;
    const int64_t n = g.frames[b];
    const int64_t tile = FULL ? blockIdx.x % tiles_per_block : n / M1_TILE;
    const int64_t f0 = tile * M1_TILE;
    if (FULL ? f0 + M1_TILE > n : f0 >= n) return;
    const K lo = (K)st[b].lo_m, hi = (K)st[b].hi_m;
    if (lo > hi) return;
    const T mhat = (T)st[b].mhat, alo = (T)st[b].alo, ahi = (T)st[b].ahi;
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid == 0) {
        s_below_m = 0;
        s_below_a = 0;
        s_fill_m = 0;
        s_fill_a = 0;
        s_nan = 0;
    }
    __syncthreads();
    const T *in = x + g.in_off[b] + 2 * f0;
    T d[BATCH];
    bool live[BATCH];
    if (FULL) {
        // a frame (two samples) is one load when the block starts at an even element of an aligned buffer
        V2 q[BATCH];
        if ((reinterpret_cast<uintptr_t>(in) & (sizeof(V2) - 1)) == 0) {
            const V2 *v2 = reinterpret_cast<const V2 *>(in);
#pragma unroll
            for (int u = 0; u < BATCH; ++u) q[u] = v2[u * 256 + tid];
        } else {
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                q[u].x = in[2 * (u * 256 + tid)];
                q[u].y = in[2 * (u * 256 + tid) + 1];
            }
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            d[u] = m1_d1(w0, w1, q[u].x, q[u].y);
            live[u] = true;
        }
    } else {
        T v0[BATCH], v1[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int64_t f = (int64_t)u * 256 + tid;
            live[u] = f0 + f < n;
            v0[u] = live[u] ? in[2 * f] : (T)0;
            v1[u] = live[u] ? in[2 * f + 1] : (T)0;
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) d[u] = m1_d1(w0, w1, v0[u], v1[u]);
    }
    m1_gather_finish<T, FULL>(d, live, b, f0, n, lo, hi, mhat, alo, ahi, st, cand_m, cap_m, cand_a, cap_a, stage_m, stage_a,
                              &s_below_m, &s_below_a, &s_fill_m, &s_fill_a, &s_base_m, &s_base_a, &s_nan);
}

// The same pass for the finest detail band of mono blocks.  A workgroup takes the tile of M1_TILE coefficients
// that starts at coefficient k0 = tile * M1_TILE of block b; a thread forms runs of ADJ neighbouring coefficients
// from the (F + 2 ADJ) / VW aligned 16-byte pieces that hold their F + 2 ADJ - 2 samples (VW samples a piece),
// straight from memory - neighbouring threads' pieces overlap in the cache (plain loads: with the streaming hint
// the overlapping pieces miss it, +12 %).  Measured alternatives on 2560 blocks of 144 000 samples, eight taps:
// samples staged in LDS and coefficients from 8-byte LDS reads 0.56 ms (the 33 KB stage leaves four workgroups on
// a CU), runs of two 0.41 ms, of four 0.38 ms, of eight 0.42 ms.  Tiles that wrap, touch the repeated last sample of
// an odd block or start at an unaligned address form their coefficients one by one with the periodization index.
static constexpr int D1_ADJ = 4;           // neighbouring coefficients a thread of the direct gather forms from one set of pieces
template <typename T, int F, int ADJ>
__global__ __launch_bounds__(256) void k_d1_gather_direct(const T *__restrict__ x, BlockGeom g, SrcMono<T, F, 0> src,
                                                          M1State *__restrict__ st, T *__restrict__ cand_m, int64_t cap_m,
                                                          T *__restrict__ cand_a, int64_t cap_a, int64_t tiles_per_block, WresPlan skip) {
    using K = typename KeyT<T>::type;
    constexpr int VW = 16 / (int)sizeof(T), NCH = (F + 2 * ADJ) / VW;   // samples per piece, pieces per run
    using VQ = T __attribute__((ext_vector_type(VW)));
    constexpr int BATCH = 16, NU = BATCH / ADJ;                       // coefficients, runs per thread
    constexpr int REGS = NCH * VW * (int)sizeof(T) / 4;               // registers the pieces of a run take
    constexpr int G = REGS <= 12 ? 4 : (REGS <= 24 && NU >= 2 ? 2 : 1);   // runs whose pieces are in flight together
    static_assert(NU % G == 0 && (F + 2 * ADJ) % VW == 0, "runs per thread");
    __shared__ T stage_m[M1_STAGE_M], stage_a[M1_STAGE_A];
    __shared__ unsigned long long s_below_m, s_below_a;
    __shared__ unsigned int s_fill_m, s_fill_a, s_base_m, s_base_a;
    __shared__ int s_nan;
    const int64_t b = blockIdx.x / tiles_per_block;
    const int64_t frames = g.frames[b];
    if (skip.window > 0 && wres_ring(frames, skip.window, skip.run)) return;     // gathered by the wave-resident windows
    int64_t n = (frames + 1) >> 1;                          // coefficients of the block (of this tile's part of it, below)
    int64_t k0 = (blockIdx.x % tiles_per_block) * M1_TILE;
    if (skip.window > 0 && frames >= skip.window) {
        // the windows have gathered the middle of the block: the first tile takes the coefficients before it, the
        // second those after it (fewer than M1_TILE: less than a window of samples)
        const int64_t tile = blockIdx.x % tiles_per_block;
        if (tile >= 2) return;
        if (tile == 0) n = skip.lead >> 1;
        else k0 = wres_tail(frames, skip) >> 1;
    }
    if (k0 >= n) return;
    const K lo = (K)st[b].lo_m, hi = (K)st[b].hi_m;
    if (lo > hi) return;
    const T mhat = (T)st[b].mhat, alo = (T)st[b].alo, ahi = (T)st[b].ahi;
    const int tid = threadIdx.x;
    if (tid == 0) {
        s_below_m = 0;
        s_below_a = 0;
        s_fill_m = 0;
        s_fill_a = 0;
        s_nan = 0;
    }
    __syncthreads();
    const T *in = x + g.in_off[b];
    const int64_t base = 2 * k0 - F / 2;                    // unwrapped index of the sample before the first window
    T d[BATCH];
    bool live[BATCH];
    const bool direct = base >= 0 && base + 2 * M1_TILE + F <= frames &&
                        (reinterpret_cast<uintptr_t>(in + base) & (sizeof(VQ) - 1)) == 0;
    if (direct) {
        const VQ *vq = reinterpret_cast<const VQ *>(in + base);
#pragma unroll
        for (int q0 = 0; q0 < NU; q0 += G) {
            VQ w[G][NCH];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < G; ++q) {
                // run rr = (q0 + q) * 256 + tid: coefficients k0 + ADJ rr + a, samples base + 2 ADJ rr + [1, F + 2 ADJ - 2]
                const VQ *pc = vq + (int64_t)((q0 + q) * 256 + tid) * (2 * ADJ / VW);
#pragma unroll
                for (int t = 0; t < NCH; ++t) w[q][t] = pc[t];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < G; ++q) {
                T v[NCH * VW];
#pragma unroll
                for (int t = 0; t < NCH; ++t)
#pragma unroll
                    for (int e = 0; e < VW; ++e) v[t * VW + e] = w[q][t][e];
                // tap j of the run's coefficient a meets sample base + 2 ADJ rr + F + 2 a - j
#pragma unroll
                for (int a = 0; a < ADJ; ++a) {
                    T sd = 0;
#pragma unroll
                    for (int j = 0; j < F; ++j) sd = fma(src.g[j], v[F + 2 * a - j], sd);
                    d[ADJ * (q0 + q) + a] = sd;
                    live[ADJ * (q0 + q) + a] = true;
                }
            }
        }
    } else {
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int64_t k = k0 + ADJ * ((int64_t)(u / ADJ) * 256 + tid) + u % ADJ;
            live[u] = k < n;
            d[u] = live[u] ? src.at(in, frames, k) : (T)0;
        }
    }
    if (k0 + M1_TILE <= n)
        m1_gather_finish<T, true, BATCH, 256, ADJ>(d, live, b, k0, n, lo, hi, mhat, alo, ahi, st, cand_m, cap_m, cand_a, cap_a,
                                                   stage_m, stage_a, &s_below_m, &s_below_a, &s_fill_m, &s_fill_a, &s_base_m,
                                                   &s_base_a, &s_nan);
    else
        m1_gather_finish<T, false, BATCH, 256, ADJ>(d, live, b, k0, n, lo, hi, mhat, alo, ahi, st, cand_m, cap_m, cand_a, cap_a,
                                                    stage_m, stage_a, &s_below_m, &s_below_a, &s_fill_m, &s_fill_a, &s_base_m,
                                                    &s_base_a, &s_nan);
}

// Per block: are the wanted ranks inside the gathered lists?  Descriptors for the two selections.
__global__ void k_m1_check(const M1State *st, const int64_t *frames, int64_t n_blocks, int64_t cap_m, int64_t cap_a,
                           int64_t *len_m, int64_t *len_a, unsigned long long *below_m, unsigned long long *below_a,
                           int *valid_m, int *valid_a, int *resolved, double *med, double *mad, int all_to_fallback) {
    const int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    M1State s = st[b];
    if (all_to_fallback) s.usable = 0;               // test hook
    const int64_t n = frames[b], r0 = (n - 1) / 2, r1 = n / 2;
    const double nan = __longlong_as_double(0x7FF8000000000000ll);
    const bool in_m = r0 >= (int64_t)s.below_m && r1 - (int64_t)s.below_m < (int64_t)s.n_m;
    const bool in_a = r0 >= (int64_t)s.below_a && r1 - (int64_t)s.below_a < (int64_t)s.n_a;
    const bool sane = s.usable && !s.has_nan && !s.spill;
    // tied median bracket: the counts alone prove the median; otherwise it is selected from the list
    const bool known_m = sane && s.tie_m && in_m;
    const bool ok_m = sane && !s.tie_m && (int64_t)s.n_m <= cap_m && in_m;
    const bool known_a = known_m && s.tie_a && in_a;
    const bool ok_a = (ok_m || known_m) && !s.tie_a && (int64_t)s.n_a <= cap_a && in_a;
    len_m[b] = ok_m ? (int64_t)s.n_m : 0;
    len_a[b] = ok_a ? (int64_t)s.n_a : 0;
    below_m[b] = s.below_m;
    below_a[b] = s.below_a;
    valid_m[b] = ok_m ? 1 : 0;
    valid_a[b] = ok_a ? 1 : 0;
    // a NaN anywhere in the block: both medians are NaN (numpy), nothing left to select
    const bool is_nan = s.usable && s.has_nan;
    resolved[b] = (is_nan || known_a) ? 1 : 0;
    med[b] = known_m ? s.mhat : nan;
    mad[b] = known_a ? s.alo : nan;
}

// Does the second selection of a block stand now that its median is known?
template <typename T>
__device__ __forceinline__ bool m1_verified(const M1State &s, double med, double t0, double t1) {
    const double eps_t = sizeof(T) == 4 ? 1.1920929e-7 : 2.220446049250313e-16;
    const double c = (double)(T)med;
    const double span = fabs(t1) + fabs(c) + fabs(s.mhat) + (s.ahi <= 1.7e308 ? fabs(s.ahi) : 0.0);
    const double e = fabs(c - s.mhat) * (1.0 + 4.0 * eps_t) + 8.0 * eps_t * span + 1e-300;
    const bool ok_lo = s.alo < 0.0 || t0 > s.alo + e;
    const bool ok_hi = !(s.ahi <= 1.7e308) || t1 < s.ahi - e;
    return ok_lo && ok_hi && t0 == t0 && t1 == t1;
}

// Exact fallback: one workgroup per unresolved block streams the block's coefficients (computed on
// the fly): 11-bit MSD radix passes find the key of rank r0; the key of rank r1 = r0 or r0 + 1 is the
// same key when enough keys tie with it, else the smallest larger key (one more sweep, no atomics).
// Visits every key of the block: f(i, key) for i in [0, n), whole waves take every step and the loads
// of UB steps are in flight together.
template <typename T, typename Src, typename F>
__device__ __forceinline__ void m1_sweep(const Src &src, const T *in, int64_t frames, int64_t n, bool absdev, T c, F f) {
    if constexpr (Src::STEREO) {
        using V2 = T __attribute__((ext_vector_type(2)));
        constexpr int UB = 16;
        const int64_t span = (int64_t)UB * M1_THREADS;
        const int64_t n_pad = (n + span - 1) / span * span;
        const bool vec = (reinterpret_cast<uintptr_t>(in) & (sizeof(V2) - 1)) == 0;      // one load per frame
        for (int64_t i0 = threadIdx.x; i0 < n_pad; i0 += span) {
            V2 q[UB];
            if (vec && i0 - threadIdx.x + span <= n) {
                const V2 *v2 = reinterpret_cast<const V2 *>(in);
#pragma unroll
                for (int u = 0; u < UB; ++u) q[u] = v2[i0 + (int64_t)u * M1_THREADS];
            } else {
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int64_t i = i0 + (int64_t)u * M1_THREADS;
                    q[u].x = i < n ? in[2 * i] : (T)0;
                    q[u].y = i < n ? in[2 * i + 1] : (T)0;
                }
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                T v = m1_d1(src.w0, src.w1, q[u].x, q[u].y);
                if (absdev) v = fabs(v - c);
                f(i0 + (int64_t)u * M1_THREADS < n, m1_key(v));
            }
        }
    } else {
        // any other source: the coefficients one by one (neighbouring lanes form neighbouring coefficients,
        // their samples overlap in the cache)
        constexpr int UB = 4;
        const int64_t span = (int64_t)UB * M1_THREADS;
        const int64_t n_pad = (n + span - 1) / span * span;
        for (int64_t i0 = threadIdx.x; i0 < n_pad; i0 += span) {
            T v[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int64_t i = i0 + (int64_t)u * M1_THREADS;
                v[u] = i < n ? src.at(in, frames, i) : (T)0;
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                T w = v[u];
                if (absdev) w = fabs(w - c);
                f(i0 + (int64_t)u * M1_THREADS < n, m1_key(w));
            }
        }
    }
}

// Visits the `cnt` keys of a list in memory: f(live, key), whole waves take every step.
template <typename K, typename F>
__device__ __forceinline__ void m1_list_sweep(const K *list, uint32_t cnt, F f) {
    const uint32_t pad = (cnt + M1_THREADS - 1) / M1_THREADS * M1_THREADS;
    for (uint32_t i = threadIdx.x; i < pad; i += M1_THREADS) f(i < cnt, i < cnt ? list[i] : (K)0);
}

// Keys of ranks r0 <= r1 <= r0 + 1 of the block (or of its absolute deviations from c).
// The first sweep histograms the leading digit; when the keys of rank r0's bin fit `list`
// (list_cap keys of scratch memory, normally a few percent of the block) a second sweep gathers
// them and the remaining digits - and usually the neighbour key of rank r1 - are found on the list;
// otherwise every digit is a sweep over the block.
template <typename T, typename Src>
__device__ void m1_slow_pair(const Src &src, const T *in, int64_t frames, int64_t n, bool absdev, T c, uint64_t r0, uint64_t r1,
                             uint32_t *hist, uint32_t *wtot, uint32_t *s_dig, uint32_t *s_bef, unsigned long long *s_min,
                             uint32_t *s_fill, int *nan_seen, typename KeyT<T>::type *list, int64_t list_cap,
                             typename KeyT<T>::type *k0, typename KeyT<T>::type *k1) {
    using K = typename KeyT<T>::type;
    constexpr int BITS = KeyT<T>::bits, PASSES = (BITS + 10) / 11;
    const int tid = threadIdx.x;
    K prefix = 0;
    uint32_t rank = (uint32_t)r0, in_bin = 0, rank_top = 0, in_top = 0;
    bool on_list = false;
    uint32_t cnt = 0;
    // digits of 11 bits from the top (the last one takes what is left: 10 bits of a 32-bit key), so that
    // the leading bin is a quarter of an octave of values
    int prev_shift = BITS;
    for (int p = 0; p < PASSES; ++p) {
        const int shift = BITS - 11 * (p + 1) > 0 ? BITS - 11 * (p + 1) : 0;
        const int width = prev_shift - shift;
        const int up = prev_shift < BITS ? prev_shift : 0;      // keys match when (k >> up) == prefix (p > 0)
        for (int i = tid; i < M1_BINS; i += M1_THREADS) hist[i] = 0;
        __syncthreads();
        if (on_list) {
            m1_list_sweep<K>(list, cnt, [&](bool live, K k) {
                const bool match = live && (K)(k >> up) == prefix;
                m1_hist_add(hist, (uint32_t)(k >> shift) & ((1u << width) - 1u), match);
            });
        } else {
            m1_sweep<T>(src, in, frames, n, absdev, c, [&](bool live, K k) {
                if (live && k == ~(K)0) *nan_seen = 1;
                const bool match = live && (p == 0 || (K)(k >> up) == prefix);
                m1_hist_add(hist, (uint32_t)(k >> shift) & ((1u << width) - 1u), match);
            });
        }
        __syncthreads();
        m1_find_bin(hist, rank, wtot, s_dig, s_bef);
        prefix = (prefix << width) | (K)(*s_dig);
        prev_shift = shift;
        rank -= *s_bef;
        in_bin = hist[*s_dig];
        __syncthreads();
        if (p == 0) {
            rank_top = rank;
            in_top = in_bin;
            if (PASSES > 1 && (int64_t)in_bin <= list_cap) {
                // gather the keys of the leading bin: the other digits are found among them
                if (tid == 0) *s_fill = 0;
                __syncthreads();
                m1_sweep<T>(src, in, frames, n, absdev, c, [&](bool live, K k) {
                    const bool match = live && (K)(k >> shift) == prefix;
                    const unsigned long long bal = __ballot(match);
                    if (bal) {
                        uint32_t base = 0;
                        if ((tid & 63) == (int)__ffsll((long long)bal) - 1) base = atomicAdd(s_fill, (uint32_t)__popcll(bal));
                        base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)__ffsll((long long)bal) - 1);
                        if (match)
                            list[base + __builtin_amdgcn_mbcnt_hi((unsigned int)(bal >> 32),
                                                                  __builtin_amdgcn_mbcnt_lo((unsigned int)bal, 0u))] = k;
                    }
                });
                __syncthreads();
                cnt = *s_fill;
                on_list = true;
                __syncthreads();
            }
        }
    }
    *k0 = prefix;
    if (r1 == r0 || rank + 1 < in_bin) {        // enough keys tie with rank r0's
        *k1 = prefix;
        return;
    }
    // rank r1 holds the smallest key above k0: on the list when the leading bin has a key beyond rank r0
    if (tid == 0) *s_min = ~0ull;
    __syncthreads();
    unsigned long long mn = ~0ull;
    if (on_list && rank_top + 1 < in_top) {
        m1_list_sweep<K>(list, cnt, [&](bool live, K k) {
            if (live && k > prefix && (unsigned long long)k < mn) mn = (unsigned long long)k;
        });
    } else {
        m1_sweep<T>(src, in, frames, n, absdev, c, [&](bool live, K k) {
            if (live && k > prefix && (unsigned long long)k < mn) mn = (unsigned long long)k;
        });
    }
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = __shfl_xor(mn, d);
        mn = o < mn ? o : mn;
    }
    if ((tid & 63) == 0) atomicMin(s_min, mn);
    __syncthreads();
    *k1 = (K)(*s_min);
    __syncthreads();
}

// Last kernel of the statistics, one workgroup per block: the block's second selection is verified;
// a block that is not resolved by then is selected exactly here; the universal threshold
// sigma * sqrt(2 ln frames) is written (frames == 1 gives 0).
template <typename T, typename Src>
__global__ __launch_bounds__(M1_THREADS) void k_m1_fallback(const T *__restrict__ x, BlockGeom g, Src src,
                                                            const M1State *__restrict__ st, const int *__restrict__ resolved,
                                                            const int *__restrict__ valid_a, const double *__restrict__ pair_a,
                                                            double *med, double *mad, double *thresh, double *sigma_out,
                                                            double *thresh_out, T *__restrict__ scratch, int64_t scratch_cap) {
    using K = typename KeyT<T>::type;
    __shared__ uint32_t hist[M1_BINS];
    __shared__ uint32_t wtot[M1_THREADS / 64];
    __shared__ uint32_t s_dig, s_bef, s_fill;
    __shared__ unsigned long long s_min;
    __shared__ int s_nan;
    const int64_t b = blockIdx.x;
    const int64_t frames = g.frames[b];
    const int64_t n = Src::count(frames);                  // coefficients of the block
    const double nan = __longlong_as_double(0x7FF8000000000000ll);
    auto finish = [&](double sigma) {
        const double t = sigma * sqrt(2.0 * log((double)frames));
        thresh[b] = t;
        if (sigma_out) sigma_out[b] = sigma;
        if (thresh_out) thresh_out[b] = t;
    };
    const bool done = resolved[b] || (valid_a[b] && m1_verified<T>(st[b], med[b], pair_a[2 * b], pair_a[2 * b + 1]));
    if (done) {
        if (threadIdx.x == 0) finish(mad[b]);
        return;
    }
    const T *in = x + g.in_off[b];
    if (threadIdx.x == 0) s_nan = 0;
    __syncthreads();
    if (n <= 0) {
        if (threadIdx.x == 0) {
            med[b] = nan;
            mad[b] = nan;
            finish(nan);
        }
        return;
    }
    const uint64_t r0 = (uint64_t)((n - 1) / 2), r1 = (uint64_t)(n / 2);
    K k0, k1;
    // the block's own candidate list (useless now) is the scratch of the exact selection
    K *list = reinterpret_cast<K *>(scratch + blockIdx.x * scratch_cap);
    m1_slow_pair<T>(src, in, frames, n, false, (T)0, r0, r1, hist, wtot, &s_dig, &s_bef, &s_min, &s_fill, &s_nan, list, scratch_cap,
                    &k0, &k1);
    __syncthreads();
    if (s_nan) {
        if (threadIdx.x == 0) {
            med[b] = nan;
            mad[b] = nan;
            finish(nan);
        }
        return;
    }
    const T a = val_of(k0), bb = val_of(k1);
    const T m = r0 == r1 ? a : (T)((a + bb) / (T)2);
    K q0, q1;
    m1_slow_pair<T>(src, in, frames, n, true, m, r0, r1, hist, wtot, &s_dig, &s_bef, &s_min, &s_fill, &s_nan, list, scratch_cap,
                    &q0, &q1);
    if (threadIdx.x == 0) {
        const T ta = val_of(q0), tb = val_of(q1);
        const double sg = s_nan ? nan : (double)(r0 == r1 ? ta : (T)((ta + tb) / (T)2));     // inf - inf
        med[b] = (double)m;
        mad[b] = sg;
        finish(sg);
    }
}

// ---------------------------------------------------------------- host side
template <typename T, typename Src>
static int64_t m1_cap(int64_t max_count, double frac) {
    // a block with no more coefficients than the sample is gathered whole; longer ones hold `frac` of their keys in the bracket
    const int64_t want = (int64_t)((double)max_count * frac) + 1024;
    const int64_t whole = max_count < Src::S ? max_count : Src::S;
    return want > whole ? want : whole;
}
// Which batches take the single-pass route (max_count: coefficients of the longest block).  Its sampling runs one
// workgroup per block; lists that fit LDS (stereo blocks up to ~299 K frames in binary32, ~145 K in binary64) are
// selected there, longer lists are bracketed again first (select_subset_ws_bytes).
template <typename T, typename Src>
static bool m1_src_applicable(const Ctx *c, int64_t n_blocks, int64_t max_count) {
    const size_t kb = sizeof(typename KeyT<T>::type);
    const size_t cap = (size_t)m1_cap<T, Src>(max_count, Src::FRAC_A);
    return !c->force_generic && max_count < (int64_t(1) << 30) && cap <= select_subset_max_keys(kb);
}
template <typename T, typename Src>
static size_t m1_src_ws_bytes(int64_t n_blocks, int64_t max_count) {
    const size_t nb = (size_t)n_blocks;
    return ws_need(nb, sizeof(M1State)) + ws_need(nb * (size_t)m1_cap<T, Src>(max_count, Src::FRAC_M), sizeof(T)) +
           ws_need(nb * (size_t)m1_cap<T, Src>(max_count, Src::FRAC_A), sizeof(T)) + 5 * ws_need(nb, 8) + 3 * ws_need(nb, 4) +
           3 * ws_need(nb, 8) + ws_need(2 * nb, 8) +
           select_subset_ws_bytes(n_blocks, m1_cap<T, Src>(max_count, Src::FRAC_M), sizeof(typename KeyT<T>::type)) +
           select_subset_ws_bytes(n_blocks, m1_cap<T, Src>(max_count, Src::FRAC_A), sizeof(typename KeyT<T>::type));
}
template <typename T>
static bool m1_applicable(const Ctx *c, int64_t n_blocks, int64_t max_frames) {
    return m1_src_applicable<T, SrcStereo<T>>(c, n_blocks, max_frames);
}
template <typename T>
static size_t m1_ws_bytes(int64_t n_blocks, int64_t max_frames) {
    return m1_src_ws_bytes<T, SrcStereo<T>>(n_blocks, max_frames);
}

// Median absolute deviation (sigma) and universal threshold of every block from coefficients formed
// on the fly by `src`; gather(st, cand_m, cap_m, cand_a, cap_a) enqueues the source's gather pass.
// *thresh_ws receives the per-block thresholds (device, binary64).
template <typename T, typename Src, typename Gather>
static int mad_single_pass(Ctx *c, const T *d_in, const BlockGeom &g, int64_t max_count, const Src &src,
                           const char *sample_name, Gather gather, double *d_sigma_out, double *d_thresh_out,
                           double **thresh_ws) {
    const size_t nb = (size_t)g.n_blocks;
    const int64_t cap_m = m1_cap<T, Src>(max_count, Src::FRAC_M), cap_a = m1_cap<T, Src>(max_count, Src::FRAC_A);
    M1State *st = ws_take<M1State>(c, nb);
    T *cand_m = ws_take<T>(c, nb * (size_t)cap_m);
    T *cand_a = ws_take<T>(c, nb * (size_t)cap_a);
    int64_t *len_m = ws_take<int64_t>(c, nb), *len_a = ws_take<int64_t>(c, nb), *counts = ws_take<int64_t>(c, nb);
    unsigned long long *below_m = ws_take<unsigned long long>(c, nb), *below_a = ws_take<unsigned long long>(c, nb);
    int *valid_m = ws_take<int>(c, nb), *valid_a = ws_take<int>(c, nb), *resolved = ws_take<int>(c, nb);
    double *med = ws_take<double>(c, nb), *mad = ws_take<double>(c, nb), *thresh = ws_take<double>(c, nb);
    double *pair_a = ws_take<double>(c, 2 * nb);
    if (c->ws_used > c->ws_bytes) {
        set_error("internal: workspace not reserved for the single-pass threshold statistics");
        return ADN_ERR_NOMEM;
    }
    const unsigned nbk = (unsigned)ceil_div(g.n_blocks, 256);
    {
        auto kern = k_m1_sample<T, Src>;
        const size_t dyn = (size_t)Src::S * sizeof(typename KeyT<T>::type);
        ADN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
        ADN_LAUNCH_NAMED(sample_name, kern, dim3((unsigned)g.n_blocks), dim3(Src::SAMPLE_THREADS), dyn, c->stream, d_in, g, src, st, counts);
    }
    ADN_TRY(gather(st, cand_m, cap_m, cand_a, cap_a));
    ADN_LAUNCH(k_m1_check, dim3(nbk), dim3(256), 0, c->stream, st, counts, g.n_blocks, cap_m, cap_a, len_m, len_a,
                       below_m, below_a, valid_m, valid_a, resolved, med, mad, c->force_fallback ? 1 : 0);
    ADN_HIP(hipGetLastError());
    SubsetSelect sm;
    sm.n_seg = g.n_blocks;
    sm.cap = cap_m;
    sm.d_len = len_m;
    sm.d_orig_len = counts;
    sm.d_below = below_m;
    sm.d_valid = valid_m;
    sm.d_out = med;
    ADN_TRY(launch_select_subset<T>(c, cand_m, sm));
    SubsetSelect sa;
    sa.n_seg = g.n_blocks;
    sa.cap = cap_a;
    sa.d_len = len_a;
    sa.d_orig_len = counts;
    sa.d_below = below_a;
    sa.d_valid = valid_a;
    sa.xform = XF_ABSDEV;
    sa.d_center = med;
    sa.d_out = mad;
    sa.d_pair = pair_a;
    ADN_TRY(launch_select_subset<T>(c, cand_a, sa));
    ADN_LAUNCH((k_m1_fallback<T, Src>), dim3((unsigned)g.n_blocks), dim3(M1_THREADS), 0, c->stream, d_in, g, src, st,
                       resolved, valid_a, pair_a, med, mad, thresh, d_sigma_out, d_thresh_out, cand_a, cap_a);
    ADN_HIP(hipGetLastError());
    if (c->ablate & 2048) {      // diagnostics: which blocks went to the exact fallback, and why (synchronises)
        ADN_HIP(hipStreamSynchronize(c->stream));
        std::vector<M1State> hs(nb);
        std::vector<int> hvm(nb), hva(nb), hres(nb);
        std::vector<double> hmed(nb), hpair(2 * nb);
        std::vector<int64_t> hcnt(nb);
        ADN_HIP(hipMemcpy(hs.data(), st, nb * sizeof(M1State), hipMemcpyDeviceToHost));
        ADN_HIP(hipMemcpy(hvm.data(), valid_m, nb * sizeof(int), hipMemcpyDeviceToHost));
        ADN_HIP(hipMemcpy(hva.data(), valid_a, nb * sizeof(int), hipMemcpyDeviceToHost));
        ADN_HIP(hipMemcpy(hres.data(), resolved, nb * sizeof(int), hipMemcpyDeviceToHost));
        ADN_HIP(hipMemcpy(hmed.data(), med, nb * sizeof(double), hipMemcpyDeviceToHost));
        ADN_HIP(hipMemcpy(hpair.data(), pair_a, 2 * nb * sizeof(double), hipMemcpyDeviceToHost));
        ADN_HIP(hipMemcpy(hcnt.data(), counts, nb * sizeof(int64_t), hipMemcpyDeviceToHost));
        size_t shown = 0;
        for (size_t b = 0; b < nb && shown < 24; ++b) {
            const M1State &m = hs[b];
            const double eps_t = sizeof(T) == 4 ? 1.1920929e-7 : 2.220446049250313e-16;
            const double cc = (double)(T)hmed[b];
            const double span = fabs(hpair[2 * b + 1]) + fabs(cc) + fabs(m.mhat) + (m.ahi <= 1.7e308 ? fabs(m.ahi) : 0.0);
            const double e = fabs(cc - m.mhat) * (1.0 + 4.0 * eps_t) + 8.0 * eps_t * span + 1e-300;
            const bool ver = (m.alo < 0.0 || hpair[2 * b] > m.alo + e) && (!(m.ahi <= 1.7e308) || hpair[2 * b + 1] < m.ahi - e) &&
                             hpair[2 * b] == hpair[2 * b] && hpair[2 * b + 1] == hpair[2 * b + 1];
            if (hres[b] || (hva[b] && ver)) continue;
            ++shown;
            fprintf(stderr, "m1 block %zu: n %lld usable %d nan %d spill %d tie %d/%d n_m %u (cap %lld) below_m %llu n_a %u (cap %lld) below_a %llu "
                            "valid %d/%d med %.9g mhat %.9g alo %.9g ahi %.9g pair %.9g %.9g\n", b, (long long)hcnt[b], m.usable,
                    m.has_nan, m.spill, m.tie_m, m.tie_a, m.n_m, (long long)cap_m, m.below_m, m.n_a, (long long)cap_a, m.below_a,
                    hvm[b], hva[b], hmed[b], m.mhat, m.alo, m.ahi, hpair[2 * b], hpair[2 * b + 1]);
        }
    }
    *thresh_ws = thresh;
    return ADN_OK;
}

template <typename T>
static int stereo_mad_single_pass(Ctx *c, const T *d_in, const BlockGeom &g, int64_t max_frames, T w0, T w1,
                                  double *d_sigma_out, double *d_thresh_out, double **thresh_ws) {
    const int64_t tiles = max_frames / M1_TILE;            // whole tiles of the longest block
    ADN_REQUIRE(g.n_blocks * (tiles + 1) < (int64_t)0x7FFFFFFF, "wavelet batch too large for one launch");
    SrcStereo<T> src;
    src.w0 = w0;
    src.w1 = w1;
    auto gather = [&](M1State *st, T *cand_m, int64_t cap_m, T *cand_a, int64_t cap_a) -> int {
        if (tiles > 0)
            ADN_LAUNCH((k_m1_gather<T, true>), dim3((unsigned)(g.n_blocks * tiles)), dim3(256), 0, c->stream, d_in, g,
                               w0, w1, st, cand_m, cap_m, cand_a, cap_a, tiles);
        ADN_LAUNCH((k_m1_gather<T, false>), dim3((unsigned)g.n_blocks), dim3(256), 0, c->stream, d_in, g, w0, w1, st,
                           cand_m, cap_m, cand_a, cap_a, (int64_t)1);
        return ADN_OK;
    };
    return mad_single_pass<T>(c, d_in, g, max_frames, src, "k_m1_sample", gather, d_sigma_out, d_thresh_out, thresh_ws);
}

// The same statistics for the finest detail band of mono blocks (periodization), taps g[0..F) of the
// decomposition high-pass filter: no coefficient array, one pass over the samples.
template <typename T, int F, int S>
static bool mono_single_pass_applicable(const Ctx *c, int64_t n_blocks, int64_t max_frames) {
    return m1_src_applicable<T, SrcMono<T, F, S>>(c, n_blocks, (max_frames + 1) / 2);
}
template <typename T, int F, int S>
static size_t mono_single_pass_ws_bytes(int64_t n_blocks, int64_t max_frames) {
    return m1_src_ws_bytes<T, SrcMono<T, F, S>>(n_blocks, (max_frames + 1) / 2);
}
// fused_TL > 0: the gather pass is the wave-resident windows of fused_TL levels, which also write the level
// approximation of every window to `top`; after_windows(plan) then launches whatever covers the rest of `top`, and the
// direct gather takes the blocks the windows do not cover whole.
template <typename T, int F, int S, typename After>
static int mono_mad_single_pass(Ctx *c, const T *d_in, const BlockGeom &g, int64_t max_frames, const T *taps_g,
                                double *d_sigma_out, double *d_thresh_out, double **thresh_ws, int wavelet, int fused_TL, T *top,
                                int64_t top_stride, After after_windows) {
    const int64_t max_count = (max_frames + 1) / 2;
    const int64_t tiles = ceil_div(max_count, (int64_t)M1_TILE);
    ADN_REQUIRE(g.n_blocks * tiles < (int64_t)0x7FFFFFFF, "wavelet batch too large for one launch");
    SrcMono<T, F, S> src;
    SrcMono<T, F, 0> gsrc;
    for (int j = 0; j < F; ++j) src.g[j] = gsrc.g[j] = taps_g[j];
    auto gather = [&](M1State *st, T *cand_m, int64_t cap_m, T *cand_a, int64_t cap_a) -> int {
        // coefficients that the wave-resident windows own are gathered by them, the rest here
        WresPlan skip{0, 0, 0, 1};
        {
            if (fused_TL > 0) {
                ADN_TRY(wres_gather(c, F, fused_TL, d_in, g, max_frames, wavelet, WresGather{st, cand_m, cap_m, cand_a, cap_a}, top,
                                    top_stride, &skip));
                ADN_TRY(after_windows(skip));
            }
        }
        // (blocks with windows want two tiles, for their ends)
        const int64_t gt = skip.window > 0 && tiles < 2 ? 2 : tiles;
        if (gt > 0)
            ADN_LAUNCH((k_d1_gather_direct<T, F, D1_ADJ>), dim3((unsigned)(g.n_blocks * gt)), dim3(256), 0, c->stream, d_in, g, gsrc, st,
                               cand_m, cap_m, cand_a, cap_a, gt, skip);
        return ADN_OK;
    };
    return mad_single_pass<T>(c, d_in, g, max_count, src, "k_d1_sample", gather, d_sigma_out, d_thresh_out, thresh_ws);
}
