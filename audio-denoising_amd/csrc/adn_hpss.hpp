// Harmonic-percussive separation stage (src/hpss_separate.py:41-67, 115-140, 142-190, 233-283).
// Included by adn_stft.hip (inside namespace adn, after rows_dispatch / apply_dispatch).
//
//   |Z| rows  ->  median over 31 frames (harmonic) and over 31 bins (percussive), reflected edges
//             ->  soft / hard / binary masks, frame-major           (k_hpss_masks)
//             ->  ISTFT(Z * mask_h), ISTFT(Z * mask_p)               (the fused apply kernels, per-frame gain rows)
//             ->  transient preservation, joint normalisation, RMS   (k_hpss_energy .. k_hpss_scale)
#pragma once

namespace hpss {

constexpr int TB = 32, TT = 32;        // bins x frames of one mask tile
constexpr int HMAX = 15;               // largest half kernel (kernel_size <= 31)
constexpr int LW = TT + 2 * HMAX + 1;  // LDS row pitch (odd: the column walks of the bin medians spread over banks)

template <typename T>
__device__ __forceinline__ T inf_of() { return (T)INFINITY; }

// scipy.ndimage 'reflect' (d c b a | a b c d | d c b a) of index i into [0, n)
__device__ __forceinline__ int reflect_idx(int i, int n) {
    if (i >= 0 && i < n) return i;
    const int p = 2 * n;
    int m = i % p;
    if (m < 0) m += p;
    return m < n ? m : p - 1 - m;
}

template <typename T>
__device__ __forceinline__ void sort30(T (&v)[30]) {
#pragma unroll
    for (int c = 0; c < SORTNET30_N; ++c) {
        const T a = v[SORTNET30[c].i], b = v[SORTNET30[c].j];
        v[SORTNET30[c].i] = fmin(a, b);          // magnitudes: no NaN ordering to preserve
        v[SORTNET30[c].j] = fmax(a, b);
    }
}
template <typename T>
__device__ __forceinline__ T med3(T a, T b, T c) {
    return fmax(fmin(a, b), fmin(fmax(a, b), c));
}

struct BandTable {           // band of every bin: reflect range [lo, hi], lo < 0: bin belongs to no band
    const int *lo, *hi;
};

// Medians of two neighbouring windows that share ks-1 values: sort the shared values once (padded
// with +inf to 30), then the median of each window is the median of three:
// (shared[(ks-1)/2 - 1], the window's own extra value, shared[(ks-1)/2]).
template <typename T, typename F>
__device__ __forceinline__ void median_pair(F at, int ks, T &m0, T &m1) {
    // at(d): value at offset d from the first centre, d in [-h, h + 1]
    const int h = ks / 2;
    T v[30];
#pragma unroll
    for (int q = 0; q < 30; ++q) v[q] = q < ks - 1 ? at(q - h + 1) : inf_of<T>();
    sort30(v);
    T lo = v[0], hi = v[0];
#pragma unroll
    for (int q = 0; q < 30; ++q) {           // v[h-1], v[h] without dynamic register indexing
        lo = q == h - 1 ? v[q] : lo;
        hi = q == h ? v[q] : hi;
    }
    if (ks == 1) {
        m0 = at(0);
        m1 = at(1);
    } else {
        m0 = med3(lo, at(-h), hi);
        m1 = med3(lo, at(h + 1), hi);
    }
}

// One tile of TB bins x TT frames of one signal: both medians from an LDS copy of the tile and its
// halo, masks written frame-major: M[(sig * Tmax + t) * F + k].
// KS: kernel size known at compile time (the reference's default 31: predicates and the picks of the
// two middle values fold away), 0: taken from ks_rt.
template <typename T, int KS>
__global__ __launch_bounds__(256) void k_hpss_masks(const T *__restrict__ S, int F, int64_t Ts, int64_t Tmax,
                                                    const int64_t *__restrict__ Tn, BandTable bt, int ks_rt,
                                                    T margin_h, T margin_p, T power, int mask_type,
                                                    T *__restrict__ Mh, T *__restrict__ Mp) {
    __shared__ T tile[(TB + 2 * HMAX) * LW];
    __shared__ T sh[TB][TT + 1], sp[TB][TT + 1];
    const int64_t sig = blockIdx.z;
    const int T_ = (int)Tn[sig];
    const int t0 = blockIdx.x * TT, k0 = blockIdx.y * TB;
    if (t0 >= T_) return;
    const int ks = KS ? KS : ks_rt;
    const int h = ks / 2, tid = threadIdx.x;
    const T *Ssig = S + sig * F * Ts;
    // rows k0-h .. k0+TB-1+h (clamped to the spectrum; reflection is applied when the windows are
    // read, per band), columns = frames t0-h .. t0+TT+h reflected into [0, T_)
    const int rows = TB + 2 * h, cols = TT + 2 * h + 1;
    {
        const int c = tid & 63;                          // cols <= 63: one column per lane, four rows per step
        const int t = reflect_idx(t0 - h + c, T_);
        for (int r = tid >> 6; r < rows; r += 4) {
            int k = k0 - h + r;
            k = k < 0 ? 0 : (k > F - 1 ? F - 1 : k);
            if (c < cols) tile[r * LW + c] = Ssig[(int64_t)k * Ts + t];
        }
    }
    __syncthreads();
    // ---- harmonic: median along time, two neighbouring frames per step
    for (int p = tid; p < TB * (TT / 2); p += 256) {
        const int kk = p / (TT / 2), tt = (p % (TT / 2)) * 2;
        const T *rowp = tile + (kk + h) * LW + (tt + h);
        T m0, m1;
        median_pair<T>([&](int d) { return rowp[d]; }, ks, m0, m1);
        sh[kk][tt] = m0;
        sh[kk][tt + 1] = m1;
    }
    // ---- percussive: median along frequency, reflected at the edges of the bin's band
    for (int p = tid; p < (TB / 2) * TT; p += 256) {
        const int kk = (p / TT) * 2, tt = p % TT;
        const int k = k0 + kk;
        const T *colp = tile + (tt + h);
        T m0 = 0, m1 = 0;
        const int ka = k < F ? k : F - 1, kb = k + 1 < F ? k + 1 : F - 1;
        const int lo_a = bt.lo[ka], hi_a = bt.hi[ka], lo_b = bt.lo[kb], hi_b = bt.hi[kb];
        const bool same = lo_a == lo_b && hi_a == hi_b && lo_a >= 0 && k + 1 < F && k - h >= lo_a && k + 1 + h <= hi_a;
        if (same) {
            median_pair<T>([&](int d) { return colp[(kk + h + d) * LW]; }, ks, m0, m1);
        } else {
            // band edge (or no band): each bin on its own, window indices reflected into the band
            for (int e = 0; e < 2; ++e) {
                const int kc = k + e;
                if (kc >= F) break;
                const int lo = bt.lo[kc], hi = bt.hi[kc];
                T me = 0;
                if (lo >= 0) {
                    const int nb = hi - lo + 1;
                    T v[30];
#pragma unroll
                    for (int q = 0; q < 30; ++q) {
                        T val = inf_of<T>();
                        if (q < ks - 1) {
                            const int src = lo + reflect_idx(kc - h + 1 + q - lo, nb);
                            val = colp[(src - (k0 - h)) * LW];
                        }
                        v[q] = val;
                    }
                    sort30(v);
                    T a = v[0], b = v[0];
#pragma unroll
                    for (int q = 0; q < 30; ++q) {
                        a = q == h - 1 ? v[q] : a;
                        b = q == h ? v[q] : b;
                    }
                    const T extra = colp[(lo + reflect_idx(kc - h - lo, nb) - (k0 - h)) * LW];
                    me = ks == 1 ? colp[(kc - (k0 - h)) * LW] : med3(a, extra, b);
                }
                if (e == 0) m0 = me;
                else m1 = me;
            }
        }
        sp[kk][tt] = m0;
        sp[kk + 1][tt] = m1;
    }
    __syncthreads();
    // ---- masks, written frame-major (runs of TB bins per frame)
    for (int i = tid; i < TB * TT; i += 256) {
        const int tt = i / TB, kk = i % TB;
        const int k = k0 + kk, t = t0 + tt;
        if (k >= F || t >= T_) continue;
        T mh = 0, mp = 0;
        if (bt.lo[k] >= 0) {
            const T a = sh[kk][tt] * margin_h, b = sp[kk][tt] * margin_p;
            const T ph = power == (T)2 ? a * a : (T)pow((double)a, (double)power);
            const T pp = power == (T)2 ? b * b : (T)pow((double)b, (double)power);
            const T total = ph + pp + (T)1e-10;
            mh = ph / total;
            mp = pp / total;
        }
        if (mask_type == ADN_MASK_BINARY) {
            mh = mh > (T)0.5 ? (T)1 : (T)0;
            mp = mp > (T)0.5 ? (T)1 : (T)0;
        } else if (mask_type == ADN_MASK_HARD) {
            mh = mh > mp ? (T)1 : (T)0;
            mp = (T)1 - mh;
        }
        const int64_t o = (sig * Tmax + t) * F + k;
        Mh[o] = mh;
        Mp[o] = mp;
    }
}

// ---- transient preservation (src/hpss_separate.py:142-190)
// frames i * 512 .. + 2048 for i in range(0, m - 2048, 512); one wave per frame
template <typename T>
__global__ __launch_bounds__(256) void k_hpss_energy(const T *__restrict__ x, Signals sg, const T *__restrict__ yh,
                                                     const int64_t *__restrict__ y_off, int64_t max_frames,
                                                     double *__restrict__ eo, double *__restrict__ eh) {
    const int64_t sig = blockIdx.y;
    const int64_t i = blockIdx.x * 4ll + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int64_t n = sg.len[sig];
    const int64_t m = (stft_frames(n, sg.n_fft, sg.hop, 1) - 1) * sg.hop;      // ISTFT length <= n
    const int64_t cnt = m > 2048 ? (m - 2048 + 511) / 512 : 0;
    if (i >= cnt) return;
    const T *xs = x + sg.off[sig];
    const T *ys = yh + y_off[sig];
    double so = 0.0, shh = 0.0;
    for (int j = lane; j < 2048; j += 64) {
        const int64_t p = i * 512 + j;
        const double a = (double)fold_sample(xs, p, n, sg.n_ch), b = (double)ys[p];
        so += a * a;
        shh += b * b;
    }
    for (int d = 32; d >= 1; d >>= 1) {
        so += __shfl_xor(so, d);
        shh += __shfl_xor(shh, d);
    }
    if (lane == 0) {
        const double vo = sqrt(so / 2048.0), vh = sqrt(shh / 2048.0);
        eo[sig * max_frames + i] = vo;
        eh[sig * max_frames + i] = vh;
    }
}

// max over the frames of the original's energies; one workgroup per signal
__global__ __launch_bounds__(256) void k_hpss_emax(Signals sg, int64_t max_frames, const double *__restrict__ eo,
                                                   const double *__restrict__ eh, double threshold, int preserve,
                                                   unsigned long long *__restrict__ eo_max,
                                                   int64_t *__restrict__ n_tr, unsigned char *__restrict__ cov) {
    __shared__ double part[4];
    __shared__ int cpart[4];
    const int64_t sig = blockIdx.x;
    const int64_t n = sg.len[sig];
    const int64_t m = (stft_frames(n, sg.n_fft, sg.hop, 1) - 1) * sg.hop;
    const int64_t cnt = m > 2048 ? (m - 2048 + 511) / 512 : 0;
    double mx = 0.0;
    for (int64_t i = threadIdx.x; i < cnt; i += 256) mx = fmax(mx, eo[sig * max_frames + i]);
    for (int d = 32; d >= 1; d >>= 1) mx = fmax(mx, __shfl_xor(mx, d));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mx;
    __syncthreads();
    const double emax = fmax(fmax(part[0], part[1]), fmax(part[2], part[3]));
    if (threadIdx.x == 0) eo_max[sig] = (unsigned long long)key_of(emax);
    // number of transient frames (for the log)
    const double thr = threshold * emax;
    int c = 0;
    for (int64_t i = threadIdx.x; i < cnt; i += 256) c += (eo[sig * max_frames + i] - eh[sig * max_frames + i]) > thr;
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
    if ((threadIdx.x & 63) == 0) cpart[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) n_tr[sig] = preserve ? cpart[0] + cpart[1] + cpart[2] + cpart[3] : 0;
    // cov[hop]: some transient frame covers the 512 samples of this hop (frame i covers hops i .. i+3)
    for (int64_t hp = threadIdx.x; hp < max_frames + 4; hp += 256) {
        bool any = false;
        for (int64_t i = hp - 3 > 0 ? hp - 3 : 0; i <= hp && i < cnt; ++i)
            any |= (eo[sig * max_frames + i] - eh[sig * max_frames + i]) > thr;
        cov[sig * (max_frames + 4) + hp] = (preserve && any) ? 1 : 0;
    }
}

// blend the original into the harmonic output where a covering frame is transient; |y| maxima
template <typename T>
__global__ __launch_bounds__(256) void k_hpss_blend(const T *__restrict__ x, Signals sg, T *__restrict__ yh,
                                                    const T *__restrict__ yp, const int64_t *__restrict__ y_off,
                                                    int64_t max_frames, const unsigned char *__restrict__ cov,
                                                    double blend, unsigned long long *__restrict__ ymax) {
    const int64_t sig = blockIdx.y;
    const int64_t n = sg.len[sig];
    const int64_t m = (stft_frames(n, sg.n_fft, sg.hop, 1) - 1) * sg.hop;
    const T *xs = x + sg.off[sig];
    T *hs = yh + y_off[sig];
    const T *ps = yp + y_off[sig];
    double mx = 0.0;
    for (int64_t j = blockIdx.x * 256ll + threadIdx.x; j < m; j += (int64_t)gridDim.x * 256) {
        T v = hs[j];
        if (cov[sig * (max_frames + 4) + (j >> 9)]) {
            v = (T)(blend * (double)fold_sample(xs, j, n, sg.n_ch) + (1.0 - blend) * (double)v);
            hs[j] = v;
        }
        const double a = fabs((double)v), b = fabs((double)ps[j]);
        mx = fmax(mx, fmax(a, b));
    }
    __shared__ double part[4];
    for (int d = 32; d >= 1; d >>= 1) mx = fmax(mx, __shfl_xor(mx, d));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicMax(&ymax[sig], (unsigned long long)key_of(fmax(fmax(part[0], part[1]), fmax(part[2], part[3]))));
}

// scale both outputs by 0.95 / max when the maximum exceeds 0.95; sums of squares for the log
template <typename T>
__global__ __launch_bounds__(256) void k_hpss_scale(const T *__restrict__ x, Signals sg, T *__restrict__ yh,
                                                    T *__restrict__ yp, const int64_t *__restrict__ y_off,
                                                    const unsigned long long *__restrict__ ymax,
                                                    double *__restrict__ sums) {
    const int64_t sig = blockIdx.y;
    const int64_t n = sg.len[sig];
    const int64_t m = (stft_frames(n, sg.n_fft, sg.hop, 1) - 1) * sg.hop;
    const T *xs = x + sg.off[sig];
    T *hs = yh + y_off[sig], *ps = yp + y_off[sig];
    const double mx = val_of((uint64_t)ymax[sig]);
    const bool scale = mx > 0.95;
    const T f = (T)(0.95 / mx);
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int64_t j = blockIdx.x * 256ll + threadIdx.x; j < n; j += (int64_t)gridDim.x * 256) {
        if (j < m) {
            T a = hs[j], b = ps[j];
            if (scale) {
                a *= f;
                b *= f;
                hs[j] = a;
                ps[j] = b;
            }
            s0 += (double)a * (double)a;
            s1 += (double)b * (double)b;
        }
        const double c = (double)fold_sample(xs, j, n, sg.n_ch);
        s2 += c * c;
    }
    for (int d = 32; d >= 1; d >>= 1) {
        s0 += __shfl_xor(s0, d);
        s1 += __shfl_xor(s1, d);
        s2 += __shfl_xor(s2, d);
    }
    __shared__ double part[4][3];
    if ((threadIdx.x & 63) == 0) {
        part[threadIdx.x >> 6][0] = s0;
        part[threadIdx.x >> 6][1] = s1;
        part[threadIdx.x >> 6][2] = s2;
    }
    __syncthreads();
    if (threadIdx.x < 3)
        atomicAdd(&sums[sig * 3 + threadIdx.x],
                  (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]));
}

__global__ void k_hpss_stats(Signals sg, int64_t n_sig, const double *__restrict__ sums,
                             const unsigned long long *__restrict__ ymax, const int64_t *__restrict__ n_tr,
                             double *__restrict__ stats) {
    const int64_t sig = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (sig >= n_sig) return;
    const int64_t n = sg.len[sig];
    const int64_t m = (stft_frames(n, sg.n_fft, sg.hop, 1) - 1) * sg.hop;
    stats[sig * 5 + 0] = sqrt(sums[sig * 3 + 0] / (double)(m > 0 ? m : 1));
    stats[sig * 5 + 1] = sqrt(sums[sig * 3 + 1] / (double)(m > 0 ? m : 1));
    stats[sig * 5 + 2] = sqrt(sums[sig * 3 + 2] / (double)(n > 0 ? n : 1));
    stats[sig * 5 + 3] = val_of((uint64_t)ymax[sig]);
    stats[sig * 5 + 4] = (double)n_tr[sig];
}

}  // namespace hpss

template <typename T>
int hpss_separate(Ctx *c, const T *d_x, int64_t n_sig, const int64_t *d_off, const int64_t *d_len, int64_t max_len,
                  int n_ch, int sample_rate, int n_fft, int hop, int kernel_size, double margin_h, double margin_p,
                  double power, int mask_type, int n_bands, const double *h_lo_hz, const double *h_hi_hz,
                  int preserve_transients, double threshold, double blend, T *d_yh, T *d_yp,
                  const int64_t *d_y_off, double *d_stats) {
    ADN_REQUIRE(c && d_x && d_off && d_len && d_yh && d_yp && d_y_off, "null pointer argument");
    ADN_REQUIRE(kernel_size >= 1 && kernel_size <= 31 && (kernel_size & 1), "kernel_size must be odd and <= 31, got %d", kernel_size);
    ADN_REQUIRE(mask_type == ADN_MASK_SOFT || mask_type == ADN_MASK_HARD || mask_type == ADN_MASK_BINARY, "unknown mask type %d", mask_type);
    ADN_REQUIRE(n_bands >= 0 && (n_bands == 0 || (h_lo_hz && h_hi_hz)), "band arguments missing");
    ADN_TRY(check_stft_args(n_fft, hop, n_ch));
    if (n_sig <= 0) return ADN_OK;
    if (max_len < n_fft) {
        set_error("window is longer than input signal (n_fft=%d, longest signal=%lld)", n_fft, (long long)max_len);
        return ADN_ERR_SHORT;
    }
    DeviceGuard dg(c->device);
    ws_reset(c);
    Tables<T> tb;
    ADN_TRY(get_tables<T>(c, n_fft, &tb));
    const int F = n_fft / 2 + 1;
    // band of every bin: the whole spectrum, or the LAST listed band that contains the bin
    // (src/hpss_separate.py:96-111 writes the bands in order; both band edges are inclusive)
    std::vector<int> blo(F, 0), bhi(F, F - 1);
    if (n_bands > 0) {
        std::fill(blo.begin(), blo.end(), -1);
        std::fill(bhi.begin(), bhi.end(), -1);
        const double val = 1.0 / ((double)n_fft * (1.0 / (double)sample_rate));
        for (int b = 0; b < n_bands; ++b) {
            int lo = -1, hi = -1;
            for (int k = 0; k < F; ++k) {
                const double f = (double)k * val;
                if (f >= h_lo_hz[b] && f <= h_hi_hz[b]) {
                    if (lo < 0) lo = k;
                    hi = k;
                }
            }
            for (int k = lo; lo >= 0 && k <= hi; ++k) {
                blo[k] = lo;
                bhi[k] = hi;
            }
        }
    }
    const int64_t Tmax = stft_frames(max_len, n_fft, hop, 1);
    const int64_t Ts = row_stride(Tmax);
    const int64_t out_max = (Tmax - 1) * hop;
    const int64_t max_frames = out_max > 2048 ? (out_max - 2048 + 511) / 512 : 1;
    const size_t nS = (size_t)n_sig * F * Ts, nM = (size_t)n_sig * Tmax * F;
    ADN_TRY(ws_reserve(c, ws_need(nS, sizeof(T)) + 2 * ws_need(nM, sizeof(T)) + ws_need(n_sig, 8) * 4 + 2 * ws_need(F, 4) +
                              2 * ws_need((size_t)n_sig * max_frames, 8) + ws_need((size_t)n_sig * 3, 8) +
                              ws_need((size_t)n_sig * 5, 8) + ws_need((size_t)n_sig * (max_frames + 4), 1) +
                              2 * APPLY_WS_BYTES));
    T *S = ws_take<T>(c, nS);
    T *Mh = ws_take<T>(c, nM), *Mp = ws_take<T>(c, nM);
    int64_t *Tn = ws_take<int64_t>(c, n_sig);
    unsigned long long *eo_max = ws_take<unsigned long long>(c, n_sig), *ymax = ws_take<unsigned long long>(c, n_sig);
    int64_t *n_tr = ws_take<int64_t>(c, n_sig);
    int *d_blo = ws_take<int>(c, F), *d_bhi = ws_take<int>(c, F);
    double *eo = ws_take<double>(c, (size_t)n_sig * max_frames), *eh = ws_take<double>(c, (size_t)n_sig * max_frames);
    double *sums = ws_take<double>(c, (size_t)n_sig * 3);
    unsigned char *cov = ws_take<unsigned char>(c, (size_t)n_sig * (max_frames + 4));
    double *stats_ws = ws_take<double>(c, (size_t)n_sig * 5);
    ADN_HIP(hipMemcpyAsync(d_blo, blo.data(), sizeof(int) * F, hipMemcpyHostToDevice, c->stream));
    ADN_HIP(hipMemcpyAsync(d_bhi, bhi.data(), sizeof(int) * F, hipMemcpyHostToDevice, c->stream));
    ADN_HIP(hipStreamSynchronize(c->stream));          // blo / bhi are stack-owned host vectors
    ADN_HIP(hipMemsetAsync(eo_max, 0, sizeof(unsigned long long) * n_sig, c->stream));
    ADN_HIP(hipMemsetAsync(ymax, 0, sizeof(unsigned long long) * n_sig, c->stream));
    ADN_HIP(hipMemsetAsync(sums, 0, sizeof(double) * 3 * n_sig, c->stream));
    Signals sg{d_off, d_len, n_ch, n_fft, hop, 1};
    ADN_LAUNCH(k_frame_counts, dim3((unsigned)ceil_div(n_sig, 256)), dim3(256), 0, c->stream, sg, n_sig, Tn);
    ADN_TRY(rows_dispatch<T>(c, d_x, sg, tb, n_sig, Ts, 0, S, nullptr));
    hpss::BandTable bt{d_blo, d_bhi};
    {
        const dim3 grid((unsigned)ceil_div(Tmax, hpss::TT), (unsigned)ceil_div(F, hpss::TB), (unsigned)n_sig);
        if (kernel_size == 31)
            ADN_LAUNCH((hpss::k_hpss_masks<T, 31>), grid, dim3(256), 0, c->stream, S, F, Ts, Tmax, Tn, bt,
                               kernel_size, (T)margin_h, (T)margin_p, (T)power, mask_type, Mh, Mp);
        else
            ADN_LAUNCH((hpss::k_hpss_masks<T, 0>), grid, dim3(256), 0, c->stream, S, F, Ts, Tmax, Tn, bt,
                               kernel_size, (T)margin_h, (T)margin_p, (T)power, mask_type, Mh, Mp);
    }
    ADN_HIP(hipGetLastError());
    ApplyArgs<T> a{};
    a.x = d_x;
    a.y_off = d_y_off;
    a.sg = sg;
    a.tb = tb;
    a.row_stride = Tmax * F;
    a.row_t_stride = F;
    a.no_clip = 1;
    a.st = 1;
    a.sf = 1;
    a.y = d_yh;
    a.row = Mh;
    ADN_TRY((apply_dispatch<T, APPLY_GAIN>(c, a, n_sig, Tmax)));
    a.y = d_yp;
    a.row = Mp;
    ADN_TRY((apply_dispatch<T, APPLY_GAIN>(c, a, n_sig, Tmax)));
    const unsigned gx = (unsigned)std::min<int64_t>(ceil_div(max_len, 256 * 8), std::max<int64_t>(8, 4096 / n_sig));
    ADN_LAUNCH((hpss::k_hpss_energy<T>), dim3((unsigned)ceil_div(max_frames, 4), (unsigned)n_sig), dim3(256), 0,
                       c->stream, d_x, sg, d_yh, d_y_off, max_frames, eo, eh);
    ADN_LAUNCH(hpss::k_hpss_emax, dim3((unsigned)n_sig), dim3(256), 0, c->stream, sg, max_frames, eo, eh, threshold,
                       preserve_transients, eo_max, n_tr, cov);
    ADN_LAUNCH((hpss::k_hpss_blend<T>), dim3(gx, (unsigned)n_sig), dim3(256), 0, c->stream, d_x, sg, d_yh, d_yp,
                       d_y_off, max_frames, cov, blend, ymax);
    ADN_LAUNCH((hpss::k_hpss_scale<T>), dim3(gx, (unsigned)n_sig), dim3(256), 0, c->stream, d_x, sg, d_yh, d_yp,
                       d_y_off, ymax, sums);
    double *stats = d_stats ? d_stats : stats_ws;
    ADN_LAUNCH(hpss::k_hpss_stats, dim3((unsigned)ceil_div(n_sig, 256)), dim3(256), 0, c->stream, sg, n_sig, sums,
                       ymax, n_tr, stats);
    ADN_HIP(hipGetLastError());
    return ADN_OK;
}
