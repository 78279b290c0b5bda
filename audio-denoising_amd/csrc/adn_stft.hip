// STFT stages: noise-PSD profile (phase 1), spectral subtraction (phase 2b) and
// Wiener filtering (phase 2c), batched over independent signals.
//
// Reference behaviour restated here (not its code): scipy.signal.stft/istft as
// called from src/noise_profile.py:11-14, src/spectral_subtract.py:82,108 and
// src/wiener_filter.py:72,128 (periodic Hann window, boundary=None, padded=True on
// the way in, boundary=True on the way out), and the per-bin arithmetic of those
// three files.
//
// Execution model: ONE WAVEFRONT PER FRAME.  A frame of n_fft real samples is packed
// into n_fft/2 complex points in the wave's private LDS buffer and transformed there
// (radix-2^2 decimation in frequency, bit-reversed spectrum; the inverse is the
// matching decimation-in-time network, so no reordering pass exists).  Waves of a
// workgroup exchange only what the stage needs across frames: a row of clean
// magnitudes per frame for the time smoothing, and the overlap-add accumulators.
// A workgroup slides along one segment of one signal, so each sample is loaded once
// per overlapping frame from L1/L2 and every spectrum stays on chip:
// HBM sees the input once and the output once.
#include <math.h>

#include <algorithm>

#include <vector>

#include <type_traits>

#include "adn_kernels.hpp"
#include "adn_stft_shared.hpp"
#include "adn_sortnet30.hpp"

namespace adn {

// ---------------------------------------------------------------- tables
// Device layout of one table set: window[N] | tw[Nc/2] | rtw[Nc/2+1]
//   window[k] = 0.5 - 0.5 cos(2 pi k / N)            (scipy get_window('hann', N))
//   tw[k]     = exp(-2 pi i k / Nc)                  (complex FFT of length Nc = N/2)
//   rtw[k]    = exp(-2 pi i k / N)                   (real-FFT split / merge)
template <typename T>
static int get_tables(Ctx *c, int n_fft, Tables<T> *out) {
    using V = typename Vec2<T>::type;
    const int Nc = n_fft / 2;
    int logc = 0;
    while ((1 << logc) < Nc) ++logc;
    StftTable *e = nullptr;
    for (int i = 0; i < c->n_tables; ++i)
        if (c->tables[i].n_fft == n_fft && c->tables[i].elem == (int)sizeof(T)) e = &c->tables[i];
    if (!e) {
        ADN_REQUIRE(c->n_tables < 16, "too many distinct (n_fft, precision) pairs on one context");
        // n_fft = 2048: the inter-pass twiddles of the register-resident transform (15 * 64 + 15 * 4
        // complex values, the layout of fast::fill_tw_table) are appended, ready to be copied to LDS
        const size_t fast_n = n_fft == 2048 ? 2 * (15 * 64 + 15 * 4) : 0;
        const size_t fast_off = ((size_t)n_fft + 2 * (Nc / 2) + 2 * (Nc / 2 + 1) + 3) & ~size_t(3);   // 16-byte aligned
        const size_t count = fast_off + fast_n;
        std::vector<T> h(count);
        double wsum = 0.0;
        for (int k = 0; k < n_fft; ++k) {
            const double w = 0.5 - 0.5 * cos(2.0 * M_PI * (double)k / (double)n_fft);
            wsum += w;
            h[k] = (T)w;
        }
        T *tw = h.data() + n_fft;
        for (int k = 0; k < Nc / 2; ++k) {
            const double a = -2.0 * M_PI * (double)k / (double)Nc;
            tw[2 * k] = (T)cos(a);
            tw[2 * k + 1] = (T)sin(a);
        }
        T *rtw = tw + 2 * (Nc / 2);
        for (int k = 0; k <= Nc / 2; ++k) {
            const double a = -2.0 * M_PI * (double)k / (double)n_fft;
            rtw[2 * k] = (T)cos(a);
            rtw[2 * k + 1] = (T)sin(a);
        }
        if (fast_n) {
            T *ft = h.data() + fast_off;
            auto w1024 = [&](int k, T *dst) {          // exp(-2 pi i k / 1024) from the rounded table above
                k &= 1023;
                const T re = tw[2 * (k & 511)], im = tw[2 * (k & 511) + 1];
                dst[0] = (k & 512) ? -re : re;
                dst[1] = (k & 512) ? -im : im;
            };
            for (int i = 0; i < 15 * 64; ++i) w1024((i & 63) * ((i >> 6) + 1), ft + 2 * i);
            for (int i = 0; i < 15 * 4; ++i) w1024(16 * (i & 3) * ((i >> 2) + 1), ft + 2 * (15 * 64 + i));
        }
        void *dev = nullptr;
        ADN_HIP(hipMalloc(&dev, count * sizeof(T)));
        ADN_HIP(hipMemcpy(dev, h.data(), count * sizeof(T), hipMemcpyHostToDevice));
        e = &c->tables[c->n_tables++];
        e->n_fft = n_fft;
        e->elem = (int)sizeof(T);
        e->dev = dev;
        e->wsum = wsum;
    }
    const T *base = static_cast<const T *>(e->dev);
    out->window = base;
    out->tw = reinterpret_cast<const V *>(base + n_fft);
    out->rtw = reinterpret_cast<const V *>(base + n_fft + 2 * (Nc / 2));
    out->fast_tw = n_fft == 2048
                       ? reinterpret_cast<const V *>(base + (((size_t)n_fft + 2 * (Nc / 2) + 2 * (Nc / 2 + 1) + 3) & ~size_t(3)))
                       : nullptr;
    out->wsum = e->wsum;
    out->n_fft = n_fft;
    out->logc = logc;
    return ADN_OK;
}

// ------------------------------------------------------- wave-level FFTs
// Position of complex point i in a wave's LDS buffer.  One spare slot after every 16 and every 64
// points keeps the strided sweeps of the transforms and the bit-reversed spectrum accesses spread
// over the banks (unpadded, the 64 lanes of a bit-reversed access all hit one bank).
__device__ __forceinline__ int lp(int i) { return i + (i >> 4) + (i >> 6); }
__host__ __device__ constexpr int lp_size(int Nc) { return Nc + (Nc >> 4) + (Nc >> 6) + 2; }

// In-place complex FFT of Nc = 2^logn points in LDS, natural order in, bit-reversed
// order out (decimation in frequency, two radix-2 stages fused per sweep).
// Every sweep handles the butterflies of a lane in batches of four: all LDS reads and twiddle
// loads of a batch are issued before the first result is needed, and the batch is written back
// afterwards (the butterflies of one sweep touch disjoint elements).
template <typename V>
__device__ void fft_dif(V *buf, const V *tw, int logn, int lane) {
    const int Nc = 1 << logn, Q = Nc >> 2;
    constexpr int U = 4;
    int s = logn;
    while (s >= 2) {
        const int m = 1 << (s - 1), hm = m >> 1, twstep = Nc >> s;
#pragma unroll 1
        for (int q0 = lane; q0 < Q; q0 += 64 * U) {
            V a0[U], a1[U], a2[U], a3[U], w1[U], w2[U];
            int i0[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = q0 + 64 * u < Q ? q0 + 64 * u : q0;         // out of range: repeat the first one
                const int j = q & (hm - 1);
                i0[u] = ((q >> (s - 2)) << s) + j;
                a0[u] = buf[lp(i0[u])];
                a1[u] = buf[lp(i0[u] + hm)];
                a2[u] = buf[lp(i0[u] + m)];
                a3[u] = buf[lp(i0[u] + m + hm)];
                w1[u] = tw[j * twstep];
                w2[u] = tw[2 * j * twstep];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const V b0 = cadd(a0[u], a2[u]), b2 = cmul(csub(a0[u], a2[u]), w1[u]);
                const V b1 = cadd(a1[u], a3[u]), d = csub(a1[u], a3[u]);
                const V b3 = cmul(V{d.y, -d.x}, w1[u]);                 // (a1 - a3) * (-i) * w1
                a0[u] = cadd(b0, b1);
                a1[u] = cmul(csub(b0, b1), w2[u]);
                a2[u] = cadd(b2, b3);
                a3[u] = cmul(csub(b2, b3), w2[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (u == 0 || q0 + 64 * u < Q) {
                    buf[lp(i0[u])] = a0[u];
                    buf[lp(i0[u] + hm)] = a1[u];
                    buf[lp(i0[u] + m)] = a2[u];
                    buf[lp(i0[u] + m + hm)] = a3[u];
                }
        }
        wave_sync();
        s -= 2;
    }
    if (s == 1) {
        for (int q = lane; q < Nc / 2; q += 64) {
            const V a = buf[lp(2 * q)], b = buf[lp(2 * q + 1)];
            buf[lp(2 * q)] = cadd(a, b);
            buf[lp(2 * q + 1)] = csub(a, b);
        }
        wave_sync();
    }
}

// Inverse network: bit-reversed order in, natural order out, conjugate twiddles, no scaling.
template <typename V>
__device__ void fft_dit_inv(V *buf, const V *tw, int logn, int lane) {
    const int Nc = 1 << logn, Q = Nc >> 2;
    constexpr int U = 4;
    int s = 1;
    if (logn & 1) {
        for (int q = lane; q < Nc / 2; q += 64) {
            const V a = buf[lp(2 * q)], b = buf[lp(2 * q + 1)];
            buf[lp(2 * q)] = cadd(a, b);
            buf[lp(2 * q + 1)] = csub(a, b);
        }
        wave_sync();
        s = 2;
    }
    while (s + 1 <= logn) {
        const int m = 1 << (s - 1);
#pragma unroll 1
        for (int q0 = lane; q0 < Q; q0 += 64 * U) {
            V a0[U], a1[U], a2[U], a3[U], w1[U], w2[U];
            int i0[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = q0 + 64 * u < Q ? q0 + 64 * u : q0;
                const int j = q & (m - 1);
                i0[u] = ((q >> (s - 1)) << (s + 1)) + j;
                a0[u] = buf[lp(i0[u])];
                a1[u] = buf[lp(i0[u] + m)];
                a2[u] = buf[lp(i0[u] + 2 * m)];
                a3[u] = buf[lp(i0[u] + 3 * m)];
                w1[u] = tw[j * (Nc >> s)];
                w2[u] = tw[j * (Nc >> (s + 1))];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const V t1 = cmulc(a1[u], w1[u]), t3 = cmulc(a3[u], w1[u]);
                const V b0 = cadd(a0[u], t1), b1 = csub(a0[u], t1), b2 = cadd(a2[u], t3), b3 = csub(a2[u], t3);
                const V u2 = cmulc(b2, w2[u]), v3 = cmulc(b3, w2[u]);
                const V u3 = V{-v3.y, v3.x};                          // * (+i)
                a0[u] = cadd(b0, u2);
                a2[u] = csub(b0, u2);
                a1[u] = cadd(b1, u3);
                a3[u] = csub(b1, u3);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (u == 0 || q0 + 64 * u < Q) {
                    buf[lp(i0[u])] = a0[u];
                    buf[lp(i0[u] + 2 * m)] = a2[u];
                    buf[lp(i0[u] + m)] = a1[u];
                    buf[lp(i0[u] + 3 * m)] = a3[u];
                }
        }
        wave_sync();
        s += 2;
    }
}

// Load frame t (windowed) into the wave's buffer as Nc packed complex points and
// transform it; on return buf holds X[k] * scale at brev(k) for k < Nc and X[Nc] at buf[Nc].
template <typename T, typename V>
__device__ void wave_rfft_frame(V *buf, const T *__restrict__ x, int64_t n, int n_ch, int64_t t,
                                const Tables<T> &tb, int hop, T scale, int lane, int ablate = 0,
                                const T *win = nullptr) {
    const int Nc = tb.n_fft / 2, logc = tb.logc;
    const int64_t p0 = t * hop;
    constexpr int LB = 8;                                    // loads in flight per lane and batch
    if (win == nullptr) win = tb.window;
    const V *wv = reinterpret_cast<const V *>(win);
    if (n_ch == 1 && p0 + tb.n_fft <= n && (reinterpret_cast<uintptr_t>(x + p0) % sizeof(V)) == 0) {
        const V *xv = reinterpret_cast<const V *>(x + p0);
#pragma unroll 1
        for (int i0 = lane; i0 < Nc; i0 += 64 * LB) {
            V a[LB], w[LB];
#pragma unroll
            for (int u = 0; u < LB; ++u) {
                const int i = i0 + 64 * u < Nc ? i0 + 64 * u : i0;
                a[u] = xv[i];
                w[u] = wv[i];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < LB; ++u)
                if (u == 0 || i0 + 64 * u < Nc) buf[lp(i0 + 64 * u)] = V{a[u].x * w[u].x, a[u].y * w[u].y};
        }
    } else if (n_ch == 2 && p0 + tb.n_fft <= n && (reinterpret_cast<uintptr_t>(x + 2 * p0) % sizeof(V)) == 0) {
        const V *xv = reinterpret_cast<const V *>(x + 2 * p0);   // (left, right) pairs
#pragma unroll 1
        for (int i0 = lane; i0 < Nc; i0 += 64 * LB) {
            V a[LB], b[LB], w[LB];
#pragma unroll
            for (int u = 0; u < LB; ++u) {
                const int i = i0 + 64 * u < Nc ? i0 + 64 * u : i0;
                a[u] = xv[2 * i];
                b[u] = xv[2 * i + 1];
                w[u] = wv[i];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < LB; ++u)
                if (u == 0 || i0 + 64 * u < Nc)
                    buf[lp(i0 + 64 * u)] = V{(a[u].x + a[u].y) / (T)2 * w[u].x, (b[u].x + b[u].y) / (T)2 * w[u].y};
        }
    } else {
        for (int i = lane; i < Nc; i += 64) {
            const T a = fold_sample(x, p0 + 2 * i, n, n_ch), b = fold_sample(x, p0 + 2 * i + 1, n, n_ch);
            buf[lp(i)] = V{a * win[2 * i], b * win[2 * i + 1]};
        }
    }
    wave_sync();
    if (!(ablate & 128)) fft_dif(buf, tb.tw, logc, lane);
    if (ablate & 256) return;
    // split the packed transform into the real-signal spectrum, in place (pairs k <-> Nc-k),
    // four pairs per lane and batch; lane 0 then handles the two self-paired bins
    constexpr int U = 4;
#pragma unroll 1
    for (int k0 = lane; k0 < Nc / 2; k0 += 64 * U) {
        V zk[U], zm[U], w[U];
        int pk[U], pm[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kk = k0 + 64 * u;
            const int k = (kk > 0 && kk < Nc / 2) ? kk : 1;
            pk[u] = lp(brev(k, logc));
            pm[u] = lp(brev(Nc - k, logc));
            zk[u] = buf[pk[u]];
            zm[u] = buf[pm[u]];
            w[u] = tb.rtw[k];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kk = k0 + 64 * u;
            if (kk > 0 && kk < Nc / 2) {
                const V e = V{(zk[u].x + zm[u].x) * (T)0.5, (zk[u].y - zm[u].y) * (T)0.5};
                const V o = V{(zk[u].y + zm[u].y) * (T)0.5, (zm[u].x - zk[u].x) * (T)0.5};   // -i/2 (zk - conj zm)
                const V wo = cmul(w[u], o);
                buf[pk[u]] = V{(e.x + wo.x) * scale, (e.y + wo.y) * scale};
                buf[pm[u]] = V{(e.x - wo.x) * scale, -(e.y - wo.y) * scale};
            }
        }
    }
    if (lane == 0) {
        const V z = buf[0];
        buf[0] = V{(z.x + z.y) * scale, (T)0};
        buf[lp(Nc)] = V{(z.x - z.y) * scale, (T)0};
        const int ph = lp(brev(Nc / 2, logc));
        const V zh = buf[ph];
        buf[ph] = V{zh.x * scale, -zh.y * scale};
    }
    wave_sync();
}

// Inverse of the split + inverse FFT: buf holds X'[k] as laid out above; on return the
// buffer, read as 2*Nc reals, holds Nc * irfft(X') (natural order, unscaled).
template <typename T, typename V>
__device__ void wave_irfft_frame(V *buf, const Tables<T> &tb, int lane) {
    const int Nc = tb.n_fft / 2, logc = tb.logc;
    constexpr int U = 4;
#pragma unroll 1
    for (int k0 = lane; k0 < Nc / 2; k0 += 64 * U) {
        V xk[U], xm[U], w[U];
        int pk[U], pm[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kk = k0 + 64 * u;
            const int k = (kk > 0 && kk < Nc / 2) ? kk : 1;
            pk[u] = lp(brev(k, logc));
            pm[u] = lp(brev(Nc - k, logc));
            xk[u] = buf[pk[u]];
            xm[u] = buf[pm[u]];
            w[u] = tb.rtw[k];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kk = k0 + 64 * u;
            if (kk > 0 && kk < Nc / 2) {
                const V e = V{(xk[u].x + xm[u].x) * (T)0.5, (xk[u].y - xm[u].y) * (T)0.5};
                const V d = V{(xk[u].x - xm[u].x) * (T)0.5, (xk[u].y + xm[u].y) * (T)0.5};   // (xk - conj xm)/2
                const V o = cmulc(d, w[u]);                                      // conj(w) * d
                buf[pk[u]] = V{e.x - o.y, e.y + o.x};                            // e + i o
                buf[pm[u]] = V{e.x + o.y, -(e.y - o.x)};                         // conj(e - i o)
            }
        }
    }
    if (lane == 0) {
        const T a = buf[0].x, b = buf[lp(Nc)].x;              // imaginary parts are ignored (irfft)
        buf[0] = V{(a + b) * (T)0.5, (a - b) * (T)0.5};
        const int ph = lp(brev(Nc / 2, logc));
        const V z = buf[ph];
        buf[ph] = V{z.x, -z.y};
    }
    wave_sync();
    fft_dit_inv(buf, tb.tw, logc, lane);
}

// ------------------------------------------- spectra rows (|X| or |X|^2), transposed
// Workgroup: W waves, FT frames of one signal (FT a multiple of W).
// Output S[(sig*F + k)*Tmax + t].

template <typename T, int W>
__global__ __launch_bounds__(W * 64) void k_stft_rows(const T *__restrict__ x, Signals sg,
                                                      Tables<T> tb, int64_t groups_per_sig,
                                                      int ROW_FT, int64_t Tmax, int power,
                                                      T *__restrict__ S,
                                                      unsigned long long *__restrict__ sig_max) {
    using V = typename Vec2<T>::type;
    extern __shared__ __align__(16) unsigned char smem[];
    const int Nc = tb.n_fft / 2, F = Nc + 1, logc = tb.logc;
    const int bstride = lp_size(Nc);
    V *bufs = reinterpret_cast<V *>(smem);
    T *tile = reinterpret_cast<T *>(bufs + (size_t)W * bstride);       // [F][ROW_FT + 1]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t sig = blockIdx.x / groups_per_sig, grp = blockIdx.x % groups_per_sig;
    const int64_t n = sg.len[sig];
    const int64_t Tn = stft_frames(n, tb.n_fft, sg.hop, sg.padded);
    const int64_t t0 = grp * ROW_FT;
    if (t0 >= Tn) return;
    const T *xs = x + sg.off[sig];
    V *buf = bufs + (size_t)wave * bstride;
    const T scale = (T)(1.0 / tb.wsum);
    T vmax = 0;
    for (int it = 0; it < ROW_FT / W; ++it) {
        const int ft = it * W + wave;
        const int64_t t = t0 + ft;
        if (t < Tn) {
            wave_rfft_frame<T, V>(buf, xs, n, sg.n_ch, t, tb, sg.hop, scale, lane);
            for (int k = lane; k <= Nc; k += 64) {
                const V z = buf[lp(k == Nc ? Nc : brev(k, logc))];
                const T p2 = z.x * z.x + z.y * z.y;
                const T v = power ? p2 : sqrt(p2);
                tile[k * (ROW_FT + 1) + ft] = v;
                vmax = v > vmax ? v : vmax;
            }
        }
        wave_sync();
    }
    __syncthreads();
    const int nft = (int)((Tn - t0) < ROW_FT ? (Tn - t0) : ROW_FT);
    for (int idx = threadIdx.x; idx < F * ROW_FT; idx += W * 64) {
        const int k = idx / ROW_FT, ft = idx % ROW_FT;
        if (ft < nft) S[((size_t)sig * F + k) * Tmax + t0 + ft] = tile[k * (ROW_FT + 1) + ft];
    }
    if (sig_max != nullptr) {
        for (int d = 32; d >= 1; d >>= 1) {
            const T o = __shfl_xor(vmax, d);
            vmax = o > vmax ? o : vmax;
        }
        // non-negative IEEE values order like their bit patterns
        if (lane == 0) atomicMax(&sig_max[sig], (unsigned long long)key_of((double)vmax));
    }
}

// Per-signal frame counts as int64 (segment lengths for the row statistics).
__global__ void k_frame_counts(Signals sg, int64_t n_sig, int64_t *Tn) {
    const int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (s < n_sig) Tn[s] = stft_frames(sg.len[s], sg.n_fft, sg.hop, sg.padded);
}

// ------------------------------------------------------ phase 1 epilogue
// B[k] = max(P[k], floor * max|Z|) -> box mean over freq_bins (edge replicated) ->
// box mean over time_frames of a time-constant row.
template <typename T>
__global__ __launch_bounds__(256) void k_profile_finish(const double *__restrict__ P,
                                                        const unsigned long long *__restrict__ sig_max,
                                                        int F, double floor_frac, int freq_bins,
                                                        int time_frames, T *__restrict__ out) {
    const int64_t sig = blockIdx.x;
    const T mx = (T)val_of((uint64_t)sig_max[sig]);
    const T lim = (T)floor_frac * mx;
    const int left = freq_bins / 2, right = freq_bins - left - 1;
    for (int k = threadIdx.x; k < F; k += 256) {
        T acc = 0;
        for (int o = -left; o <= right; ++o) {
            int kk = k + o;
            kk = kk < 0 ? 0 : (kk > F - 1 ? F - 1 : kk);
            const T p = (T)P[sig * F + kk];
            acc += p > lim ? p : lim;
        }
        const T v = acc / (T)freq_bins;
        T tsum = 0;
        for (int i = 0; i < time_frames; ++i) tsum += v;
        out[sig * F + k] = time_frames > 0 ? tsum / (T)time_frames : v;
    }
}

// -------------------------------------------------------------- VAD (2c)
// speech[i] = 20 log10(sqrt(mean(frame^2) + 1e-12) + 1e-12) > -40   (src/wiener_filter.py:17-26)
template <typename T>
__global__ __launch_bounds__(256) void k_vad_energy(const T *__restrict__ x, Signals sg,
                                                    int64_t n_sig, int64_t frame_len,
                                                    int64_t nv_max,
                                                    unsigned char *__restrict__ speech) {
    const int64_t gid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int64_t sig = gid / nv_max, i = gid % nv_max;
    if (sig >= n_sig) return;
    const int64_t n = sg.len[sig];
    int64_t nv = (n + frame_len - 1) / frame_len;
    nv = nv < 1 ? 1 : nv;
    if (i >= nv) return;
    const T *xs = x + sg.off[sig];
    const int64_t s = i * frame_len, e = (s + frame_len < n) ? s + frame_len : n;
    double acc = 0.0;
    constexpr int VEC = 16 / (int)sizeof(T);
    if (sg.n_ch == 1 && ((e - s) % VEC) == 0 && (reinterpret_cast<uintptr_t>(xs + s) & 15) == 0) {
        // mono, aligned: 16-byte loads, all of a lane's loads in flight together
        using VT = typename std::conditional<sizeof(T) == 4, float4, double2>::type;
        const VT *xv = reinterpret_cast<const VT *>(xs + s);
        const int nv4 = (int)((e - s) / VEC);
        constexpr int UB = 4;
        for (int q0 = lane; q0 < nv4; q0 += 64 * UB) {
            VT v[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) v[u] = xv[q0 + 64 * u < nv4 ? q0 + 64 * u : q0];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                if (u && q0 + 64 * u >= nv4) break;
                const T *ve = reinterpret_cast<const T *>(&v[u]);
#pragma unroll
                for (int c = 0; c < VEC; ++c) acc += (double)ve[c] * (double)ve[c];
            }
        }
    } else {
        for (int64_t p = s + lane; p < e; p += 64) {
            const double v = (double)fold_sample(xs, p, n, sg.n_ch);
            acc += v * v;
        }
    }
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    if (lane == 0) {
        unsigned char f = 0;
        if (e > s) {
            const double rms = sqrt(acc / (double)(e - s) + 1e-12);
            f = 20.0 * log10(rms + 1e-12) > -40.0;
        }
        speech[sig * nv_max + i] = f;
    }
}

// The same decision from energies that another kernel has added up (energy[sig * nv_max + i] = sum of squares).
__global__ void k_vad_flags(Signals sg, int64_t n_sig, int64_t frame_len, int64_t nv_max, const double *__restrict__ energy,
                            unsigned char *__restrict__ speech) {
    const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (gid >= n_sig * nv_max) return;
    const int64_t sig = gid / nv_max, i = gid % nv_max;
    const int64_t n = sg.len[sig];
    const int64_t s = i * frame_len, e = (s + frame_len < n) ? s + frame_len : n;
    unsigned char f = 0;
    if (e > s) {
        const double rms = sqrt(energy[gid] / (double)(e - s) + 1e-12);
        f = 20.0 * log10(rms + 1e-12) > -40.0;
    }
    speech[gid] = f;
}

// Hangover (a frame is speech if any of frames i-5..i fired) and the map from STFT
// frames to VAD frames: idx = min(int(t*hop/frame_len), n_vad-1)  (src/wiener_filter.py:27-33, 92-98).
__global__ void k_vad_map(Signals sg, int64_t n_sig, int64_t frame_len, int64_t nv_max, int64_t Tmax,
                          const unsigned char *__restrict__ speech, unsigned char *__restrict__ vad) {
    const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (gid >= n_sig * Tmax) return;
    const int64_t sig = gid / Tmax, t = gid % Tmax;
    const int64_t n = sg.len[sig];
    const int64_t Tn = stft_frames(n, sg.n_fft, sg.hop, sg.padded);
    if (t >= Tn) return;
    int64_t nv = (n + frame_len - 1) / frame_len;
    nv = nv < 1 ? 1 : nv;
    int64_t idx = (t * sg.hop) / frame_len;
    idx = idx > nv - 1 ? nv - 1 : idx;
    unsigned char v = 0;
    for (int64_t j = idx > 5 ? idx - 5 : 0; j <= idx; ++j) v |= speech[sig * nv_max + j];
    vad[gid] = v;
}

// speech_raw[row] = sum over VAD frames of S[row][t] / max(count, 1)  (src/wiener_filter.py:99-102)
template <typename T>
__global__ __launch_bounds__(256) void k_masked_mean(const T *__restrict__ S, int F, int64_t Tmax,
                                                     int64_t s_stride, int64_t n_rows, const int64_t *__restrict__ Tn,
                                                     const unsigned char *__restrict__ vad,
                                                     double *__restrict__ out) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n_rows) return;
    const int64_t sig = row / F;
    const int64_t T_ = Tn[sig];
    const T *r = S + row * s_stride;
    const unsigned char *m = vad + sig * Tmax;
    T acc = 0;
    int c = 0;                     // frames flagged as speech (every row recounts: no atomics)
    for (int64_t t = lane; t < T_; t += 64) {
        const bool on = m[t] != 0;
        acc += on ? r[t] : (T)0;
        c += on;
    }
    for (int d = 32; d >= 1; d >>= 1) {
        acc += __shfl_xor(acc, d);
        c += __shfl_xor(c, d);
    }
    if (lane == 0) out[row] = (double)(acc / (T)(c > 1 ? c : 1));
}

// gain[k] of the reference Wiener stage from the per-bin statistics (src/wiener_filter.py:84-86,
// 108-115, 35-39): three zero-padded running medians around the SNR arithmetic.
__device__ inline double medfilt_at(const double *v, int F, int k, int ksz) {
    if (ksz == 5) {
        // the reference's kernel size: five values in registers, a nine-comparator sorting network
        double w0 = k >= 2 ? v[k - 2] : 0.0, w1 = k >= 1 ? v[k - 1] : 0.0, w2 = v[k];
        double w3 = k + 1 < F ? v[k + 1] : 0.0, w4 = k + 2 < F ? v[k + 2] : 0.0;
        auto cx = [](double &x, double &y) {
            const bool sw = y < x;
            const double lo = sw ? y : x, hi = sw ? x : y;
            x = lo;
            y = hi;
        };
        cx(w0, w1); cx(w3, w4); cx(w2, w4); cx(w2, w3); cx(w0, w3); cx(w0, w2); cx(w1, w4); cx(w1, w3); cx(w1, w2);
        return w2;
    }
    double w[31];
    const int h = ksz / 2;
    for (int i = 0; i < ksz; ++i) {
        const int kk = k - h + i;
        w[i] = (kk < 0 || kk >= F) ? 0.0 : v[kk];
    }
    for (int i = 1; i < ksz; ++i) {       // insertion sort (ksz <= 31)
        const double key = w[i];
        int j = i - 1;
        while (j >= 0 && w[j] > key) {
            w[j + 1] = w[j];
            --j;
        }
        w[j + 1] = key;
    }
    return w[h];
}

// (a bin per thread for n_fft = 2048: the three running medians are short serial loops, and a signal is one workgroup)
static constexpr int GAIN_THREADS = 1024;
template <typename T>
__global__ __launch_bounds__(GAIN_THREADS) void k_wiener_gain(const double *__restrict__ p10,
                                                     const double *__restrict__ speech_raw, int F,
                                                     int ksz, T *__restrict__ gain) {
    extern __shared__ __align__(16) unsigned char smem[];
    double *a = reinterpret_cast<double *>(smem), *b = a + F, *g = b + F;
    const int64_t sig = blockIdx.x;
    for (int k = threadIdx.x; k < F; k += GAIN_THREADS) {
        a[k] = (double)(T)p10[sig * F + k];
        b[k] = (double)(T)speech_raw[sig * F + k];
    }
    __syncthreads();
    double nz[9], sp[9];
    int cnt = 0;
    for (int k = threadIdx.x; k < F; k += GAIN_THREADS, ++cnt) {
        nz[cnt] = (double)(T)fmax(medfilt_at(a, F, k, ksz), 1e-12);
        sp[cnt] = medfilt_at(b, F, k, ksz);
    }
    __syncthreads();
    cnt = 0;
    for (int k = threadIdx.x; k < F; k += GAIN_THREADS, ++cnt) {
        const T noise = (T)nz[cnt];
        T speech = (T)sp[cnt];
        speech = speech > noise ? speech : noise;
        const T snr = speech / (noise + (T)1e-12);
        T gv = snr / ((T)1 + snr);
        gv = gv < (T)0 ? (T)0 : (gv > (T)1 ? (T)1 : gv);
        g[k] = (double)gv;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < F; k += GAIN_THREADS) gain[sig * F + k] = (T)medfilt_at(g, F, k, ksz);
}

// 1 / sqrt(x) for x > 0 to within a few ulp: hardware estimate (about 2^-26 relative) refined by
// two Newton steps; replaces the far longer correctly rounded sqrt and division sequences in the
// per-bin magnitude arithmetic of the binary64 kernels.  x = 0 gives +inf (callers select).
__device__ __forceinline__ double inv_sqrt(double x) {
    double r = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    r = r * (1.5 - hx * r * r);
    r = r * (1.5 - hx * r * r);
    return r;
}
__device__ __forceinline__ float inv_sqrt(float x) { return 1.0f / sqrtf(x); }

// ---------------------------------------- fused STFT -> modify -> ISTFT overlap-add
static constexpr int MAG_PAD = 16;     // replicated bins kept beyond both ends of a magnitude row (box <= 33 bins)
template <typename T, int W, int MODE>
__global__ __launch_bounds__(W * 64) void k_stft_apply(ApplyArgs<T> a) {
    using V = typename Vec2<T>::type;
    extern __shared__ __align__(16) unsigned char smem[];
    const Tables<T> &tb = a.tb;
    const int N = tb.n_fft, Nc = N / 2, F = Nc + 1, logc = tb.logc;
    const int H = a.sg.hop, R = N / H;
    const int RING = W + R;
    const int bstride = lp_size(Nc), mstride = F + 1 + 2 * MAG_PAD;
    const int h = MODE == APPLY_SUBTRACT ? a.st / 2 : 0;
    const int hr = MODE == APPLY_SUBTRACT ? a.st - h - 1 : 0;
    // Frame t lives in buffer t % NB and its smoothed magnitudes in row t % MR.  An iteration
    // transforms W new frames; the hr newest ones wait (parked in their buffers) for their right
    // neighbours and are synthesised by the next iteration, so that every iteration after the
    // first synthesises W frames and no frame is transformed twice inside a segment.
    const int NB = W + hr, MR = W + h + hr;
    V *bufs = reinterpret_cast<V *>(smem);                                // [NB][bstride]
    // [MR][mstride]: a row holds bins 0..Nc at offset MAG_PAD, with replicated edge bins on both sides
    T *mags = reinterpret_cast<T *>(bufs + (size_t)NB * bstride) + MAG_PAD;
    T *acc = mags - MAG_PAD + (MODE == APPLY_SUBTRACT ? (size_t)MR * mstride : 0);   // [RING][H]
    T *wsq = acc + (size_t)RING * H;                                      // [H] 1 / norm of a fully overlapped hop
    // the window is staged in LDS when the host found room for it (generic pointer: either memory)
    const T *win = a.win_lds ? wsq + H : tb.window;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, tid = threadIdx.x;
    const int64_t sig = blockIdx.x / a.segs_per_sig, seg = blockIdx.x % a.segs_per_sig;
    const int64_t n = a.sg.len[sig];
    const int64_t Tn = stft_frames(n, N, H, 1);
    const int64_t out_hops = Tn - 1;                       // output length (Tn-1)*H
    const int64_t g0 = seg * a.seg_hops;
    if (Tn < 1 || g0 >= out_hops) return;
    const int64_t g1 = (g0 + a.seg_hops < out_hops) ? g0 + a.seg_hops : out_hops;
    // padded-coordinate hops [ph0, ph1) are produced by this workgroup
    const int64_t ph0 = g0 + R / 2, ph1 = g1 + R / 2;
    int64_t next_u = ph0 - R + 1;
    next_u = next_u < 0 ? 0 : next_u;
    const int64_t last_u = (ph1 - 1 < Tn - 1) ? ph1 - 1 : Tn - 1;
    const T *xs = a.x + a.sg.off[sig];
    T *ys = a.y + a.y_off[sig];
    int16_t *ps = a.pcm ? a.pcm + a.y_off[sig] : nullptr;
    const T *row = a.row + sig * a.row_stride;
    const T fscale = (T)(1.0 / tb.wsum);
    const T iscale = (T)(tb.wsum / (double)Nc);

    for (int i = tid; i < RING * H; i += W * 64) acc[i] = 0;
    if (a.win_lds)
        for (int i = tid; i < N; i += W * 64) wsq[H + i] = tb.window[i];
    for (int i = tid; i < H; i += W * 64) {
        T norm = 0;
        for (int r = R - 1; r >= 0; --r) {                 // same order as the edge loop below
            const T w = tb.window[r * H + i];
            norm += w * w;
        }
        wsq[i] = (T)1 / (norm > (T)1e-10 ? norm : (T)1);    // reciprocal: one multiplication per output sample
    }
    __syncthreads();

    int64_t base = next_u - h;                              // first new frame of the iteration
    base = base < 0 ? 0 : base;
    while (next_u <= last_u) {
        const int64_t ta = base + wave;
        int64_t a_last = last_u + hr;
        a_last = a_last > Tn - 1 ? Tn - 1 : a_last;
        // ---- stage A: spectrum of frame ta (and its clean magnitudes)
        if (ta <= a_last && !(a.ablate & 1)) {
            V *buf = bufs + (size_t)(ta % NB) * bstride;
            wave_rfft_frame<T, V>(buf, xs, n, a.sg.n_ch, ta, tb, H, fscale, lane, a.ablate, win);
            if (MODE == APPLY_SUBTRACT && !(a.ablate & 32)) {
                // clean magnitudes of the frame, then their box mean along frequency (edge
                // replicated), in place: every lane reads the neighbourhoods of its bins before any
                // lane overwrites a bin.  The time mean over the neighbouring frames follows in
                // stage B, in the same order as the two 1-D passes of scipy's uniform_filter.
                T *mrow = mags + (size_t)(ta % MR) * mstride;
                constexpr int MB = 6;                            // bins per lane and batch, loads first
#pragma unroll 1
                for (int k0 = lane; k0 <= Nc; k0 += 64 * MB) {
                    V z[MB];
                    T nz[MB];
#pragma unroll
                    for (int u = 0; u < MB; ++u) {
                        const int k = k0 + 64 * u <= Nc ? k0 + 64 * u : Nc;
                        z[u] = buf[lp(k == Nc ? Nc : brev(k, logc))];
                        nz[u] = row[k];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < MB; ++u) {
                        const T p2 = z[u].x * z[u].x + z[u].y * z[u].y;
                        const T mag = p2 > (T)0 ? p2 * inv_sqrt(p2) : (T)0;
                        const T c1 = mag - a.alpha * nz[u], c2 = a.beta * mag;
                        if (k0 + 64 * u <= Nc) mrow[k0 + 64 * u] = c1 > c2 ? c1 : c2;
                    }
                }
                const int fl = a.sf / 2, fr = a.sf - fl - 1;
                wave_sync();
                if (lane < MAG_PAD) {                            // edge replication
                    mrow[-1 - lane] = mrow[0];
                    mrow[Nc + 1 + lane] = mrow[Nc];
                }
                wave_sync();
                const T inv_sf = (T)1 / (T)a.sf;
                constexpr int KB = 33;                           // bins per lane: ceil((4096 / 2 + 1) / 64)
                T box[KB];
                if (a.sf == 5) {                                 // the reference's box: all reads in flight together
#pragma unroll
                    for (int u = 0; u < KB; ++u) {
                        if (64 * u > Nc) break;                  // uniform
                        const int k = lane + 64 * u <= Nc ? lane + 64 * u : Nc;
                        const T v0 = mrow[k - 2], v1 = mrow[k - 1], v2 = mrow[k], v3 = mrow[k + 1], v4 = mrow[k + 2];
                        box[u] = ((((((T)0 + v0) + v1) + v2) + v3) + v4) * inv_sf;
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < KB; ++u) {
                        if (64 * u > Nc) break;
                        const int k = lane + 64 * u <= Nc ? lane + 64 * u : Nc;
                        T fs = 0;
                        for (int df = -fl; df <= fr; ++df) fs += mrow[k + df];
                        box[u] = fs * inv_sf;
                    }
                }
                wave_sync();
#pragma unroll
                for (int u = 0; u < KB; ++u) {
                    if (64 * u > Nc) break;
                    if (lane + 64 * u <= Nc) mrow[lane + 64 * u] = box[u];
                }
            }
        }
        __syncthreads();
        // ---- frames [ub, ue] can be finalised: all their right neighbours have been transformed
        // (ends of the signal: the neighbours are clamped to its first / last frame)
        const int64_t ub = next_u;
        int64_t ue = base + W - 1 - hr;
        ue = ue > last_u ? last_u : ue;
        // ---- stage B: wave w modifies the spectrum of frame ub + w and transforms it back
        const int64_t t = ub + wave;
        V *buf = bufs + (size_t)(t % NB) * bstride;
        if (t <= ue && !(a.ablate & 2)) {
            if (a.ablate & 64) {
            } else if (MODE == APPLY_SUBTRACT) {
                const T inv_st = (T)1 / (T)a.st;
                // rows of the time neighbours (clamped at the signal's first / last frame)
                const T *trow[7];
#pragma unroll
                for (int i = 0; i < 7; ++i) {
                    int64_t tt = t - h + i;
                    tt = tt < 0 ? 0 : (tt > Tn - 1 ? Tn - 1 : tt);
                    trow[i] = mags + (size_t)(tt % MR) * mstride;
                }
                constexpr int MB = 6;                            // bins per lane and batch, loads first
#pragma unroll 1
                for (int k0 = lane; k0 <= Nc; k0 += 64 * MB) {
                    V z[MB];
                    T tsum[MB];
                    int pk[MB];
#pragma unroll
                    for (int u = 0; u < MB; ++u) {
                        const int k = k0 + 64 * u <= Nc ? k0 + 64 * u : Nc;
                        pk[u] = lp(k == Nc ? Nc : brev(k, logc));
                        z[u] = buf[pk[u]];
                        T ts = 0;
#pragma unroll
                        for (int i = 0; i < 7; ++i)
                            if (i < a.st) ts += trow[i][k];
                        tsum[u] = ts;
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < MB; ++u) {
                        const T sm = tsum[u] * inv_st;
                        const T p2 = z[u].x * z[u].x + z[u].y * z[u].y;
                        // clean * exp(i*angle(Z)) = clean * Z / |Z|; angle(0) = 0
                        const T g = sm * inv_sqrt(p2);
                        if (k0 + 64 * u <= Nc) buf[pk[u]] = p2 > (T)0 ? V{z[u].x * g, z[u].y * g} : V{sm, (T)0};
                    }
                }
            } else {
                for (int k = lane; k <= Nc; k += 64) {
                    const int pk = lp(k == Nc ? Nc : brev(k, logc));
                    const V z = buf[pk];
                    const T gk = row[k + (int64_t)t * a.row_t_stride];
                    buf[pk] = V{z.x * gk, z.y * gk};
                }
            }
            wave_sync();
            wave_irfft_frame<T, V>(buf, tb, lane);
        }
        __syncthreads();
        // ---- overlap-add of frames [ub, ue] into padded hops [ub, ue + R - 1]
        if (ue >= ub && !(a.ablate & 4)) {
            const bool final_group = (ue == Tn - 1);
            const int nj = (int)(ue - ub) + R;              // hops touched, < RING
            const int LOGH = 31 - __clz(H);                 // the hop is a power of two
            const int slot0 = (int)(ub % RING);
            const T *frames = reinterpret_cast<const T *>(bufs);
            const int fb0 = (int)(ub % NB);                 // buffer of frame ub
#pragma unroll 1
            for (int idx = tid; idx < (nj << LOGH); idx += W * 64) {
                const int jj = idx >> LOGH, i = idx & (H - 1);
                const int64_t j = ub + jj;
                int sl = slot0 + jj;
                sl = sl >= RING ? sl - RING : sl;
                T *slot = acc + ((size_t)sl << LOGH) + i;
                T v = *slot;
                // frames ub + f, f in [f_lo, f_hi], overlap hop j
                const int f_lo = jj - R + 1 > 0 ? jj - R + 1 : 0;
                const int f_hi = jj < (int)(ue - ub) ? jj : (int)(ue - ub);
                int fb = fb0 + f_lo;
                fb = fb >= NB ? fb - NB : fb;
                for (int f = f_lo; f <= f_hi; ++f) {
                    const int pos = ((jj - f) << LOGH) + i;
                    v += frames[(size_t)fb * (2 * bstride) + 2 * lp(pos >> 1) + (pos & 1)] * iscale * win[pos];
                    fb = fb + 1 >= NB ? 0 : fb + 1;
                }
                if (j <= ue || final_group) {
                    if (j >= ph0 && j < ph1) {
                        T o = v * wsq[i];
                        if (j - R + 1 < 0 || j > Tn - 1) {            // first / last hops: fewer frames overlap
                            T norm = 0;
                            int64_t t_lo = j - R + 1;
                            t_lo = t_lo < 0 ? 0 : t_lo;
                            const int64_t t_hi = j < Tn - 1 ? j : Tn - 1;
                            for (int64_t tt = t_lo; tt <= t_hi; ++tt) {
                                const T w = win[((int)(j - tt) << LOGH) + i];
                                norm += w * w;
                            }
                            o = v / (norm > (T)1e-10 ? norm : (T)1);
                        }
                        if (!a.no_clip) o = o < (T)-1 ? (T)-1 : (o > (T)1 ? (T)1 : o);      // np.clip keeps NaN
                        const int64_t yo = ((j - R / 2) << LOGH) + i;
                        if (ps) {
                            const int k = pcm_int32(o) >> 16;
                            ps[yo] = (int16_t)k;
                            o = (T)k * (T)(1.0 / 32768.0);
                        }
                        ys[yo] = o;
                    }
                    *slot = 0;
                } else {
                    *slot = v;
                }
            }
        }
        __syncthreads();
        next_u = ue + 1;
        base += W;
    }
}

// ---------------------------------------------------------- host drivers
static int check_stft_args(int n_fft, int hop, int n_ch) {
    ADN_REQUIRE(n_fft >= 64 && n_fft <= 4096 && (n_fft & (n_fft - 1)) == 0,
                "n_fft must be a power of two in [64, 4096], got %d", n_fft);
    // the overlap-add tiles assume that the n_fft/2 samples istft trims are whole hops
    if (hop < 1 || hop > n_fft / 2 || (n_fft / 2) % hop != 0) {
        set_error("hop must divide n_fft/2 (got n_fft=%d hop=%d)", n_fft, hop);
        return ADN_ERR_UNSUPPORTED;
    }
    ADN_REQUIRE(n_ch >= 1, "n_ch must be >= 1");
    return ADN_OK;
}

template <typename T, int W>
static int launch_rows(Ctx *c, const T *d_x, const Signals &sg, const Tables<T> &tb, int64_t n_sig,
                       int ROW_FT, int64_t Tmax, int power, T *S, unsigned long long *sig_max) {
    using V = typename Vec2<T>::type;
    const int Nc = tb.n_fft / 2;
    const size_t lds = (size_t)W * lp_size(Nc) * sizeof(V) + (size_t)(Nc + 1) * (ROW_FT + 1) * sizeof(T);
    const int64_t groups = ceil_div(Tmax, ROW_FT);
    auto kern = k_stft_rows<T, W>;
    ADN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ADN_LAUNCH_NAMED("k_stft_rows", kern, dim3((unsigned)(groups * n_sig)), dim3(W * 64), lds, c->stream, d_x, sg,
                       tb, groups, ROW_FT, Tmax, power, S, sig_max);
    ADN_HIP(hipGetLastError());
    return ADN_OK;
}

// Row stride of the transposed magnitude / power matrix: a multiple of 32 frames, so that the
// 16-frame segments a workgroup writes are aligned sectors of the row (no partial-line writes).
static inline int64_t row_stride(int64_t Tmax) { return (Tmax + 31) & ~int64_t(31); }

template <typename T>
static int rows_dispatch(Ctx *c, const T *d_x, const Signals &sg, const Tables<T> &tb, int64_t n_sig,
                         int64_t Tmax, int power, T *S, unsigned long long *sig_max) {
    using V = typename Vec2<T>::type;
    const int Nc = tb.n_fft / 2;
    if constexpr (std::is_same<T, float>::value) {
        if (tb.n_fft == 2048 && !c->force_generic) return rows_fast(c, d_x, sg, tb, n_sig, Tmax, power, S, sig_max);
    }
    const size_t per_wave = (size_t)lp_size(Nc) * sizeof(V);
    // widest transposed tile (frames per workgroup) and most waves that fit the CU's LDS
    for (int ft = 16; ft >= 2; ft >>= 1) {
        const size_t tile = (size_t)(Nc + 1) * (ft + 1) * sizeof(T);
        const size_t budget = 150 * 1024;
        if (ft >= 8 && 8 * per_wave + tile <= budget) return launch_rows<T, 8>(c, d_x, sg, tb, n_sig, ft, Tmax, power, S, sig_max);
        if (ft >= 4 && 4 * per_wave + tile <= budget) return launch_rows<T, 4>(c, d_x, sg, tb, n_sig, ft, Tmax, power, S, sig_max);
        if (2 * per_wave + tile <= budget) return launch_rows<T, 2>(c, d_x, sg, tb, n_sig, ft, Tmax, power, S, sig_max);
    }
    set_error("n_fft=%d does not fit the 160 KB LDS of a CU", tb.n_fft);
    return ADN_ERR_UNSUPPORTED;
}

// LDS of k_stft_apply: frame buffers (W + the parked ones), magnitude rows, hop accumulators + norm
template <typename T>
static size_t apply_lds_bytes(int W, int mode, int Nc, int R, int hop, int st) {
    using V = typename Vec2<T>::type;
    const int h = mode == APPLY_SUBTRACT ? st / 2 : 0, hr = mode == APPLY_SUBTRACT ? st - h - 1 : 0;
    return (size_t)(W + hr) * lp_size(Nc) * sizeof(V) +
           (mode == APPLY_SUBTRACT ? (size_t)(W + h + hr) * (Nc + 2 + 2 * MAG_PAD) * sizeof(T) : 0) +
           (size_t)(W + R + 1) * hop * sizeof(T);
}

template <typename T, int W, int MODE>
static int launch_apply(Ctx *c, ApplyArgs<T> a, int64_t n_sig, int64_t Tmax) {
    const int N = a.tb.n_fft, Nc = N / 2, R = N / a.sg.hop;
    size_t lds = apply_lds_bytes<T>(W, MODE, Nc, R, a.sg.hop, a.st);
    a.win_lds = lds + (size_t)N * sizeof(T) <= 160 * 1024;
    if (a.win_lds) lds += (size_t)N * sizeof(T);
    const int64_t out_hops = Tmax - 1;
    if (out_hops <= 0) return ADN_OK;
    // enough workgroups to fill the chip several times, but long enough segments to
    // amortise the (R - 1 + smoothing halo) warm-up frames of every segment
    int64_t seg = ceil_div(n_sig * out_hops, (int64_t)c->num_cus * 8);
    seg = seg < 16 ? 16 : (seg > 128 ? 128 : seg);
    a.seg_hops = (int)seg;
    a.segs_per_sig = ceil_div(out_hops, seg);
    a.ablate = c->ablate;
    auto kern = k_stft_apply<T, W, MODE>;
    ADN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ADN_LAUNCH_NAMED((MODE == APPLY_SUBTRACT ? "k_stft_apply<SUBTRACT>" : "k_stft_apply<GAIN>"), kern, dim3((unsigned)(a.segs_per_sig * n_sig)), dim3(W * 64), lds, c->stream, a);
    ADN_HIP(hipGetLastError());
    return ADN_OK;
}

template <typename T, int MODE>
static int apply_dispatch(Ctx *c, const ApplyArgs<T> &a, int64_t n_sig, int64_t Tmax) {
    const int N = a.tb.n_fft, Nc = N / 2, R = N / a.sg.hop;
    auto need = [&](int W) { return apply_lds_bytes<T>(W, MODE, Nc, R, a.sg.hop, a.st); };
    const int min_w = MODE == APPLY_SUBTRACT ? a.st : 1;
    if constexpr (std::is_same<T, float>::value) {
        // the register-resident kernel is built for the reference's 3-frame time smoothing
        const bool st_ok = MODE == APPLY_GAIN || a.st == 3;
        if (!c->force_generic && st_ok && apply_fast_ok(N, a.sg.hop, a.st, a.sf)) return apply_fast<MODE>(c, a, n_sig, Tmax);
    } else {
        if (!c->force_generic && (c->ablate & 256) == 0 && apply_wave64_ok(N, a.sg.hop, a.st, a.sf, MODE))
            return apply_wave64<MODE>(c, a, n_sig, Tmax);
    }
    if (need(8) <= 160 * 1024) return launch_apply<T, 8, MODE>(c, a, n_sig, Tmax);
    if constexpr (MODE == APPLY_SUBTRACT) {      // a time smoothing of 5 ... 7 frames needs as many waves
        if (need(6) <= 160 * 1024 && min_w > 4 && min_w <= 6) return launch_apply<T, 6, MODE>(c, a, n_sig, Tmax);
    }
    if (need(4) <= 160 * 1024 && min_w <= 4) return launch_apply<T, 4, MODE>(c, a, n_sig, Tmax);
    if (need(3) <= 160 * 1024 && min_w <= 3) return launch_apply<T, 3, MODE>(c, a, n_sig, Tmax);
    if (need(2) <= 160 * 1024 && min_w <= 2) return launch_apply<T, 2, MODE>(c, a, n_sig, Tmax);
    set_error("n_fft=%d hop=%d smoothing=%d does not fit the 160 KB LDS of a CU", N, a.sg.hop, a.st);
    return ADN_ERR_UNSUPPORTED;
}

template <typename T>
int noise_psd(Ctx *c, const T *d_x, int64_t n_sig, const int64_t *d_off, const int64_t *d_len,
              int64_t max_len, int n_ch, int n_fft, int hop, int padded, double percentile,
              double spectral_floor, int freq_bins, int time_frames, T *d_baseline) {
    ADN_REQUIRE(c && d_x && d_off && d_len && d_baseline, "null pointer argument");
    ADN_TRY(check_stft_args(n_fft, hop, n_ch));
    ADN_REQUIRE(percentile >= 0.0 && percentile <= 100.0, "percentile must be in [0, 100]");
    ADN_REQUIRE(freq_bins >= 1 && time_frames >= 1, "smoothing sizes must be >= 1");
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
    const int64_t Tmax = stft_frames(max_len, n_fft, hop, padded);
    const int64_t Ts = row_stride(Tmax);
    Signals sg{d_off, d_len, n_ch, n_fft, hop, padded};
    // rows of up to a few thousand frames take the grouped layout (rows written by waves that work alone, read by
    // the register-resident row select), like the statistics of the Wiener stage
    const bool grouped = !c->force_generic && (c->ablate & 64) == 0 && rows_wave_ok(sg, n_fft) && Tmax <= select_grouped_limit(sizeof(T));
    const int64_t sig_floats = std::max<int64_t>((int64_t)F * Ts, rows_wave_sig_floats(Tmax));
    const size_t nS = (size_t)n_sig * sig_floats;
    ADN_TRY(ws_reserve(c, ws_need(nS, sizeof(T)) + ws_need(n_sig, 8) * 2 + ws_need((size_t)n_sig * F, 8) +
                              select_ws_bytes(n_sig * F, Tmax) + APPLY_WS_BYTES));
    T *S = ws_take<T>(c, nS);
    int64_t *Tn = ws_take<int64_t>(c, n_sig);
    unsigned long long *sig_max = ws_take<unsigned long long>(c, n_sig);
    double *P = ws_take<double>(c, (size_t)n_sig * F);
    ADN_HIP(hipMemsetAsync(sig_max, 0, sizeof(unsigned long long) * n_sig, c->stream));
    ADN_LAUNCH(k_frame_counts, dim3((unsigned)ceil_div(n_sig, 256)), dim3(256), 0, c->stream, sg, n_sig, Tn);
    SelectArgs sa;
    if (grouped) {
        ADN_TRY(rows_wave<T>(c, d_x, sg, tb, n_sig, Tmax, 0, S, sig_floats, nullptr, 0, 0, sig_max));
        sa.quad_stride = ROWS_GROUP_STRIDE;
        sa.blk_rows = F;
        sa.blk_sig_stride = sig_floats;
    } else {
        ADN_TRY(rows_dispatch<T>(c, d_x, sg, tb, n_sig, Ts, 0, S, sig_max));
    }
    sa.n_seg = n_sig * F;
    sa.seg_stride = Ts;
    sa.d_seg_len = Tn;
    sa.len_div = F;
    sa.max_len = Tmax;
    sa.mode = SEL_PERCENTILE;
    sa.q = percentile;
    sa.d_out = P;
    ADN_TRY(launch_select<T>(c, S, sa));
    ADN_LAUNCH((k_profile_finish<T>), dim3((unsigned)n_sig), dim3(256), 0, c->stream, P, sig_max, F,
                       spectral_floor, freq_bins, time_frames, d_baseline);
    ADN_HIP(hipGetLastError());
    return ADN_OK;
}

template <typename T>
int spectral_subtract(Ctx *c, const T *d_x, int64_t n_sig, const int64_t *d_off,
                      const int64_t *d_len, int64_t max_len, int n_ch, const T *d_noise,
                      int64_t noise_stride, int n_fft, int hop, double alpha, double beta,
                      int smooth_time, int smooth_freq, T *d_y, const int64_t *d_y_off, int16_t *d_pcm16) {
    ADN_REQUIRE(c && d_x && d_off && d_len && d_noise && d_y && d_y_off, "null pointer argument");
    ADN_TRY(check_stft_args(n_fft, hop, n_ch));
    ADN_REQUIRE(smooth_time >= 1 && smooth_time <= 7 && smooth_freq >= 1 && smooth_freq <= 33,
                "smoothing sizes out of range (time 1..7, frequency 1..33)");
    if (n_sig <= 0) return ADN_OK;
    if (max_len < n_fft) {
        set_error("window is longer than input signal (n_fft=%d, longest signal=%lld)", n_fft, (long long)max_len);
        return ADN_ERR_SHORT;
    }
    DeviceGuard dg(c->device);
    ws_reset(c);
    ADN_TRY(ws_reserve(c, APPLY_WS_BYTES));
    Tables<T> tb;
    ADN_TRY(get_tables<T>(c, n_fft, &tb));
    ApplyArgs<T> a{};
    a.x = d_x;
    a.y = d_y;
    a.y_off = d_y_off;
    a.sg = Signals{d_off, d_len, n_ch, n_fft, hop, 1};
    a.tb = tb;
    a.pcm = d_pcm16;
    a.row = d_noise;
    a.row_stride = noise_stride;
    a.alpha = (T)alpha;
    a.beta = (T)beta;
    a.st = smooth_time;
    a.sf = smooth_freq;
    return apply_dispatch<T, APPLY_SUBTRACT>(c, a, n_sig, stft_frames(max_len, n_fft, hop, 1));
}

template <typename T>
int wiener(Ctx *c, const T *d_x, int64_t n_sig, const int64_t *d_off, const int64_t *d_len,
           int64_t max_len, int n_ch, int sample_rate, int n_fft, int hop, double update_ms,
           int kernel_bins, int vad_guided, T *d_y, const int64_t *d_y_off, T *d_gain, int16_t *d_pcm16) {
    ADN_REQUIRE(c && d_x && d_off && d_len && d_y && d_y_off, "null pointer argument");
    ADN_TRY(check_stft_args(n_fft, hop, n_ch));
    ADN_REQUIRE(kernel_bins >= 1 && kernel_bins <= 31 && (kernel_bins & 1), "median kernel must be odd and <= 31");
    ADN_REQUIRE(n_fft / 2 + 1 <= 9 * 256, "n_fft too large for the gain kernel");
    const int64_t frame_len = (int64_t)((double)sample_rate * update_ms / 1000.0);
    ADN_REQUIRE(!vad_guided || frame_len >= 1, "VAD frame of %lld samples (sample_rate*update_ms/1000)", (long long)frame_len);
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
    const int64_t Tmax = stft_frames(max_len, n_fft, hop, 1);
    const int64_t nv_max = vad_guided ? (ceil_div(max_len, frame_len) < 1 ? 1 : ceil_div(max_len, frame_len)) : 1;
    const int64_t Ts = row_stride(Tmax);
    // (the grouped layout of the rows pads the bins: slightly more per signal than F * Ts)
    const int64_t sig_floats = std::max<int64_t>((int64_t)F * Ts, rows_wave_sig_floats(Tmax));
    const size_t nS = (size_t)n_sig * sig_floats, nR = (size_t)n_sig * F;
    ADN_TRY(ws_reserve(c, ws_need(nS, sizeof(T)) + ws_need(n_sig, 8) + 2 * ws_need(nR, 8) + ws_need(nR, sizeof(T)) +
                              ws_need((size_t)n_sig * nv_max, 1) + ws_need((size_t)n_sig * Ts, 1) + ws_need(n_sig, 4) +
                              2 * select_ws_bytes(n_sig * F, Tmax) + 2 * APPLY_WS_BYTES + ws_need((size_t)n_sig * nv_max, 8) +
                              2 * select_rows_stream_ws_bytes(c, n_sig, F, Tmax, sizeof(T))));
    T *S = ws_take<T>(c, nS);
    double *vad_e = ws_take<double>(c, (size_t)n_sig * nv_max);
    int64_t *Tn = ws_take<int64_t>(c, n_sig);
    double *p10 = ws_take<double>(c, nR), *sraw = ws_take<double>(c, nR);
    T *gain = d_gain ? d_gain : ws_take<T>(c, nR);
    unsigned char *speech = ws_take<unsigned char>(c, (size_t)n_sig * nv_max);
    unsigned char *vad = ws_take<unsigned char>(c, (size_t)n_sig * Ts);   // rows padded like S
    Signals sg{d_off, d_len, n_ch, n_fft, hop, 1};
    ADN_LAUNCH(k_frame_counts, dim3((unsigned)ceil_div(n_sig, 256)), dim3(256), 0, c->stream, sg, n_sig, Tn);
    SelectArgs sa;
    // rows of up to a few thousand frames: power rows in groups of four frames, written by waves that
    // work alone and read by the register-resident row select; the rows kernel adds up the voice-activity
    // frames' energies on the way when those frames are at least a hop long
    const bool grouped = !c->force_generic && (c->ablate & 64) == 0 && rows_wave_ok(sg, n_fft) && Tmax <= select_grouped_limit_stream(c, sizeof(T));
    const bool vad_in_rows = grouped && vad_guided && (c->ablate & 128) == 0 && frame_len >= hop &&
                             frame_len <= rows_wave_vad_max_frame();
    if (vad_guided && !vad_in_rows) {
        ADN_LAUNCH((k_vad_energy<T>), dim3((unsigned)ceil_div(n_sig * nv_max, 4)), dim3(256), 0, c->stream,
                           d_x, sg, n_sig, frame_len, nv_max, speech);
        ADN_LAUNCH(k_vad_map, dim3((unsigned)ceil_div(n_sig * Ts, 256)), dim3(256), 0, c->stream, sg, n_sig,
                           frame_len, nv_max, Ts, speech, vad);
    }
    if (grouped) {
        if (vad_in_rows) ADN_HIP(hipMemsetAsync(vad_e, 0, (size_t)n_sig * nv_max * sizeof(double), c->stream));
        ADN_TRY(rows_wave<T>(c, d_x, sg, tb, n_sig, Tmax, 1, S, sig_floats, vad_in_rows ? vad_e : nullptr, frame_len, nv_max));
        if (vad_in_rows) {
            ADN_LAUNCH(k_vad_flags, dim3((unsigned)ceil_div(n_sig * nv_max, 256)), dim3(256), 0, c->stream, sg, n_sig,
                       frame_len, nv_max, vad_e, speech);
            ADN_LAUNCH(k_vad_map, dim3((unsigned)ceil_div(n_sig * Ts, 256)), dim3(256), 0, c->stream, sg, n_sig,
                       frame_len, nv_max, Ts, speech, vad);
        }
        sa.quad_stride = ROWS_GROUP_STRIDE;
        sa.blk_rows = F;
        sa.blk_sig_stride = sig_floats;
        sa.rows_stream = true;
    } else {
        ADN_TRY(rows_dispatch<T>(c, d_x, sg, tb, n_sig, Ts, 1, S, nullptr));
    }
    sa.n_seg = n_sig * F;
    sa.seg_stride = Ts;
    sa.d_seg_len = Tn;
    sa.len_div = F;
    sa.max_len = Tmax;
    sa.mode = SEL_PERCENTILE;
    sa.q = 10.0;
    sa.d_out = p10;
    // rows that fit one workgroup get the VAD-masked mean from the same pass over the row
    const bool fuse_mean = vad_guided && (grouped || Tmax <= select_short_limit());
    if (fuse_mean) {
        sa.d_mask = vad;
        sa.mask_div = F;
        sa.mask_stride = Ts;
        sa.d_masked_mean = sraw;
    }
    ADN_TRY(launch_select<T>(c, S, sa));
    sa.d_mask = nullptr;
    sa.d_masked_mean = nullptr;
    if (vad_guided && !fuse_mean) {
        ADN_LAUNCH((k_masked_mean<T>), dim3((unsigned)ceil_div(n_sig * F, 4)), dim3(256), 0, c->stream, S, F,
                           Ts, Ts, n_sig * F, Tn, vad, sraw);
    } else if (!vad_guided) {
        sa.mode = SEL_MEDIAN;
        sa.d_out = sraw;
        ADN_TRY(launch_select<T>(c, S, sa));
    }
    ADN_LAUNCH((k_wiener_gain<T>), dim3((unsigned)n_sig), dim3(GAIN_THREADS), 3 * F * sizeof(double), c->stream, p10,
                       sraw, F, kernel_bins, gain);
    ADN_HIP(hipGetLastError());
    ApplyArgs<T> a{};
    a.x = d_x;
    a.y = d_y;
    a.y_off = d_y_off;
    a.sg = sg;
    a.tb = tb;
    a.pcm = d_pcm16;
    a.row = gain;
    a.row_stride = F;
    a.st = 1;
    a.sf = 1;
    return apply_dispatch<T, APPLY_GAIN>(c, a, n_sig, Tmax);
}

// ------------------------------------------------ the Wiener stage in parts
// One recording split into time chunks over several devices (or calls): every holder computes the
// statistics of its own frames with these pieces, the per-bin order statistic is a distributed
// selection over the power rows (select_hist / select_resolve), the masked sums are added up
// outside, and every holder applies the same gain to its chunk.

// |Z| (power = 0) or |Z|^2 (power = 1) of every frame of every signal, transposed: d_S[(s * F + k) * row_stride + t].
template <typename T>
int stft_rows(Ctx *c, const T *d_x, int64_t n_sig, const int64_t *d_off, const int64_t *d_len, int64_t max_len, int n_ch,
              int n_fft, int hop, int padded, int power, T *d_S, int64_t row_stride) {
    ADN_REQUIRE(c && d_x && d_off && d_len && d_S, "null pointer argument");
    ADN_TRY(check_stft_args(n_fft, hop, n_ch));
    if (n_sig <= 0) return ADN_OK;
    if (max_len < n_fft) {
        set_error("window is longer than input signal (n_fft=%d, longest signal=%lld)", n_fft, (long long)max_len);
        return ADN_ERR_SHORT;
    }
    const int64_t Tmax = stft_frames(max_len, n_fft, hop, padded);
    ADN_REQUIRE(row_stride >= Tmax && row_stride % 32 == 0, "row_stride must be a multiple of 32 and at least %lld frames",
                (long long)Tmax);
    DeviceGuard dg(c->device);
    ws_reset(c);
    Tables<T> tb;
    ADN_TRY(get_tables<T>(c, n_fft, &tb));
    Signals sg{d_off, d_len, n_ch, n_fft, hop, padded};
    return rows_dispatch<T>(c, d_x, sg, tb, n_sig, row_stride, power, d_S, nullptr);
}

// Energy VAD flags of STFT frames t0 .. t0 + n_frames - 1 of ONE signal of n_total samples
// (src/wiener_filter.py:12-33, 92-98) from a buffer that holds its samples from `base_sample` on:
// the buffer must reach back to VAD frame min(t0 * hop / frame_len, last) - 5 (hangover) and forward to
// the end of the VAD frame of the last STFT frame (or the end of the signal).
template <typename T>
__global__ __launch_bounds__(256) void k_vad_part_energy(const T *__restrict__ xv, int n_ch, int64_t n_total, int64_t frame_len,
                                                         int64_t i0, int64_t n_vad, unsigned char *__restrict__ speech) {
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (w >= n_vad) return;
    const int64_t i = i0 + w;
    const int64_t s = i * frame_len, e = (s + frame_len < n_total) ? s + frame_len : n_total;
    double acc = 0.0;
    for (int64_t p = s + lane; p < e; p += 64) {
        const double v = (double)fold_sample(xv, p, n_total, n_ch);
        acc += v * v;
    }
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    if (lane == 0) {
        unsigned char f = 0;
        if (e > s) {
            const double rms = sqrt(acc / (double)(e - s) + 1e-12);
            f = 20.0 * log10(rms + 1e-12) > -40.0;
        }
        speech[w] = f;
    }
}

__global__ void k_vad_part_map(int64_t n_total, int64_t frame_len, int hop, int64_t t0, int64_t n_frames, int64_t i0,
                               const unsigned char *__restrict__ speech, unsigned char *__restrict__ vad) {
    const int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (q >= n_frames) return;
    int64_t nv = (n_total + frame_len - 1) / frame_len;
    nv = nv < 1 ? 1 : nv;
    int64_t idx = ((t0 + q) * hop) / frame_len;
    idx = idx > nv - 1 ? nv - 1 : idx;
    unsigned char v = 0;
    for (int64_t j = idx > 5 ? idx - 5 : 0; j <= idx; ++j) v |= speech[j - i0];
    vad[q] = v;
}

template <typename T>
int vad_frames(Ctx *c, const T *d_x, int64_t n_buf, int n_ch, int64_t base_sample, int64_t n_total, int sample_rate,
               double update_ms, int hop, int64_t t0, int64_t n_frames, unsigned char *d_vad) {
    ADN_REQUIRE(c && d_x && d_vad, "null pointer argument");
    ADN_REQUIRE(n_ch >= 1 && hop >= 1 && t0 >= 0 && n_frames >= 0 && base_sample >= 0 && n_total >= 1, "bad size");
    const int64_t frame_len = (int64_t)((double)sample_rate * update_ms / 1000.0);
    ADN_REQUIRE(frame_len >= 1, "VAD frame of %lld samples (sample_rate*update_ms/1000)", (long long)frame_len);
    if (n_frames == 0) return ADN_OK;
    int64_t nv = ceil_div(n_total, frame_len);
    nv = nv < 1 ? 1 : nv;
    int64_t first = (t0 * hop) / frame_len, last = ((t0 + n_frames - 1) * hop) / frame_len;
    first = first > nv - 1 ? nv - 1 : first;
    last = last > nv - 1 ? nv - 1 : last;
    const int64_t i0 = first > 5 ? first - 5 : 0, n_vad = last - i0 + 1;
    const int64_t need_lo = i0 * frame_len, need_hi = ((last + 1) * frame_len < n_total) ? (last + 1) * frame_len : n_total;
    ADN_REQUIRE(base_sample <= need_lo && base_sample + n_buf >= need_hi,
                "the buffer holds samples [%lld, %lld) but the flags need [%lld, %lld)", (long long)base_sample,
                (long long)(base_sample + n_buf), (long long)need_lo, (long long)need_hi);
    DeviceGuard dg(c->device);
    ws_reset(c);
    ADN_TRY(ws_reserve(c, ws_need(n_vad, 1)));
    unsigned char *speech = ws_take<unsigned char>(c, n_vad);
    const T *xv = d_x - base_sample * n_ch;                // sample p of the signal is xv[p * n_ch + channel]
    ADN_LAUNCH((k_vad_part_energy<T>), dim3((unsigned)ceil_div(n_vad, 4)), dim3(256), 0, c->stream, xv, n_ch, n_total,
               frame_len, i0, n_vad, speech);
    ADN_LAUNCH(k_vad_part_map, dim3((unsigned)ceil_div(n_frames, 256)), dim3(256), 0, c->stream, n_total, frame_len, hop, t0,
               n_frames, i0, speech, d_vad);
    ADN_HIP(hipGetLastError());
    return ADN_OK;
}

// d_sum[row] = sum of d_S[row * row_stride + t] over the t < row_len with d_mask[t] != 0 (binary64 sums);
// *d_count = number of such t.
template <typename T>
__global__ __launch_bounds__(256) void k_rows_masked_sum(const T *__restrict__ S, int64_t n_rows, int64_t row_len,
                                                         int64_t row_stride, const unsigned char *__restrict__ mask,
                                                         double *__restrict__ sum, long long *__restrict__ count) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n_rows) return;
    const T *r = S + row * row_stride;
    double acc = 0.0;
    long long cn = 0;
    for (int64_t t = lane; t < row_len; t += 64) {
        const bool on = mask[t] != 0;
        acc += on ? (double)r[t] : 0.0;
        cn += on;
    }
    for (int d = 32; d >= 1; d >>= 1) {
        acc += __shfl_xor(acc, d);
        cn += __shfl_xor(cn, d);
    }
    if (lane == 0) {
        sum[row] = acc;
        if (row == 0) *count = cn;
    }
}

template <typename T>
int rows_masked_sum(Ctx *c, const T *d_S, int64_t n_rows, int64_t row_len, int64_t row_stride, const unsigned char *d_mask,
                    double *d_sum, long long *d_count) {
    ADN_REQUIRE(c && d_S && d_mask && d_sum && d_count, "null pointer argument");
    ADN_REQUIRE(n_rows >= 1 && row_len >= 0 && row_stride >= row_len, "bad size");
    DeviceGuard dg(c->device);
    ADN_LAUNCH((k_rows_masked_sum<T>), dim3((unsigned)ceil_div(n_rows, 4)), dim3(256), 0, c->stream, d_S, n_rows, row_len,
               row_stride, d_mask, d_sum, d_count);
    ADN_HIP(hipGetLastError());
    return ADN_OK;
}

// The per-bin gain from the two per-bin statistics (src/wiener_filter.py:84-86, 108-115, 35-39).
template <typename T>
int wiener_gain(Ctx *c, const double *d_noise_psd, const double *d_speech_psd, int64_t n_sig, int n_bins, int kernel_bins,
                T *d_gain) {
    ADN_REQUIRE(c && d_noise_psd && d_speech_psd && d_gain, "null pointer argument");
    ADN_REQUIRE(kernel_bins >= 1 && kernel_bins <= 31 && (kernel_bins & 1), "median kernel must be odd and <= 31");
    ADN_REQUIRE(n_bins >= 1 && n_bins <= 9 * 256, "bin count out of range");
    if (n_sig <= 0) return ADN_OK;
    DeviceGuard dg(c->device);
    ADN_LAUNCH((k_wiener_gain<T>), dim3((unsigned)n_sig), dim3(GAIN_THREADS), 3 * n_bins * sizeof(double), c->stream, d_noise_psd,
               d_speech_psd, n_bins, kernel_bins, d_gain);
    ADN_HIP(hipGetLastError());
    return ADN_OK;
}

// STFT -> Z * gain -> ISTFT -> clip with gain rows handed in (the apply half of the Wiener stage).
template <typename T>
int stft_apply_gain(Ctx *c, const T *d_x, int64_t n_sig, const int64_t *d_off, const int64_t *d_len, int64_t max_len,
                    int n_ch, int n_fft, int hop, const T *d_gain, int64_t gain_stride, T *d_y, const int64_t *d_y_off,
                    int16_t *d_pcm16) {
    ADN_REQUIRE(c && d_x && d_off && d_len && d_gain && d_y && d_y_off, "null pointer argument");
    ADN_TRY(check_stft_args(n_fft, hop, n_ch));
    if (n_sig <= 0) return ADN_OK;
    if (max_len < n_fft) {
        set_error("window is longer than input signal (n_fft=%d, longest signal=%lld)", n_fft, (long long)max_len);
        return ADN_ERR_SHORT;
    }
    DeviceGuard dg(c->device);
    ws_reset(c);
    ADN_TRY(ws_reserve(c, APPLY_WS_BYTES));
    Tables<T> tb;
    ADN_TRY(get_tables<T>(c, n_fft, &tb));
    ApplyArgs<T> a{};
    a.x = d_x;
    a.y = d_y;
    a.y_off = d_y_off;
    a.sg = Signals{d_off, d_len, n_ch, n_fft, hop, 1};
    a.tb = tb;
    a.pcm = d_pcm16;
    a.row = d_gain;
    a.row_stride = gain_stride;
    a.st = 1;
    a.sf = 1;
    return apply_dispatch<T, APPLY_GAIN>(c, a, n_sig, stft_frames(max_len, n_fft, hop, 1));
}

// ------------------------------------------------ phase-1 analysis summaries
// (src/analyze_audio.py:50-62 summarize_band_energy, :161 noise_profile_time_mean_mag)
struct BandDesc {
    int k_lo[16], n_bins[16];      // first bin and bin count per band
    int64_t bin_prefix[17];        // bins of the bands before band b
    int n_bands;
};

// seg_off / seg_len of the flattened band values: band b of signal s holds n_bins[b] * Tn[s] values
__global__ void k_band_segments(BandDesc bd, const int64_t *__restrict__ Tn, int64_t n_sig, int64_t Tmax,
                                int64_t *seg_off, int64_t *seg_len) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n_sig * bd.n_bands) return;
    const int64_t sig = i / bd.n_bands;
    const int b = (int)(i % bd.n_bands);
    seg_off[i] = (sig * bd.bin_prefix[bd.n_bands] + bd.bin_prefix[b]) * Tmax;
    seg_len[i] = (int64_t)bd.n_bins[b] * Tn[sig];
}

// Copies the band's rows (frames < Tn) into one contiguous segment and accumulates their sum.
template <typename T>
__global__ __launch_bounds__(256) void k_band_gather(const T *__restrict__ S, int F, int64_t Ts, BandDesc bd,
                                                     const int64_t *__restrict__ Tn,
                                                     const int64_t *__restrict__ seg_off, T *__restrict__ B,
                                                     double *__restrict__ sum) {
    const int64_t sig = blockIdx.z;
    const int b = blockIdx.y;
    const int64_t T_ = Tn[sig];
    const int64_t seg = sig * bd.n_bands + b;
    const int64_t total = (int64_t)bd.n_bins[b] * T_;
    double acc = 0.0;
    for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t k = i / T_, t = i % T_;
        const T v = S[(sig * F + bd.k_lo[b] + k) * Ts + t];
        B[seg_off[seg] + i] = v;
        acc += (double)v;
    }
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    if ((threadIdx.x & 63) == 0 && acc != 0.0) atomicAdd(&sum[seg], acc);
}

__global__ void k_band_mean(const double *__restrict__ sum, const int64_t *__restrict__ seg_len, int64_t n,
                            double *__restrict__ mean) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) mean[i] = seg_len[i] > 0 ? sum[i] / (double)seg_len[i] : __longlong_as_double(0x7FF8000000000000ll);
}

// time_mean[sig * Tmax + t] = mean over the F bins of frame t
template <typename T>
__global__ __launch_bounds__(256) void k_time_mean(const T *__restrict__ S, int F, int64_t Ts, int64_t Tmax,
                                                   const int64_t *__restrict__ Tn, T *__restrict__ out) {
    const int64_t sig = blockIdx.y;
    const int64_t t = blockIdx.x * 256ll + threadIdx.x;
    if (t >= Tn[sig]) return;
    // eight rows in flight per thread; the partial sums are combined in a fixed order
    const T *p = S + sig * F * Ts + t;
    double a8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int k = 0;
    for (; k + 8 <= F; k += 8) {
        T v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(k + u) * Ts];
#pragma unroll
        for (int u = 0; u < 8; ++u) a8[u] += (double)v[u];
    }
    for (; k < F; ++k) a8[0] += (double)p[(int64_t)k * Ts];
    const double acc = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
    out[sig * Tmax + t] = (T)(acc / (double)F);
}

template <typename T>
int band_summary(Ctx *c, const T *d_x, int64_t n_sig, const int64_t *d_off, const int64_t *d_len,
                 int64_t max_len, int n_ch, int sample_rate, int n_fft, int hop, int padded, int n_bands,
                 const double *h_lo_hz, const double *h_hi_hz, int n_perc, const double *h_perc,
                 double *d_mean_mag, double *d_perc_db, T *d_time_mean, T *d_mag) {
    ADN_REQUIRE(c && d_x && d_off && d_len, "null pointer argument");
    ADN_REQUIRE(n_bands >= 0 && n_bands <= 16 && n_perc >= 0 && n_perc <= 16, "at most 16 bands and 16 percentiles");
    ADN_REQUIRE(n_bands == 0 || (h_lo_hz && h_hi_hz && d_mean_mag), "band arguments missing");
    ADN_REQUIRE(n_perc == 0 || (h_perc && d_perc_db), "percentile arguments missing");
    ADN_REQUIRE(sample_rate >= 1, "sample_rate must be >= 1");
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
    // bins of a band: lo <= f_k <= hi with f_k = k * (1 / (n_fft * (1 / fs))) as scipy.fft.rfftfreq builds it
    BandDesc bd{};
    bd.n_bands = n_bands;
    const double val = 1.0 / ((double)n_fft * (1.0 / (double)sample_rate));
    int64_t band_bins = 0;
    for (int b = 0; b < n_bands; ++b) {
        int lo = -1, cnt = 0;
        for (int k = 0; k < F; ++k) {
            const double f = (double)k * val;
            if (f >= h_lo_hz[b] && f <= h_hi_hz[b]) {
                if (lo < 0) lo = k;
                ++cnt;
            }
        }
        bd.k_lo[b] = lo < 0 ? 0 : lo;
        bd.n_bins[b] = cnt;
        bd.bin_prefix[b] = band_bins;
        band_bins += cnt;
    }
    bd.bin_prefix[n_bands] = band_bins;
    const int64_t Tmax = stft_frames(max_len, n_fft, hop, padded);
    const int64_t Ts = row_stride(Tmax);
    const size_t nS = (size_t)n_sig * F * Ts, nB = (size_t)n_sig * band_bins * Tmax, nseg = (size_t)n_sig * n_bands;
    int64_t max_seg = 1;
    for (int b = 0; b < n_bands; ++b) max_seg = std::max<int64_t>(max_seg, (int64_t)bd.n_bins[b] * Tmax);
    ADN_TRY(ws_reserve(c, ws_need(nS, sizeof(T)) + ws_need(nB, sizeof(T)) + ws_need(n_sig, 8) + 3 * ws_need(nseg, 8) +
                              (size_t)n_perc * select_ws_bytes((int64_t)nseg, max_seg) + 256));
    T *S = ws_take<T>(c, nS);
    T *B = ws_take<T>(c, nB);
    int64_t *Tn = ws_take<int64_t>(c, n_sig);
    int64_t *seg_off = ws_take<int64_t>(c, nseg), *seg_len = ws_take<int64_t>(c, nseg);
    double *sum = ws_take<double>(c, nseg);
    Signals sg{d_off, d_len, n_ch, n_fft, hop, padded};
    ADN_LAUNCH(k_frame_counts, dim3((unsigned)ceil_div(n_sig, 256)), dim3(256), 0, c->stream, sg, n_sig, Tn);
    ADN_TRY(rows_dispatch<T>(c, d_x, sg, tb, n_sig, Ts, 0, S, nullptr));
    if (d_time_mean)
        ADN_LAUNCH((k_time_mean<T>), dim3((unsigned)ceil_div(Tmax, 256), (unsigned)n_sig), dim3(256), 0, c->stream,
                           S, F, Ts, Tmax, Tn, d_time_mean);
    if (d_mag)      // the magnitude matrix itself (for the host-side plots), rows of T_max frames
        ADN_HIP(hipMemcpy2DAsync(d_mag, (size_t)Tmax * sizeof(T), S, (size_t)Ts * sizeof(T), (size_t)Tmax * sizeof(T),
                                 (size_t)n_sig * F, hipMemcpyDeviceToDevice, c->stream));
    if (n_bands > 0) {
        ADN_HIP(hipMemsetAsync(sum, 0, sizeof(double) * nseg, c->stream));
        ADN_LAUNCH(k_band_segments, dim3((unsigned)ceil_div((int64_t)nseg, 256)), dim3(256), 0, c->stream, bd, Tn,
                           n_sig, Tmax, seg_off, seg_len);
        const unsigned gx = (unsigned)std::min<int64_t>(ceil_div(max_seg, 256 * 8), 1024);
        ADN_LAUNCH((k_band_gather<T>), dim3(gx, (unsigned)n_bands, (unsigned)n_sig), dim3(256), 0, c->stream, S, F,
                           Ts, bd, Tn, seg_off, B, sum);
        ADN_LAUNCH(k_band_mean, dim3((unsigned)ceil_div((int64_t)nseg, 256)), dim3(256), 0, c->stream, sum, seg_len,
                           (int64_t)nseg, d_mean_mag);
        for (int p = 0; p < n_perc; ++p) {
            SelectArgs sa;
            sa.n_seg = (int64_t)nseg;
            sa.d_seg_off = seg_off;
            sa.d_seg_len = seg_len;
            sa.max_len = max_seg;
            sa.mode = SEL_PERCENTILE_DB20;
            sa.q = h_perc[p];
            sa.d_out = d_perc_db + (size_t)p * nseg;
            ADN_TRY(launch_select<T>(c, B, sa));
        }
    }
    ADN_HIP(hipGetLastError());
    return ADN_OK;
}

#include "adn_hpss.hpp"

#define ADN_INST(T)                                                                                    \
    template int noise_psd<T>(Ctx *, const T *, int64_t, const int64_t *, const int64_t *, int64_t,   \
                              int, int, int, int, double, double, int, int, T *);                     \
    template int spectral_subtract<T>(Ctx *, const T *, int64_t, const int64_t *, const int64_t *,    \
                                      int64_t, int, const T *, int64_t, int, int, double, double,     \
                                      int, int, T *, const int64_t *, int16_t *);                     \
    template int wiener<T>(Ctx *, const T *, int64_t, const int64_t *, const int64_t *, int64_t, int, \
                           int, int, int, double, int, int, T *, const int64_t *, T *, int16_t *);    \
    template int hpss_separate<T>(Ctx *, const T *, int64_t, const int64_t *, const int64_t *, int64_t, int, int, \
                                  int, int, int, double, double, double, int, int, const double *,     \
                                  const double *, int, double, double, T *, T *, const int64_t *, double *); \
    template int band_summary<T>(Ctx *, const T *, int64_t, const int64_t *, const int64_t *, int64_t, \
                                 int, int, int, int, int, int, const double *, const double *, int,   \
                                 const double *, double *, double *, T *, T *);                        \
    template int stft_rows<T>(Ctx *, const T *, int64_t, const int64_t *, const int64_t *, int64_t, int, int, int, \
                              int, int, T *, int64_t);                                                 \
    template int vad_frames<T>(Ctx *, const T *, int64_t, int, int64_t, int64_t, int, double, int, int64_t, int64_t, \
                               unsigned char *);                                                       \
    template int rows_masked_sum<T>(Ctx *, const T *, int64_t, int64_t, int64_t, const unsigned char *, double *, \
                                    long long *);                                                      \
    template int wiener_gain<T>(Ctx *, const double *, const double *, int64_t, int, int, T *);        \
    template int stft_apply_gain<T>(Ctx *, const T *, int64_t, const int64_t *, const int64_t *, int64_t, int, int, \
                                    int, const T *, int64_t, T *, const int64_t *, int16_t *);
ADN_INST(float)
ADN_INST(double)

}  // namespace adn
