// Fused tile kernels of the mono (time-axis, periodization) VisuShrink block denoiser.
// Included by adn_dwt.hip (inside namespace adn, after Taps / shrink / BlockGeom / mad_threshold).
//
// One workgroup owns S consecutive samples of one block and keeps TL <= 4 transform levels
// in LDS: pywt.wavedec (denoise.py:77), pywt.threshold (denoise.py:84) and pywt.waverec
// (denoise.py:87) of those levels cost one read of the block for the statistics pass and one
// read + one write for the apply pass.  Levels above TL (level > 4) run on the level-TL
// approximation of the whole block, in one workgroup per block (k_deep).
//
// Index scheme.  For each in-tile level l the kernel holds a CANONICAL CONTIGUOUS RANGE of
// the level's arrays: elements (s + i) mod N_l, i < c.  A tile whose ranges neither wrap nor
// touch the replicated pad sample of an odd-length level ("interior") uses direct indexing;
// every other tile evaluates pywt's periodization formulas with exact modular arithmetic,
// so any block length (odd lengths at any level, blocks shorter than a tile) is covered.
#pragma once

namespace tiles {

constexpr int TL_MAX = 4;
constexpr int THREADS_MAX = 256;
constexpr int TILE_SMALL_TL = 2;

// canonical contiguous range: indices (s + i) mod N for i < c
struct Rg {
    int s, c, N;
};
// s mod N for an s that a tile's halo can only take one period outside [0, N) (any s is still reduced correctly: the
// division is the rare path).  An integer division costs some thirty instructions, a tile's plan needs seven or more
// of them, and a one-wave tile has fewer than a thousand instructions in all.
__device__ __forceinline__ int mod_near(int s, int N) {
    if (s >= -N && s < 2 * N) return s < 0 ? s + N : (s >= N ? s - N : s);
    int m = s % N;
    return m < 0 ? m + N : m;
}
__device__ __forceinline__ Rg rg_make(int s, int c, int N) {
    if (c >= N) return Rg{0, N, N};
    return Rg{mod_near(s, N), c, N};
}
__device__ __forceinline__ int rg_local(const Rg &r, int q) {
    int li = q - r.s;
    if (li < 0) li += r.N;
    return li < r.c ? li : r.c - 1;   // a miss cannot happen (ranges carry margins); stay in bounds anyway
}
__device__ __forceinline__ int rg_canon(const Rg &r, int i) {
    const int q = r.s + i;
    return q >= r.N ? q - r.N : q;
}
__device__ __forceinline__ bool rg_wraps(const Rg &r) { return r.s + r.c > r.N; }

// Unwrapped first / last coefficient index of level l feeding outputs [lo, lo+cnt) of level
// l-1.  One extra index on each side: when the output range wraps around an ODD length the
// coefficient index of the wrapped outputs shifts by one (2 * N_l = N_{l-1} + 1).
template <int F>
__host__ __device__ constexpr int inv_lo(int lo) { return ((lo - F / 2) >> 1) - 1; }
template <int F>
__host__ __device__ constexpr int inv_hi(int lo, int cnt) { return ((lo + cnt - 1 + F / 2 - 1) >> 1) + 1; }

// compile-time upper bounds of the range counts (mirrors make_plan) and the LDS layout
template <int S, int F, int TL>
struct Cap {
    int ic[TL + 2], gc[TL + 2];     // bounds of I_l.c / G_l.c
    int offA[TL + 2], offD[TL + 2]; // element offsets of A_l (l = 0..TL) and D_l (l = 1..TL)
    int total;
    constexpr Cap() : ic{}, gc{}, offA{}, offD{}, total(0) {
        ic[0] = S;
        for (int l = 1; l <= TL; ++l) ic[l] = (ic[l - 1] + F) / 2 + 3;
        gc[TL] = ic[TL];
        for (int l = TL - 1; l >= 0; --l) gc[l] = 2 * (gc[l + 1] - 1) + F + 3;
        // Level 0 (the samples) is dead once level 1 exists, and the arrays of all deeper levels together are
        // shorter than it: they take its place, which lets a third more tiles share a CU.
        int deeper = 0;
        for (int l = 2; l <= TL; ++l) deeper += ((gc[l] + 3) & ~1) + ((ic[l] + 3) & ~1);
        const int sizeA0 = (gc[0] + 3) & ~1;
        const bool alias = TL >= 2 && deeper <= sizeA0;
        int off = 0;
        offA[0] = 0;
        off += sizeA0;
        offD[0] = off;
        if (TL >= 1) {
            offA[1] = off;
            off += (gc[1] + 3) & ~1;    // +1 for the alignment shift, rounded to even
            offD[1] = off;
            off += (ic[1] + 3) & ~1;
        }
        int o2 = alias ? 0 : off;
        for (int l = 2; l <= TL; ++l) {
            offA[l] = o2;
            o2 += (gc[l] + 3) & ~1;
            offD[l] = o2;
            o2 += (ic[l] + 3) & ~1;
        }
        total = alias ? off : o2;
    }
};

template <int TL>
struct Plan {
    int N[TL + 1], M[TL + 1];
    Rg G[TL + 1];      // range of A_l (and of the level-l outputs the forward pass computes)
    Rg I[TL + 1];      // apply: range of R_l / D_l the synthesis needs; stats: owned coefficients
    int own_lo[TL + 1], own_hi[TL + 1];   // coefficients this tile owns (counted / exported once)
    bool interior;
};

template <int F, int TL, int MODE>
__device__ __forceinline__ Plan<TL> make_plan(int n, int n0, int n1, int c0, int c1) {
    Plan<TL> p;
    p.N[0] = n;
#pragma unroll
    for (int l = 0; l < TL; ++l) {
        p.M[l] = p.N[l] + (p.N[l] & 1);
        p.N[l + 1] = p.M[l] >> 1;
    }
    p.M[TL] = p.N[TL] + (p.N[TL] & 1);
    bool interior = true;
    {
        // coefficients above the samples [c0, c1) are this tile's to count and export ([n0, n1) unless the tile
        // borders the wave-resident tiles, whose stored samples sit one position right of their coefficients)
        int lo = c0, hi = c1;
        p.own_lo[0] = c0;
        p.own_hi[0] = c1;
#pragma unroll
        for (int l = 1; l <= TL; ++l) {
            lo >>= 1;
            hi = (hi + 1) >> 1;
            p.own_lo[l] = lo;
            p.own_hi[l] = hi < p.N[l] ? hi : p.N[l];
        }
    }
    if (MODE == MODE_APPLY) {
        int lo = n0, cnt = n1 - n0;
#pragma unroll
        for (int l = 1; l <= TL; ++l) {
            const int klo = inv_lo<F>(lo), khi = inv_hi<F>(lo, cnt);
            p.I[l] = rg_make(klo, khi - klo + 1, p.N[l]);
            interior = interior && klo >= 0 && khi <= p.N[l] - 1;
            lo = klo;
            cnt = khi - klo + 1;
        }
    } else {
#pragma unroll
        for (int l = 1; l <= TL; ++l) {
            const int c = p.own_hi[l] - p.own_lo[l];
            p.I[l] = rg_make(p.own_lo[l], c > 0 ? c : 0, p.N[l]);
        }
    }
    p.I[0] = Rg{0, 0, 1};
    p.G[TL] = p.I[TL];
#pragma unroll
    for (int l = TL - 1; l >= 0; --l) {
        const Rg &g = p.G[l + 1];
        const int lo = 2 * g.s - F / 2 + 1;
        const int cnt = 2 * (g.c - 1) + F + 2;   // +2: the replicated pad sample shifts indices by one
        int lo_c = mod_near(lo, p.M[l]);
        if (lo_c == p.N[l]) lo_c = p.N[l] - 1;   // landed on the pad position
        p.G[l] = rg_make(lo_c - 1, cnt + 1, p.N[l]);
        interior = interior && lo >= 0 && 2 * (g.s + g.c - 1) + F / 2 <= p.N[l] - 1;
    }
#pragma unroll
    for (int l = 0; l <= TL; ++l) interior = interior && !rg_wraps(p.G[l]);
    p.interior = interior;
    return p;
}

template <typename T> struct Vec2T;
template <> struct Vec2T<float> { using type = float2; };
template <> struct Vec2T<double> { using type = double2; };

// One tile.  STATS: writes the owned finest-band details to d1_out (and, when top != null, the
// owned level-TL approximation to top).  APPLY: analysis, shrink, synthesis of TL levels; the
// level-TL approximation comes from `top` when use_top (output of k_deep), else from the
// tile's own analysis.
template <typename T, int F, int TL, int S, int THREADS, int MODE>
__global__ __launch_bounds__(THREADS) void k_tile(const T *__restrict__ x, T *__restrict__ y,
                                                  BlockGeom g, int tiles_per_block,
                                                  TapsF<T, F> tp, const double *__restrict__ thresh,
                                                  int thr_mode, int64_t *__restrict__ zeroed,
                                                  T *__restrict__ d1_out, int64_t d1_stride,
                                                  T *__restrict__ top, int64_t top_stride,
                                                  int use_top, int ablate, WresPlan wp) {
    constexpr Cap<S, F, TL> cap{};
    using V2 = typename Vec2T<T>::type;
    using P = typename Pk<T>::type;
    __shared__ __attribute__((aligned(16))) T lds[cap.total];
    const int tid = threadIdx.x;
    // grid: x = tile of the block, (z, y) = block (no division by the runtime tile count)
    const int64_t b = (int64_t)blockIdx.z * 32768 + blockIdx.y;
    const int tile = (int)blockIdx.x;
    if (b >= g.n_blocks) return;
    const int n = (int)g.frames[b];
    const int n_lim = MODE == MODE_APPLY ? n + (n & 1) : n;
    // samples [n0, n1) are this tile's to produce, the coefficients above [c0, c1) its to count and export
    int n0, n1, c0, c1;
    if (wp.window > 0 && wres_ring(n, wp.window, wp.run)) return;          // the wave-resident tiles cover this block
    const int n_windows = wp.window > 0 && n >= wp.window ? (n - wp.window) / wp.advance + 1 : 0;
    if (n_windows == 0) {
        n0 = tile * S;
        if (n0 >= n_lim) return;
        n1 = n0 + S < n_lim ? n0 + S : n_lim;
        c0 = n0;
        c1 = n1;
    } else {
        // the wave-resident tiles have the middle of the block: this kernel runs on the head [0, lead) and on the
        // tail [wres_tail, n_lim)
        const int head_tiles = (wp.lead + S - 1) / S;
        if (tile < head_tiles) {
            n0 = tile * S;
            n1 = n0 + S < wp.lead ? n0 + S : wp.lead;
        } else {
            n0 = (int)wres_tail(n, wp) + (tile - head_tiles) * S;
            if (n0 >= n_lim) return;
            n1 = n0 + S < n_lim ? n0 + S : n_lim;
        }
        c0 = n0;
        c1 = n1;
    }
    const Plan<TL> p = make_plan<F, TL, MODE>(n, n0, n1, c0, c1);
    const bool interior = p.interior;

    // alignment shift of every forward source array: window starts land on even LDS indices
    int sh[TL + 1];
#pragma unroll
    for (int l = 0; l < TL; ++l) sh[l] = (2 * p.G[l + 1].s + F / 2 - p.G[l].s - F + 1) & 1;
    sh[TL] = 0;

    // ---- level 0 from memory
    {
        const T *in = x + g.in_off[b];
        T *A0 = lds + cap.offA[0] + sh[0];
        const Rg r = p.G[0];
        if (interior) {
            for (int i = tid; i < r.c; i += THREADS) A0[i] = in[r.s + i];
        } else {
            for (int i = tid; i < r.c; i += THREADS) A0[i] = in[rg_canon(r, i)];
        }
    }
    __syncthreads();

    // ---- analysis (apply: details are shrunk as they are produced)
    const T thr = MODE == MODE_APPLY ? (T)thresh[b] : (T)0;
    int nz = 0;
#pragma unroll
    for (int l = 1; l <= TL; ++l) {
        if (ablate & 1) break;
        if (ablate & 2) thr_mode = -1;      // timing only: neither soft nor hard
        const T *src = lds + cap.offA[l - 1] + sh[l - 1];
        T *dA = lds + cap.offA[l] + sh[l];
        T *dD = lds + cap.offD[l];
        const Rg sr = p.G[l - 1], dr = p.G[l], ir = p.I[l];
        const int Ns = p.N[l - 1], Ms = p.M[l - 1];
        if (interior) {
            const int off_lo = 2 * dr.s + F / 2 - sr.s - F + 1;   // window start of output 0
            const int dI = ir.s - dr.s;
            auto put = [&](int i, T sa, T sd) {
                dA[i] = sa;
                if (MODE == MODE_APPLY) {
                    // shrink on the way into LDS; a tile counts the zeroed coefficients it owns
                    const int il = i - dI;
                    if (il >= 0 && il < ir.c) {
                        const int k = dr.s + i;
                        nz += (k >= p.own_lo[l] && k < p.own_hi[l] && fabs(sd) < thr) ? 1 : 0;
                        dD[il] = shrink_fast(sd, thr, thr_mode);
                    }
                } else if (l == 1 && d1_out != nullptr) {
                    const int k = dr.s + i;
                    if (k >= p.own_lo[1] && k < p.own_hi[1]) d1_out[b * d1_stride + k] = sd;
                }
            };
            // A thread takes two neighbouring outputs: their windows share all but two elements, so the pair
            // costs F / 2 + 1 eight-byte LDS reads instead of F.  (The read past the window of a last, single
            // output stays inside the LDS arrays and is not used.)
            for (int i = 2 * tid; i < dr.c; i += 2 * THREADS) {
                const V2 *w = reinterpret_cast<const V2 *>(src + off_lo + 2 * i);
                // window element t pairs with tap F-1-t; pywt's summation order (tap 0 first);
                // both filters advance together in one two-wide fused multiply-add per tap
                T v[F + 2];
#pragma unroll
                for (int t = 0; t < F / 2 + 1; ++t) {
                    const V2 q = w[t];
                    v[2 * t] = q.x;
                    v[2 * t + 1] = q.y;
                }
                P acc0 = {(T)0, (T)0}, acc1 = {(T)0, (T)0};
                if constexpr (sizeof(T) == 4) {
                    // window element e is one half of the pair it was read in: the instruction's operand select
                    // feeds that half to both filters, no register moves
                    P qp[F / 2 + 1];
#pragma unroll
                    for (int t = 0; t < F / 2 + 1; ++t) qp[t] = P{v[2 * t], v[2 * t + 1]};
#pragma unroll
                    for (int j = 0; j < F; ++j) {
                        acc0 = pk_fma_half((F - 1 - j) & 1, tp.hg[j], qp[(F - 1 - j) >> 1], acc0);
                        acc1 = pk_fma_half((F + 1 - j) & 1, tp.hg[j], qp[(F + 1 - j) >> 1], acc1);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < F; ++j) {
                        const P v0 = {v[F - 1 - j], v[F - 1 - j]};
                        const P v1 = {v[F + 1 - j], v[F + 1 - j]};
                        acc0 = __builtin_elementwise_fma(tp.hg[j], v0, acc0);
                        acc1 = __builtin_elementwise_fma(tp.hg[j], v1, acc1);
                    }
                }
                put(i, acc0.x, acc0.y);
                if (i + 1 < dr.c) put(i + 1, acc1.x, acc1.y);
            }
        } else {
            for (int i = tid; i < dr.c; i += THREADS) {
                const int k = rg_canon(dr, i);
                int idx = 2 * k + F / 2;
                while (idx >= Ms) idx -= Ms;   // at most F/2/Ms + 1 rounds
                T sa = 0, sd = 0;
#pragma unroll
                for (int j = 0; j < F; ++j) {
                    const int q = idx < Ns ? idx : Ns - 1;
                    const T v = src[rg_local(sr, q)];
                    sa = fma(tp.h[j], v, sa);           // the interior path's operation order, so that a
                    sd = fma(tp.g[j], v, sd);           // sample's result does not depend on its tile
                    idx = idx == 0 ? Ms - 1 : idx - 1;
                }
                dA[i] = sa;
                if (MODE == MODE_APPLY) {
                    int il = k - ir.s;
                    if (il < 0) il += ir.N;
                    if (il < ir.c) {
                        nz += (k >= p.own_lo[l] && k < p.own_hi[l] && fabs(sd) < thr) ? 1 : 0;
                        dD[il] = shrink_fast(sd, thr, thr_mode);
                    }
                } else if (l == 1 && d1_out != nullptr) {
                    if (k >= p.own_lo[1] && k < p.own_hi[1]) d1_out[b * d1_stride + k] = sd;
                }
            }
        }
        __syncthreads();
    }

    if (MODE == MODE_STATS) {
        if (top != nullptr) {
            const T *aT = lds + cap.offA[TL];
            const Rg r = p.G[TL];
            for (int i = tid; i < r.c; i += THREADS) {
                const int k = rg_canon(r, i);
                if (k >= p.own_lo[TL] && k < p.own_hi[TL]) top[b * top_stride + k] = aT[i];
            }
        }
        return;
    }

    if (use_top) {
        T *aT = lds + cap.offA[TL];
        const Rg r = p.I[TL];
        for (int i = tid; i < r.c; i += THREADS) aT[i] = top[b * top_stride + rg_canon(r, i)];
    }
    if (zeroed != nullptr) {
        for (int d = 32; d >= 1; d >>= 1) nz += __shfl_xor(nz, d);
        if ((tid & 63) == 0 && nz) atomicAdd((unsigned long long *)&zeroed[b], (unsigned long long)nz);
    }
    __syncthreads();

    // ---- synthesis: R_{l-1} from R_l (in A_l's storage, I_l-local) and the shrunk D_l
#pragma unroll
    for (int l = TL; l >= 1; --l) {
        if ((ablate & 4) && l > 1) continue;
        const T *ra = lds + cap.offA[l] + (l == TL ? sh[TL] : 0);
        const T *rd = lds + cap.offD[l];
        const Rg ir = p.I[l];
        const Rg orr = l > 1 ? p.I[l - 1] : Rg{n0, n1 - n0, 2 * p.N[1]};
        T *out = l > 1 ? lds + cap.offA[l - 1] : y + g.out_off[b] + n0;
        const int K = p.N[l];
        // the level-1 output pair (y[2m], y[2m+1]) goes out as one store when that address is aligned
        const bool pair_store = l == 1 && ((orr.s & 1) == 0) &&
                                ((reinterpret_cast<uintptr_t>(out) & (2 * sizeof(T) - 1)) == 0);
        if (interior) {
            const int m0 = orr.s >> 1, m1 = (orr.s + orr.c - 1) >> 1;
            auto put = [&](int m, P acc) {
                const int io = 2 * m - orr.s;
                if (l == 1 && pair_store && io >= 0 && io + 1 < orr.c) {
                    V2 o2;
                    o2.x = acc.x;
                    o2.y = acc.y;
                    *reinterpret_cast<V2 *>(out + io) = o2;
                } else {
                    if (io >= 0 && io < orr.c) out[io] = acc.x;
                    if (io + 1 >= 0 && io + 1 < orr.c) out[io + 1] = acc.y;
                }
            };
            // A thread takes two neighbouring output pairs m, m + 1: they share all but one coefficient of each
            // band.  The first pair starts at an even index of the coefficient arrays (one pair earlier than m0
            // if need be: that pair's outputs lie before the range and are dropped), so that the F / 2 + 2
            // coefficients of a band are F / 4 + 1 eight-byte LDS reads.  Reads one element past either end of the
            // needed coefficients stay inside the LDS arrays and feed dropped outputs only.
            const int ma = m0 - ((m0 - F / 4 - ir.s) & 1);
            for (int m = ma + 2 * tid; m <= m1; m += 2 * THREADS) {
                if (ablate & 8) {
                    const int io = 2 * m - orr.s;
                    if (io >= 0) out[io] = ra[(m - ir.s) & 1023];
                    if (io + 1 < orr.c) out[io + 1] = rd[(m - ir.s) & 1023];
                    continue;
                }
                const int lb = m - F / 4 - ir.s;
                const V2 *ra2 = reinterpret_cast<const V2 *>(ra + lb), *rd2 = reinterpret_cast<const V2 *>(rd + lb);
                T a[F / 2 + 2], d[F / 2 + 2];
#pragma unroll
                for (int q = 0; q < F / 4 + 1; ++q) {
                    const V2 qa = ra2[q], qd = rd2[q];
                    a[2 * q] = qa.x;
                    a[2 * q + 1] = qa.y;
                    d[2 * q] = qd.x;
                    d[2 * q + 1] = qd.y;
                }
                P acc0 = {(T)0, (T)0}, acc1 = {(T)0, (T)0};           // (y[2m], y[2m+1]), (y[2m+2], y[2m+3])
#pragma unroll
                for (int q = 0; q < F / 2; ++q) {
                    const P av0 = {a[F / 2 - 1 - q], a[F / 2 - q]}, av1 = {a[F / 2 - q], a[F / 2 + 1 - q]};
                    const P dv0 = {d[F / 2 - 1 - q], d[F / 2 - q]}, dv1 = {d[F / 2 - q], d[F / 2 + 1 - q]};
                    acc0 = __builtin_elementwise_fma(tp.rlp[q], av0, acc0);
                    acc0 = __builtin_elementwise_fma(tp.rhp[q], dv0, acc0);
                    acc1 = __builtin_elementwise_fma(tp.rlp[q], av1, acc1);
                    acc1 = __builtin_elementwise_fma(tp.rhp[q], dv1, acc1);
                }
                put(m, acc0);
                if (m + 1 <= m1) put(m + 1, acc1);
            }
        } else {
            for (int i = tid; i < orr.c; i += THREADS) {
                const int nn = rg_canon(orr, i);
                const int par = (nn + F / 2 - 1) & 1;
                int k = (nn + F / 2 - 1 - par) >> 1;
                while (k >= K) k -= K;       // at most (F/4)/K + 1 rounds; K >= 1
                T s = 0;
#pragma unroll
                for (int q = 0; q < F / 2; ++q) {
                    const int li = rg_local(ir, k);
                    const T cl = par ? tp.rl[2 * q + 1] : tp.rl[2 * q];
                    const T ch = par ? tp.rh[2 * q + 1] : tp.rh[2 * q];
                    s = fma(cl, ra[li], s);
                    s = fma(ch, rd[li], s);
                    k = k == 0 ? K - 1 : k - 1;
                }
                out[i] = s;
            }
        }
        if (l > 1) __syncthreads();
    }
}

// Levels TL+1 .. L of one block on the level-TL approximation (whole arrays in LDS, exact
// modular indexing): analysis, shrink + zero count, synthesis back to level TL, in place in `top`.
static constexpr int DEEP_THREADS = 1024;   // one workgroup per block and CU: all the waves a CU takes
__host__ __device__ static inline int64_t deep_pad(int64_t n) { return (n + 3) & ~(int64_t)3; }
template <typename T, int F>
__global__ __launch_bounds__(DEEP_THREADS) void k_deep(T *__restrict__ top, int64_t top_stride, BlockGeom g,
                                              int TL, int L, TapsF<T, F> tp,
                                              const double *__restrict__ thresh, int thr_mode,
                                              int64_t *__restrict__ zeroed) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *lds = reinterpret_cast<T *>(smem_raw);
    __shared__ int offA[34], offD[34], NN[34];
    const int tid = threadIdx.x, NT = DEEP_THREADS;
    const int64_t b = blockIdx.x;
    if (tid == 0) {
        int n = (int)g.frames[b];
        for (int l = 0; l < TL; ++l) n = (n + 1) >> 1;
        // A_TL at 0, then A and D of level TL + 1.  A_TL is dead from the analysis of level TL + 2 until the last
        // synthesis step writes it again, and the arrays of all levels beyond TL + 1 together are no longer than it
        // (unless the block is tiny): they take its place, so that two workgroups share a CU.  Every array starts
        // at a multiple of four elements (the 16-byte reads of the paths below).
        const int n1 = (n + 1) >> 1;
        int deeper = 0;
        for (int l = TL + 2, m = n1; l <= L; ++l) {
            m = (m + 1) >> 1;
            deeper += 2 * deep_pad(m);
        }
        NN[TL] = n;
        offA[TL] = 0;
        NN[TL + 1] = n1;
        offA[TL + 1] = deep_pad(n);
        offD[TL + 1] = deep_pad(n) + deep_pad(n1);
        int off = deeper <= n ? 0 : deep_pad(n) + 2 * deep_pad(n1);
        for (int l = TL + 2, m = n1; l <= L; ++l) {
            m = (m + 1) >> 1;
            NN[l] = m;
            offA[l] = off;
            off += deep_pad(m);
            offD[l] = off;
            off += deep_pad(m);
        }
    }
    __syncthreads();
    T *tp_io = top + b * top_stride;
    // 16-byte pieces when the block's array is aligned
    constexpr int VW = 16 / (int)sizeof(T);
    typedef T V16 __attribute__((ext_vector_type(VW)));
    typedef T V2 __attribute__((ext_vector_type(2)));
    const bool vec_io = (reinterpret_cast<uintptr_t>(tp_io) & 15) == 0;
    if (vec_io) {
        const int nv = NN[TL] / VW;
#pragma unroll 4
        for (int i = tid; i < nv; i += NT) reinterpret_cast<V16 *>(lds)[i] = reinterpret_cast<const V16 *>(tp_io)[i];
        for (int i = nv * VW + tid; i < NN[TL]; i += NT) lds[i] = tp_io[i];
    } else {
        for (int i = tid; i < NN[TL]; i += NT) lds[i] = tp_io[i];
    }
    __syncthreads();
    // Away from the two ends of a level no window wraps or meets the repeated last sample, and a thread takes a
    // group of neighbouring outputs whose windows it reads once, in aligned 16-byte (analysis) or two-element
    // (synthesis) pieces: 20 elements for two analysis outputs of sixteen taps instead of 32, ten of each band for
    // four synthesis outputs instead of 64.  Per output the products are added in the order of the exact path.
    // The few outputs at the ends take the exact path, on the threads counted from the last one (those that the
    // last round of groups leaves idle).
    constexpr bool GROUPS = F % 8 == 0;
    for (int l = TL + 1; l <= L; ++l) {
        const T *src = lds + offA[l - 1];
        T *dA = lds + offA[l], *dD = lds + offD[l];
        const int Ns = NN[l - 1], Ms = Ns + (Ns & 1), K = NN[l];
        // group p: outputs 2p and 2p + 1, from src[4p - F/2 .. 4p + F/2 + 3]
        int p_lo = F / 8, p_hi = min((Ns - F / 2 - 4) >> 2, (K - 2) >> 1);
        if (!GROUPS || p_hi < p_lo) {
            p_lo = 0;
            p_hi = -1;
        }
        if constexpr (GROUPS) {
            for (int p = p_lo + tid; p <= p_hi; p += NT) {
                T w[F + 4];
                const V16 *wv = reinterpret_cast<const V16 *>(src + 4 * p - F / 2);
#pragma unroll
                for (int i = 0; i < (F + 4) / VW; ++i) {
                    const V16 v = wv[i];
#pragma unroll
                    for (int e = 0; e < VW; ++e) w[i * VW + e] = v[e];
                }
                T sa0 = 0, sd0 = 0, sa1 = 0, sd1 = 0;
#pragma unroll
                for (int j = 0; j < F; ++j) {
                    sa0 = fma(tp.h[j], w[F - j], sa0);
                    sd0 = fma(tp.g[j], w[F - j], sd0);
                    sa1 = fma(tp.h[j], w[F + 2 - j], sa1);
                    sd1 = fma(tp.g[j], w[F + 2 - j], sd1);
                }
                V2 oa, od;
                oa[0] = sa0;
                oa[1] = sa1;
                od[0] = sd0;
                od[1] = sd1;
                *reinterpret_cast<V2 *>(dA + 2 * p) = oa;
                *reinterpret_cast<V2 *>(dD + 2 * p) = od;
            }
        }
        const int e_lo = 2 * p_lo, e_hi = 2 * (p_hi + 1);
        for (int e = NT - 1 - tid; e < e_lo + K - e_hi; e += NT) {
            const int k = e < e_lo ? e : e_hi + (e - e_lo);
            int idx = 2 * k + F / 2;
            while (idx >= Ms) idx -= Ms;
            T sa = 0, sd = 0;
#pragma unroll
            for (int j = 0; j < F; ++j) {
                const T v = src[idx < Ns ? idx : Ns - 1];
                sa = fma(tp.h[j], v, sa);
                sd = fma(tp.g[j], v, sd);
                idx = idx == 0 ? Ms - 1 : idx - 1;
            }
            dA[k] = sa;
            dD[k] = sd;
        }
        __syncthreads();
    }
    const T t = (T)thresh[b];
    int nz = 0;
    for (int l = TL + 1; l <= L; ++l) {
        T *dD = lds + offD[l];
        for (int k = tid; k < NN[l]; k += NT) {
            const T d = dD[k];
            nz += fabs(d) < t ? 1 : 0;
            dD[k] = shrink(d, t, thr_mode);
        }
    }
    if (zeroed != nullptr) {
        for (int d = 32; d >= 1; d >>= 1) nz += __shfl_xor(nz, d);
        if ((tid & 63) == 0 && nz) atomicAdd((unsigned long long *)&zeroed[b], (unsigned long long)nz);
    }
    __syncthreads();
    for (int l = L; l > TL; --l) {
        const T *ra = lds + offA[l], *rd = lds + offD[l];
        T *out = lds + offA[l - 1];
        const int K = NN[l], n_out = NN[l - 1];   // waverec drops the extra sample of an odd level
        // group m: outputs 4m .. 4m + 3, from coefficients 2m - F/4 .. 2m + F/4 + 1 of either band (output 4m
        // starts at coefficient 2m + F/4 - 1 with the odd taps, 4m + 1 and 4m + 2 at the next one with the even
        // and the odd taps, 4m + 3 at the one after with the even taps)
        int m_lo = F / 8, m_hi = min((K - 2 - F / 4) >> 1, (n_out - 4) >> 2);
        if (!GROUPS || m_hi < m_lo) {
            m_lo = 0;
            m_hi = -1;
        }
        if constexpr (GROUPS) {
            for (int m = m_lo + tid; m <= m_hi; m += NT) {
                T a[F / 2 + 2], d[F / 2 + 2];
                const V2 *pa = reinterpret_cast<const V2 *>(ra + 2 * m - F / 4);
                const V2 *pd = reinterpret_cast<const V2 *>(rd + 2 * m - F / 4);
#pragma unroll
                for (int i = 0; i < F / 4 + 1; ++i) {
                    const V2 va = pa[i], vd = pd[i];
                    a[2 * i] = va[0];
                    a[2 * i + 1] = va[1];
                    d[2 * i] = vd[0];
                    d[2 * i + 1] = vd[1];
                }
                T s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
                for (int q = 0; q < F / 2; ++q) {
                    const int i0 = F / 2 - 1 - q;
                    s0 = fma(tp.rl[2 * q + 1], a[i0], s0);
                    s0 = fma(tp.rh[2 * q + 1], d[i0], s0);
                    s1 = fma(tp.rl[2 * q], a[i0 + 1], s1);
                    s1 = fma(tp.rh[2 * q], d[i0 + 1], s1);
                    s2 = fma(tp.rl[2 * q + 1], a[i0 + 1], s2);
                    s2 = fma(tp.rh[2 * q + 1], d[i0 + 1], s2);
                    s3 = fma(tp.rl[2 * q], a[i0 + 2], s3);
                    s3 = fma(tp.rh[2 * q], d[i0 + 2], s3);
                }
                V2 o0, o1;
                o0[0] = s0;
                o0[1] = s1;
                o1[0] = s2;
                o1[1] = s3;
                reinterpret_cast<V2 *>(out + 4 * m)[0] = o0;
                reinterpret_cast<V2 *>(out + 4 * m)[1] = o1;
            }
        }
        const int e_lo = 4 * m_lo, e_hi = 4 * (m_hi + 1);
        for (int e = NT - 1 - tid; e < e_lo + n_out - e_hi; e += NT) {
            const int nn = e < e_lo ? e : e_hi + (e - e_lo);
            const int par = (nn + F / 2 - 1) & 1;
            int k = (nn + F / 2 - 1 - par) >> 1;
            while (k >= K) k -= K;
            T s = 0;
#pragma unroll
            for (int q = 0; q < F / 2; ++q) {
                const T cl = par ? tp.rl[2 * q + 1] : tp.rl[2 * q];
                const T ch = par ? tp.rh[2 * q + 1] : tp.rh[2 * q];
                s = fma(cl, ra[k], s);
                s = fma(ch, rd[k], s);
                k = k == 0 ? K - 1 : k - 1;
            }
            out[nn] = s;
        }
        __syncthreads();
    }
    if (vec_io) {
        const int nv = NN[TL] / VW;
#pragma unroll 4
        for (int i = tid; i < nv; i += NT) reinterpret_cast<V16 *>(tp_io)[i] = reinterpret_cast<const V16 *>(lds)[i];
        for (int i = nv * VW + tid; i < NN[TL]; i += NT) tp_io[i] = lds[i];
    } else {
        for (int i = tid; i < NN[TL]; i += NT) tp_io[i] = lds[i];
    }
}

// Samples and threads of a tile.  Up to two levels in the tile: one wave per tile of about 1024 (binary32) samples,
// so that the barriers between the levels cost nothing and a CU holds a dozen independent tiles; the halo a tile
// recomputes grows with 2^levels, so deeper tiles stay large (about 4096 samples, four waves).
// Four-wave APPLY tiles: every per-level loop hands its outputs (the owned ones plus the halo) to the threads in
// whole rounds, so a count just above a multiple of a round costs a nearly empty round: 4096 samples at four
// levels and sixteen taps take 22 rounds of the workgroup, 3584 samples 17.  Their sizes are those for which the
// rounds of all levels are (nearly) full.  (One-wave tiles were measured at 448, 960 and 1472 samples as well, with
// fewer rounds per sample in every case, and ran 15 %, 12 % and 20 % slower than at 1024; two-wave tiles of 1920
// samples 25 % slower.)
template <typename T, int F, int TL, int MODE> struct TileS {
    static constexpr bool SMALL = TL <= TILE_SMALL_TL;
    static constexpr int THREADS = SMALL ? 64 : 256;
    static constexpr int S32 = SMALL || MODE != MODE_APPLY ? (SMALL ? 1024 : 4096) : F == 8 ? (TL == 3 ? 3904 : 3776) : (TL == 3 ? 3840 : 3584);
    static constexpr int S64 = SMALL || MODE != MODE_APPLY ? (SMALL ? 512 : 2048) : F == 8 ? (TL == 3 ? 1888 : 1728) : (TL == 3 ? 1792 : 2016);
    static constexpr int S = sizeof(T) == 4 ? S32 : S64;
};
static constexpr int TILE_MIN_S = 512;      // the smallest of the sizes above

template <typename T, int F, int TL, int MODE>
static void launch_tile(Ctx *c, const T *x, T *y, const BlockGeom &g, int64_t max_frames, int wavelet,
                        const double *thresh, int thr_mode, int64_t *zeroed, T *d1, int64_t d1_stride,
                        T *top, int64_t top_stride, int use_top, WresPlan wp) {
    constexpr int S = TileS<T, F, TL, MODE>::S, THREADS = TileS<T, F, TL, MODE>::THREADS;
    int tiles = (int)ceil_div(max_frames + 1, S);
    if (wp.window > 0 && max_frames >= wp.window) {
        // blocks that hold a window: head and tail only; shorter blocks: all of them
        const int short_tiles = (int)ceil_div(wp.window + 1, S);
        const int edge_tiles = (int)(ceil_div(wp.lead, S) + ceil_div(wp.window - wp.lead - wp.advance + wp.run + 2, S));
        tiles = short_tiles > edge_tiles ? short_tiles : edge_tiles;
    }
    const TapsF<T, F> tp = make_taps_f<T, F>(wavelet);
    const int64_t by = g.n_blocks < 32768 ? g.n_blocks : 32768;
    ADN_LAUNCH((k_tile<T, F, TL, S, THREADS, MODE>), dim3((unsigned)tiles, (unsigned)by, (unsigned)ceil_div(g.n_blocks, by)), dim3(THREADS),
                       0, c->stream, x, y, g, tiles, tp, thresh, thr_mode, zeroed, d1, d1_stride, top,
                       top_stride, use_top, c->ablate, wp);
}

template <typename T, int F, int MODE>
static void launch_tile_tl(Ctx *c, int TL, const T *x, T *y, const BlockGeom &g, int64_t max_frames,
                           int wavelet, const double *thresh, int thr_mode, int64_t *zeroed, T *d1,
                           int64_t d1_stride, T *top, int64_t top_stride, int use_top, const WresPlan *given = nullptr) {
    // binary32: the wave-resident tiles take every window that fits its block, the LDS tiles what is left
    // (given: the windows have been launched by the caller)
    WresPlan wp{0, 0, 0, 1};
    if (given != nullptr) wp = *given;
    else {
        if (!(c->ablate & 4096))
            wres_tiles(c, F, TL < 1 ? 1 : (TL > 4 ? 4 : TL), MODE, x, y, g, max_frames, wavelet, thresh, thr_mode, zeroed, d1, d1_stride, top,
                       top_stride, use_top, &wp);
    }
    switch (TL) {
        case 1: launch_tile<T, F, 1, MODE>(c, x, y, g, max_frames, wavelet, thresh, thr_mode, zeroed, d1, d1_stride, top, top_stride, use_top, wp); break;
        case 2: launch_tile<T, F, 2, MODE>(c, x, y, g, max_frames, wavelet, thresh, thr_mode, zeroed, d1, d1_stride, top, top_stride, use_top, wp); break;
        case 3: launch_tile<T, F, 3, MODE>(c, x, y, g, max_frames, wavelet, thresh, thr_mode, zeroed, d1, d1_stride, top, top_stride, use_top, wp); break;
        default: launch_tile<T, F, 4, MODE>(c, x, y, g, max_frames, wavelet, thresh, thr_mode, zeroed, d1, d1_stride, top, top_stride, use_top, wp); break;
    }
}

// LDS bytes k_deep needs for the longest block (its layout, see the kernel; tiny blocks do not fold the deeper
// levels into the level-TL array and may need more than a longer block: a floor covers them)
template <typename T>
static size_t deep_lds_bytes(int64_t max_frames, int TL, int L) {
    int64_t n = max_frames;
    for (int l = 0; l < TL; ++l) n = (n + 1) / 2;
    const int64_t n1 = (n + 1) / 2;
    int64_t deeper = 0;
    for (int64_t l = TL + 2, m = n1; l <= L; ++l) {
        m = (m + 1) / 2;
        deeper += 2 * deep_pad(m);
    }
    size_t total = (size_t)(deep_pad(n) + 2 * deep_pad(n1) + (deeper <= n ? 0 : deeper));
    // a block that does not fold has fewer than 16 (L - TL) + 2 elements at level TL and needs at most 3.5 times that
    if (total < 2048) total = 2048;
    return total * sizeof(T) + 16;
}

template <typename T, int F>
static int denoise_f(Ctx *c, const T *d_in, T *d_out, BlockGeom g, int64_t max_frames, int wavelet,
                     int level, int thr_mode, double *d_sigma, double *d_thresh, int64_t *d_zeroed,
                     const double *d_thresh_given, bool *handled) {
    const int TL = level < TL_MAX ? level : TL_MAX;
    const bool deep = level > TL;
    size_t deep_lds = 0;
    if (deep) {
        deep_lds = deep_lds_bytes<T>(max_frames, TL, level);
        if (deep_lds > 158 * 1024) return ADN_OK;   // the level-TL approximation does not fit one CU's LDS
    }
    const int64_t d1_stride = (max_frames + 1) / 2 + 1;
    int64_t top_stride = max_frames;
    for (int l = 0; l < TL; ++l) top_stride = (top_stride + 1) / 2;
    top_stride += 1;
    g.d_stride = d1_stride;
    double *thresh = const_cast<double *>(d_thresh_given);
    // Blocks whose levels all live in the tiles: the threshold statistics come from one pass over the samples
    // (finest-band coefficients formed on the fly, bracket candidates gathered), no coefficient array.  The
    // small bracket sample keeps the per-block sampling cheap; its longer candidate lists must fit LDS.
    constexpr int S_SMALL = 4096, S_LARGE = 16384;
    auto no_windows = [](const WresPlan &) -> int { return ADN_OK; };
    // Deeper transforms: the same statistics, their gather pass fused into the wave-resident windows that
    // write the level-TL approximation for the deep levels (no finest band in memory, one pass over the samples for
    // both); the LDS tiles write the approximation of what the windows leave, the direct gather takes those blocks.
    const bool fused = !(c->ablate & 4096);
    if (deep && fused && !d_thresh_given && !(c->ablate & 512) && max_frames >= 4096) {
        const bool small = mono_single_pass_applicable<T, F, S_SMALL>(c, 0, max_frames);
        const bool large = !small && mono_single_pass_applicable<T, F, S_LARGE>(c, g.n_blocks, max_frames);
        if (small || large) {
            const TapsF<T, F> tp = make_taps_f<T, F>(wavelet);
            const size_t nT = (size_t)g.n_blocks * top_stride;
            ADN_TRY(ws_reserve(c, ws_need(nT, sizeof(T)) + (small ? mono_single_pass_ws_bytes<T, F, S_SMALL>(g.n_blocks, max_frames)
                                                                   : mono_single_pass_ws_bytes<T, F, S_LARGE>(g.n_blocks, max_frames))));
            T *top = ws_take<T>(c, nT);
            auto rest_of_top = [&](const WresPlan &wp) -> int {
                launch_tile_tl<T, F, MODE_STATS>(c, TL, d_in, (T *)nullptr, g, max_frames, wavelet, nullptr, thr_mode, nullptr, (T *)nullptr,
                                                 d1_stride, top, top_stride, 0, &wp);
                return ADN_OK;
            };
            if (small)
                ADN_TRY((mono_mad_single_pass<T, F, S_SMALL>(c, d_in, g, max_frames, tp.g, d_sigma, d_thresh, &thresh, wavelet, TL, top,
                                                             top_stride, rest_of_top)));
            else
                ADN_TRY((mono_mad_single_pass<T, F, S_LARGE>(c, d_in, g, max_frames, tp.g, d_sigma, d_thresh, &thresh, wavelet, TL, top,
                                                             top_stride, rest_of_top)));
            if (deep_lds > 48 * 1024)
                ADN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_deep<T, F>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)deep_lds));
            ADN_LAUNCH((k_deep<T, F>), dim3((unsigned)g.n_blocks), dim3(DEEP_THREADS), deep_lds, c->stream, top, top_stride, g, TL, level, tp,
                       thresh, thr_mode, d_zeroed);
            launch_tile_tl<T, F, MODE_APPLY>(c, TL, d_in, d_out, g, max_frames, wavelet, thresh, thr_mode, d_zeroed, (T *)nullptr, 0, top,
                                             top_stride, 1);
            ADN_HIP(hipGetLastError());
            *handled = true;
            return ADN_OK;
        }
    }
    const bool want_single = !deep && !d_thresh_given && !(c->ablate & 512);
    const bool single_small = want_single && mono_single_pass_applicable<T, F, S_SMALL>(c, 0, max_frames);
    const bool single_large = want_single && !single_small && mono_single_pass_applicable<T, F, S_LARGE>(c, g.n_blocks, max_frames);
    if (single_small || single_large) {
        const TapsF<T, F> tp = make_taps_f<T, F>(wavelet);
        if (single_small) {
            ADN_TRY(ws_reserve(c, mono_single_pass_ws_bytes<T, F, S_SMALL>(g.n_blocks, max_frames)));
            ADN_TRY((mono_mad_single_pass<T, F, S_SMALL>(c, d_in, g, max_frames, tp.g, d_sigma, d_thresh, &thresh, wavelet, 0, (T *)nullptr, 0, no_windows)));
        } else {
            ADN_TRY(ws_reserve(c, mono_single_pass_ws_bytes<T, F, S_LARGE>(g.n_blocks, max_frames)));
            ADN_TRY((mono_mad_single_pass<T, F, S_LARGE>(c, d_in, g, max_frames, tp.g, d_sigma, d_thresh, &thresh, wavelet, 0, (T *)nullptr, 0, no_windows)));
        }
        launch_tile_tl<T, F, MODE_APPLY>(c, TL, d_in, d_out, g, max_frames, wavelet, thresh, thr_mode, d_zeroed,
                                         (T *)nullptr, 0, (T *)nullptr, 0, 0);
        ADN_HIP(hipGetLastError());
        *handled = true;
        return ADN_OK;
    }
    const size_t nD = (size_t)g.n_blocks * d1_stride, nT = deep ? (size_t)g.n_blocks * top_stride : 0;
    ADN_TRY(ws_reserve(c, ws_need(nD, sizeof(T)) + ws_need(nT, sizeof(T)) +
                              mad_ws_bytes(g.n_blocks, (max_frames + 1) / 2)));
    T *D1 = ws_take<T>(c, nD);
    T *top = deep ? ws_take<T>(c, nT) : nullptr;
    // thresholds handed in by the caller (a block shared between devices): no statistics; the first
    // pass still runs when the deep levels need the level-TL approximation it writes
    if (deep)
        launch_tile_tl<T, F, MODE_STATS>(c, TL, d_in, (T *)nullptr, g, max_frames, wavelet, nullptr, thr_mode,
                                         nullptr, D1, d1_stride, top, top_stride, 0);
    else if (!d_thresh_given)
        launch_tile_tl<T, F, MODE_STATS>(c, 1, d_in, (T *)nullptr, g, max_frames, wavelet, nullptr, thr_mode,
                                         nullptr, D1, d1_stride, (T *)nullptr, 0, 0);
    if (!d_thresh_given)
        ADN_TRY(mad_threshold<T>(c, D1, g, F, ADN_EXT_PERIODIZATION, 0, (max_frames + 1) / 2, d_sigma, d_thresh,
                                 &thresh));
    if (deep) {
        const TapsF<T, F> tp = make_taps_f<T, F>(wavelet);
        if (deep_lds > 48 * 1024)
            ADN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_deep<T, F>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)deep_lds));
        ADN_LAUNCH((k_deep<T, F>), dim3((unsigned)g.n_blocks), dim3(DEEP_THREADS), deep_lds, c->stream, top,
                           top_stride, g, TL, level, tp, thresh, thr_mode, d_zeroed);
    }
    launch_tile_tl<T, F, MODE_APPLY>(c, TL, d_in, d_out, g, max_frames, wavelet, thresh, thr_mode, d_zeroed,
                                     (T *)nullptr, 0, top, top_stride, deep ? 1 : 0);
    ADN_HIP(hipGetLastError());
    *handled = true;
    return ADN_OK;
}

}  // namespace tiles

// Mono blocks through the tile kernels.  *handled stays false when the batch is outside
// their limits (block length >= 2^30, or level > 4 with a level-4 approximation too long
// for one CU's LDS); the caller then takes the level-by-level path.
template <typename T>
static int wavelet_denoise_fused(Ctx *c, const T *d_in, T *d_out, BlockGeom g, int64_t max_frames,
                                 int wavelet, int level, int thr_mode, double *d_sigma,
                                 double *d_thresh, int64_t *d_zeroed, const double *d_thresh_given, bool *handled) {
    *handled = false;
    if (c->force_generic || max_frames >= (int64_t(1) << 30)) return ADN_OK;
    const int64_t tiles = ceil_div(max_frames + 1, (int64_t)tiles::TILE_MIN_S);       // the smallest tile
    if (tiles * g.n_blocks >= (int64_t)0x7FFFFFFF) return ADN_OK;
    if (filter_bank(wavelet).F == 8)
        return tiles::denoise_f<T, 8>(c, d_in, d_out, g, max_frames, wavelet, level, thr_mode, d_sigma,
                                      d_thresh, d_zeroed, d_thresh_given, handled);
    return tiles::denoise_f<T, 16>(c, d_in, d_out, g, max_frames, wavelet, level, thr_mode, d_sigma,
                                   d_thresh, d_zeroed, d_thresh_given, handled);
}
