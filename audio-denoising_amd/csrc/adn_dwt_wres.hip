// Wave-resident tiles of the mono (time-axis, periodization) VisuShrink block denoiser, both precisions, up to four
// levels (deeper transforms run their first four levels here).
//
// pywt.wavedec (denoise.py:77), pywt.threshold (denoise.py:84) and pywt.waverec (denoise.py:87) of up to four
// levels for the part of a block that lies away from its ends (all of it, when its length is a multiple of a lane's
// run).  One wavefront owns a window of 64 * C consecutive samples; lane l holds samples [l C, (l + 1) C) and every
// coefficient of every level that sits above them, all in registers.  There is no LDS array, no barrier and no loop:
// the code of a tile is one straight line of two-wide fused multiply-adds whose operands are selected at compile
// time, and what a lane needs from beyond its own run (F / 2 - 1 values on either side for an analysis level, F / 4
// for a synthesis level) comes from the neighbouring lanes through whole-wave shifts (the adjacent ones; binary64
// holds 32 samples per lane at most, and sixteen taps at four levels then reach two lanes away at the deepest levels).
//
// Lane 0 has no left neighbour and lane 63 no right one: what they receive instead is wrong, and the error spreads
// by the filter length per level.  CONE is the number of output samples at either end of the window that can be
// affected; lanes [LV0, LV1) are clean in full and are the ones that store.  Successive windows of a block advance
// by A = (LV1 - LV0) C samples, so the clean lanes tile the block.  A block whose length is a multiple of C is
// periodic at every level without pywt's repeated last sample: its windows go round it.  Any other block gets the
// windows that fit and one more that ends within a run of its end; its head and tail (less than a window in all),
// and blocks shorter than a window, are left to the LDS tile kernel k_tile, which is launched on those ranges only.
// Results are bit-identical to k_tile's: the same products are added in the same order (tap 0 first in the analysis,
// pywt's order in the synthesis).
//
// Index rules (periodization, away from the ends; N even or odd makes no difference there):
//   analysis   a_l[k], d_l[k] = sum_j (h[j], g[j]) * a_{l-1}[2k + F/2 - j],               j = 0 .. F-1
//   synthesis  a_{l-1}[2m+1]  = sum_q rl[2q]   a_l[m + F/4 - q] + rh[2q]   d_l[m + F/4 - q], q = 0 .. F/2-1
//              a_{l-1}[2m+2]  = sum_q rl[2q+1] a_l[m + F/4 - q] + rh[2q+1] d_l[m + F/4 - q]
// The two synthesis outputs that share their coefficients are an odd sample and the NEXT even one, so between the
// levels a lane's synthesis results sit one position to the right of its analysis run; the last level pairs an even
// sample with the next odd one instead (operand pairs put together from neighbouring coefficients) and ends in place.
#include "adn_dwt_shared.hpp"

namespace adn {
namespace wres {

using tiles::MODE_APPLY;
using tiles::MODE_STATS;
enum { MODE_GATHER = 2 };       // statistics mode with the finest details classified against the blocks' brackets instead of stored
using tiles::TapsF;

// lane l receives lane l - 1's value (lane 0: zero)
__device__ __forceinline__ float lane_prev(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));     // wave_shr:1
}
// lane l receives lane l + 1's value (lane 63: zero)
__device__ __forceinline__ float lane_next(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));     // wave_shl:1
}
// the same for binary64: the two halves of the value
__device__ __forceinline__ double lane_prev(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned int lo = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(unsigned int)u, 0x138, 0xf, 0xf, true);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(unsigned int)(u >> 32), 0x138, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double lane_next(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned int lo = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(unsigned int)u, 0x130, 0xf, 0xf, true);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(unsigned int)(u >> 32), 0x130, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// orders a wave's LDS writes before its own later reads (no other wave touches the slot)
__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// acc + coef * (q.half, q.half)
template <typename P>
__device__ __forceinline__ P fma_half(int half, P coef, P q, P acc) { return tiles::pk_fma_half(half, coef, q, acc); }

template <int F, int TL, int C>
struct WresGeom {
    static constexpr int W = 64 * C;
    static constexpr int cone() {
        int dinv[TL + 1] = {};
        int inv = 0;
        for (int l = 1; l <= TL; ++l) {
            inv = (inv + F / 2) / 2;            // ceil((inv + F/2 - 1) / 2): coefficients of level l that see beyond the window
            dinv[l] = inv;
        }
        int s = dinv[TL];
        for (int l = TL; l >= 1; --l) {
            const int m = s > dinv[l] ? s : dinv[l];
            s = 2 * m + F / 2 - 1;
        }
        return s;
    }
    static constexpr int CONE = cone();
    static constexpr int LV0 = (CONE + C - 1) / C;         // lane l stores samples [l C, l C + C)
    static constexpr int LV1 = (W - CONE) / C;
    static constexpr int A = (LV1 - LV0) * C;
    static constexpr int LEAD = LV0 * C;                   // first sample a block's first window stores
    static_assert(C % (1 << TL) == 0 && 2 * (C >> (TL - 1)) >= F / 2 && 2 * (C >> TL) >= F / 4 && C >= F / 2,
                  "halos come from the two lanes on either side at most");
    static_assert(LV1 > LV0, "no clean lane");
};

template <typename T, int F, int TL, int C, int MODE>
struct Wres {
    typedef WresGeom<F, TL, C> G;
    typedef typename tiles::Pk<T>::type P;
    static constexpr int Q = F / 4;

    struct St {
        P s0[C / 2];               // samples (x[2p], x[2p+1])
        P lev[TL + 1][C / 2];      // lev[l][i] = (a_l[i], d_l[i]), i < C >> l: as the analysis leaves them
        // what stays alive through the turn: the details of every level (shrunk) and the last approximation, two to a
        // pair -- half the registers of the (a, d) pairs, whose approximation halves are dead once the next level exists
        P dp[TL + 1][C / 4];       // dp[l][p] = (d_l[2p], d_l[2p+1])
        P ap[C / 4];               // (a_TL[2p], a_TL[2p+1])
    };
    // element i of an array packed two to a pair
    static __device__ __forceinline__ T at(const P *v, int i) { return (i & 1) ? v[i >> 1].y : v[i >> 1].x; }
    // the value of the lane d (1 or 2) lanes before / after this one
    static __device__ __forceinline__ T before(T v, int d) {
        v = lane_prev(v);
        return d >= 2 ? lane_prev(v) : v;
    }
    static __device__ __forceinline__ T after(T v, int d) {
        v = lane_next(v);
        return d >= 2 ? lane_next(v) : v;
    }
    // Element idx of an array of c per lane that continues over the lanes (idx < 0: in a lane before, idx >= c: after).
    // elem(e) returns this lane's element e.
    template <typename Elem>
    static __device__ __forceinline__ T across(int idx, int c, Elem elem) {
        if (idx < 0) {
            const int d = (-idx + c - 1) / c;
            return before(elem(idx + d * c), d);
        }
        if (idx >= c) {
            const int d = idx / c;
            return after(elem(idx - d * c), d);
        }
        return elem(idx);
    }

    // ---- analysis of level L from level L - 1
    template <int L>
    static __device__ __forceinline__ void ana(St &s, const TapsF<T, F> &tp) {
        constexpr int cs = C >> (L - 1), c = C >> L;
        if constexpr (L == 1) {
            // neighbours' samples, packed as they are held: Q pairs on either side (one sample more than needed)
            P hl[Q], hr[Q];
#pragma unroll
            for (int h = 0; h < Q; ++h) {
                hl[h] = P{lane_prev(s.s0[C / 2 - Q + h].x), lane_prev(s.s0[C / 2 - Q + h].y)};
                hr[h] = P{lane_next(s.s0[h].x), lane_next(s.s0[h].y)};
            }
#pragma unroll
            for (int i = 0; i < c; ++i) {
                P acc = {(T)0, (T)0};
#pragma unroll
                for (int j = 0; j < F; ++j) {
                    const int idx = 2 * i + F / 2 - j;                    // sample relative to the lane's first
                    const int pi = ((idx + 2 * Q) >> 1) - Q, half = idx & 1;
                    const P q = pi < 0 ? hl[pi + Q] : (pi >= C / 2 ? hr[pi - C / 2] : s.s0[pi]);
                    acc = fma_half(half, tp.hg[j], q, acc);
                }
                s.lev[1][i] = acc;
            }
        } else {
            // neighbours' approximation coefficients (the low halves of their pairs), packed two to a pair:
            // slot idx + F/2 on the left (idx = -F/2+1 .. -1), slot idx - cs on the right (idx = cs .. cs + F/2 - 2)
            constexpr int HV = F / 2;
            P hl[HV / 2], hr[HV / 2];
            auto a_prev = [&](int e) { return s.lev[L - 1][e].x; };
#pragma unroll
            for (int sl = 0; sl < HV; ++sl) {
                const T vl = sl >= 1 ? across(sl - HV, cs, a_prev) : (T)0;
                const T vr = sl <= HV - 2 ? across(cs + sl, cs, a_prev) : (T)0;
                if (sl & 1) {
                    hl[sl >> 1].y = vl;
                    hr[sl >> 1].y = vr;
                } else {
                    hl[sl >> 1].x = vl;
                    hr[sl >> 1].x = vr;
                }
            }
#pragma unroll
            for (int i = 0; i < c; ++i) {
                P acc = {(T)0, (T)0};
#pragma unroll
                for (int j = 0; j < F; ++j) {
                    const int idx = 2 * i + F / 2 - j;
                    if (idx < 0) acc = fma_half((idx + HV) & 1, tp.hg[j], hl[(idx + HV) >> 1], acc);
                    else if (idx >= cs) acc = fma_half((idx - cs) & 1, tp.hg[j], hr[(idx - cs) >> 1], acc);
                    else acc = fma_half(0, tp.hg[j], s.lev[L - 1][idx], acc);
                }
                s.lev[L][i] = acc;
            }
        }
    }
    // ---- the details of level L, shrunk (KIND 1: soft, 2: hard, 0: as they are), into their packed array; returns
    // how many fall below the threshold
    template <int L, int KIND>
    static __device__ __forceinline__ int pack(St &s, T thr) {
        constexpr int c = C >> L;
        int nz = 0;
#pragma unroll
        for (int p = 0; p < c / 2; ++p) {
            T d0 = s.lev[L][2 * p].y, d1 = s.lev[L][2 * p + 1].y;
            if (KIND != 0) {
                nz += (fabs(d0) < thr ? 1 : 0) + (fabs(d1) < thr ? 1 : 0);
                d0 = tiles::shrink_fast(d0, thr, KIND == 1 ? ADN_THR_SOFT : ADN_THR_HARD);
                d1 = tiles::shrink_fast(d1, thr, KIND == 1 ? ADN_THR_SOFT : ADN_THR_HARD);
            }
            s.dp[L][p] = P{d0, d1};
        }
        if constexpr (L == TL) {
#pragma unroll
            for (int p = 0; p < c / 2; ++p) s.ap[p] = P{s.lev[L][2 * p].x, s.lev[L][2 * p + 1].x};
        }
        return nz;
    }
    // thr_mode: ADN_THR_SOFT / ADN_THR_HARD, or -1 for the statistics pass (no shrinking)
    template <int L>
    static __device__ __forceinline__ int ana_all(St &s, const TapsF<T, F> &tp, T thr, int thr_mode) {
        if constexpr (L <= TL) {
            ana<L>(s, tp);
            int nz;
            if (MODE != MODE_APPLY) nz = pack<L, 0>(s, thr);
            else if (thr_mode == ADN_THR_SOFT) nz = pack<L, 1>(s, thr);
            else nz = pack<L, 2>(s, thr);
            return nz + ana_all<L + 1>(s, tp, thr, thr_mode);
        } else {
            return 0;
        }
    }

    // ---- synthesis of level L - 1 from level L.  rin: the level-L approximation as left by the previous stage
    // (positions 1 .. c of the lane's run, two to a pair) -- unused for L == TL, whose approximation is ap.
    // rout[m] = (position 2m + 1, position 2m + 2) of level L - 1.
    template <int L>
    static __device__ __forceinline__ void syn(const St &s, const TapsF<T, F> &tp, const P (&rin)[C / 2], P (&rout)[C / 2]) {
        constexpr int c = C >> L;
        // details: slot idx + Q on the left (idx = -Q+1 .. -1), slot idx - c on the right (idx = c .. c + Q - 1)
        P dl[Q / 2], dr[Q / 2], al[Q / 2], ar[Q / 2];
        auto d_own = [&](int e) { return at(s.dp[L], e); };
        auto a_own = [&](int e) { return at(s.ap, e); };
        auto r_own = [&](int e) { return at(rin, e); };        // held index = position - 1
#pragma unroll
        for (int sl = 0; sl < Q; ++sl) {
            const T vl = sl >= 1 ? across(sl - Q, c, d_own) : (T)0;
            const T vr = across(c + sl, c, d_own);
            T wl, wr;
            if constexpr (L == TL) {
                wl = sl >= 1 ? across(sl - Q, c, a_own) : (T)0;
                wr = across(c + sl, c, a_own);
            } else {
                // approximation at positions -Q+1 .. 0 (slot pos + Q - 1) and c+1 .. c+Q-1 (slot pos - c - 1); a lane
                // holds positions 1 .. c, position p at index p - 1
                wl = across(sl - Q, c, r_own);
                wr = sl <= Q - 2 ? across(c + sl, c, r_own) : (T)0;
            }
            if (sl & 1) {
                dl[sl >> 1].y = vl;
                dr[sl >> 1].y = vr;
                al[sl >> 1].y = wl;
                ar[sl >> 1].y = wr;
            } else {
                dl[sl >> 1].x = vl;
                dr[sl >> 1].x = vr;
                al[sl >> 1].x = wl;
                ar[sl >> 1].x = wr;
            }
        }
#pragma unroll
        for (int m = 0; m < c; ++m) {
            P acc = {(T)0, (T)0};
#pragma unroll
            for (int q = 0; q < F / 2; ++q) {
                const int idx = m + Q - q;                     // coefficient relative to the lane's first
                const P rl2 = {tp.rl[2 * q], tp.rl[2 * q + 1]}, rh2 = {tp.rh[2 * q], tp.rh[2 * q + 1]};
                if constexpr (L == TL) {
                    if (idx < 0) acc = fma_half((idx + Q) & 1, rl2, al[(idx + Q) >> 1], acc);
                    else if (idx >= c) acc = fma_half((idx - c) & 1, rl2, ar[(idx - c) >> 1], acc);
                    else acc = fma_half(idx & 1, rl2, s.ap[idx >> 1], acc);
                } else {
                    if (idx <= 0) acc = fma_half((idx + Q - 1) & 1, rl2, al[(idx + Q - 1) >> 1], acc);
                    else if (idx >= c + 1) acc = fma_half((idx - c - 1) & 1, rl2, ar[(idx - c - 1) >> 1], acc);
                    else acc = fma_half((idx - 1) & 1, rl2, rin[(idx - 1) >> 1], acc);
                }
                if (idx < 0) acc = fma_half((idx + Q) & 1, rh2, dl[(idx + Q) >> 1], acc);
                else if (idx >= c) acc = fma_half((idx - c) & 1, rh2, dr[(idx - c) >> 1], acc);
                else acc = fma_half(idx & 1, rh2, s.dp[L][idx >> 1], acc);
            }
            rout[m] = acc;
        }
    }
    // ---- the last synthesis level with its results in place: rout[m] = (sample 2m, sample 2m + 1) of the lane's run.
    //   x[2m]   = sum_q rl[2q+1] a_1[m + F/4 - 1 - q] + rh[2q+1] d_1[m + F/4 - 1 - q]
    //   x[2m+1] = sum_q rl[2q]   a_1[m + F/4 - q]     + rh[2q]   d_1[m + F/4 - q]
    // The two results take neighbouring coefficients, so an operand is a pair (c[j], c[j+1]): every other one is a pair
    // as held, the rest are put together from two registers; one level pays that so that a lane's samples leave at
    // 16-byte aligned addresses.
    static __device__ __forceinline__ void syn_last(const St &s, const TapsF<T, F> &tp, const P (&rin)[C / 2], P (&rout)[C / 2]) {
        constexpr int c = C >> 1;
        T av[c + 2 * Q], dv[c + 2 * Q];               // coefficients -Q .. c + Q - 1 at index + Q
#pragma unroll
        for (int j = 0; j < c; ++j) dv[j + Q] = at(s.dp[1], j);
        auto d_own = [&](int e) { return at(s.dp[1], e); };
#pragma unroll
        for (int h = 0; h < Q; ++h) {
            dv[h] = across(h - Q, c, d_own);
            dv[c + Q + h] = across(c + h, c, d_own);
        }
        if constexpr (TL == 1) {
            auto a_own = [&](int e) { return at(s.ap, e); };
#pragma unroll
            for (int j = 0; j < c; ++j) av[j + Q] = at(s.ap, j);
#pragma unroll
            for (int h = 0; h < Q; ++h) {
                av[h] = across(h - Q, c, a_own);
                av[c + Q + h] = across(c + h, c, a_own);
            }
        } else {
            // rin holds positions 1 .. c: position p is element p - 1
            auto r_own = [&](int e) { return at(rin, e); };
#pragma unroll
            for (int p = 1; p <= c; ++p) av[p + Q] = at(rin, p - 1);
#pragma unroll
            for (int h = 0; h <= Q; ++h) av[h] = across(h - Q - 1, c, r_own);            // positions -Q .. 0
#pragma unroll
            for (int h = 1; h < Q; ++h) av[c + Q + h] = across(c + h - 1, c, r_own);     // positions c + 1 .. c + Q - 1
        }
#pragma unroll
        for (int m = 0; m < c; ++m) {
            P acc = {(T)0, (T)0};
#pragma unroll
            for (int q = 0; q < F / 2; ++q) {
                const int j = m + Q - 1 - q + Q;           // index of the pair's first coefficient in av / dv
                acc = __builtin_elementwise_fma(tp.rlp[q], P{av[j], av[j + 1]}, acc);
                acc = __builtin_elementwise_fma(tp.rhp[q], P{dv[j], dv[j + 1]}, acc);
            }
            rout[m] = acc;
        }
    }
    template <int L>
    static __device__ __forceinline__ void syn_all(const St &s, const TapsF<T, F> &tp, const P (&rin)[C / 2], P (&rout)[C / 2]) {
        if constexpr (L == 1) {
            syn_last(s, tp, rin, rout);
        } else {
            P mid[C / 2];
            syn<L>(s, tp, rin, mid);
            syn_all<L - 1>(s, tp, mid, rout);
        }
    }
};

// 16-byte pieces of T
template <typename T> struct PieceOf;
template <> struct PieceOf<float> { typedef float type __attribute__((ext_vector_type(4))); static constexpr int VW = 4; };
template <> struct PieceOf<double> { typedef double type __attribute__((ext_vector_type(2))); static constexpr int VW = 2; };

template <typename T, int N>
__device__ __forceinline__ void load_run(T (&v)[N], const T *__restrict__ p) {
    typedef typename PieceOf<T>::type V;
    constexpr int VW = PieceOf<T>::VW;
    if constexpr (N % VW == 0) {
#pragma unroll
        for (int i = 0; i < N / VW; ++i) {
            V q;
            __builtin_memcpy(&q, p + VW * i, 16);            // the element's alignment is all the address has
#pragma unroll
            for (int e = 0; e < VW; ++e) v[VW * i + e] = q[e];
        }
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = p[i];
    }
}
template <typename T, int N>
__device__ __forceinline__ void store_run(T *__restrict__ p, const T (&v)[N]) {
    typedef typename PieceOf<T>::type V;
    constexpr int VW = PieceOf<T>::VW;
    if constexpr (N % VW == 0) {
#pragma unroll
        for (int i = 0; i < N / VW; ++i) {
            V q;
#pragma unroll
            for (int e = 0; e < VW; ++e) q[e] = v[VW * i + e];
            __builtin_memcpy(p + VW * i, &q, 16);
        }
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) p[i] = v[i];
    }
}

// The NV values every lane holds for consecutive positions (lane l: positions first + l NV ...) to dst, through the
// wave's LDS slot so that a store instruction writes 1 KB of consecutive memory instead of sixteen bytes out of every
// lane's values: OWN lanes at a time put their values at element (l % OWN) (NV + VW) of the slot (the 16-byte pad keeps
// the 16-byte accesses of neighbouring lanes on different banks), then the k-th instruction writes 16-byte piece
// f = 64 k + lane of those OWN NV values.  Positions at or beyond `wrap` continue at 0 (windows that go round their
// block); lanes outside [lv0, lv1), and those whose first position is before `from` or not before `stop`, hold
// nothing to keep.  dst + first must be 16-byte aligned.
template <typename T, int NV>
struct Staged {
    typedef typename PieceOf<T>::type V;
    static constexpr int VW = PieceOf<T>::VW;
    static constexpr int OWN = NV * (int)sizeof(T) >= 256 ? 32 : 64;
    static constexpr int SLOT = OWN * (NV + VW);           // elements per wave
    static __device__ __forceinline__ void store(const T (&v)[NV], T *slot, T *__restrict__ dst, int64_t first,
                                                 int64_t wrap, int64_t from, int64_t stop, int lv0, int lv1, int lane) {
        V *mine = reinterpret_cast<V *>(slot + (lane % OWN) * (NV + VW));
#pragma unroll
        for (int h = 0; h < 64 / OWN; ++h) {
            if (h > 0) wave_sync_lds();
            if (OWN == 64 || lane / OWN == h) {
#pragma unroll
                for (int k = 0; k < NV / VW; ++k) {
                    V q;
#pragma unroll
                    for (int e = 0; e < VW; ++e) q[e] = v[VW * k + e];
                    mine[k] = q;
                }
            }
            wave_sync_lds();
#pragma unroll
            for (int k = 0; k < OWN * (NV / VW) / 64; ++k) {
                const int f = 64 * k + lane, o = f / (NV / VW), within = f % (NV / VW), owner = OWN * h + o;
                const V q = *reinterpret_cast<const V *>(slot + o * (NV + VW) + VW * within);
                const int64_t run = first + (int64_t)owner * NV;
                int64_t at = run + VW * within;
                at = at >= wrap ? at - wrap : at;
                if (owner >= lv0 && owner < lv1 && run >= from && run < stop) *reinterpret_cast<V *>(dst + at) = q;
            }
        }
    }
};

static constexpr int GSTAGE_M = 512, GSTAGE_A = 1024;    // candidates of a window that are written out through LDS, per list
template <typename T, int NV>
__device__ __forceinline__ void gather_details(const T (&d)[NV], bool own, int64_t b, int64_t k0, const WresGather &gat, int lane,
                                               T *stage_m, T *stage_a, int ablate) {
    using K = typename KeyT<T>::type;
    M1State *st = gat.st + b;
    T *const cand_m = static_cast<T *>(gat.cand_m), *const cand_a = static_cast<T *>(gat.cand_a);
    const K lo = (K)st->lo_m, hi = (K)st->hi_m;
    if (st->lo_m > st->hi_m) return;                       // no usable bracket: the block goes to the exact fallback
    const unsigned long long owners = __ballot(own);
    if (owners == 0) return;
    if (lo == 0 && hi == ~(K)0) {
        // short block, every coefficient is a candidate of both lists: stored in place
        bool nan = false;
        if (own) {
            store_run(cand_m + b * gat.cap_m + k0, d);
            store_run(cand_a + b * gat.cap_a + k0, d);
#pragma unroll
            for (int i = 0; i < NV; ++i) nan |= d[i] != d[i];
        }
        if (nan) st->has_nan = 1;
        if (lane == 0) {
            const unsigned int cnt = (unsigned int)__popcll(owners) * NV;
            atomicAdd(m1_counts(st), (unsigned long long)cnt | ((unsigned long long)cnt << 32));
        }
        return;
    }
    const T mhat = (T)st->mhat, alo = (T)st->alo, ahi = (T)st->ahi;
    const bool tie_m = st->tie_m != 0, tie_a = st->tie_a != 0;
    const K span = hi - lo;
    unsigned int below_m = 0, below_a = 0, cm = 0, ca = 0;
    unsigned int km = 0, ka = 0;                           // bit i: coefficient i is kept (NV <= 32)
    bool nan = false;
    static_assert(NV <= 32, "keep masks are 32 bits");
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const T x = d[i];
        nan |= x != x;
        const K k = m1_key(x);
        below_m += k < lo ? 1u : 0u;
        const bool keep_m = (K)(k - lo) <= span;
        const T a = fabs(x - mhat);
        below_a += a < alo ? 1u : 0u;
        const bool keep_a = a >= alo && a <= ahi;
        km |= keep_m ? 1u << i : 0u;
        ka |= keep_a ? 1u << i : 0u;
    }
    if (!own) km = ka = below_m = below_a = 0, nan = false;
    cm = __popc(km);
    ca = __popc(ka);
    // inclusive prefix sums over the lanes, both lists in one word (a list of a window holds fewer than 65 536 keys)
    unsigned int both = cm | (ca << 16);
#pragma unroll
    for (int dlt = 1; dlt < 64; dlt <<= 1) {
        const unsigned int o = __shfl_up(both, dlt);
        both += lane >= dlt ? o : 0u;
    }
    const unsigned int tot = __shfl(both, 63);
    const unsigned int tm = tot & 0xFFFFu, ta = tot >> 16;
    unsigned int base_m = 0, base_a = 0;
    if (lane == 0 && (tm | ta)) {
        // both list lengths in one returning atomic (they share an aligned 64-bit word)
        const unsigned long long was = atomicAdd(m1_counts(st), (unsigned long long)tm | ((unsigned long long)ta << 32));
        base_m = (unsigned int)was;
        base_a = (unsigned int)(was >> 32);
    }
    base_m = __shfl(base_m, 0);
    base_a = __shfl(base_a, 0);
    for (int dlt = 32; dlt >= 1; dlt >>= 1) {
        below_m += __shfl_xor(below_m, dlt);
        below_a += __shfl_xor(below_a, dlt);
    }
    if (lane == 0) {
        if (below_m) atomicAdd(&st->below_m, (unsigned long long)below_m);
        if (below_a) atomicAdd(&st->below_a, (unsigned long long)below_a);
    }
    if (nan) st->has_nan = 1;
    if (ablate & 8192) return;                             // timing only
    // The kept keys leave through the wave's LDS stage (compacted there, then written as consecutive elements of the
    // list); a window with more of them than the stage holds writes them one by one.
    auto emit = [&](unsigned int keep, unsigned int mine, unsigned int incl, unsigned int total, unsigned int base, T *stage,
                    unsigned int stage_cap, T *dst, int64_t cap) {
        if (total == 0) return;
        unsigned int p = incl - mine;                       // the lane's first position among the window's kept keys
        if (total <= stage_cap) {
#pragma unroll
            for (int i = 0; i < NV; ++i)
                if (keep >> i & 1u) stage[p++] = d[i];
            wave_sync_lds();
            for (unsigned int q = lane; q < total; q += 64)
                if ((int64_t)base + q < cap) dst[(int64_t)base + q] = stage[q];     // beyond the capacity: seen by n_m > cap_m
            wave_sync_lds();
        } else {
#pragma unroll
            for (int i = 0; i < NV; ++i)
                if (keep >> i & 1u) {
                    if ((int64_t)base + p < cap) dst[(int64_t)base + p] = d[i];
                    ++p;
                }
        }
    };
    if (!tie_m) emit(km, cm, both & 0xFFFFu, tm, base_m, stage_m, GSTAGE_M, cand_m + b * gat.cap_m, gat.cap_m);
    if (!tie_a) emit(ka, ca, both >> 16, ta, base_a, stage_a, GSTAGE_A, cand_a + b * gat.cap_a, gat.cap_a);
}

// One window per wave, four waves per workgroup.  STATS: the finest-band details of the clean lanes to d1_out and,
// when top != null, their level-TL approximation to top.  APPLY: analysis, shrink, synthesis; the level-TL
// approximation comes from `top` when use_top (the deep levels' result), else from the window's own analysis.
template <typename T, int F, int TL, int C, int MODE>
__global__ __launch_bounds__(256) void k_wres(const T *__restrict__ x, T *__restrict__ y, BlockGeom g,
                                              TapsF<T, F> tp, const double *__restrict__ thresh, int thr_mode,
                                              int64_t *__restrict__ zeroed, T *__restrict__ d1_out, int64_t d1_stride,
                                              T *__restrict__ top, int64_t top_stride, int use_top, int ablate, WresGather gat) {
    typedef Wres<T, F, TL, C, MODE> K;
    typedef typename K::G G;
    typedef typename K::P P;
    constexpr int NV = MODE == MODE_APPLY ? C : C / 2;      // values a lane sends out through the stage
    __shared__ __attribute__((aligned(16))) T stage[MODE == MODE_GATHER ? 4 * (GSTAGE_M + GSTAGE_A) : 4 * Staged<T, NV>::SLOT];
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.z * 32768 + blockIdx.y;
    if (b >= g.n_blocks) return;
    const int64_t n = g.frames[b];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t wi = (int64_t)blockIdx.x * 4 + wave;
    int64_t W0 = wi * G::A;                                                        // first sample of the window
    // A block whose length is a multiple of the lanes' run is periodic at every level without a repeated last sample:
    // its windows go round the block (the last ones wrap to its start, a lane's run never straddles the end) and
    // cover it whole.  Any other block: the windows that fit and one more that ends within a run of the block's end;
    // the LDS tiles take what is left at the two ends.
    const bool ring = wres_ring(n, G::W, C);
    int64_t own_from = 0, own_stop = INT64_MAX;           // first samples of the runs this window owns lie in [own_from, own_stop)
    if (ring) {
        if (W0 >= n) return;
        own_stop = n + (int64_t)G::LV0 * C;               // the last window stops where the first one's clean lanes begin
    } else {
        if (n < G::W) return;
        const int64_t regular = (n - G::W) / G::A + 1;
        if (wi > regular) return;
        if (wi == regular) {
            W0 = (n - G::W) / C * C;
            own_from = regular * G::A + G::LEAD;
            if (W0 + (int64_t)G::LV1 * C <= own_from) return;
        }
    }
    int64_t start = W0 + lane * C;                                                  // the lane's first sample
    const bool clean = lane >= G::LV0 && lane < G::LV1 && start >= own_from && start < own_stop;
    if (ring) start = start >= n ? start - n : start;

    typename K::St s;
    {
        T v[C];
        load_run(v, x + g.in_off[b] + start);
#pragma unroll
        for (int p = 0; p < C / 2; ++p) s.s0[p] = P{v[2 * p], v[2 * p + 1]};
    }
    const T thr = MODE == MODE_APPLY ? (T)thresh[b] : (T)0;
    int nz = K::template ana_all<1>(s, tp, thr, thr_mode);

    if (MODE == MODE_GATHER) {
        {
            T d[C / 2];
#pragma unroll
            for (int i = 0; i < C / 2; ++i) d[i] = K::at(s.dp[1], i);
            T *gs = stage + wave * (GSTAGE_M + GSTAGE_A);
            gather_details(d, clean, b, start >> 1, gat, lane, gs, gs + GSTAGE_M, ablate);
        }
        if (clean && top != nullptr) {
            T a[C >> TL];
#pragma unroll
            for (int i = 0; i < (C >> TL); ++i) a[i] = K::at(s.ap, i);
            store_run(top + b * top_stride + (start >> TL), a);
        }
        return;
    }
    if (MODE == MODE_STATS) {
        if (d1_out != nullptr) {
            T d[C / 2];
#pragma unroll
            for (int i = 0; i < C / 2; ++i) d[i] = K::at(s.dp[1], i);
            T *db = d1_out + b * d1_stride;
            if ((reinterpret_cast<uintptr_t>(db) & 15) == 0)
                Staged<T, C / 2>::store(d, stage + wave * Staged<T, C / 2>::SLOT, db, W0 >> 1, ring ? n >> 1 : INT64_MAX, own_from >> 1,
                                     ring ? own_stop >> 1 : INT64_MAX, G::LV0, G::LV1, lane);
            else if (clean)
                store_run(db + (start >> 1), d);
        }
        if (!clean) return;
        if (top != nullptr) {
            T a[C >> TL];
#pragma unroll
            for (int i = 0; i < (C >> TL); ++i) a[i] = K::at(s.ap, i);
            store_run(top + b * top_stride + (start >> TL), a);
        }
        return;
    }

    if (zeroed != nullptr) {
        nz = clean ? nz : 0;
        for (int d = 32; d >= 1; d >>= 1) nz += __shfl_xor(nz, d);
        if (lane == 0 && nz) atomicAdd((unsigned long long *)&zeroed[b], (unsigned long long)nz);
    }
    if (use_top) {
        T a[C >> TL];
        load_run(a, top + b * top_stride + (start >> TL));
#pragma unroll
        for (int p = 0; p < (C >> TL) / 2; ++p) s.ap[p] = P{a[2 * p], a[2 * p + 1]};
    }
    P out[C / 2];
    {
        P none[C / 2];
#pragma unroll
        for (int i = 0; i < C / 2; ++i) none[i] = P{(T)0, (T)0};
        K::template syn_all<TL>(s, tp, none, out);
    }
    T v[C];
#pragma unroll
    for (int p = 0; p < C / 2; ++p) {
        v[2 * p] = out[p].x;
        v[2 * p + 1] = out[p].y;
    }
    if (ablate & 8192) return;                             // timing only
    T *yb = y + g.out_off[b];
    if ((reinterpret_cast<uintptr_t>(yb) & 15) == 0)
        Staged<T, C>::store(v, stage + wave * Staged<T, C>::SLOT, yb, W0, ring ? n : INT64_MAX, own_from, own_stop, G::LV0, G::LV1, lane);
    else if (clean)
        store_run(yb + start, v);
}

// samples per lane: the smallest run whose halos come from the adjacent lanes only, but at least sixteen; binary64
// holds 32 at most (sixteen taps at four levels then reach two lanes away at the deepest levels)
template <typename T, int F, int TL> struct RunOf {
    static constexpr int NEED = F * (1 << TL) / 4;
    static constexpr int C = sizeof(T) == 8 ? (NEED <= 16 ? 16 : 32) : (NEED < 16 ? 16 : NEED);
};

template <typename T, int F, int TL, int MODE>
static int launch_f_tl(Ctx *c, const T *x, T *y, const BlockGeom &g, int64_t max_frames, int wavelet, const double *thresh,
                       int thr_mode, int64_t *zeroed, T *d1, int64_t d1_stride, T *top, int64_t top_stride, int use_top,
                       WresPlan *plan) {
    constexpr int C = RunOf<T, F, TL>::C;
    if constexpr (C == 0) {
        plan->window = 0;
        return ADN_OK;
    } else {
        typedef WresGeom<F, TL, C> G;
        plan->window = G::W;
        plan->advance = G::A;
        plan->lead = G::LEAD;
        plan->run = C;
        if (max_frames < G::W) return ADN_OK;                  // no block holds a window
        const int64_t windows = ceil_div(max_frames, G::A) + 1;    // of a ring; a block with LDS tiles at its ends has no more
        const TapsF<T, F> tp = tiles::make_taps_f<T, F>(wavelet);
        const int64_t by = g.n_blocks < 32768 ? g.n_blocks : 32768;
        ADN_LAUNCH_NAMED(MODE == MODE_APPLY ? "k_wres<APPLY>" : "k_wres<STATS>", (k_wres<T, F, TL, C, MODE>), dim3((unsigned)ceil_div(windows, 4), (unsigned)by, (unsigned)ceil_div(g.n_blocks, by)), dim3(256),
                   0, c->stream, x, y, g, tp, thresh, thr_mode, zeroed, d1, d1_stride, top, top_stride, use_top, c->ablate, WresGather{nullptr, nullptr, 0, nullptr, 0});
        return ADN_OK;
    }
}

template <typename T, int F, int MODE>
static int launch_f(Ctx *c, int TL, const T *x, T *y, const BlockGeom &g, int64_t max_frames, int wavelet, const double *thresh,
                    int thr_mode, int64_t *zeroed, T *d1, int64_t d1_stride, T *top, int64_t top_stride, int use_top,
                    WresPlan *plan) {
    switch (TL) {
        case 1: return launch_f_tl<T, F, 1, MODE>(c, x, y, g, max_frames, wavelet, thresh, thr_mode, zeroed, d1, d1_stride, top, top_stride, use_top, plan);
        case 2: return launch_f_tl<T, F, 2, MODE>(c, x, y, g, max_frames, wavelet, thresh, thr_mode, zeroed, d1, d1_stride, top, top_stride, use_top, plan);
        case 3: return launch_f_tl<T, F, 3, MODE>(c, x, y, g, max_frames, wavelet, thresh, thr_mode, zeroed, d1, d1_stride, top, top_stride, use_top, plan);
        default: return launch_f_tl<T, F, 4, MODE>(c, x, y, g, max_frames, wavelet, thresh, thr_mode, zeroed, d1, d1_stride, top, top_stride, use_top, plan);
    }
}

template <typename T>
static int tiles_t(Ctx *c, int F, int TL, int mode, const T *x, T *y, const BlockGeom &g, int64_t max_frames, int wavelet,
                   const double *thresh, int thr_mode, int64_t *zeroed, T *d1, int64_t d1_stride, T *top, int64_t top_stride,
                   int use_top, WresPlan *plan) {
    if (F == 8) {
        if (mode == tiles::MODE_STATS)
            return launch_f<T, 8, tiles::MODE_STATS>(c, TL, x, y, g, max_frames, wavelet, thresh, thr_mode, zeroed, d1, d1_stride, top, top_stride, use_top, plan);
        return launch_f<T, 8, tiles::MODE_APPLY>(c, TL, x, y, g, max_frames, wavelet, thresh, thr_mode, zeroed, d1, d1_stride, top, top_stride, use_top, plan);
    }
    if (mode == tiles::MODE_STATS)
        return launch_f<T, 16, tiles::MODE_STATS>(c, TL, x, y, g, max_frames, wavelet, thresh, thr_mode, zeroed, d1, d1_stride, top, top_stride, use_top, plan);
    return launch_f<T, 16, tiles::MODE_APPLY>(c, TL, x, y, g, max_frames, wavelet, thresh, thr_mode, zeroed, d1, d1_stride, top, top_stride, use_top, plan);
}

template <typename T, int F, int TL, int C>
static int launch_gather(Ctx *c, const T *x, const BlockGeom &g, int64_t max_frames, int wavelet, WresGather gat, T *top,
                         int64_t top_stride, WresPlan *plan) {
    typedef WresGeom<F, TL, C> G;
    plan->window = G::W;
    plan->advance = G::A;
    plan->lead = G::LEAD;
    plan->run = C;
    if (max_frames < G::W) {
        plan->window = 0;                                  // no block holds a window: nothing is gathered here
        return ADN_OK;
    }
    const int64_t windows = ceil_div(max_frames, G::A) + 1;
    const TapsF<T, F> tp = tiles::make_taps_f<T, F>(wavelet);
    const int64_t by = g.n_blocks < 32768 ? g.n_blocks : 32768;
    ADN_LAUNCH_NAMED("k_wres<GATHER>", (k_wres<T, F, TL, C, MODE_GATHER>), dim3((unsigned)ceil_div(windows, 4), (unsigned)by, (unsigned)ceil_div(g.n_blocks, by)),
                     dim3(256), 0, c->stream, x, (T *)nullptr, g, tp, (const double *)nullptr, 0, (int64_t *)nullptr, (T *)nullptr,
                     (int64_t)0, top, top_stride, 0, c->ablate, gat);
    return ADN_OK;
}

// (a gather of one level alone, without the approximation, was measured against the direct gather k_d1_gather_direct on
// 2560 blocks of 144 000 binary32 samples: 0.46 ms against 0.37 ms -- classifying costs both the same, and the direct
// kernel's loads are the better coalesced; so only the four-level form is built, for the blocks that need `top` anyway)
template <typename T>
static int gather_t(Ctx *c, int F, int TL, const T *x, const BlockGeom &g, int64_t max_frames, int wavelet, WresGather gat, T *top,
                    int64_t top_stride, WresPlan *plan) {
    if (TL != 4) {
        set_error("internal: the wave-resident gather is built for four levels");
        return ADN_ERR_ARG;
    }
    if (F == 8) return launch_gather<T, 8, 4, RunOf<T, 8, 4>::C>(c, x, g, max_frames, wavelet, gat, top, top_stride, plan);
    return launch_gather<T, 16, 4, RunOf<T, 16, 4>::C>(c, x, g, max_frames, wavelet, gat, top, top_stride, plan);
}

}  // namespace wres

int wres_gather(Ctx *c, int F, int TL, const float *x, const BlockGeom &g, int64_t max_frames, int wavelet, WresGather gat,
                float *top, int64_t top_stride, WresPlan *plan) {
    return wres::gather_t<float>(c, F, TL, x, g, max_frames, wavelet, gat, top, top_stride, plan);
}
int wres_gather(Ctx *c, int F, int TL, const double *x, const BlockGeom &g, int64_t max_frames, int wavelet, WresGather gat,
                double *top, int64_t top_stride, WresPlan *plan) {
    return wres::gather_t<double>(c, F, TL, x, g, max_frames, wavelet, gat, top, top_stride, plan);
}

// Launches the wave-resident tiles of TL levels on every window that fits its block and reports, in *plan, how the
// windows tile a block (window length, advance, first stored sample): the caller launches the LDS tiles on what is left.
// plan->window == 0: this transform has no wave-resident form in this precision.
int wres_tiles(Ctx *c, int F, int TL, int mode, const float *x, float *y, const BlockGeom &g, int64_t max_frames, int wavelet,
               const double *thresh, int thr_mode, int64_t *zeroed, float *d1, int64_t d1_stride, float *top, int64_t top_stride,
               int use_top, WresPlan *plan) {
    return wres::tiles_t<float>(c, F, TL, mode, x, y, g, max_frames, wavelet, thresh, thr_mode, zeroed, d1, d1_stride, top, top_stride, use_top, plan);
}
int wres_tiles(Ctx *c, int F, int TL, int mode, const double *x, double *y, const BlockGeom &g, int64_t max_frames, int wavelet,
               const double *thresh, int thr_mode, int64_t *zeroed, double *d1, int64_t d1_stride, double *top, int64_t top_stride,
               int use_top, WresPlan *plan) {
    return wres::tiles_t<double>(c, F, TL, mode, x, y, g, max_frames, wavelet, thresh, thr_mode, zeroed, d1, d1_stride, top, top_stride, use_top, plan);
}

}  // namespace adn
