// Register-resident STFT kernels for the headline shape: n_fft = 2048, hop 512, both precisions.
//
// One wavefront owns one frame at a time and a run of consecutive frames of one signal.  The 2048 real samples
// are packed into 1024 complex points, 16 per lane, and transformed as 16 x 16 x 4:
//   pass 1  16-point DFT in registers over the stride-64 samples of a lane,
//           twiddle by W1024^(lane*k1), transpose through LDS (padded rows: conflict free);
//   pass 2  16-point DFT in registers, twiddle by W64^(b*c), second exchange through LDS;
//   pass 3  four radix-4 butterflies per lane.
// The inverse transform is the same network with conjugated twiddles.  Spectra are exchanged between lanes only
// through a natural-order LDS image with 8 entries of padding per 256 (all accesses of a wave hit distinct
// banks); the inter-pass twiddles are one table per workgroup in LDS, the real-FFT split factors of a lane are
// one value times compile-time constants.
// Kernels: k_apply_wave (STFT -> modify -> ISTFT, one wave per run of output hops), k_rows_wave (power / magnitude
// rows in groups of four frames, one wave per run of frames), k_rows_fast (transposed rows, one workgroup per
// sixteen frames), k_apply_wg_map (run sizes for batches of signals of different lengths).
#include "adn_kernels.hpp"
#include "adn_stft_shared.hpp"

namespace adn {
namespace fast {

// A complex value is a native two-float vector, so that it occupies an aligned register pair and
// the packed binary32 instructions (two lanes of the pair per instruction) apply to it directly.
// Component swaps and sign changes of complex arithmetic (multiplication by +-i, the cross terms
// of a complex product) are expressed with the instructions' own operand modifiers (op_sel picks
// the half of a source each result half reads, neg_lo / neg_hi negate it), which costs nothing;
// written as plain component arithmetic the same operations compile to separate moves.
typedef float c2 __attribute__((ext_vector_type(2)));
static_assert(sizeof(c2) == sizeof(float2), "c2 is layout-compatible with float2");
static constexpr int NC = 1024;     // complex points per frame
static constexpr int RS = 68;       // exchange row stride (float2): 64 + 4 keeps quads on distinct banks
static constexpr int WBUF = 1088;   // float2 per wave buffer: 16 * RS >= padded natural image + 1
static constexpr int FAST_ST = 3;   // time-smoothing frames this specialisation is built for
static constexpr int ROWS_GPW = 8;  // groups of frames per workgroup of the rows kernel
static constexpr int PARK = 1026;   // float2 slots of a parked spectrum (1025 used)
static constexpr int RAWPAD = 8;    // replicated bins kept beyond each end of a raw magnitude row

__device__ __forceinline__ int nat(int k) { return k + ((k >> 8) << 3); }        // float2 index
__device__ __forceinline__ int natf(int s) { return s + ((s >> 9) << 4); }       // float index of sample s

__device__ __forceinline__ const c2 *as_c2(const float2 *p) { return reinterpret_cast<const c2 *>(p); }

// a + (-i) b  (forward transforms: INV = false)  or  a + i b  (inverse)
template <bool INV>
__device__ __forceinline__ c2 add_rot(c2 a, c2 b) {
    c2 r;
    if (INV) asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    else asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a - (-i) b  (INV = false)  or  a - i b  (INV = true)
template <bool INV>
__device__ __forceinline__ c2 sub_rot(c2 a, c2 b) { return add_rot<!INV>(a, b); }

// complex products: two packed instructions each
__device__ __forceinline__ c2 mul(c2 v, c2 w) {            // v * w
    c2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(v), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(v), "v"(w), "v"(t));
    return r;
}
__device__ __forceinline__ c2 mulc(c2 v, c2 w) {           // v * conj(w)
    c2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[1,0]" : "=v"(t) : "v"(v), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(v), "v"(w), "v"(t));
    return r;
}
// the same with the second factor in a scalar register pair (compile-time constants)
__device__ __forceinline__ c2 mul_k(c2 v, c2 w) {
    c2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(v), "s"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(v), "s"(w), "v"(t));
    return r;
}
__device__ __forceinline__ c2 mulc_k(c2 v, c2 w) {
    c2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1] neg_hi:[1,0]" : "=v"(t) : "v"(v), "s"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(v), "s"(w), "v"(t));
    return r;
}
// (a.x + b.x, a.y - b.y)  and  (a.y + b.y, b.x - a.x): the even / odd parts of a real-signal spectrum pair
__device__ __forceinline__ c2 add_conj(c2 a, c2 b) {
    c2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ c2 odd_part(c2 a, c2 b) {
    c2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// (a.x + b.y, b.x - a.y) = conj(a - i b)
__device__ __forceinline__ c2 conj_sub_rot(c2 a, c2 b) {
    c2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// (explicit fused multiply-adds where a kernel's result must not depend on how the compiler contracts the
// expression in that kernel: the rows kernels and the fused kernels have to agree bit for bit)
__device__ __forceinline__ float norm2(c2 z) { return __builtin_fmaf(z.x, z.x, z.y * z.y); }
// (xk.x - xm.x, xk.y + xm.y)
__device__ __forceinline__ c2 sub_add(c2 a, c2 b) {
    c2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// -i v (forward) / i v (inverse)
template <bool INV>
__device__ __forceinline__ c2 rot90(c2 v) {
    c2 r;
    if (INV) asm("v_pk_add_f32 %0, %1, 0 op_sel:[1,0] op_sel_hi:[0,0] neg_lo:[1,0]" : "=v"(r) : "v"(v));
    else asm("v_pk_add_f32 %0, %1, 0 op_sel:[1,0] op_sel_hi:[0,0] neg_hi:[1,0]" : "=v"(r) : "v"(v));
    return r;
}

// The same vocabulary for binary64 (the reference's precision): plain component arithmetic, there are no packed
// binary64 instructions.  The kernels below are written once against this vocabulary.
typedef double z2 __attribute__((ext_vector_type(2)));
typedef double z4 __attribute__((ext_vector_type(4)));
typedef float f4v __attribute__((ext_vector_type(4)));
static_assert(sizeof(z2) == sizeof(double2), "z2 is layout-compatible with double2");
__device__ __forceinline__ const z2 *as_c2(const double2 *p) { return reinterpret_cast<const z2 *>(p); }
template <bool INV>
__device__ __forceinline__ z2 add_rot(z2 a, z2 b) { return INV ? z2{a.x - b.y, a.y + b.x} : z2{a.x + b.y, a.y - b.x}; }
template <bool INV>
__device__ __forceinline__ z2 sub_rot(z2 a, z2 b) { return add_rot<!INV>(a, b); }
__device__ __forceinline__ z2 mul(z2 v, z2 w) { return z2{fma(-v.y, w.y, v.x * w.x), fma(v.y, w.x, v.x * w.y)}; }
__device__ __forceinline__ z2 mulc(z2 v, z2 w) { return z2{fma(v.y, w.y, v.x * w.x), fma(v.y, w.x, -(v.x * w.y))}; }
__device__ __forceinline__ z2 mul_k(z2 v, z2 w) { return mul(v, w); }
__device__ __forceinline__ z2 mulc_k(z2 v, z2 w) { return mulc(v, w); }
__device__ __forceinline__ z2 add_conj(z2 a, z2 b) { return z2{a.x + b.x, a.y - b.y}; }
__device__ __forceinline__ z2 odd_part(z2 a, z2 b) { return z2{a.y + b.y, b.x - a.x}; }
__device__ __forceinline__ z2 conj_sub_rot(z2 a, z2 b) { return z2{a.x + b.y, b.x - a.y}; }
__device__ __forceinline__ double norm2(z2 z) { return fma(z.x, z.x, z.y * z.y); }
__device__ __forceinline__ z2 sub_add(z2 a, z2 b) { return z2{a.x - b.x, a.y + b.y}; }
template <bool INV>
__device__ __forceinline__ z2 rot90(z2 v) { return INV ? z2{-v.y, v.x} : z2{v.y, -v.x}; }

// scalar type of a complex type, and the types a frame is fetched in
template <typename C> struct ReOf;
template <> struct ReOf<c2> { typedef float type; };
template <> struct ReOf<z2> { typedef double type; };
template <typename C> using Re = typename ReOf<C>::type;
template <typename R> struct QuadOf;                   // four values of R as one vector
template <typename R> struct CxOf;
template <> struct CxOf<float> { typedef c2 type; };
template <> struct CxOf<double> { typedef z2 type; };
template <> struct QuadOf<float> { typedef f4v type; };
template <> struct QuadOf<double> { typedef z4 type; };
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
// binary64: the hardware's reciprocal square root (about 26 good bits) and two Newton steps, eight instructions where
// the library's sqrt followed by a division takes some forty; the result is within two units in the last place for
// every positive finite argument (squared magnitudes here: no overflow of x * y * y near the ends of the range
// matters, x y y stays about 1).  Zero gives NaN (0 * inf): the callers select around it.
__device__ __forceinline__ double fast_rsq(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    y = y * fma(-(h * y), y, 1.5);
    y = y * fma(-(h * y), y, 1.5);
    return y;
}
__device__ __forceinline__ double fast_sqrt(double x) {
    const double r = x * fast_rsq(x);
    return (x > 0.0 && x < __builtin_inf()) ? r : x;       // sqrt(+-0) = +-0, sqrt(inf) = inf, sqrt(NaN) = NaN
}

// v * W16^M (forward) or its conjugate (inverse)
template <int M, bool INV, typename C>
__device__ __forceinline__ C mul_w16(C v) {
    typedef Re<C> R;
    constexpr R C1 = (R)0.9238795325112867, S1 = (R)0.3826834323650898, RH = (R)0.7071067811865476;
    if (M == 0) return v;
    if (M == 4) return rot90<INV>(v);                      // -i v  /  i v
    if (M == 2) return add_rot<INV>(v, v) * RH;            // (x + y, y - x) R  /  (x - y, x + y) R
    if (M == 6) return add_rot<!INV>(v, v) * (-RH);        // (y - x, -x - y) R  /  (-x - y, x - y) R
    // generic: W = (c, -s) forward
    constexpr R c = M == 1 ? C1 : (M == 3 ? S1 : -C1);     // M in {1, 3, 9}
    constexpr R sn = M == 1 ? S1 : (M == 3 ? C1 : -S1);
    return INV ? mulc_k(v, C{c, -sn}) : mul_k(v, C{c, -sn});
}

template <bool INV, typename C>
__device__ __forceinline__ void bfly4(C &a0, C &a1, C &a2, C &a3) {
    const C s02 = a0 + a2, d02 = a0 - a2, s13 = a1 + a3, d13 = a1 - a3;
    a0 = s02 + s13;
    a1 = add_rot<INV>(d02, d13);
    a2 = s02 - s13;
    a3 = sub_rot<INV>(d02, d13);
}

// 16-point DFT in registers.  Input r[n]; output X[k] at r[P16(k)].
__host__ __device__ constexpr int P16(int k) { return 4 * (k & 3) + (k >> 2); }

template <bool INV, typename C>
__device__ __forceinline__ void dft16(C (&r)[16]) {
    bfly4<INV>(r[0], r[4], r[8], r[12]);
    bfly4<INV>(r[1], r[5], r[9], r[13]);
    bfly4<INV>(r[2], r[6], r[10], r[14]);
    bfly4<INV>(r[3], r[7], r[11], r[15]);
    // position 4*k1 + n2 holds y[n2][k1]; twiddle by W16^(n2*k1)
    r[5] = mul_w16<1, INV>(r[5]);
    r[6] = mul_w16<2, INV>(r[6]);
    r[7] = mul_w16<3, INV>(r[7]);
    r[9] = mul_w16<2, INV>(r[9]);
    r[10] = mul_w16<4, INV>(r[10]);
    r[11] = mul_w16<6, INV>(r[11]);
    r[13] = mul_w16<3, INV>(r[13]);
    r[14] = mul_w16<6, INV>(r[14]);
    r[15] = mul_w16<9, INV>(r[15]);
    bfly4<INV>(r[0], r[1], r[2], r[3]);
    bfly4<INV>(r[4], r[5], r[6], r[7]);
    bfly4<INV>(r[8], r[9], r[10], r[11]);
    bfly4<INV>(r[12], r[13], r[14], r[15]);
}

// Twiddles of the two inter-pass multiplications, one copy per workgroup in LDS:
//   t1[(k1-1)*64 + lane] = W1024^(lane*k1),  t2[(c-1)*4 + b] = W64^(b*c),  k1, c = 1..15
static constexpr int TWTAB = 15 * 64 + 15 * 4;     // complex entries
template <typename C>
struct LaneTwT {
    const C *t1;   // + lane
    const C *t2;   // + (lane & 3)
};

// Cooperative fill by the whole workgroup from the precomputed image (16-byte copies, the table
// offset inside the blob is 16-byte aligned); the caller synchronises before first use.
__device__ __forceinline__ void fill_tw_table(c2 *tab, const float2 *__restrict__ image, int tid, int nthreads) {
    const float4 *src = reinterpret_cast<const float4 *>(image);
    float4 *dst = reinterpret_cast<float4 *>(tab);
    for (int i = tid; i < TWTAB / 2; i += nthreads) dst[i] = src[i];
}
__device__ __forceinline__ void fill_tw_table(z2 *tab, const double2 *__restrict__ image, int tid, int nthreads) {
    const z2 *src = reinterpret_cast<const z2 *>(image);
    for (int i = tid; i < TWTAB; i += nthreads) tab[i] = src[i];
}

template <typename C>
__device__ __forceinline__ LaneTwT<C> lane_tw(const C *tab, int lane) {
    return LaneTwT<C>{tab + lane, tab + 15 * 64 + (lane & 3)};
}

// In: lane l, r[j] = x[l + 64 j].
// Out: lane l = 16*cq + k1, r[4*cc + d] = X[k1 + 16 * (4*cq + cc) + 256 * d].
// Three passes (16 x 16 x 4 points), all butterflies in registers, two exchanges through the
// wave's LDS buffer E (16 rows of RS float2).
template <bool INV, typename C>
__device__ __forceinline__ void wave_fft1024(C (&r)[16], C *E, const LaneTwT<C> &t, int lane) {
    dft16<INV>(r);
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) {
        C v = r[P16(k1)];
        if (k1) v = INV ? mulc(v, t.t1[(k1 - 1) * 64]) : mul(v, t.t1[(k1 - 1) * 64]);
        E[k1 * RS + lane] = v;
    }
    wave_sync();
    const int row = (lane >> 2) * RS + (lane & 3);
#pragma unroll
    for (int a = 0; a < 16; ++a) r[a] = E[row + 4 * a];
    wave_sync();
    dft16<INV>(r);
    // lane = 4*k1 + b now holds Y[c] (at r[P16(c)]) of the 16-point transforms over a (source index
    // b + 4a): twiddle by W64^(b*c) and regroup so that one lane holds all four b of four values of
    // c; row k1 of E takes the values at column 4*c + b
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        C v = r[P16(c)];
        if (c) v = INV ? mulc(v, t.t2[(c - 1) * 4]) : mul(v, t.t2[(c - 1) * 4]);
        E[row + 4 * c] = v;
    }
    wave_sync();
    if constexpr (sizeof(C) == 8) {
        const float4 *src = reinterpret_cast<const float4 *>(E + (lane & 15) * RS + (lane >> 4) * 16);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 q = src[i];
            r[2 * i] = C{q.x, q.y};
            r[2 * i + 1] = C{q.z, q.w};
        }
    } else {
        const C *src = E + (lane & 15) * RS + (lane >> 4) * 16;
#pragma unroll
        for (int i = 0; i < 16; ++i) r[i] = src[i];
    }
    wave_sync();
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) bfly4<INV>(r[4 * cc], r[4 * cc + 1], r[4 * cc + 2], r[4 * cc + 3]);
}

// Natural-order (padded) store of the layout produced by wave_fft1024.
template <typename C>
__device__ __forceinline__ void store_natural(const C (&r)[16], C *buf, int lane) {
    const int base = (lane & 15) + 64 * (lane >> 4);          // k1 + 16 * 4 * cq
#pragma unroll
    for (int cc = 0; cc < 4; ++cc)
#pragma unroll
        for (int d = 0; d < 4; ++d) buf[base + 16 * cc + 264 * d] = r[4 * cc + d];   // nat(k1 + 16 c + 256 d)
}

// Real-signal spectrum of the packed transform held in buf (natural, padded).
// lo[j] = X[k], hi[j] = X[1024-k] for k = lane + 64 j (lane 0, j 0: X[0] and X[1024]); mid = X[512].
template <typename C>
struct SpecT {
    C lo[8], hi[8], mid;
};

// Addresses of the two kinds of accesses to the natural (padded) spectrum image: bin lane + 64 j
// and its mirror 1024 - lane - 64 j.  Both are one base address per lane plus compile-time
// offsets (so that they cost two registers, not one per access): nat(lane + 64 j) = lane + 64 j +
// 8 (j >> 2); for lane >= 1 the mirror is (64 - lane) + 64 (15 - j) + 8 ((15 - j) >> 2); lane 0
// sits exactly on the padding steps and needs 8 more for j = 4, 8, 12 (its j = 0 mirror is bin 0,
// handled by the callers, and the address formed for it stays inside the buffer).
template <typename C>
struct NatAddrT {
    const C *fwd;     // buf + lane
    const C *mir;     // buf + 64 - lane
    const C *mir4;    // the same, + 8 for lane 0: mirror base of j = 4, 8, 12
};
template <typename C>
__device__ __forceinline__ NatAddrT<C> nat_addr(const C *buf, int lane) {
    return NatAddrT<C>{buf + lane, buf + 64 - lane, buf + 64 - lane + (lane == 0 ? 8 : 0)};
}
template <int J> __device__ __forceinline__ int nat_fwd_off() { return 64 * J + 8 * (J >> 2); }
template <int J> __device__ __forceinline__ int nat_mir_off() { return 64 * (15 - J) + 8 * ((15 - J) >> 2); }

template <int J, typename C>
__device__ __forceinline__ void split_bin(SpecT<C> &s, const NatAddrT<C> &na, const C (&rtw)[8], Re<C> hs, Re<C> scale, int lane) {
    const C zk = na.fwd[nat_fwd_off<J>()];
    const C zm = ((J & 3) == 0 ? na.mir4 : na.mir)[nat_mir_off<J>()];
    const C e = add_conj(zk, zm);                          // (zk.x + zm.x, zk.y - zm.y)
    const C o = odd_part(zk, zm);                          // (zk.y + zm.y, zm.x - zk.x)
    const C wo = mul(o, rtw[J]);
    s.lo[J] = (e + wo) * hs;
    s.hi[J] = (e - wo) * C{hs, -hs};                       // conjugated
    if (J == 0 && lane == 0) {
        s.lo[0] = C{(zk.x + zk.y) * scale, 0};
        s.hi[0] = C{(zk.x - zk.y) * scale, 0};
    }
}

template <typename C>
__device__ __forceinline__ void split_spectrum(SpecT<C> &s, const C *buf, const C (&rtw)[8], Re<C> scale, int lane) {
    // Z[k] = E[k] + W^k O[k] with E = (z[k] + conj(z[N-k])) / 2, O = -i (z[k] - conj(z[N-k])) / 2;
    // the halves are folded into the scale
    const Re<C> hs = (Re<C>)0.5 * scale;
    const NatAddrT<C> na = nat_addr(buf, lane);
    split_bin<0>(s, na, rtw, hs, scale, lane);
    split_bin<1>(s, na, rtw, hs, scale, lane);
    split_bin<2>(s, na, rtw, hs, scale, lane);
    split_bin<3>(s, na, rtw, hs, scale, lane);
    split_bin<4>(s, na, rtw, hs, scale, lane);
    split_bin<5>(s, na, rtw, hs, scale, lane);
    split_bin<6>(s, na, rtw, hs, scale, lane);
    split_bin<7>(s, na, rtw, hs, scale, lane);
    const C z = buf[nat(512)];
    s.mid = C{z.x * scale, -z.y * scale};
}

// Inverse of split_spectrum for a wave that transforms the result itself: the lower half of the packed spectrum, Z'[lane + 64 j]
// for j < 8, is what the inverse transform wants in r[j] of this very lane, so it stays in registers; only the
// mirrored bins (the upper half, owned by other lanes) go through buf.
template <int J, typename C>
__device__ __forceinline__ void merge_bin_keep(const SpecT<C> &s, C (&r)[16], const NatAddrT<C> &na, const C (&rtw)[8], int lane) {
    typedef Re<C> R;
    const C xk = s.lo[J], xm = s.hi[J];
    C *mir = const_cast<C *>((J & 3) == 0 ? na.mir4 : na.mir);
    const C e = add_conj(xk, xm) * (R)0.5;                 // ((xk.x + xm.x) / 2, (xk.y - xm.y) / 2)
    const C d = sub_add(xk, xm);                           // (xk.x - xm.x, xk.y + xm.y)
    const C o = mulc(d * (R)0.5, rtw[J]);
    r[J] = add_rot<true>(e, o);                            // e + i o = (e.x - o.y, e.y + o.x)
    const C m = conj_sub_rot(e, o);                        // conj(e - i o) = (e.x + o.y, o.x - e.y)
    if (J == 0) {
        // lane 0: bins 0 and 1024 are real and pack into Z'[0]; its mirror slot is not a bin
        if (lane == 0) r[0] = C{(xk.x + xm.x) * (R)0.5, (xk.x - xm.x) * (R)0.5};
        else mir[nat_mir_off<J>()] = m;
    } else {
        mir[nat_mir_off<J>()] = m;
    }
}

template <typename C>
__device__ __forceinline__ void merge_spectrum_keep(const SpecT<C> &s, C (&r)[16], C *buf, const C (&rtw)[8], int lane) {
    const NatAddrT<C> na = nat_addr(buf, lane);
    merge_bin_keep<0>(s, r, na, rtw, lane);
    merge_bin_keep<1>(s, r, na, rtw, lane);
    merge_bin_keep<2>(s, r, na, rtw, lane);
    merge_bin_keep<3>(s, r, na, rtw, lane);
    merge_bin_keep<4>(s, r, na, rtw, lane);
    merge_bin_keep<5>(s, r, na, rtw, lane);
    merge_bin_keep<6>(s, r, na, rtw, lane);
    merge_bin_keep<7>(s, r, na, rtw, lane);
    if (lane == 0) buf[nat(512)] = C{s.mid.x, -s.mid.y};
}

// exp(-2 pi i (lane + 64 j) / 2048) = rtw0 * exp(-i pi j / 16)
template <typename C>
__device__ __forceinline__ void derive_rtw(C (&rtw)[8], C rtw0) {
    typedef Re<C> R;
    rtw[0] = rtw0;
    rtw[1] = mul_k(rtw0, C{(R)0.9807852804032304, (R)-0.19509032201612825});
    rtw[2] = mul_k(rtw0, C{(R)0.9238795325112867, (R)-0.3826834323650898});
    rtw[3] = mul_k(rtw0, C{(R)0.8314696123025452, (R)-0.5555702330196022});
    rtw[4] = mul_k(rtw0, C{(R)0.7071067811865476, (R)-0.7071067811865476});
    rtw[5] = mul_k(rtw0, C{(R)0.5555702330196022, (R)-0.8314696123025452});
    rtw[6] = mul_k(rtw0, C{(R)0.3826834323650898, (R)-0.9238795325112867});
    rtw[7] = mul_k(rtw0, C{(R)0.19509032201612825, (R)-0.9807852804032304});
}

// Raw samples of the frame at padded position p0, requested into registers (no arithmetic, so
// that the loads stay in flight).  NCH = 1: pf[j] is packed point lane + 64 j; NCH = 2: pf[j] holds
// the two (left, right) pairs of packed point lane + 64 j.  Returns false when the frame needs the
// generic loader instead (tail of the signal, odd alignment, another channel count).
template <int NCH, typename R> struct FetchT;
template <> struct FetchT<1, float> { typedef c2 type; };
template <> struct FetchT<2, float> { typedef float4 type; };
template <> struct FetchT<1, double> { typedef z2 type; };
template <> struct FetchT<2, double> { typedef z4 type; };

__device__ __forceinline__ c2 load_streaming(const c2 *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ float4 load_streaming(const float4 *p) {
    const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(p));
    return float4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ z2 load_streaming(const z2 *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ z4 load_streaming(const z4 *p) { return __builtin_nontemporal_load(p); }

template <int NCH, typename R>
__device__ __forceinline__ bool fetch_frame(typename FetchT<NCH, R>::type (&pf)[16], const R *__restrict__ xs, int64_t n,
                                            int n_ch, int64_t p0, int lane) {
    typedef typename FetchT<NCH, R>::type V;
    const bool ok = p0 + 2048 <= n && n_ch == NCH && (reinterpret_cast<uintptr_t>(xs + NCH * p0) & (sizeof(V) - 1)) == 0;
    // No branch around the loads (a join between the loads and their first use would make the
    // compiler wait for every younger memory operation as well): a frame that cannot be fetched this
    // way reads the first 2048 frames of the signal instead, from the address rounded down to the
    // load size, and the result is ignored.  Every signal holds at least one whole frame; rounding
    // down stays inside the caller's buffer because only element offsets inside it can be odd.
    // (pointer arithmetic only: an integer round trip would turn the loads into flat ones)
    const R *src = ok ? xs + NCH * p0 : xs - ((reinterpret_cast<uintptr_t>(xs) & (sizeof(V) - 1)) / sizeof(R));
    const V *xv = reinterpret_cast<const V *>(src) + lane;
    __builtin_amdgcn_sched_barrier(0);
    // the first quarter of a frame is the hop that no later frame of the run reads again: it is loaded
    // with the streaming hint, so that the cache keeps the three hops the next steps come back for
#pragma unroll
    for (int j = 0; j < 4; ++j) pf[j] = load_streaming(xv + 64 * j);
#pragma unroll
    for (int j = 4; j < 16; ++j) pf[j] = xv[64 * j];
    __builtin_amdgcn_sched_barrier(0);
    return ok;
}

__device__ __forceinline__ c2 fold_point(const c2 &q) { return q; }
__device__ __forceinline__ c2 fold_point(const float4 &q) { return c2{(q.x + q.y) / 2.0f, (q.z + q.w) / 2.0f}; }
__device__ __forceinline__ z2 fold_point(const z2 &q) { return q; }
__device__ __forceinline__ z2 fold_point(const z4 &q) { return z2{(q.x + q.y) / 2.0, (q.z + q.w) / 2.0}; }

// Windowed packed frame r[j] = z[lane + 64 j] from the fetched samples; wv = first half of the window
template <typename V, typename C>
__device__ __forceinline__ void window_fetched(C (&r)[16], const V (&pf)[16], const C *wv, int lane) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const C w = wv[lane + 64 * j];
        const C lo = fold_point(pf[j]), hi = fold_point(pf[j + 8]);
        r[j] = lo * w;
        r[j + 8] = __builtin_elementwise_fma(-hi, w, hi);  // w[n + 1024] = 1 - w[n]
    }
}

// Any frame (zero-padded tail, odd alignment, any channel count): staged through the wave's buffer.
template <typename C, typename R>
__device__ __noinline__ void stage_frame_half_window(C *buf, const R *__restrict__ xs, int64_t n, int n_ch,
                                                     int64_t p0, const R *__restrict__ whalf, int lane) {
#pragma unroll 1
    for (int i = lane; i < NC; i += 64) {
        const R u = fold_sample(xs, p0 + 2 * i, n, n_ch), v = fold_sample(xs, p0 + 2 * i + 1, n, n_ch);
        const int wi = 2 * (i & 511);
        const R w0 = whalf[wi], w1 = whalf[wi + 1];
        buf[nat(i)] = i < 512 ? C{u * w0, v * w1} : C{fma(-u, w0, u), fma(-v, w1, v)};
    }
}

// --------------------------------------------------- transposed power / magnitude rows
template <int W>
__global__ __launch_bounds__(W * 64) void k_rows_fast(const float *__restrict__ x, Signals sg,
                                                      Tables<float> tb, int64_t groups_per_sig,
                                                      int FT, int64_t Tmax, int power,
                                                      float *__restrict__ S,
                                                      unsigned long long *__restrict__ sig_max) {
    extern __shared__ __align__(16) unsigned char smem[];
    c2 *bufs = reinterpret_cast<c2 *>(smem);
    float *tile = reinterpret_cast<float *>(bufs + (size_t)W * WBUF);      // [1025][FT + 1]
    c2 *twtab = reinterpret_cast<c2 *>(tile + 1025 * (FT + 1) + ((1025 * (FT + 1)) & 1));
    float *wtab = reinterpret_cast<float *>(twtab + TWTAB);                // [2048] window
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // a workgroup handles ROWS_GPW consecutive groups of FT frames of one signal, so that the
    // tables are staged once per ROWS_GPW groups
    const int64_t wgs_per_sig = (groups_per_sig + ROWS_GPW - 1) / ROWS_GPW;
    const int64_t sig = blockIdx.x / wgs_per_sig, grp0 = (blockIdx.x % wgs_per_sig) * ROWS_GPW;
    const int64_t n = sg.len[sig];
    const int64_t Tn = stft_frames(n, 2048, sg.hop, sg.padded);
    if (grp0 * FT >= Tn) return;
    const float *xs = x + sg.off[sig];
    c2 *buf = bufs + (size_t)wave * WBUF;
    fill_tw_table(twtab, tb.fast_tw, threadIdx.x, W * 64);
    for (int i = threadIdx.x; i < 2048; i += W * 64) wtab[i] = tb.window[i];
    __syncthreads();
    const LaneTwT<c2> tw = lane_tw(twtab, lane);
    c2 rtw[8];
    derive_rtw(rtw, as_c2(tb.rtw)[lane]);            // as the one-wave-per-run kernels derive them
    const float scale = (float)(1.0 / tb.wsum);
    const int ts = FT + 1;
    float vmax = 0.0f;
#pragma unroll 1
    for (int g = 0; g < ROWS_GPW; ++g) {
    const int64_t t0 = (grp0 + g) * FT;
    if (t0 >= Tn) break;
    if (g) __syncthreads();                                  // the tile of the previous group has been stored
    for (int it = 0; it < FT / W; ++it) {
        const int ft = it * W + wave;
        const int64_t t = t0 + ft;
        if (t < Tn) {
            // the loader and window arithmetic of the one-wave-per-run kernels (second half of the window applied as
            // x - x w[n - 1024]), so that a frame gives the same bits whichever kernel transforms it
            c2 r[16];
            bool direct;
            const c2 *wv = reinterpret_cast<const c2 *>(wtab);
            if (sg.n_ch == 2) {
                float4 pf[16];
                direct = fetch_frame<2>(pf, xs, n, 2, t * sg.hop, lane);
                if (direct) window_fetched(r, pf, wv, lane);
            } else {
                c2 pf[16];
                direct = fetch_frame<1>(pf, xs, n, sg.n_ch, t * sg.hop, lane);
                if (direct) window_fetched(r, pf, wv, lane);
            }
            if (!direct) {
                stage_frame_half_window(buf, xs, n, sg.n_ch, t * sg.hop, wtab, lane);
                wave_sync();
                const c2 *nb = buf + lane;
#pragma unroll
                for (int j = 0; j < 16; ++j) r[j] = nb[64 * j + 8 * (j >> 2)];        // nat(lane + 64 j)
                wave_sync();
            }
            wave_fft1024<false>(r, buf, tw, lane);
            store_natural(r, buf, lane);
            wave_sync();
            SpecT<c2> s;
            split_spectrum(s, buf, rtw, scale, lane);
            wave_sync();
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = lane + 64 * j;
                float p0 = norm2(s.lo[j]);
                float p1 = norm2(s.hi[j]);
                if (!power) {
                    p0 = __builtin_amdgcn_sqrtf(p0);
                    p1 = __builtin_amdgcn_sqrtf(p1);
                }
                tile[k * ts + ft] = p0;
                tile[(1024 - k) * ts + ft] = p1;
                vmax = fmaxf(vmax, fmaxf(p0, p1));
            }
            if (lane == 0) {
                float pm = norm2(s.mid);
                if (!power) pm = __builtin_amdgcn_sqrtf(pm);
                tile[512 * ts + ft] = pm;
                vmax = fmaxf(vmax, pm);
            }
        }
    }
    __syncthreads();
    const int nft = (int)((Tn - t0) < FT ? (Tn - t0) : FT);
    float *Srow = S + (size_t)sig * 1025 * Tmax + t0;
    if (nft == FT && (FT & 3) == 0 && (reinterpret_cast<uintptr_t>(Srow) & 15) == 0 && (Tmax & 3) == 0) {
        // four frames of one bin per 16-byte store
        const int q4 = FT / 4;
        for (int idx = threadIdx.x; idx < 1025 * q4; idx += W * 64) {
            const int k = idx / q4, f4 = (idx % q4) * 4;
            const float *tp = tile + k * ts + f4;
            *reinterpret_cast<float4 *>(Srow + (size_t)k * Tmax + f4) = float4{tp[0], tp[1], tp[2], tp[3]};
        }
    } else {
        for (int idx = threadIdx.x; idx < 1025 * FT; idx += W * 64) {
            const int k = idx / FT, ft = idx % FT;
            if (ft < nft) Srow[(size_t)k * Tmax + ft] = tile[k * ts + ft];
        }
    }
    }
    if (sig_max != nullptr) {
        for (int d = 32; d >= 1; d >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, d));
        if (lane == 0) atomicMax(&sig_max[sig], (unsigned long long)key_of((double)vmax));
    }
}

// ------------------------------------ fused STFT -> modify -> ISTFT, one wave per run of hops
// hop 512, four-fold overlap.  A wave owns a run of consecutive output hops of one signal and
// walks its frames one after another, alone: no workgroup barrier after the tables are staged,
// so the waves of a SIMD drift apart and one wave's LDS / memory round trips are covered by the
// other's arithmetic.
//   * overlap-add: the inverse transform leaves the four hop-quarters of a frame at the same
//     in-hop positions of a lane (r[4 cc + d], d = quarter), so the three partial hops are
//     twelve complex registers per lane and a finished hop leaves in four 8-byte stores; no
//     accumulator ring in LDS, no pass over it.
//   * a finished hop is stored by the NEXT step, right after that step has requested its frame's
//     samples: memory operations complete in order, so a wave that asked for its frame after the
//     stores would wait for the stores' acknowledgement as well.
//   * SUBTRACT smooths over the two neighbouring frames: the frequency-smoothed magnitudes of a
//     lane's own 17 bins stay in registers (the sum of the two previous rows and the previous
//     row), and the spectrum of the newest frame, whose right neighbour is not known yet, is
//     swapped with the one parked in the wave's LDS slot.  The waves of a workgroup work on one
//     signal and share its noise row in LDS.
//   * GAIN keeps the signal's gain row in the wave's LDS slot (per-frame gain rows are read from
//     memory).
// LDS bandwidth is the scarcest resource of this kernel (six exchanges of a whole frame per frame),
// so tables are kept out of it where arithmetic can stand in: only the first half of the periodic
// Hann window is stored (w[n + 1024] = 1 - w[n], applied as x - x w), and the real-FFT split
// factors of a lane are one value times compile-time constants.
static constexpr int GROW = 1032;                 // floats of a gain / noise row slot (1025 used)


// NCH: the channel count whose frames are fetched straight into registers (other counts, and frames
// at the zero-padded tail of a signal, take the staged loader)
// R: binary32 (two or three waves per SIMD) or binary64 (one wave per SIMD: a frame alone is 64 registers)
template <typename R, int W, int MODE, int NCH, bool PROF>
__global__ __launch_bounds__(W * 64)
__attribute__((amdgpu_waves_per_eu(sizeof(R) == 8 ? 1 : (MODE == APPLY_SUBTRACT ? 2 : 3), sizeof(R) == 8 ? 1 : (MODE == APPLY_SUBTRACT ? 2 : 3))))
void k_apply_wave(ApplyArgs<R> a) {
    typedef typename CxOf<R>::type C;
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int H = 512;
    // binary64 SUBTRACT: four waves' exchange buffers and parked spectra fill the LDS of a CU to the last byte when
    // the parked spectrum takes exactly its 1025 slots and the signal's noise row is read from memory instead
    constexpr bool ROW_IN_LDS = !(sizeof(R) == 8 && MODE == APPLY_SUBTRACT);
    // a finished hop waits for the next step's loads to be out before it is stored (binary32 GAIN stores it at once)
    constexpr bool DEFER = !(sizeof(R) == 4 && MODE == APPLY_GAIN);
    constexpr int SLOT = MODE == APPLY_SUBTRACT ? (sizeof(R) == 8 ? 1025 : PARK) : 0;   // complex slots per wave beside the exchange buffer
    const Tables<R> &tb = a.tb;
    C *bufs = reinterpret_cast<C *>(smem);                                          // [W][WBUF] exchange
    C *slots = bufs + (size_t)W * WBUF;                                              // [W][SLOT] parked spectrum / gain row
    C *twtab = slots + (size_t)W * SLOT;
    R *wtab = reinterpret_cast<R *>(twtab + TWTAB);                           // [1024] first half of the window
    R *ntab = wtab + 1024;                                                        // [512] scale / full-overlap norm
    R *nrow_lds = ntab + 512;                                                     // [GROW] the signal's noise row (SUBTRACT) / gain row (GAIN)
    // the wave index as a scalar: the run's bounds, pointers and counters then live in scalar registers
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane0 = threadIdx.x & 63, tid = threadIdx.x;
    // the waves of a workgroup work on one signal (segs_per_sig is a multiple of W) and share its noise / gain row
    int64_t sig;
    int run;
    bool have_sig;
    if (a.wg_start != nullptr) {
        // signals of different lengths: signal s owns workgroups wg_start[s] .. wg_start[s + 1] - 1 (as many as its own
        // frames need); the grid is sized for the longest signal and the workgroups past the last one do nothing
        const int b = (int)blockIdx.x, ns = (int)a.n_sig;
        have_sig = b < a.wg_start[ns];
        int lo = 0, hi = ns;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (a.wg_start[mid] <= b) lo = mid;
            else hi = mid;
        }
        sig = lo;
        run = (b - a.wg_start[lo]) * W + wave;
    } else {
        const int64_t gid = (int64_t)blockIdx.x * W + wave;
        sig = gid / a.segs_per_sig;
        run = (int)(gid % a.segs_per_sig);
        have_sig = sig < a.n_sig;
    }
    const R *row = a.row + (have_sig ? sig : 0) * a.row_stride;
    fill_tw_table(twtab, tb.fast_tw, tid, W * 64);
    for (int i = tid; i < 1024; i += W * 64) wtab[i] = tb.window[i];
    const R fscale = (R)(1.0 / tb.wsum);
    const R iscale = (R)(tb.wsum / 1024.0);
    for (int i = tid; i < 512; i += W * 64) {
        R sq = (R)0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const R w = tb.window[i + q * H];
            sq += w * w;
        }
        ntab[i] = iscale / (sq > (R)1e-10 ? sq : (R)1.0);
    }
    if (MODE == APPLY_SUBTRACT || a.row_t_stride == 0)
        if (ROW_IN_LDS)
            for (int i = tid; i <= 1024; i += W * 64) nrow_lds[i] = row[i];
    __syncthreads();                                       // the only workgroup barrier: every wave reaches it

    if (!have_sig) return;
    const int64_t n = a.sg.len[sig];
    const int Tn = (int)stft_frames(n, 2048, H, 1);        // frame / hop indices of one signal fit 32 bits
    // output hops are the padded hops j = 2 .. Tn (istft drops n_fft/2 at both ends); hop j is
    // complete once frame min(j, Tn - 1) has been added
    const int seg_hops = a.wg_start != nullptr ? a.wg_start[a.n_sig + 1] : a.seg_hops;
    const int ja = 2 + run * seg_hops;
    if (Tn < 2 || ja > Tn) return;
    const int jb = (ja + seg_hops < Tn + 1) ? ja + seg_hops : Tn + 1;            // exclusive
    const int ta = ja - 3 < 0 ? 0 : ja - 3;                // first frame synthesised (fills the partial hops)
    const int tb_ = jb - 1 < Tn - 1 ? jb - 1 : Tn - 1;     // last frame synthesised
    const R *xs = a.x + a.sg.off[sig];
    R *ys = a.y + a.y_off[sig];
    int16_t *ps = a.pcm ? a.pcm + a.y_off[sig] : nullptr;
    C *buf = bufs + wave * WBUF;
    R *bufr = reinterpret_cast<R *>(buf);
    C *park = slots + wave * SLOT;
    const R *nrow = ROW_IN_LDS ? nrow_lds : row;
    const R *grow_lds = nrow;
    const C *wv = reinterpret_cast<const C *>(wtab);
    const C *nv = reinterpret_cast<const C *>(ntab);
    const C rtw0 = as_c2(tb.rtw)[lane0];
    const bool y_vec = (reinterpret_cast<uintptr_t>(ys) & 7) == 0;
    const bool p_vec = (reinterpret_cast<uintptr_t>(ps) & 3) == 0;
    const bool gain_in_lds = MODE == APPLY_GAIN && a.row_t_stride == 0;
    auto wfull = [&](int idx) { return idx < 1024 ? wtab[idx] : (R)1.0 - wtab[idx - 1024]; };

    // one finished hop: normalise, clip, optional PCM16, store
    auto emit = [&](int j, const C (&v)[4], int mlane) {
        const bool full = j >= 3 && j <= Tn - 1;
        const int64_t base = (int64_t)(j - 2) * H + 2 * mlane;               // the lane's first sample of the hop
        R *yp0 = ys + base;
        int16_t *pp0 = ps + base;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const int i = 2 * (mlane + 16 * cc);           // sample of the hop (and i + 1)
            C o;
            if (full) {
                o = v[cc] * nv[mlane + 16 * cc];
            } else {
                R nx_ = (R)0.0, ny_ = (R)0.0;
                int t_lo = j - 3;
                t_lo = t_lo < 0 ? 0 : t_lo;
                const int t_hi = j < Tn - 1 ? j : Tn - 1;
#pragma unroll 1
                for (int tt = t_lo; tt <= t_hi; ++tt) {
                    const R w0 = wfull((j - tt) * H + i), w1 = wfull((j - tt) * H + i + 1);
                    nx_ += w0 * w0;
                    ny_ += w1 * w1;
                }
                o = C{v[cc].x * iscale / (nx_ > (R)1e-10 ? nx_ : (R)1.0), v[cc].y * iscale / (ny_ > (R)1e-10 ? ny_ : (R)1.0)};
            }
            if (!a.no_clip) {
                o.x = o.x < -(R)1.0 ? -(R)1.0 : (o.x > (R)1.0 ? (R)1.0 : o.x);
                o.y = o.y < -(R)1.0 ? -(R)1.0 : (o.y > (R)1.0 ? (R)1.0 : o.y);
            }
            if (ps) {
                const int k0 = pcm_int32(o.x) >> 16, k1 = pcm_int32(o.y) >> 16;
                int16_t *pp = pp0 + 32 * cc;
                if (p_vec) {
                    *reinterpret_cast<unsigned *>(pp) = (unsigned)(k0 & 0xFFFF) | ((unsigned)k1 << 16);
                } else {
                    pp[0] = (int16_t)k0;
                    pp[1] = (int16_t)k1;
                }
                o = C{(R)k0 * ((R)1.0 / (R)32768.0), (R)k1 * ((R)1.0 / (R)32768.0)};
            }
            R *yp = yp0 + 32 * cc;
            if (y_vec) {
                *reinterpret_cast<C *>(yp) = o;
            } else {
                yp[0] = o.x;
                yp[1] = o.y;
            }
        }
    };

    // frames analysed: SUBTRACT also needs the magnitude rows of the frames around [ta, tb_]
    const int u0 = MODE == APPLY_SUBTRACT ? (ta > 0 ? ta - 1 : 0) : ta;
    const int u1 = MODE == APPLY_SUBTRACT ? (tb_ + 1 < Tn - 1 ? tb_ + 1 : Tn - 1) : tb_;
    C acc0[4], acc1[4], acc2[4];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) acc0[cc] = acc1[cc] = acc2[cc] = C{(R)0.0, (R)0.0};
    // SUBTRACT: frequency-smoothed magnitude rows at the lane's bins (lo: k = lane + 64 j, hi: 1024 - k, mid: 512)
    R m1_lo[8], m1_hi[8], m1_mid = (R)0.0;               // row of the previous frame
    R s2_lo[8], s2_hi[8], s2_mid = (R)0.0;               // sum of the rows of the two previous frames
    const int s_end = MODE == APPLY_SUBTRACT ? tb_ + 1 : tb_;
    // diagnostics (PROF): shader-clock totals of the phases of this wave's frames
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tk = PROF ? clock64() : 0;
    auto tick = [&](int i) {
        if (PROF) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned long long now = clock64();
            ph[i] += now - tk;
            tk = now;
        }
    };
    C pend[4];                                            // finished hop waiting to be stored
    int pend_j = -1;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) pend[cc] = C{(R)0.0, (R)0.0};

#pragma unroll 1
    for (int s = u0; s <= s_end; ++s) {
        // Everything derived from the lane index (LDS and memory addresses of a dozen access patterns)
        // is re-derived in every step from an opaque copy: kept across the steps as loop invariants,
        // those values alone would take some 150 registers from the transforms.
        int lane = lane0;
        asm volatile("" : "+v"(lane));
        const int mlane = (lane & 15) + 64 * (lane >> 4);  // packed point of r[4 cc + d] is mlane + 16 cc + 256 d
        const LaneTwT<C> tw = lane_tw(twtab, lane);
        SpecT<C> sp;
        const bool analyse = s <= u1;
        // the frame's samples are requested first, THEN the hop finished by the previous step is stored
        typename FetchT<NCH, R>::type pf[16];
        bool pf_mode = false;
        // (a step without analysis fetches, and ignores, the run's last frame again)
        pf_mode = fetch_frame<NCH>(pf, xs, n, a.sg.n_ch, (int64_t)(analyse ? s : u1) * H, lane) && analyse;
        // The common case of the pending hop (full overlap, aligned output, no PCM copy) is stored by
        // straight-line code: four unconditional stores, no branch.  The wait for the frame
        // then allows exactly those four stores to be outstanding; any branch between the loads and
        // their first use makes the compiler's wait conservative (it would wait for the stores too).
        const bool pend_fast = pend_j >= 3 && pend_j <= Tn - 1 && y_vec && ps == nullptr;
        if constexpr (DEFER) {
            const R lim = a.no_clip ? (R)__builtin_inf() : (R)1.0;
            // a hop that is not stored here goes to a scratch line instead (unconditional stores)
            R *yp0 = pend_fast ? ys + ((int64_t)(pend_j - 2) * H + 2 * mlane) : a.dummy + 2 * mlane;
            C o[4];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                o[cc] = pend[cc] * nv[mlane + 16 * cc];
                o[cc].x = o[cc].x < -lim ? -lim : (o[cc].x > lim ? lim : o[cc].x);
                o[cc].y = o[cc].y < -lim ? -lim : (o[cc].y > lim ? lim : o[cc].y);
            }
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) __builtin_nontemporal_store(o[cc], reinterpret_cast<C *>(yp0 + 32 * cc));
        }
        __builtin_amdgcn_sched_barrier(0);
        C r[16];                                          // the frame being transformed (either direction)
        if (analyse) {
            // ---- frame s: window the fetched samples, forward transform, real-signal spectrum
            if (pf_mode) {
                window_fetched(r, pf, wv, lane);
            } else {
                stage_frame_half_window(buf, xs, n, a.sg.n_ch, (int64_t)s * H, wtab, lane);
                wave_sync();
                const C *nb = buf + lane;
#pragma unroll
                for (int j = 0; j < 16; ++j) r[j] = nb[64 * j + 8 * (j >> 2)];        // nat(lane + 64 j)
                wave_sync();
            }
            tick(0);
        }
        // the other cases of the pending hop (first / last hops of a signal, PCM copy, odd alignment)
        if (pend_j >= 0 && !pend_fast) emit(pend_j, pend, mlane);
        pend_j = -1;
        if (analyse) {
            wave_fft1024<false>(r, buf, tw, lane);
            tick(1);
            store_natural(r, buf, lane);
            wave_sync();
            C rtw[8];
            derive_rtw(rtw, rtw0);
            split_spectrum(sp, buf, rtw, fscale, lane);
            wave_sync();
            tick(2);
        }
        __builtin_amdgcn_sched_barrier(0);
        int t = s;                                          // frame synthesised in this step
        if (MODE == APPLY_SUBTRACT) {
            R mc_lo[8], mc_hi[8], mc_mid;
            if (analyse) {
                // clean magnitudes of the lane's bins -> raw row in the wave's buffer (RAWPAD spare
                // bins at both ends for the box filter's edge replication)
                R *rawm = bufr + RAWPAD;
                {
                    // mirrored bins 1024 - lane - 64 j are addressed upwards from the lowest one (j = 7):
                    // LDS instructions take unsigned offsets only
                    const R *nlo = nrow + lane, *nhi = nrow + 576 - lane;
                    R *wlo = rawm + lane, *whi = rawm + 576 - lane;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const R m0 = fast_sqrt(norm2(sp.lo[j]));
                        const R m1 = fast_sqrt(norm2(sp.hi[j]));
                        wlo[64 * j] = fmax(m0 - a.alpha * nlo[64 * j], a.beta * m0);
                        whi[64 * (7 - j)] = fmax(m1 - a.alpha * nhi[64 * (7 - j)], a.beta * m1);
                    }
                }
                if (lane == 0) {
                    const R mm = fast_sqrt(norm2(sp.mid));
                    rawm[512] = fmax(mm - a.alpha * nrow[512], a.beta * mm);
                }
                // park the new spectrum, take the previous frame's
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const C plo = park[j * 64 + lane], phi = park[(8 + j) * 64 + lane];
                    park[j * 64 + lane] = sp.lo[j];
                    park[(8 + j) * 64 + lane] = sp.hi[j];
                    sp.lo[j] = plo;
                    sp.hi[j] = phi;
                }
                const C pm = park[1024];
                wave_sync();
                if (lane == 0) park[1024] = sp.mid;
                sp.mid = pm;
                if (lane < RAWPAD) {
                    rawm[-1 - lane] = rawm[0];
                    rawm[1025 + lane] = rawm[1024];
                }
                wave_sync();
                __builtin_amdgcn_sched_barrier(0);
                // box mean over five bins (the only frequency smoothing this kernel is launched for);
                // same summation order as the shape-generic kernel
                const R inv_sf = (R)1.0 / (R)5.0;
                const R *blo = rawm + lane - 2, *bhi = rawm + 576 - lane - 2;     // first bin of each window
                auto box = [&](const R *p) {
                    R rv[5];
#pragma unroll
                    for (int d = 0; d < 5; ++d) rv[d] = p[d];
                    R fs = (R)0.0;
#pragma unroll
                    for (int d = 0; d < 5; ++d) fs += rv[d];
                    return fs * inv_sf;
                };
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    mc_lo[j] = box(blo + 64 * j);
                    mc_hi[j] = box(bhi + 64 * (7 - j));
                    if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);     // at most 40 row values in flight
                }
                mc_mid = box(rawm + 510);
                wave_sync();                                // the row has been read: the buffer is free again
            } else {
                // s == Tn: the row beyond the last frame is the last frame's (edge replication); its
                // spectrum is the parked one
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    mc_lo[j] = m1_lo[j];
                    mc_hi[j] = m1_hi[j];
                    sp.lo[j] = park[j * 64 + lane];
                    sp.hi[j] = park[(8 + j) * 64 + lane];
                }
                mc_mid = m1_mid;
                sp.mid = park[1024];
            }
            if (s == u0) {                                  // the row before the first one: replicated if frame 0, unused otherwise
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    m1_lo[j] = mc_lo[j];
                    m1_hi[j] = mc_hi[j];
                    s2_lo[j] = s2_hi[j] = (R)0.0;
                }
                m1_mid = mc_mid;
                s2_mid = (R)0.0;
            }
            t = s - 1;
            if (t >= ta) {
                // smoothed magnitude of frame t = (row[t-1] + row[t] + row[t+1]) / 3, phase of its spectrum kept:
                // clean * exp(i angle(Z)) = clean * Z / |Z|, and angle(0) = 0
                const R inv_st = (R)1.0 / (R)FAST_ST;
                auto rephase = [&](C z, R sm) {
                    const R p2 = norm2(z);
                    const R g = sm * fast_rsq(p2);
                    return p2 > (R)0.0 ? z * g : C{sm, (R)0.0};
                };
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    sp.lo[j] = rephase(sp.lo[j], (s2_lo[j] + mc_lo[j]) * inv_st);
                    sp.hi[j] = rephase(sp.hi[j], (s2_hi[j] + mc_hi[j]) * inv_st);
                }
                sp.mid = rephase(sp.mid, (s2_mid + mc_mid) * inv_st);
            }
            // rows for the next step: s2 = row[s-1] + row[s], m1 = row[s]
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                s2_lo[j] = m1_lo[j] + mc_lo[j];
                s2_hi[j] = m1_hi[j] + mc_hi[j];
                m1_lo[j] = mc_lo[j];
                m1_hi[j] = mc_hi[j];
            }
            s2_mid = m1_mid + mc_mid;
            m1_mid = mc_mid;
        } else {
            if (gain_in_lds) {
                const R *glo = grow_lds + lane, *ghi = grow_lds + 576 - lane;     // upwards from bin 1024 - lane - 448
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    sp.lo[j] = sp.lo[j] * glo[64 * j];
                    sp.hi[j] = sp.hi[j] * ghi[64 * (7 - j)];
                }
                sp.mid = sp.mid * grow_lds[512];
            } else {
                const R *grow = row + (int64_t)t * a.row_t_stride;     // per-frame gain rows (masks)
                const R *glo = grow + lane, *ghi = grow + 1024 - lane;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    sp.lo[j] = sp.lo[j] * glo[64 * j];
                    sp.hi[j] = sp.hi[j] * ghi[-64 * j];
                }
                sp.mid = sp.mid * grow[512];
            }
        }
        tick(3);
        const bool synth = t >= ta;
        if (synth) {
            __builtin_amdgcn_sched_barrier(0);
            // ---- frame t: back to the packed spectrum, inverse transform
            {
                C rtw[8];
                derive_rtw(rtw, rtw0);
                merge_spectrum_keep(sp, r, buf, rtw, lane);
            }
            wave_sync();
            {
                const C *nb = buf + lane;
#pragma unroll
                for (int j = 8; j < 16; ++j) r[j] = nb[64 * j + 8 * (j >> 2)];        // nat(lane + 64 j)
            }
            wave_sync();
            tick(4);
            wave_fft1024<true>(r, buf, tw, lane);
            tick(5);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (synth) {
            // ---- synthesis window (first half stored: w[n + 1024] = 1 - w[n]), overlap-add
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const C w0 = wv[mlane + 16 * cc], w1 = wv[mlane + 16 * cc + 256];
                const C q0 = r[4 * cc] * w0, q1 = r[4 * cc + 1] * w1;
                const C q2 = r[4 * cc + 2] - r[4 * cc + 2] * w0, q3 = r[4 * cc + 3] - r[4 * cc + 3] * w1;
                pend[cc] = acc0[cc] + q0;                  // oldest frame first, like a sequential overlap-add
                acc0[cc] = acc1[cc] + q1;
                acc1[cc] = acc2[cc] + q2;
                acc2[cc] = q3;
            }
            if (DEFER) {
                if (t >= ja) pend_j = t;                   // stored by the next step, after its loads are out
            } else if (t >= ja) {
                // stored at once: the eight registers of a pending hop are what this instantiation lacks
                if (t >= 3 && t <= Tn - 1 && y_vec && ps == nullptr) {
                    const R lim = a.no_clip ? (R)__builtin_inf() : (R)1.0;
                    R *yp0 = ys + ((int64_t)(t - 2) * H + 2 * mlane);
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        C o = pend[cc] * nv[mlane + 16 * cc];
                        o.x = o.x < -lim ? -lim : (o.x > lim ? lim : o.x);
                        o.y = o.y < -lim ? -lim : (o.y > lim ? lim : o.y);
                        __builtin_nontemporal_store(o, reinterpret_cast<C *>(yp0 + 32 * cc));
                    }
                } else {
                    emit(t, pend, mlane);
                }
            }
        }
        tick(6);
    }
    const int mlane0 = (lane0 & 15) + 64 * (lane0 >> 4);
    if (pend_j >= 0) emit(pend_j, pend, mlane0);
    // the hop after the last frame is complete without a frame of its own
    if (jb == Tn + 1) emit(Tn, acc0, mlane0);
    if (PROF && lane0 == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&a.dbg[i], ph[i]);
}

// Sum of a binary64 value over the wave, delivered in lane 63: four shifts inside the rows of sixteen lanes, then
// the two row broadcasts (data-parallel-primitive moves of the two halves, no LDS round trips).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
    const double o = __hiloint2double(hi, lo);
    // lanes outside the row mask (or without a source lane) receive +0.0
    return v + o;
}
__device__ __forceinline__ double wave_sum_to_last_lane(double v) {
    v = dpp_add_f64<0x111, 0xf>(v);     // row_shr:1
    v = dpp_add_f64<0x112, 0xf>(v);     // row_shr:2
    v = dpp_add_f64<0x114, 0xf>(v);     // row_shr:4
    v = dpp_add_f64<0x118, 0xf>(v);     // row_shr:8: lane 15 of every row holds the row's sum
    v = dpp_add_f64<0x142, 0xa>(v);     // row_bcast:15 into rows 1 and 3
    v = dpp_add_f64<0x143, 0xc>(v);     // row_bcast:31 into rows 2 and 3: lane 63 holds the total
    return v;
}

// ------------------------------------ power / magnitude rows, one wave per run of frames
// The statistics pass of the Wiener stage.  A wave walks consecutive frames alone (no workgroup barrier after
// the tables are staged), keeps the 17 values of its bins for four frames in registers and writes them as one
// 16-byte store per bin: S[signal][frame / 4][bin][frame % 4].  Every store instruction of a wave is 1 KB of
// consecutive bytes, no transposition through LDS and nothing for the waves to wait for each other; the row
// statistics read a bin's values as one 16-byte piece per group of four frames.
// VADE: the kernel also adds up the energy of the voice-activity frames (src/wiener_filter.py:17-26: frames of
// frame_len samples, frame_len >= the hop) from the samples it holds anyway.  A run owns the first hop of each of
// its frames (the signal's last run: everything up to the signal's end); per lane two binary64 accumulators, for
// the voice-activity frame the run is in and for the next one; a finished frame's sum goes to vad_e[frame] with
// one atomic addition per run that holds a part of it -- at most two runs, so the sum does not depend on their order.
template <typename R, int W, int NCH, bool VADE>
__global__ __launch_bounds__(W * 64) __attribute__((amdgpu_waves_per_eu(sizeof(R) == 8 ? 1 : 3, sizeof(R) == 8 ? 1 : 3)))
void k_rows_wave(const R *__restrict__ x, Signals sg, Tables<R> tb, int64_t n_sig, int64_t runs_per_sig,
                 int groups_per_run, int64_t sig_stride, int power, R *__restrict__ S, R *__restrict__ dummy,
                 double *__restrict__ vad_e, int64_t frame_len, int64_t nv_max, unsigned long long *__restrict__ sig_max,
                 const int *__restrict__ wave_start) {
    typedef typename CxOf<R>::type C;
    typedef typename QuadOf<R>::type Q;                    // the four frames of a bin: one store
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int H = 512;
    C *bufs = reinterpret_cast<C *>(smem);                                          // [W][WBUF] exchange
    C *twtab = bufs + (size_t)W * WBUF;
    R *wtab = reinterpret_cast<R *>(twtab + TWTAB);                           // [1024] first half of the window
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane0 = threadIdx.x & 63, tid = threadIdx.x;
    if (wave_start != nullptr && (int64_t)blockIdx.x * W >= wave_start[n_sig]) return;     // the whole workgroup: no run left
    fill_tw_table(twtab, tb.fast_tw, tid, W * 64);
    for (int i = tid; i < 1024; i += W * 64) wtab[i] = tb.window[i];
    __syncthreads();                                       // the only workgroup barrier: every wave reaches it
    const int64_t gid = (int64_t)blockIdx.x * W + wave;
    int64_t sig;
    int run;
    if (wave_start != nullptr) {
        // signals of different lengths: signal s owns the waves wave_start[s] .. wave_start[s + 1] - 1, one per run of
        // wave_start[n_sig + 1] groups (both sized on the device from all lengths, see k_apply_wg_map)
        const int ns = (int)n_sig;
        if (gid >= wave_start[ns]) return;
        int lo = 0, hi = ns;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (wave_start[mid] <= (int)gid) lo = mid;
            else hi = mid;
        }
        sig = lo;
        run = (int)gid - wave_start[lo];
        groups_per_run = wave_start[ns + 1];
    } else {
        sig = gid / runs_per_sig;
        if (sig >= n_sig) return;
        run = (int)(gid % runs_per_sig);
    }
    const int64_t n = sg.len[sig];
    const int Tn = (int)stft_frames(n, 2048, H, sg.padded);
    const int g0 = run * groups_per_run;
    if (4 * g0 >= Tn) return;
    const int g_all = (Tn + 3) >> 2;
    const int g1 = g0 + groups_per_run < g_all ? g0 + groups_per_run : g_all;
    const R *xs = x + sg.off[sig];
    Q *Sg = reinterpret_cast<Q *>(S + sig * sig_stride);
    C *buf = bufs + wave * WBUF;
    const C *wv = reinterpret_cast<const C *>(wtab);
    const C rtw0 = as_c2(tb.rtw)[lane0];
    const R scale = (R)(1.0 / tb.wsum);
    R m0[17], m1[17], m2[17], m3[17];                  // the lane's bins of the group's four frames
#pragma unroll
    for (int i = 0; i < 17; ++i) m0[i] = m1[i] = m2[i] = m3[i] = (R)0.0;
    R vmax = 0;                                            // largest value of the run (sig_max: the signal's maximum)
    // voice-activity energies: accA belongs to frame vcur, accB to frame vcur + 1
    double accA = 0.0, accB = 0.0;
    int64_t vcur = -1;
    double *ve = VADE ? vad_e + sig * nv_max : nullptr;
    auto vad_flush = [&](double v, int64_t fr) {
        v = wave_sum_to_last_lane(v);
        if (lane0 == 63 && fr >= 0 && fr < nv_max) atomicAdd(&ve[fr], v);
    };
    // the hop at sample P; the lane holds its samples 2 lane + 128 j (+ 1) in s[j]
    auto vad_hop = [&](int64_t P, const C (&sv)[4], int lane) {
        const int64_t v0 = P / frame_len;
        if (vcur < 0) vcur = v0;
        if (v0 != vcur) {                                  // a hop is not longer than a frame: v0 == vcur + 1
            vad_flush(accA, vcur);
            accA = accB;
            accB = 0.0;
            vcur = v0;
        }
        const int64_t left = (vcur + 1) * frame_len - P;   // samples of the hop that still belong to frame vcur
        const int rel = left < H ? (int)left : H;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i0 = 2 * lane + 128 * j;
            const double d0 = (double)sv[j].x * (double)sv[j].x, d1 = (double)sv[j].y * (double)sv[j].y;
            if (i0 < rel) accA += d0; else accB += d0;
            if (i0 + 1 < rel) accA += d1; else accB += d1;
        }
    };
    auto vad_hop_from_memory = [&](int64_t P, int lane) {
        C sv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t q = P + 2 * lane + 128 * j;
            sv[j] = C{fold_sample(xs, q, n, sg.n_ch), fold_sample(xs, q + 1, n, sg.n_ch)};
        }
        vad_hop(P, sv, lane);
    };
#pragma unroll 1
    for (int t = 4 * g0; t < 4 * g1; ++t) {
        int lane = lane0;
        asm volatile("" : "+v"(lane));                     // lane-derived addresses are re-derived per frame, not kept
        R p[17];
        if (t < Tn) {
            const LaneTwT<C> tw = lane_tw(twtab, lane);
            typename FetchT<NCH, R>::type pf[16];
            const bool pf_mode = fetch_frame<NCH>(pf, xs, n, sg.n_ch, (int64_t)t * H, lane);
            C r[16];
            if (pf_mode) {
                window_fetched(r, pf, wv, lane);
                if (VADE) {
                    const C sv[4] = {fold_point(pf[0]), fold_point(pf[1]), fold_point(pf[2]), fold_point(pf[3])};
                    vad_hop((int64_t)t * H, sv, lane);
                }
            } else {
                if (VADE) vad_hop_from_memory((int64_t)t * H, lane);
                stage_frame_half_window(buf, xs, n, sg.n_ch, (int64_t)t * H, wtab, lane);
                wave_sync();
                const C *nb = buf + lane;
#pragma unroll
                for (int j = 0; j < 16; ++j) r[j] = nb[64 * j + 8 * (j >> 2)];        // nat(lane + 64 j)
                wave_sync();
            }
            wave_fft1024<false>(r, buf, tw, lane);
            store_natural(r, buf, lane);
            wave_sync();
            SpecT<C> sp;
            {
                C rtw[8];
                derive_rtw(rtw, rtw0);
                split_spectrum(sp, buf, rtw, scale, lane);
            }
            wave_sync();
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                p[j] = norm2(sp.lo[j]);
                p[8 + j] = norm2(sp.hi[j]);
            }
            p[16] = norm2(sp.mid);
            if (!power) {
#pragma unroll
                for (int i = 0; i < 17; ++i) p[i] = fast_sqrt(p[i]);
            }
            if (sig_max != nullptr) {
                // (bin 512 is lane 0's; the other lanes hold a copy of it, which changes no maximum)
#pragma unroll
                for (int i = 0; i < 17; ++i) vmax = fmax(vmax, p[i]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 17; ++i) p[i] = (R)0.0;      // frames past the signal's end inside its last group
        }
        const int e = t & 3;
        if (e == 0) {
#pragma unroll
            for (int i = 0; i < 17; ++i) m0[i] = p[i];
        } else if (e == 1) {
#pragma unroll
            for (int i = 0; i < 17; ++i) m1[i] = p[i];
        } else if (e == 2) {
#pragma unroll
            for (int i = 0; i < 17; ++i) m2[i] = p[i];
        } else {
#pragma unroll
            for (int i = 0; i < 17; ++i) m3[i] = p[i];
            Q *dst = Sg + (int64_t)(t >> 2) * ROWS_GROUP_STRIDE;
            Q *dlo = dst + lane, *dhi = dst + 1024 - lane;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                __builtin_nontemporal_store(Q{m0[j], m1[j], m2[j], m3[j]}, dlo + 64 * j);
                __builtin_nontemporal_store(Q{m0[8 + j], m1[8 + j], m2[8 + j], m3[8 + j]}, dhi - 64 * j);
            }
            // bin 512 belongs to lane 0; the other lanes' (meaningless) values go to a scratch line
            Q *dm = lane == 0 ? dst + 512 : reinterpret_cast<Q *>(dummy) + lane;
            *dm = Q{m0[16], m1[16], m2[16], m3[16]};
        }
    }
    if (sig_max != nullptr) {
        for (int d = 32; d >= 1; d >>= 1) vmax = fmax(vmax, __shfl_xor(vmax, d));
        if (lane0 == 0) atomicMax(&sig_max[sig], (unsigned long long)key_of((double)vmax));
    }
    if (VADE) {
        // the signal's last run also owns what lies beyond the first hop of the last frame
        if (g1 == g_all)
            for (int64_t P = (int64_t)Tn * H; P < n; P += H) vad_hop_from_memory(P, lane0);
        vad_flush(accA, vcur);
        vad_flush(accB, vcur + 1);
    }
}

static size_t rows_wave_lds(int W, size_t real_bytes) { return ((size_t)W * WBUF * 2 + (size_t)TWTAB * 2 + 1024) * real_bytes; }

static size_t apply_wave_lds(int W, int mode, size_t real_bytes) {
    const bool row_in_lds = !(real_bytes == 8 && mode == APPLY_SUBTRACT);
    const size_t slot = mode == APPLY_SUBTRACT ? (real_bytes == 8 ? 1025 : PARK) : 0;
    return ((size_t)W * WBUF * 2 + (size_t)W * slot * 2 + (size_t)TWTAB * 2 + (1024 + 512 + (row_in_lds ? GROW : 0))) * real_bytes;
}

}  // namespace fast

// hop 512: one wave per run of hops
// Run length and first workgroup of every signal for the one-wave-per-run kernels.  A signal of h output hops has
// ceil(h / run) runs (run r starts at output hop 2 + r * run) and takes ceil(runs / W) workgroups.  The host knows
// only the longest signal; with the lengths at hand the run is sized here so that the runs of all signals together
// fill whole rounds of the chip's wave slots (the fewest hop-steps per slot win; host_run, sized as if every signal
// were the longest, is the first candidate and is kept on ties), within the max_wgs workgroups of the launch.
// wg_start[0 .. n_sig]: first workgroup per signal and their total; wg_start[n_sig + 1]: hops per run.
// One workgroup, up to APPLY_MAP_MAX signals: signal q * 1024 + tid is thread tid's q-th.
// Workgroups of a signal of h hops (h < 2^24, exact in binary32), W = 1 << lw, inv ~ 1 / run:
__device__ __forceinline__ int wg_map_count(int h, int run, float inv, int lw) {
    int q = (int)((float)h * inv);                  // floor(h / run), off by one at most
    int r = h - q * run;
    if (r < 0) { --q; r += run; }
    if (r >= run) { ++q; r -= run; }
    q += r > 0;                                     // runs
    return (q + (1 << lw) - 1) >> lw;               // workgroups
}
// rows: the statistics kernel's variant - the unit is a group of four frames, W = 1 (a wave per run), and the run is
// what the host's rule gives for the real total (about two rounds of the wave slots, run_min .. run_max groups);
// host_run if that does not fit the launch.
__global__ __launch_bounds__(1024) void k_apply_wg_map(Signals sg, int n_sig, int host_run, int lw, int slots, int warm,
                                                       int max_wgs, int *__restrict__ wg_start, int rows, int run_min,
                                                       int run_max) {
    constexpr int PER = APPLY_MAP_MAX / 1024, NC = 25;
    __shared__ int s_hops[APPLY_MAP_MAX];
    __shared__ long long s_tot[16], s_G[NC];
    __shared__ int s_cand[NC];
    __shared__ int wtot[16];
    __shared__ int s_run;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int hops[PER];
    long long tot = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int s = q * 1024 + tid;
        hops[q] = 0;
        if (s < n_sig) {
            // stft_frames() for n_fft 2048 and hop 512, the only shape of the one-wave-per-run kernels, without divisions
            const int64_t n = sg.len[s];
            const int64_t nadd = (rows ? sg.padded : 1) ? ((512 - ((n - 2048) & 511)) & 511) : 0;
            const int64_t Tn = n < 2048 ? 0 : 1 + ((n + nadd - 2048) >> 9);
            hops[q] = rows ? (int)((Tn + 3) >> 2) : (Tn >= 2 ? (int)(Tn - 1) : 0);
            s_hops[s] = hops[q];
        }
        tot += hops[q];
    }
    if (wave * 64 < n_sig) {                        // (waves past the last signal hold zeros)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) tot += __shfl_xor(tot, d);
    }
    if (lane == 0) s_tot[wave] = tot;
    __syncthreads();
    if (tid < NC) {                                 // candidate tid: r rounds, with no, half or all of the rounding to W
        long long H = 0;                            // runs per signal allowed for
        for (int w = 0; w < 16; ++w) H += s_tot[w];
        int run = host_run;
        if (tid > 0) {
            const int r = 1 + (tid - 1) / 3, v = (tid - 1) % 3;
            const long long den = rows ? 2LL * slots : (long long)slots * r - (((long long)n_sig << lw) * v >> 1);
            // (binary64 division: exact for these magnitudes up to the rounding of a candidate, and much shorter)
            long long t = den > 0 ? (long long)ceil((double)H / (double)den) : run_max;
            run = (int)(t < run_min ? run_min : (t > run_max ? run_max : t));
        }
        s_cand[tid] = run;
    }
    __syncthreads();
    // the workgroup total of every candidate: the waves share the candidates, a wave's lanes the signals
    const int nc = rows ? 2 : NC;
    for (int i = wave; i < nc; i += 16) {
        const int run = s_cand[i];
        const float inv = __builtin_amdgcn_rcpf((float)run);
        long long g = 0;
        for (int s = lane; s < n_sig; s += 64) g += wg_map_count(s_hops[s], run, inv, lw);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) g += __shfl_xor(g, d);
        if (lane == 0) s_G[i] = g;
    }
    __syncthreads();
    if (wave == 0) {                                // lane i: candidate i; the cheapest wins, the lower index on ties
        long long key = 0x7FFFFFFFFFFFFFFFLL;
        if (lane < nc) {
            const long long G = s_G[lane];
            const long long rounds = (long long)ceil((double)(G << lw) / (double)slots);
            const long long cost = rows ? (lane == 0) : rounds * (s_cand[lane] + warm);
            if (G <= max_wgs) key = (cost << 8) | lane;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const long long o = __shfl_xor(key, d);
            key = o < key ? o : key;
        }
        if (lane == 0) s_run = key == 0x7FFFFFFFFFFFFFFFLL ? host_run : s_cand[(int)(key & 255)];
    }
    __syncthreads();
    const int run = s_run;
    const float inv = __builtin_amdgcn_rcpf((float)run);
    int base = 0;
    for (int q = 0; q < PER && q * 1024 < n_sig; ++q) {
        const int cnt = wg_map_count(hops[q], run, inv, lw);
        int incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        int excl = base + incl - cnt;
        for (int w = 0; w < wave; ++w) excl += wtot[w];
        const int s = q * 1024 + tid;
        if (s < n_sig) wg_start[s] = excl;
        for (int w = 0; w < 16; ++w) base += wtot[w];
        __syncthreads();
    }
    if (tid == 0) {
        wg_start[n_sig] = base;
        wg_start[n_sig + 1] = run;
    }
}

template <typename R, int MODE>
static int apply_wave(Ctx *c, ApplyArgs<R> a, int64_t n_sig, int64_t Tmax) {
    // binary32: eight waves (SUBTRACT, two per SIMD) / four (GAIN, three workgroups per CU); binary64: a wave's
    // exchange buffer and parked spectrum are 34 KB: four waves, one per SIMD
    constexpr int W = sizeof(R) == 8 ? 4 : (MODE == APPLY_SUBTRACT ? 8 : 4);
    constexpr int WAVES_PER_CU = sizeof(R) == 8 ? W : (MODE == APPLY_SUBTRACT ? 8 : 12);
    const int64_t out_hops = Tmax - 1;
    if (out_hops <= 0) return ADN_OK;
    // Runs per signal and hops per run.  The waves of a workgroup share one signal's noise / gain row, so a signal
    // takes a multiple of W runs (runs beyond the signal's end exit at once), and a run pays three warm-up frames
    // (and two analysis-only frames in SUBTRACT mode), so it is not made shorter than 16 hops.  A batch whose runs
    // fill the chip's wave slots exactly once is the best case (64 signals of 5624 hops: 32 runs of 176 hops; two
    // rounds measured 3 % slower); for other batch sizes the rounding to W runs per signal would leave a second,
    // nearly empty round (96 signals: 24 runs of 264 hops = 2304 waves for 2048 slots, twice the time), so the
    // number of rounds is chosen with the runs sized to fill them: the fewest hop-steps per wave slot win.
    const int64_t slots = (int64_t)c->num_cus * WAVES_PER_CU;
    const int64_t warm = MODE == APPLY_SUBTRACT ? 5 : 3;
    int64_t run = 0, segs = 0, best = -1;
    for (int64_t r = 1; r <= 8; ++r) {
        int64_t sg_try = slots * r / n_sig / W * W;                     // runs per signal that fit r rounds
        sg_try = sg_try < W ? W : sg_try;
        int64_t run_try = ceil_div(out_hops, sg_try);
        run_try = run_try < 16 ? 16 : (run_try > 1024 ? 1024 : run_try);
        sg_try = ceil_div(ceil_div(out_hops, run_try), (int64_t)W) * W;
        const int64_t cost = ceil_div(n_sig * sg_try, slots) * (run_try + warm);
        if (best < 0 || cost < best) {
            best = cost;
            run = run_try;
            segs = sg_try;
        }
    }
    a.seg_hops = (int)run;
    a.segs_per_sig = segs;
    a.n_sig = n_sig;
    // one 2 KB line that takes the stores of hops which are not stored by the kernel's fast path
    if (c->ws_used + APPLY_WS_BYTES > c->ws_bytes) {
        set_error("internal: the caller of the fused STFT kernel did not reserve its scratch line");
        return ADN_ERR_NOMEM;
    }
    a.dummy = ws_take<R>(c, 4096 / sizeof(R));
    // signals of different lengths: the runs are sized and handed out on the device, where the lengths are; the launch
    // is sized for the host's choice (every signal as long as the longest) plus the rounding of one workgroup a signal
    int *wg_start = ws_take<int>(c, APPLY_MAP_MAX + 2);
    a.wg_start = nullptr;
    int64_t wgs = ceil_div(n_sig * a.segs_per_sig, (int64_t)W);
    if (n_sig > 1 && n_sig <= APPLY_MAP_MAX && wgs + n_sig < (int64_t)0x7FFFFFFF && Tmax < (1 << 24)) {
        wgs += n_sig;
        static_assert(W == 4 || W == 8, "the map kernel takes log2(W)");
        ADN_LAUNCH(k_apply_wg_map, dim3(1), dim3(1024), 0, c->stream, a.sg, (int)n_sig, a.seg_hops, W == 8 ? 3 : 2, (int)slots,
                   (int)warm, (int)wgs, wg_start, 0, 16, 1024);
        a.wg_start = wg_start;
    }
    const size_t lds = fast::apply_wave_lds(W, MODE, sizeof(R));
    a.dbg = nullptr;
    if (c->ablate & 16) {       // diagnostics: per-phase cycle totals, printed after the launch
        ADN_HIP(hipMalloc(&a.dbg, 8 * sizeof(unsigned long long)));
        ADN_HIP(hipMemset(a.dbg, 0, 8 * sizeof(unsigned long long)));
    }
    auto kern = a.sg.n_ch == 2 ? (a.dbg ? fast::k_apply_wave<R, W, MODE, 2, true> : fast::k_apply_wave<R, W, MODE, 2, false>)
                               : (a.dbg ? fast::k_apply_wave<R, W, MODE, 1, true> : fast::k_apply_wave<R, W, MODE, 1, false>);
    ADN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ADN_REQUIRE(wgs < (int64_t)0x7FFFFFFF, "batch too large for one launch");
    ADN_LAUNCH_NAMED((MODE == APPLY_SUBTRACT ? "k_apply_wave<SUBTRACT>" : "k_apply_wave<GAIN>"), kern, dim3((unsigned)wgs),
                     dim3(W * 64), lds, c->stream, a);
    ADN_HIP(hipGetLastError());
    if (a.dbg) {
        unsigned long long hp[8];
        ADN_HIP(hipStreamSynchronize(c->stream));
        ADN_HIP(hipMemcpy(hp, a.dbg, sizeof(hp), hipMemcpyDeviceToHost));
        ADN_HIP(hipFree(a.dbg));
        const double fr = (double)n_sig * (double)out_hops;
        fprintf(stderr, "k_apply_wave<%d> shader clocks per frame and wave: load+window %.0f fft %.0f natural+split %.0f modify %.0f "
                "merge+reload %.0f ifft %.0f window+ola+emit %.0f (run %d hops)\n", MODE, hp[0] / fr, hp[1] / fr, hp[2] / fr,
                hp[3] / fr, hp[4] / fr, hp[5] / fr, hp[6] / fr, a.seg_hops);
    }
    return ADN_OK;
}

// binary32, n_fft 2048, hop 512, gain (no smoothing) or subtraction with the reference's 5 x 3 smoothing: the shapes the
// one-wave-per-run kernel is built for; every other shape runs the shape-generic kernels
template <int MODE>
int apply_fast(Ctx *c, ApplyArgs<float> a, int64_t n_sig, int64_t Tmax) {
    return apply_wave<float, MODE>(c, a, n_sig, Tmax);
}

bool apply_fast_ok(int n_fft, int hop, int st, int sf) {
    return n_fft == 2048 && hop == 512 && ((st == 1 && sf == 1) || (st == fast::FAST_ST && sf == 5));
}

int rows_fast(Ctx *c, const float *d_x, const Signals &sg, const Tables<float> &tb, int64_t n_sig,
              int64_t Tmax, int power, float *S, unsigned long long *sig_max) {
    constexpr int W = 8, FT = 16;
    const size_t lds = (size_t)W * fast::WBUF * sizeof(float2) + (size_t)(1025 * (FT + 1) + 1) * sizeof(float) +
                       (size_t)fast::TWTAB * sizeof(float2) + 2048 * sizeof(float);
    const int64_t groups = ceil_div(Tmax, FT);
    auto kern = fast::k_rows_fast<W>;
    ADN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t wgs = ceil_div(groups, (int64_t)fast::ROWS_GPW);
    ADN_LAUNCH_NAMED("k_rows_fast", kern, dim3((unsigned)(wgs * n_sig)), dim3(W * 64), lds, c->stream, d_x, sg, tb,
                       groups, FT, Tmax, power, S, sig_max);
    ADN_HIP(hipGetLastError());
    return ADN_OK;
}

bool rows_wave_ok(const Signals &sg, int n_fft) { return n_fft == 2048 && sg.hop == 512; }
int64_t rows_wave_sig_floats(int64_t Tmax) { return ceil_div(Tmax, (int64_t)4) * ROWS_GROUP_STRIDE * 4; }
int64_t rows_wave_vad_max_frame() { return 4 * 4 * 512; }      // the shortest run: four groups of four hops

// Rows in groups of four frames: S[(s * sig_stride) + ((t / 4) * ROWS_GROUP_STRIDE + k) * 4 + t % 4];
// sig_stride >= rows_wave_sig_floats(Tmax).
// The caller has reserved APPLY_WS_BYTES of workspace beyond what it took.
template <typename R>
int rows_wave(Ctx *c, const R *d_x, const Signals &sg, const Tables<R> &tb, int64_t n_sig, int64_t Tmax,
              int power, R *S, int64_t sig_stride, double *vad_e, int64_t frame_len, int64_t nv_max,
              unsigned long long *sig_max) {
    constexpr int W = 4;
    if (c->ws_used + APPLY_WS_BYTES > c->ws_bytes) {
        set_error("internal: the caller of the rows kernel did not reserve its scratch line");
        return ADN_ERR_NOMEM;
    }
    R *dummy = ws_take<R>(c, 4096 / sizeof(R));
    int *wave_start = ws_take<int>(c, APPLY_MAP_MAX + 2);
    const int64_t groups = ceil_div(Tmax, (int64_t)4);
    // about two rounds of runs over the waves a full chip holds
    const int64_t slots = (int64_t)c->num_cus * (sizeof(R) == 8 ? 4 : 12);
    int64_t gpr = ceil_div(n_sig * groups, slots * 2);
    gpr = gpr < 4 ? 4 : (gpr > 64 ? 64 : gpr);
    const int64_t runs_per_sig = ceil_div(groups, gpr);
    int64_t waves = n_sig * runs_per_sig;
    // signals of different lengths: the run is sized on the device by the same rule from the sum of the lengths
    // (shorter runs than the longest signal alone suggests), which takes at most this many waves
    const int64_t waves_dev = (2 * slots > n_sig * ceil_div(groups, (int64_t)64) ? 2 * slots : n_sig * ceil_div(groups, (int64_t)64)) + n_sig;
    const bool mapped = n_sig > 1 && n_sig <= APPLY_MAP_MAX && Tmax < (1 << 24) && (waves > waves_dev ? waves : waves_dev) < (int64_t)0x7FFFFFFF;
    if (mapped) {
        waves = waves > waves_dev ? waves : waves_dev;
        ADN_LAUNCH(k_apply_wg_map, dim3(1), dim3(1024), 0, c->stream, sg, (int)n_sig, (int)gpr, 0, (int)slots, 0, (int)waves,
                   wave_start, 1, 4, 64);
    }
    const int64_t wgs = ceil_div(waves, (int64_t)W);
    ADN_REQUIRE(wgs < (int64_t)0x7FFFFFFF / W, "batch too large for one launch");
    const size_t lds = fast::rows_wave_lds(W, sizeof(R));
    // voice-activity energies on the way (vad_e: n_sig x nv_max sums, zero on entry): a voice-activity frame must
    // not be longer than a run, so that at most two runs hold parts of it
    if (vad_e) ADN_REQUIRE(frame_len >= 512 && frame_len <= rows_wave_vad_max_frame(), "internal: voice-activity frame of %lld samples", (long long)frame_len);
    auto kern = vad_e ? (sg.n_ch == 2 ? fast::k_rows_wave<R, W, 2, true> : fast::k_rows_wave<R, W, 1, true>)
                      : (sg.n_ch == 2 ? fast::k_rows_wave<R, W, 2, false> : fast::k_rows_wave<R, W, 1, false>);
    ADN_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ADN_LAUNCH_NAMED("k_rows_wave", kern, dim3((unsigned)wgs), dim3(W * 64), lds, c->stream, d_x, sg, tb, n_sig, runs_per_sig,
                     (int)gpr, sig_stride, power, S, dummy, vad_e, frame_len, nv_max, sig_max,
                     mapped ? (const int *)wave_start : (const int *)nullptr);
    ADN_HIP(hipGetLastError());
    return ADN_OK;
}

template int rows_wave<float>(Ctx *, const float *, const Signals &, const Tables<float> &, int64_t, int64_t, int, float *,
                              int64_t, double *, int64_t, int64_t, unsigned long long *);
template int rows_wave<double>(Ctx *, const double *, const Signals &, const Tables<double> &, int64_t, int64_t, int, double *,
                               int64_t, double *, int64_t, int64_t, unsigned long long *);

// binary64, n_fft 2048, hop 512: the one-wave-per-run kernels at the reference's precision
bool apply_wave64_ok(int n_fft, int hop, int st, int sf, int mode) {
    return n_fft == 2048 && hop == 512 && (mode == APPLY_GAIN || (sf == 5 && st == fast::FAST_ST));
}
template <int MODE>
int apply_wave64(Ctx *c, ApplyArgs<double> a, int64_t n_sig, int64_t Tmax) {
    return apply_wave<double, MODE>(c, a, n_sig, Tmax);
}
template int apply_wave64<APPLY_SUBTRACT>(Ctx *, ApplyArgs<double>, int64_t, int64_t);
template int apply_wave64<APPLY_GAIN>(Ctx *, ApplyArgs<double>, int64_t, int64_t);

template int apply_fast<APPLY_SUBTRACT>(Ctx *, ApplyArgs<float>, int64_t, int64_t);
template int apply_fast<APPLY_GAIN>(Ctx *, ApplyArgs<float>, int64_t, int64_t);

}  // namespace adn
