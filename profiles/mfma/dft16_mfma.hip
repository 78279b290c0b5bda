// Bounded experiment: one radix-16 DFT stage of the 1024-point transform (64 independent 16-point complex DFTs
// per wave, the unit of work of wave_fft1024's register passes) done
//   (a) as it is today: radix-4 butterflies on the vector ALUs, 16 complex values per lane;
//   (b) as dense 16 x 16 matrix products on the matrix cores in binary32 (v_mfma_f32_16x16x4_f32), given its best
//       case: operands already in the instruction's own layout (no shuffles counted), the four real products
//       Yr = Wr Xr - Wi Xi, Yi = Wi Xr + Wr Xi as 4 x 16 MFMA instructions per wave and stage;
//   (c) the same products in bf16 with the three-way split of both operands that keeps binary32 accuracy
//       (six partial products, v_mfma_f32_16x16x16_bf16), conversions counted.
// Prints shader clocks per stage and wave (one wave per SIMD resident: issue-bound best case) and the largest
// absolute error against a binary64 DFT.   hipcc --offload-arch=gfx950 -O3 dft16_mfma.hip -o dft16_mfma
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
struct c2 { float x, y; };
__device__ inline c2 operator+(c2 a, c2 b) { return {a.x + b.x, a.y + b.y}; }
__device__ inline c2 operator-(c2 a, c2 b) { return {a.x - b.x, a.y - b.y}; }
__device__ inline c2 mulw(c2 v, float c, float s) { return {v.x * c + v.y * s, v.y * c - v.x * s}; }   // v * (c - i s)
__device__ inline void bfly4(c2 &a0, c2 &a1, c2 &a2, c2 &a3) {
    const c2 s02 = a0 + a2, d02 = a0 - a2, s13 = a1 + a3, d13 = a1 - a3;
    a0 = s02 + s13;
    a1 = c2{d02.x + d13.y, d02.y - d13.x};
    a2 = s02 - s13;
    a3 = c2{d02.x - d13.y, d02.y + d13.x};
}
__device__ inline void dft16(c2 (&r)[16]) {
    for (int i = 0; i < 4; ++i) bfly4(r[i], r[4 + i], r[8 + i], r[12 + i]);
    const float C1 = 0.9238795325112867f, S1 = 0.3826834323650898f, RH = 0.7071067811865476f;
    r[5] = mulw(r[5], C1, S1); r[6] = mulw(r[6], RH, RH); r[7] = mulw(r[7], S1, C1);
    r[9] = mulw(r[9], RH, RH); r[10] = mulw(r[10], 0.f, 1.f); r[11] = mulw(r[11], -RH, RH);
    r[13] = mulw(r[13], S1, C1); r[14] = mulw(r[14], -RH, RH); r[15] = mulw(r[15], -C1, -S1);
    for (int i = 0; i < 4; ++i) bfly4(r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]);
}

// (a) vector ALUs: x[(n * 64 + lane) * 2] -> X[k] at the register order of the production kernel (4 (k & 3) + (k >> 2))
__global__ void k_valu(const float *x, float *y, long long *clk, int reps) {
    const int lane = threadIdx.x;
    c2 r[16];
    for (int n = 0; n < 16; ++n) r[n] = {x[(n * 64 + lane) * 2], x[(n * 64 + lane) * 2 + 1]};
    const long long t0 = clock64();
    for (int it = 0; it < reps; ++it) {
        dft16(r);
        if (it + 1 < reps)
            for (int n = 0; n < 16; ++n) r[n] = {r[n].x * 0.25f, r[n].y * 0.25f};   // keep the values bounded
    }
    const long long t1 = clock64();
    for (int k = 0; k < 16; ++k) {
        const c2 v = r[4 * (k & 3) + (k >> 2)];
        y[(k * 64 + lane) * 2] = v.x;
        y[(k * 64 + lane) * 2 + 1] = v.y;
    }
    if (lane == 0) clk[blockIdx.x] = t1 - t0;
}

// (b) binary32 matrix cores.  D(16x16) += A(16x4) B(4x16): lane l holds A[l % 16][l / 16], B[l / 16][l % 16],
// D[4 (l / 16) + i][l % 16] in d[i].  W is 16 x 16 (rows k, columns n), X is 16 x 64 (rows n, columns = the wave's
// 64 transforms): four column blocks of 16, four steps of K = 4.
__global__ void k_mfma32(const float *x, float *y, long long *clk, int reps) {
    const int lane = threadIdx.x, lr = lane & 15, lq = lane >> 4;
    float wr[4], wi[4], xr[4][4], xi[4][4];        // [k step] and [column block][k step]
    for (int s = 0; s < 4; ++s) {
        const int n = 4 * s + lq, k = lr;
        const double a = -2.0 * M_PI * (double)((n * k) & 15) / 16.0;
        wr[s] = (float)cos(a);
        wi[s] = (float)sin(a);
        for (int cb = 0; cb < 4; ++cb) {
            xr[cb][s] = x[(n * 64 + 16 * cb + lr) * 2];
            xi[cb][s] = x[(n * 64 + 16 * cb + lr) * 2 + 1];
        }
    }
    f4 yr[4], yi[4];
    const long long t0 = clock64();
    for (int it = 0; it < reps; ++it) {
        for (int cb = 0; cb < 4; ++cb) {
            f4 ar = {0, 0, 0, 0}, ai = {0, 0, 0, 0};
            for (int s = 0; s < 4; ++s) {
                ar = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[s], xr[cb][s], ar, 0, 0, 0);
                ar = __builtin_amdgcn_mfma_f32_16x16x4f32(-wi[s], xi[cb][s], ar, 0, 0, 0);
                ai = __builtin_amdgcn_mfma_f32_16x16x4f32(wi[s], xr[cb][s], ai, 0, 0, 0);
                ai = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[s], xi[cb][s], ai, 0, 0, 0);
            }
            yr[cb] = ar;
            yi[cb] = ai;
        }
        if (it + 1 < reps)          // feed the result back (best case: pretend the output layout is the input layout)
            for (int cb = 0; cb < 4; ++cb)
                for (int s = 0; s < 4; ++s) {
                    xr[cb][s] = yr[cb][s] * 0.25f;
                    xi[cb][s] = yi[cb][s] * 0.25f;
                }
    }
    const long long t1 = clock64();
    for (int cb = 0; cb < 4; ++cb)
        for (int i = 0; i < 4; ++i) {
            const int k = 4 * lq + i;
            y[(k * 64 + 16 * cb + lr) * 2] = yr[cb][i];
            y[(k * 64 + 16 * cb + lr) * 2 + 1] = yi[cb][i];
        }
    if (lane == 0) clk[blockIdx.x] = t1 - t0;
}

// (c) bf16 matrix cores, K = 16 per instruction: lane l holds A[l % 16][8 (l / 16) ... + 8) and B[8 (l / 16) ...][l % 16]
// (v_mfma_f32_16x16x32_bf16 on this chip: K = 32, the upper half of K padded with zeros here would waste half of it, so
// two transforms' worth of K is not available: one instruction per product and column block, K = 16 of 32 used).
__device__ inline void split3(float v, __bf16 &a, __bf16 &b, __bf16 &c) {
    a = (__bf16)v;
    const float r1 = v - (float)a;
    b = (__bf16)r1;
    c = (__bf16)(r1 - (float)b);
}
__global__ void k_mfma_bf16(const float *x, float *y, long long *clk, int reps) {
    const int lane = threadIdx.x, lr = lane & 15, lq = lane >> 4;
    // W rows k = lr, columns n = 8 lq + j (j < 8; lanes with lq >= 2 hold the zero padding of K = 32)
    bf8 wr[3], wi[3], wni[3];
    for (int j = 0; j < 8; ++j) {
        const int n = 8 * lq + j;
        float cr = 0.f, ci = 0.f;
        if (n < 16) {
            const double a = -2.0 * M_PI * (double)((n * lr) & 15) / 16.0;
            cr = (float)cos(a);
            ci = (float)sin(a);
        }
        __bf16 a0, a1, a2;
        split3(cr, a0, a1, a2); wr[0][j] = a0; wr[1][j] = a1; wr[2][j] = a2;
        split3(ci, a0, a1, a2); wi[0][j] = a0; wi[1][j] = a1; wi[2][j] = a2;
        wni[0][j] = -wi[0][j]; wni[1][j] = -wi[1][j]; wni[2][j] = -wi[2][j];
    }
    float xr[4][8], xi[4][8];
    for (int cb = 0; cb < 4; ++cb)
        for (int j = 0; j < 8; ++j) {
            const int n = 8 * lq + j;
            xr[cb][j] = n < 16 ? x[(n * 64 + 16 * cb + lr) * 2] : 0.f;
            xi[cb][j] = n < 16 ? x[(n * 64 + 16 * cb + lr) * 2 + 1] : 0.f;
        }
    f4 yr[4], yi[4];
    const long long t0 = clock64();
    for (int it = 0; it < reps; ++it) {
        for (int cb = 0; cb < 4; ++cb) {
            bf8 br[3], bi[3];
            for (int j = 0; j < 8; ++j) {
                __bf16 a0, a1, a2;
                split3(xr[cb][j], a0, a1, a2); br[0][j] = a0; br[1][j] = a1; br[2][j] = a2;
                split3(xi[cb][j], a0, a1, a2); bi[0][j] = a0; bi[1][j] = a1; bi[2][j] = a2;
            }
            f4 ar = {0, 0, 0, 0}, ai = {0, 0, 0, 0};
            // partial products whose weight is at least 2^-16 of the leading one: (0,0) (0,1) (1,0) (0,2) (1,1) (2,0)
            const int pa[6] = {2, 1, 0, 1, 0, 0}, pb[6] = {0, 1, 2, 0, 1, 0};     // small terms first
            for (int p = 0; p < 6; ++p) {
                ar = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[pa[p]], br[pb[p]], ar, 0, 0, 0);
                ar = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wni[pa[p]], bi[pb[p]], ar, 0, 0, 0);
                ai = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wi[pa[p]], br[pb[p]], ai, 0, 0, 0);
                ai = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[pa[p]], bi[pb[p]], ai, 0, 0, 0);
            }
            yr[cb] = ar;
            yi[cb] = ai;
        }
        if (it + 1 < reps)
            for (int cb = 0; cb < 4; ++cb)
                for (int j = 0; j < 8; ++j) {
                    xr[cb][j] = yr[cb][j & 3] * 0.25f;
                    xi[cb][j] = yi[cb][j & 3] * 0.25f;
                }
    }
    const long long t1 = clock64();
    for (int cb = 0; cb < 4; ++cb)
        for (int i = 0; i < 4; ++i) {
            const int k = 4 * lq + i;
            y[(k * 64 + 16 * cb + lr) * 2] = yr[cb][i];
            y[(k * 64 + 16 * cb + lr) * 2 + 1] = yi[cb][i];
        }
    if (lane == 0) clk[blockIdx.x] = t1 - t0;
}

int main() {
    const int N = 16 * 64 * 2;
    std::vector<float> hx(N), hy(N);
    unsigned s = 12345u;
    for (int i = 0; i < N; ++i) {
        s = s * 1664525u + 1013904223u;
        hx[i] = ((float)(s >> 8) / 8388608.0f - 1.0f) * 0.5f;
    }
    std::vector<double> want(N);
    for (int col = 0; col < 64; ++col)
        for (int k = 0; k < 16; ++k) {
            double re = 0, im = 0;
            for (int n = 0; n < 16; ++n) {
                const double a = -2.0 * M_PI * (double)(n * k) / 16.0, xr = hx[(n * 64 + col) * 2], xi = hx[(n * 64 + col) * 2 + 1];
                re += xr * cos(a) - xi * sin(a);
                im += xr * sin(a) + xi * cos(a);
            }
            want[(k * 64 + col) * 2] = re;
            want[(k * 64 + col) * 2 + 1] = im;
        }
    float *dx, *dy;
    long long *dclk;
    const int blocks = 1024;                         // one wave per workgroup; four per CU = one per SIMD
    hipMalloc(&dx, N * 4); hipMalloc(&dy, N * 4); hipMalloc(&dclk, blocks * 8);
    hipMemcpy(dx, hx.data(), N * 4, hipMemcpyHostToDevice);
    std::vector<long long> hc(blocks);
    const char *names[3] = {"vector ALUs, radix-4 butterflies (today)", "matrix cores, binary32 16x16x4",
                            "matrix cores, bf16 16x16x32, 3-way split x 6 products"};
    for (int v = 0; v < 3; ++v) {
        double cyc[2];
        for (int pass = 0; pass < 2; ++pass) {       // reps 1 for the error, 1 and 65 for the cycles per stage
            const int reps = pass ? 65 : 1;
            for (int rnd = 0; rnd < 2; ++rnd) {
                if (v == 0) hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(64), 0, 0, dx, dy, dclk, reps);
                if (v == 1) hipLaunchKernelGGL(k_mfma32, dim3(blocks), dim3(64), 0, 0, dx, dy, dclk, reps);
                if (v == 2) hipLaunchKernelGGL(k_mfma_bf16, dim3(blocks), dim3(64), 0, 0, dx, dy, dclk, reps);
                hipDeviceSynchronize();
            }
            hipMemcpy(hc.data(), dclk, blocks * 8, hipMemcpyDeviceToHost);
            double sum = 0;
            for (long long c : hc) sum += (double)c;
            cyc[pass] = sum / blocks;
            if (pass == 0) {
                hipMemcpy(hy.data(), dy, N * 4, hipMemcpyDeviceToHost);
                double err = 0;
                for (int i = 0; i < N; ++i) err = fmax(err, fabs((double)hy[i] - want[i]));
                printf("%-58s max-abs error vs binary64 %.3g", names[v], err);
            }
        }
        printf("   shader clocks per stage and wave %.0f\n", (cyc[1] - cyc[0]) / 64.0);
    }
    return 0;
}
