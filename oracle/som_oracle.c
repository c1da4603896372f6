/*
 * som_oracle.c -- scalar C restatement of the batch-SOM hot loop.  TEST INFRASTRUCTURE ONLY:
 * built by oracle/Makefile into oracle/_build/libsomoracle.so and used by tests/ as a second,
 * independent checker next to oracle/som_oracle.py.  Nothing under xpysom_dask_amd/ links it.
 *
 * Parity status: PINNED -- tests/test_oracle_golden.py checks these functions against the
 * golden vectors captured from the reference (tests/golden/, oracle/make_golden.py).
 *
 * What it states, with the float32 operation ORDER made explicit (the NumPy oracle leaves
 * that to BLAS):
 *   row_sq      xp.power(a,2).sum(axis=1)            xpysom.py:529-537, distances.py:20-21,30
 *               -- squares rounded to float32, then NumPy's pairwise summation order
 *   cross       xp.dot(x, w.T)                        distances.py:22
 *               -- k-sequential fused multiply-add chain from 0 (what OpenBLAS' sgemm
 *                  micro-kernels do for one K block, and what v_mfma_f32_32x32x2_f32 does)
 *   bmu         argmin(-2*cross + w_sq)               distances.py:23, xpysom.py:416
 *   update      gaussian_rect * eta, sum_g, g^T x     neighborhoods.py:14-33, xpysom.py:434-441
 *   merge       where(den != 0, num/den, W)           xpysom.py:446-455
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* NumPy's pairwise float32 summation (numpy/_core/src/umath/loops_utils.h.src, published
 * algorithm): n < 8 sequential; n <= 128 eight strided accumulators combined as a tree and a
 * sequential tail; larger n split in two halves (first half rounded down to a multiple of 8). */
static float pairwise_sum_f32(const float* a, long n) {
    if (n < 8) {
        float res = 0.0f;
        for (long i = 0; i < n; ++i) res += a[i];
        return res;
    } else if (n <= 128) {
        float r[8];
        long i;
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    } else {
        long n2 = n / 2;
        n2 -= n2 % 8;
        return pairwise_sum_f32(a, n2) + pairwise_sum_f32(a + n2, n - n2);
    }
}

void oracle_row_sq_f32(const float* a, long rows, int D, float* out) {
    float* sq = (float*)malloc(sizeof(float) * (size_t)(D > 0 ? D : 1));
    for (long r = 0; r < rows; ++r) {
        for (int k = 0; k < D; ++k) { volatile float s = a[r * D + k] * a[r * D + k]; sq[k] = s; }
        out[r] = pairwise_sum_f32(sq, D);
    }
    free(sq);
}

void oracle_cross_f32(const float* x, const float* w, long N, long K, int D, float* out) {
    for (long n = 0; n < N; ++n)
        for (long k = 0; k < K; ++k) {
            float acc = 0.0f;
            for (int d = 0; d < D; ++d) acc = fmaf(x[n * D + d], w[k * D + d], acc);
            out[n * K + k] = acc;
        }
}

/* raveled BMU ids for the 'euclidean' activation distance, first minimum wins */
void oracle_bmu_euclid_f32(const float* x, const float* w, const float* wsq, long N, long K, int D, int32_t* ids) {
    for (long n = 0; n < N; ++n) {
        float best = INFINITY;
        int32_t bi = 0;
        for (long k = 0; k < K; ++k) {
            float acc = 0.0f;
            for (int d = 0; d < D; ++d) acc = fmaf(x[n * D + d], w[k * D + d], acc);
            float v = fmaf(-2.0f, acc, wsq[k]);
            if (v < best) { best = v; bi = (int32_t)k; }
        }
        ids[n] = bi;
    }
}

/* one mini-batch of XPySom._update with gaussian_rect (no compact support), given BMU ids.
 * wide != 0: float64 neighbourhood (numpy.float64 sigma); else float32.  num [K][D], den [K]
 * are float64 outputs of this batch (the caller casts/accumulates into float32 buffers as the
 * reference does on `+=`, xpysom.py:568-569). */
void oracle_update_gaussian(const float* x, const int32_t* bmu, long N, int X, int Y, int D, double sigma,
                            double eta, double std_coeff, int wide, double* num, double* den) {
    const long K = (long)X * Y;
    const double d = 2.0 * (std_coeff * std_coeff) * (sigma * sigma);
    memset(num, 0, sizeof(double) * (size_t)(K * D));
    memset(den, 0, sizeof(double) * (size_t)K);
    double* ax = (double*)malloc(sizeof(double) * (size_t)X);
    double* ay = (double*)malloc(sizeof(double) * (size_t)Y);
    for (long n = 0; n < N; ++n) {
        int ci = bmu[n] / Y, cj = bmu[n] % Y;
        for (int i = 0; i < X; ++i) {
            float p = (float)((i - ci) * (i - ci));
            ax[i] = wide ? exp(-(double)p / d) : (double)expf(-p / (float)d);
        }
        for (int j = 0; j < Y; ++j) {
            float p = (float)((j - cj) * (j - cj));
            ay[j] = wide ? exp(-(double)p / d) : (double)expf(-p / (float)d);
        }
        for (int i = 0; i < X; ++i)
            for (int j = 0; j < Y; ++j) {
                double g = wide ? (ax[i] * ay[j]) * eta : (double)(((float)ax[i] * (float)ay[j]) * (float)eta);
                long k = (long)i * Y + j;
                den[k] += g;
                for (int c = 0; c < D; ++c) num[k * D + c] += g * (double)x[n * D + c];
            }
    }
    free(ax);
    free(ay);
}

void oracle_merge_f32(float* w, const float* num, const float* den, long K, int D) {
    for (long k = 0; k < K; ++k)
        if (den[k] != 0.0f)
            for (int c = 0; c < D; ++c) w[k * D + c] = num[k * D + c] / den[k];
}
