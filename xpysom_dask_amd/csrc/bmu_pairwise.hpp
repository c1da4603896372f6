// Fused distance + BMU argmin for the reference's non-GEMM distances (VALU kernels):
//   'manhattan' / 'manhattan_no_opt'   distances.py:138-158 (the one hand-written CUDA kernel of the
//                                      reference, `l1norm`, :109-135, is the GPU form of the same sum)
//   'norm_p_no_opt' and odd-p 'norm_p' norm_p_power_distance_generic, distances.py:61-75
//   even-p 'norm_p'                    norm_p_power_distance_even, distances.py:77-96
// without materialising the (n, K, D) difference tensor or the (n, K) matrix.
//
// Arithmetic order is the reference's: the generic form sums float32 terms |x-w|^p over the feature
// axis in NumPy's pairwise order; the even form is the binomial expansion, p+1 float32 dot products
// (k-ordered fma chains, as sgemm) each scaled by its signed binomial coefficient in float32 and
// accumulated in FLOAT64 (`acc = xp.zeros(...)` is float64, :87).  Integer powers 0, 1, 2 follow
// NumPy's exact fast paths (1, x, x*x); higher ones are formed in float64 and rounded once (NumPy's
// powf is correctly rounded), so p <= 2 is bit-exact and larger p agrees up to rare double roundings.
//
// One thread = one sample (its row in LDS, +1 padded), units broadcast from an LDS tile.
#pragma once
#include "som_common.hpp"

namespace somhip {

enum { PW_GENERIC = 0, PW_EVEN = 1 };
constexpr int PW_SAMPLES = 128;
constexpr int PW_UNITS = 16;

// v^q for a small non-negative integer q, rounded to float32 once.  q = 0, 1, 2 reproduce NumPy's
// exact fast paths (1, v, v*v: a product of two float32 is exact in float64); higher powers are
// formed in float64 and rounded once (NumPy's powf is correctly rounded; so is this, up to rare
// double roundings).
__device__ __forceinline__ float np_ipow_f32(float v, int q) {
    double acc = 1.0;
    const double r = (double)v;
    for (int i = 0; i < q; ++i) acc *= r;
    return (float)acc;
}

// v^p for a real exponent (activation_distance_kwargs={'p': 2.5}): NumPy's float32 power is the C library's powf, which is
// correctly rounded in all but rare cases; so is float64 pow rounded once to float32
__device__ __forceinline__ float np_rpow_f32(float v, double pr) { return (float)pow((double)v, pr); }
// pr != 0: the real exponent; else the integer one
__device__ __forceinline__ float np_pow_f32(float v, int p, double pr) { return pr != 0.0 ? np_rpow_f32(v, pr) : np_ipow_f32(v, p); }

// sum_{d<n} |x_d - w_d|^p in NumPy's float32 pairwise order (cf. np_pairwise_sq_sum in bmu_f32.hpp)
__device__ float np_pairwise_absdiff_pow(const float* __restrict__ x, const float* __restrict__ w, int n, int p, double pr) {
    if (n < 8) {
        float res = 0.0f;
        for (int i = 0; i < n; ++i) res = __fadd_rn(res, np_pow_f32(__builtin_fabsf(__fsub_rn(x[i], w[i])), p, pr));
        return res;
    }
    if (n <= 128) {
        float r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = np_pow_f32(__builtin_fabsf(__fsub_rn(x[j], w[j])), p, pr);
        int i = 8;
        for (; i < n - (n % 8); i += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                r[j] = __fadd_rn(r[j], np_pow_f32(__builtin_fabsf(__fsub_rn(x[i + j], w[i + j])), p, pr));
        }
        float res = __fadd_rn(__fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3])),
                              __fadd_rn(__fadd_rn(r[4], r[5]), __fadd_rn(r[6], r[7])));
        for (; i < n; ++i) res = __fadd_rn(res, np_pow_f32(__builtin_fabsf(__fsub_rn(x[i], w[i])), p, pr));
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return __fadd_rn(np_pairwise_absdiff_pow(x, w, n2, p, pr), np_pairwise_absdiff_pow(x + n2, w + n2, n - n2, p, pr));
}

// sum_e (-1)^e C(p,e) dot(x^(p-e), w^e), float64 accumulation of float32 terms.  All p+1 dot
// products advance together over the features (each is still its own k-ordered fma chain).
// (The e-outer / d-inner nesting of the reference miscompiled under hipcc 7.2 -O3 for p >= 4 --
// every distance compared equal -- hence this loop order.)
constexpr int PW_MAX_P = 16;

__device__ double norm_p_even(const float* __restrict__ x, const float* __restrict__ w, int n, int p) {
    float dot[PW_MAX_P + 1];
    for (int e = 0; e <= p; ++e) dot[e] = 0.0f;
    for (int d = 0; d < n; ++d) {
        float xp[PW_MAX_P + 1], wp[PW_MAX_P + 1];
        for (int q = 0; q <= p; ++q) { xp[q] = np_ipow_f32(x[d], q); wp[q] = np_ipow_f32(w[d], q); }
        for (int e = 0; e <= p; ++e) dot[e] = __builtin_fmaf(xp[p - e], wp[e], dot[e]);
    }
    double acc = 0.0;
    double binom = 1.0;                                   // C(p, e), exact in float64
    for (int e = 0; e <= p; ++e) {
        const float coef = (float)((e & 1) ? -binom : binom);
        acc += (double)__fmul_rn(coef, dot[e]);
        binom = __builtin_rint(binom * (double)(p - e) / (double)(e + 1));
    }
    return acc;
}

template <int KIND>
__global__ __launch_bounds__(PW_SAMPLES) void bmu_pairwise_kernel(const float* __restrict__ X, long N, int D,
                                                                  const float* __restrict__ W, int K, int p,
                                                                  int x_in_lds, int* __restrict__ out, double pr = 0.0) {
    extern __shared__ __attribute__((aligned(16))) float smem_pw[];
    float* Ws = smem_pw;                       // [PW_UNITS][D]
    float* Xs = Ws + PW_UNITS * D;             // [PW_SAMPLES][D+1] when x_in_lds
    const int tid = threadIdx.x;
    const long s0 = (long)blockIdx.x * PW_SAMPLES;
    const long row = s0 + tid;
    const bool live = row < N;
    if (x_in_lds) {
        for (int idx = tid; idx < PW_SAMPLES * D; idx += PW_SAMPLES) {
            int r = idx / D, k = idx - r * D;
            Xs[r * (D + 1) + k] = (s0 + r < N) ? X[(s0 + r) * (long)D + k] : 0.0f;
        }
    }
    const float* x = x_in_lds ? Xs + tid * (D + 1) : X + (live ? row : 0) * (long)D;

    double best = __builtin_inf();
    int bidx = 0;
    for (int u0 = 0; u0 < K; u0 += PW_UNITS) {
        __syncthreads();
        for (int idx = tid; idx < PW_UNITS * D; idx += PW_SAMPLES) {
            int r = idx / D, k = idx - r * D;
            Ws[idx] = (u0 + r < K) ? W[(long)(u0 + r) * D + k] : 0.0f;
        }
        __syncthreads();
        const int nu = (K - u0 < PW_UNITS) ? (K - u0) : PW_UNITS;
        for (int r = 0; r < nu; ++r) {
            const float* w = Ws + r * D;
            double v = KIND == PW_EVEN ? norm_p_even(x, w, D, p) : (double)np_pairwise_absdiff_pow(x, w, D, p, pr);
            if (v < best) { best = v; bidx = u0 + r; }
        }
    }
    if (live) out[row] = bidx;
}

// float64 query rows against the float32 codebook: XPySom.winner() does not coerce its input (xpysom.py:379-396), so a
// float64 x makes NumPy compute -2 x.w^T + w_sq in float64 (dgemm on the upcast weights, the float32 w_sq promoted):
//   score(n, k) = fl64(-2 c + |w_k|^2_f32),  c = the k-ordered float64 fma chain of x_n . w_k,
// first minimum in unit order.  An analysis path (vector ALU, one thread per row, 16-unit LDS tiles).
__global__ __launch_bounds__(PW_SAMPLES) void bmu_f64_kernel(const double* __restrict__ X, long N, int D,
                                                             const float* __restrict__ W, const float* __restrict__ wsq,
                                                             int K, int* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem_pw[];
    float* Ws = smem_pw;                       // [PW_UNITS][D]
    const int tid = threadIdx.x;
    const long row = (long)blockIdx.x * PW_SAMPLES + tid;
    const bool live = row < N;
    const double* x = X + (live ? row : 0) * (long)D;
    double best = __builtin_inf();
    int bidx = 0;
    for (int u0 = 0; u0 < K; u0 += PW_UNITS) {
        __syncthreads();
        for (int idx = tid; idx < PW_UNITS * D; idx += PW_SAMPLES) {
            const int r = idx / D, k = idx - r * D;
            Ws[idx] = (u0 + r < K) ? W[(long)(u0 + r) * D + k] : 0.0f;
        }
        __syncthreads();
        const int nu = (K - u0 < PW_UNITS) ? (K - u0) : PW_UNITS;
        for (int r = 0; r < nu; ++r) {
            const float* w = Ws + r * D;
            double c = 0.0;
            for (int d = 0; d < D; ++d) c = __builtin_fma(x[d], (double)w[d], c);
            const double v = __builtin_fma(-2.0, c, (double)wsq[u0 + r]);
            if (v < best) { best = v; bidx = u0 + r; }
        }
    }
    if (live) out[row] = bidx;
}

}  // namespace somhip
