// bf16 operand preparation shared by the bf16 BMU kernels (bmu_bf16_k16.hpp, bmu_bf16_tiled.hpp).
//
// Same reference chain as bmu_f32.hpp (distances.py:11-23 + xpysom.py:416), computed as
//     d'(n,k) = B + |w~_k|^2/2 - x~_n . w~_k          (argmin_k d' == argmin_k |x~_n - w~_k|^2)
// on bf16-rounded x~, w~ with float32 accumulation in the MFMA.  The norm is taken from the SAME bf16 values as the
// products, so d' is a true distance in the rounded space.  B = max_n|x~_n| * max_k|w~_k| (Cauchy-Schwarz) makes every
// d' positive, so its bit pattern orders like an unsigned integer and B + |w~|^2/2 is simply the MFMA's initial
// accumulator, read from LDS straight into the C registers: no per-element VALU besides the argmin.
// Here: |w~|^2 and its maximum (prep_wnorm_kernel), the initial accumulators behind each stage's fragments
// (prep_wsqh_kernel), the bf16 row image and max|x~|^2 (prep_x_bf16_kernel), the LDS-DMA helper.
// (Round 1's first kernel, on v_mfma_f32_32x32x16_bf16, lived here; the 16x16x32 form holds a higher clock and
//  replaced it -- DESIGN.md 3.4.)
#pragma once
#include "som_common.hpp"

namespace somhip {

constexpr float BF_PAD_NORM = 1.0e30f;                  // |w|^2/2 of padding units: never wins

__device__ __forceinline__ void lds_dma_16(const void* gsrc, void* ldst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)ldst, 16, 0, 0);
}

// positive floats order like their bit patterns: a float max through an integer atomic
__device__ __forceinline__ void atomic_max_pos_f32(float* addr, float v) {
    // a million waves share one address: only a value that would raise the maximum pays for the atomic
    // (a NaN norm -- a NaN row or unit -- is left out: the offset B must stay finite for everybody else)
    if (v == v && __float_as_uint(v) > __hip_atomic_load((unsigned int*)addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax((unsigned int*)addr, __float_as_uint(v));
}

// precision 'exact' on IEEE half operands (bmu_exact.hpp): rows and units are scaled by powers of two so that the
// longest one has a norm in [2^13, 2^14) -- every element then fits float16 with room to spare, and only elements
// 2^-27 of the longest norm and smaller reach the subnormal range.  max2 = the largest squared norm of the set.
__device__ __forceinline__ float ex_scale(float max2) {
    const float m = __builtin_sqrtf(max2) * (1.0f + 1.0f / 1024.0f);
    if (!(m > 0.0f) || !(m < 3.0e38f)) return 1.0f;
    int e;
    (void)frexpf(m, &e);                                   // m = f * 2^e, f in [0.5, 1)
    e = 14 - e;
    e = e > 100 ? 100 : e < -100 ? -100 : e;
    return ldexpf(1.0f, e);
}

// where the exact mode's screens leave the mask of the rows whose minimum of group g they stored, for row block rb (64
// rows): [row block][group] -- the select kernel reads 64 groups' masks in one load ([group][row block] measured the same
// in the screen and no better in the select)
__device__ __forceinline__ long ex_flag_index(long rb, int g, int n_groups, long /*gm_stride*/) {
    return rb * n_groups + g;
}

// what the MFMA may see instead of the (scaled) float32 value f once it went through the 16-bit type as fb: the rounding
// error |f - fb|, or all of |f| when fb is a subnormal IEEE half (an MFMA that flushes subnormal inputs reads 0 there;
// one that does not errs by less: the larger of the two covers both)
__device__ __forceinline__ float half_operand_error(float f, float fb) {
    const float a = __builtin_fabsf(fb);
    return (a > 0.0f && a < 6.103515625e-05f) ? __builtin_fabsf(f) : __builtin_fabsf(f - fb);
}

// The exact mode's per-row error bound E(n) (derivation: bmu_exact.hpp), in scaled d' units:
//   E(n) = S (cA |x_n| wmax + cW wmax^2 + cB Bm) + cM (xerr_n w^max + (x^_n + xerr_n) werr),   S = sx sw.
// The screen (which group minima can still matter) and the scan (which do) evaluate it through this one function.
struct ExactBound {
    float cA, cW, cB, cM;
    int unit;                         // cosine: the operands are unit-length rows (|x_n| = wmax = 1 in the formula)
};
struct ExactScales {                  // per launch: from max|x|^2, max|w|^2, max_k |w^_k - w~_k|^2
    float sx, sw, wm, bmag, we, big;
};
__device__ __forceinline__ ExactScales ex_scales(const float* __restrict__ xmax2, const float* __restrict__ wmax2,
                                                 const float* __restrict__ werr2) {
    ExactScales s;
    s.sx = ex_scale(*xmax2);
    s.sw = ex_scale(*wmax2);
    s.wm = __builtin_sqrtf(*wmax2) * (1.0f + 1.0f / 1024.0f);
    const float big = __builtin_sqrtf(*wmax2) * __builtin_sqrtf(*xmax2) * (1.0f + 1.0f / 1024.0f);   // prep_wsqh_kernel's B
    s.big = big;
    s.bmag = 2.01f * big + 0.5f * s.wm * s.wm;
    s.we = __builtin_sqrtf(*werr2) * (1.0f + 1.0f / 1024.0f);
    return s;
}
// E(n), or a NaN for a row / codebook the bound does not cover (NaN or infinite norms; norms so small that float32
// products may underflow): such a row selects nothing and goes to the float32 kernel
__device__ __forceinline__ float ex_row_bound(const ExactBound& eb, const ExactScales& s, float xsq, float xerr) {
    float xn = __builtin_sqrtf(xsq) * (1.0f + 1.0f / 1024.0f);
    // cosine: unit-length operands; |x|^2 in the window that keeps |x|^2 |w|^2 a normal float32 (exact_werr_kernel)
    if (eb.unit) xn = (xsq > 0x1p-60f && xsq < 0x1p60f) ? 1.0f + 1.0f / 1024.0f : __builtin_nanf("");
    const float e = s.sx * s.sw * (eb.cA * xn * s.wm + eb.cW * s.wm * s.wm + eb.cB * s.bmag) +
                    eb.cM * (xerr * s.sw * s.wm + (s.sx * xn + xerr) * s.we);
    const bool ok = e > 0.0f && e < 3.0e38f && xn * s.wm > 1.0e-20f && s.wm * s.wm > 1.0e-20f;
    return ok ? e : __builtin_nanf("");
}

// the float32 share of E(n): S (cA |x_n| wmax + cW wmax^2) -- twice what ONE float32 evaluation of a unit's score (the
// kernel's k-ordered chain, or any other summation of the same D products) can be away from the real score, in d' units
__device__ __forceinline__ float ex_f32_share(const ExactBound& eb, const ExactScales& s, float xsq) {
    const float xn = __builtin_sqrtf(xsq) * (1.0f + 1.0f / 1024.0f);
    return s.sx * s.sw * (eb.cA * xn * s.wm + eb.cW * s.wm * s.wm);
}

// |w~|^2 of every unit (bf16-rounded values) and its maximum over the codebook
template <class EL = Bf16>
__global__ __launch_bounds__(256) void prep_wnorm_kernel(const float* __restrict__ W, int K, int D,
                                                         float* __restrict__ wn, float* __restrict__ wmax2,
                                                         const float* __restrict__ unit_wsq) {
    using E = typename EL::T;
    long u = (long)blockIdx.x * 256 + threadIdx.x;
    float s = 0.0f;
    if (u < K) {
        float scale = 1.0f;       // cosine: the stage image holds unit-length rows and no norm term
        if (unit_wsq != nullptr) { float q = unit_wsq[u]; scale = q > 0.0f ? 1.0f / __builtin_sqrtf(q) : 0.0f; }
        for (int k = 0; k < D; ++k) { float f = (float)cvt<E>(W[u * D + k] * scale); s = __builtin_fmaf(f, f, s); }
        wn[u] = unit_wsq != nullptr ? 0.0f : s;
    }
    float m = s;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomic_max_pos_f32(wmax2, m);
}

// initial accumulators  B + |w~|^2/2  written behind each stage's fragments; B from the two maxima
// (+ the launch's per-row merge keys best64[0..n_rows) = all ones, when given: one launch instead of a memset and a kernel)
__global__ __launch_bounds__(256) void prep_wsqh_kernel(const float* __restrict__ wn, int K,
                                                        const float* __restrict__ wmax2,
                                                        const float* __restrict__ xmax2, char* __restrict__ Wst,
                                                        int n_stages, int stage_bytes, int stage_units,
                                                        unsigned long long* __restrict__ best64, long n_rows,
                                                        int scaled = 0) {
    long u = (long)blockIdx.x * 256 + threadIdx.x;
    if (best64 != nullptr && u < n_rows) best64[u] = ~0ull;
    if (u >= (long)n_stages * stage_units) return;
    const float big = __builtin_sqrtf(*wmax2) * __builtin_sqrtf(*xmax2) * (1.0f + 1.0f / 1024.0f);
    // scaled (exact mode on half operands): the operands carry ex_scale factors, so does the accumulator (exactly)
    const float S = scaled ? ex_scale(*xmax2) * ex_scale(*wmax2) : 1.0f;
    float s = (u < K) ? __builtin_fmaf(0.5f * S, wn[u], S * big) : BF_PAD_NORM;
    long stage = u / stage_units;
    int within = u % stage_units;
    float* dst = (float*)(Wst + (stage + 1) * (long)stage_bytes - 1024);   // the stage's last KiB
    dst[within] = s;
}

// samples -> bf16 rows [Np][Dp] (zero padded) and max_n |x~_n|^2.  One wave per row.
// unit != 0 (cosine): the row is scaled to unit length first (a zero row stays zero).
template <class EL = Bf16>
__global__ __launch_bounds__(256) void prep_x_bf16_kernel(const float* __restrict__ X, long N, int D, int Dp,
                                                          long Np, __bf16* __restrict__ Xb,
                                                          float* __restrict__ xmax2, int unit,
                                                          const float* __restrict__ scale_max2 = nullptr,
                                                          float* __restrict__ xerr = nullptr) {
    using E = typename EL::T;
    long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (row >= Np) return;
    // scale_max2 (exact mode): rows times ex_scale(*scale_max2); the caller keeps max |x|^2 itself (xmax2 == nullptr)
    float scale = scale_max2 != nullptr ? ex_scale(*scale_max2) : 1.0f;
    if (unit) {
        float q = 0.0f;
        for (int k = lane; k < D; k += 64) { float f = row < N ? X[row * D + k] : 0.0f; q = __builtin_fmaf(f, f, q); }
        q = wave_sum(q);
        scale = q > 0.0f ? 1.0f / __builtin_sqrtf(q) : 0.0f;
    }
    float s = 0.0f, er = 0.0f;
    for (int k = lane; k < Dp; k += 64) {
        float f = (row < N && k < D) ? X[row * D + k] * scale : 0.0f;
        E b = cvt<E>(f);
        ((E*)Xb)[row * Dp + k] = b;
        float fb = (float)b;
        s = __builtin_fmaf(fb, fb, s);
        er = __builtin_fmaf(half_operand_error(f, fb), half_operand_error(f, fb), er);
    }
    s = wave_sum(s);
    if (lane == 0 && xmax2 != nullptr) atomic_max_pos_f32(xmax2, s);
    if (xerr != nullptr) {                                 // |x^ - x~| of the row, rounded up (exact mode's measured bound)
        er = wave_sum(er);
        if (lane == 0 && row < N) xerr[row] = __builtin_sqrtf(er) * (1.0f + 1.0f / 1024.0f);
    }
}

}  // namespace somhip
