// precision 'exact' beyond 128 features (129 .. 800, maps of >= 4096 units, euclidean): BLOCK SKIPPING on the wide screen.
//
// The scheme of exact_skip.hpp on bmu_bf16_wide.hpp's tiling.  tools/skip_probe_wide.py counted, for 512 x 512 x 784 with the
// gaussian neighbourhood (MNIST-shaped rows), 1-14 % of the (256-row tile, 64-unit group) blocks that a centroid / radius bound
// around last epoch's BMU cannot prove empty from a schedule's third epoch on (profiles/r05_skip_probe_wide.txt) -- where
// configs[4]'s cosine + mexican_hat schedule leaves 62-100 % (profiles/r04_skip_probe_c5.txt).  What is different here:
//   * one level: the 64-unit GROUPS (an 8 x 8 patch of the map = two 32-unit stages of the wide image);
//   * the bound on the distance to this epoch's BMU comes from the float32 score t of last epoch's BMU under the current
//     codebook (exact_seed_kernel on the sorted float32 rows: any summation order of the D products is within the float32
//     kernel's own share of the bound): U = |x|^2 + t + two float32 windows;
//   * the plan is a MODE of the wide kernel itself (bmu_bf16_wide_kernel<.., PLAN>): the centroids go in as a 32-to-a-stage
//     image, the test's cross term (sx sqrt(U))(sw r) is one fma per (row, centroid) in the epilogue -- with 25 MFMA steps
//     per stage there is room for it;
//   * the resident rows are kept sorted by their last BMU's patch as a sorted copy of the float32 rows (what the re-score
//     gathers from) and the tile image built from it.
// The screen walks each tile's list of groups (bmu_bf16_wide_kernel<.., GM, TL>), the select kernel the same list, the two
// re-score rounds the sorted float32 rows; exact_finalize_kernel scatters the ids through the order.
#pragma once
#include "bmu_bf16_wide.hpp"
#include "exact_skip.hpp"

namespace somhip {

// Centroid, radius and |c|^2 of every group of 64 units of W (patch order), any input_len: one workgroup per group, thread t
// holds features t, t + 256, ... (up to four: input_len <= 1024).  A slot without units: centroid 0, radius -1.
__global__ __launch_bounds__(256) void wide_centroids_kernel(const float* __restrict__ W, int K, int D, int n_groups,
                                                             float* __restrict__ Cc, float* __restrict__ rg, float* __restrict__ csq,
                                                             float* __restrict__ cmax2, const float* __restrict__ wmax2) {
    __shared__ float red[4];
    __shared__ float rmax_s;
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long u0 = (long)g * 64;
    const int cnt = (int)max(0L, min(64L, (long)K - u0));
    if (g == 0 && tid == 0) { cmax2[0] = *wmax2; cmax2[1] = 0.0f; }
    float c[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int k = 0; k < cnt; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int d = tid + 256 * j; if (d < D) c[j] += W[(u0 + k) * D + d]; }
    float q = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        c[j] = cnt > 0 ? c[j] / (float)cnt : 0.0f;
        const int d = tid + 256 * j;
        if (d < D) { Cc[(long)g * D + d] = c[j]; q = __builtin_fmaf(c[j], c[j], q); }
    }
    auto block_sum = [&](float v) -> float {                 // (every thread gets the sum)
        v = wave_sum(v);
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        return (red[0] + red[1]) + (red[2] + red[3]);
    };
    const float sq = block_sum(q);
    float m = 0.0f;
    for (int k = 0; k < cnt; ++k) {
        float p = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int d = tid + 256 * j; if (d < D) { const float t = W[(u0 + k) * D + d] - c[j]; p = __builtin_fmaf(t, t, p); } }
        const float d2 = block_sum(p);
        m = (d2 > m || !(d2 == d2)) ? d2 : m;                 // (a NaN unit: a NaN radius, the group is never skipped)
    }
    if (tid == 0) {
        // (sums of squares in float32, any order: relative error <= D 2^-24; the radius is rounded up by 2^-9, the plan's margins cover |c|^2)
        rg[g] = cnt > 0 ? __builtin_sqrtf(m) * (1.0f + 1.0f / 512.0f) + 1.0e-30f : -1.0f;
        csq[g] = sq;
        (void)rmax_s;
    }
}

// the tails of the centroid stage image (32 centroids to a stage; the fragments are prep_w_bf16_wide_kernel's, the measured
// rounding error exact_werr_kernel's): [0, 32) initial accumulators S'(B' + |c|^2 / 2) - hS r^2, [32, 64) sw r (rounded up);
// a slot without units: +inf and 0 (never needed); a NaN radius: NaN (always needed)
// Cst_plain (or null): a copy of the image whose tails are the plain S'(B' + |c|^2 / 2) -- the scout's nearest-centroid scan
// (its fragments: a device copy of Cst's, made by the host before this kernel).
__global__ __launch_bounds__(256) void wide_centroid_tail_kernel(const float* __restrict__ rg, const float* __restrict__ csq, int n_groups,
                                                                 char* __restrict__ Cst, int n_img_stages, int stage_bytes,
                                                                 const float* __restrict__ xmax2, const float* __restrict__ wmax2,
                                                                 char* __restrict__ Cst_plain = nullptr) {
    const long slot = (long)blockIdx.x * 256 + threadIdx.x;
    if (slot >= (long)n_img_stages * WD_STAGE_UNITS) return;
    const long stage = slot / WD_STAGE_UNITS;
    const int within = (int)(slot - stage * WD_STAGE_UNITS);
    float* tail = (float*)(Cst + (stage + 1) * (long)stage_bytes - 1024);
    float* ptail = Cst_plain != nullptr ? (float*)(Cst_plain + (stage + 1) * (long)stage_bytes - 1024) : nullptr;
    const float rad = slot < n_groups ? rg[slot] : -1.0f;
    if (rad < 0.0f) { tail[within] = __builtin_inff(); tail[WD_STAGE_UNITS + within] = 0.0f; if (ptail) ptail[within] = __builtin_inff(); return; }
    const float sw = ex_scale(*wmax2), sx = ex_scale(*xmax2);
    const float big = __builtin_sqrtf(*wmax2) * __builtin_sqrtf(*xmax2) * (1.0f + 1.0f / 1024.0f);   // (ex_scales: B')
    const float S = sx * sw, hS = 0.5f * S * (1.0f + 1.0f / 1024.0f);
    const float s0 = __builtin_fmaf(0.5f * S, csq[slot], S * big);
    tail[within] = s0 - hS * rad * rad * (1.0f + 0x1p-20f);
    tail[WD_STAGE_UNITS + within] = sw * rad * (1.0f + 0x1p-20f);
    if (ptail) ptail[within] = (s0 == s0 && s0 < 3.0e38f) ? s0 : __builtin_inff();
}

// The resident pass in sorted order: the float32 rows, |x|^2, rounding error, last epoch's BMU -- positions behind the pass's
// rows (up to the tile multiple): zero rows, NaN norms, unit 0.  One wave per row.
__global__ __launch_bounds__(256) void wide_gather_sorted_kernel(const int* __restrict__ order, long n, long np, int D,
                                                                 const float* __restrict__ X, const float* __restrict__ xsq,
                                                                 const float* __restrict__ xerr, const int* __restrict__ prev,
                                                                 float* __restrict__ Xf_s, float* __restrict__ xsq_s,
                                                                 float* __restrict__ xerr_s, int* __restrict__ prev_s) {
    const int lane = threadIdx.x & 63;
    const long p = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= np) return;
    const long r = p < n ? (long)order[p] : -1;
    if ((D & 3) == 0) {
        for (int c = lane; c < D / 4; c += 64) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (r >= 0) v = *(const f32x4*)(X + r * D + 4 * c);
            *(f32x4*)(Xf_s + p * D + 4 * c) = v;
        }
    } else {
        for (int c = lane; c < D; c += 64) Xf_s[p * D + c] = r >= 0 ? X[r * D + c] : 0.0f;
    }
    if (lane == 0) {
        const float nanv = __builtin_nanf("");
        xsq_s[p] = r >= 0 ? xsq[r] : nanv;
        xerr_s[p] = r >= 0 ? xerr[r] : nanv;
        prev_s[p] = r >= 0 ? prev[r] : 0;
    }
}

// last epoch's BMUs of the sorted rows under an order that was built in an earlier epoch
__global__ __launch_bounds__(256) void wide_prev_sorted_kernel(const int* __restrict__ order, long n, const int* __restrict__ prev,
                                                               int* __restrict__ prev_s) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p < n) prev_s[p] = prev[order[p]];
}

// Per sorted row the plan's threshold P and sx sqrt(U) (1 + 2^-10) (exact_skip.hpp, (S1)-(S4), the cross term on the vector ALU):
//   k* beats u (last epoch's BMU) in the float32 kernel, t = u's float32 score by ANY summation (exact_seed_kernel: within one
//   float32 window e32 of the real score, as the kernel's own chain is):  |x - w_k*|^2 <= U := |x|^2 + t + 3 e32 (charged: 4);
//   skip <=> d'_c - hS r^2 - (sx sqrt(U))(sw r) > P := S'(B' - |x|^2 / 2) + e_c + hS U   (+ the margins of exact_plan_kernel;
//   the fma of the cross term is charged 2^-20 S' Bm').  A row whose numbers are not finite: P = +inf (it needs every group).
// eb: the exact mode's bound constants; centroid image scales from cmax2 = {max |c|^2, max rounding error^2}, the codebook's
// from wmax2 / werr2.
__global__ __launch_bounds__(256) void wide_plan_rows_kernel(long n, const float* __restrict__ xsq_s, const float* __restrict__ xerr_s,
                                                             const float* __restrict__ tq, const float* __restrict__ xmax2,
                                                             const float* __restrict__ cmax2, const float* __restrict__ wmax2,
                                                             const float* __restrict__ werr2, ExactBound eb, int force_all,
                                                             float* __restrict__ planP, float* __restrict__ planXs) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const ExactScales sc = ex_scales(xmax2, cmax2, cmax2 + 1);
    const ExactScales sw = ex_scales(xmax2, wmax2, werr2);
    const float q = xsq_s[p], xe = xerr_s[p], t = tq[p];
    const float xn = __builtin_sqrtf(q) * (1.0f + 1.0f / 1024.0f);
    const float share = 2.0f * (eb.cA * xn * sw.wm + eb.cW * sw.wm * sw.wm);      // tau units: one float32 window (two evaluations)
    const float U0 = q * (1.0f + 1.0f / 1024.0f) + t + 2.0f * share * (1.0f + 1.0f / 1024.0f) + 0x1p-18f * (q + __builtin_fabsf(t));
    const float U = U0 < 0.0f ? 0.0f : U0;                   // (a NaN stays a NaN)
    float su = __builtin_sqrtf(U) * (1.0f + 1.0f / 1024.0f);
    const float e = ex_row_bound(eb, sc, q, xe);
    const float S = sc.sx * sc.sw;
    const float ec = 0.5f * e * (1.0f + 1.0f / 1024.0f);
    float A = S * sc.big + ec - 0.5f * S * q * (1.0f - 1.0f / 1024.0f) + 0x1p-12f * S * (sc.big + q) + 0x1p-16f * S * sc.wm * sc.wm;
    float P = A + 0.5f * S * (1.0f + 1.0f / 1024.0f) * su * su * (1.0f + 0x1p-20f) + 0x1p-20f * S * sc.bmag;
    float xs = sc.sx * su * (1.0f + 1.0f / 1024.0f);
    if (force_all || !(su == su) || !(su < 3.0e38f) || !(e == e) || !(A == A) || !(A < 3.0e38f) || !(P == P) || !(xs < 3.0e38f)) {
        P = __builtin_inff(); xs = 0.0f;                     // need everything
    }
    planP[p] = P;
    planXs[p] = xs;
}

}  // namespace somhip
