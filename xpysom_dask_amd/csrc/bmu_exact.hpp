// precision 'exact': float32-exact BMUs at split-bf16 MFMA speed.
//
// What it returns is, row for row and bit for bit, what the float32 parity kernel returns (bmu_f32_res.hpp:
// argmin_k fl(|w_k|^2 - 2 c_k), c_k the k-ordered float32 fma chain of x . w_k -- the reference's
// euclidean_squared_distance_part + argmin, distances.py:11-23, xpysom.py:416), near-ties and exact ties included.
// How: a cheap SCREEN that may be wrong by a bounded amount, then the float32 chain itself on the few units the screen
// cannot rule out.
//
//   1. screen    bmu_bf16_k16x3_kernel<.., GM = true> (hi/lo-split bf16 MFMA, bmu_bf16_k16x3.hpp) computes
//                d'(n,k) = B + |w_k|^2/2 - x_n . w_k approximately, keeps the row minimum m(n) as always, and also
//                writes, per GROUP of 64 units and row, the group's minimum: gmin[group][row] (4 bytes per row and
//                64 units: 268 MB per 65 536 rows of a 256 x 256 map, written once, read once).
//   2. candidates  exact_scan_kernel: every group with gmin <= m(n) + E(n) is a candidate of row n.  E(n) bounds
//                (float32 kernel's own rounding) + (screen's error), both relative to tau = |w|^2 - 2 x.w in real
//                arithmetic -- derivation below.  The float32 winner k* has s(k*) <= s(k) for every k, so its screen
//                value cannot exceed the screen's minimum by more than E: its group is a candidate.
//   3. re-score  exact_rescore_kernel: one wave per row; for every candidate group the 64 units' float32 scores by the
//                SAME arithmetic as the parity kernel (v_mfma_f32_32x32x2_f32 is bit for bit a k-ordered fmaf chain;
//                here the chain runs on the vector ALU, one lane per unit, reading the parity kernel's own stage
//                image), then the first minimum in unit order.  Because k* is among the re-scored units and is the
//                global first minimum, it is the re-scored first minimum.
//   4. fallback  rows the scheme cannot vouch for -- more candidate groups than the list holds, no candidate at all
//                (NaN / infinite rows or norms), a minimum that is not finite -- go to the float32 kernel itself.
//
// Error bound (euclidean, input_len <= 128).  u = 2^-24, ub = 2^-8 (bf16) or 2^-11 (f16), A(n,k) = sum_d |x_d w_kd|
// <= |x_n| max_k|w_k|.
//   float32 kernel:  |c - x.w| <= gamma_D A (fma chain of D terms), s = fl(wsq - 2c):
//                    |s - tau| <= (2 gamma_D + 2u)(1+u) A + u wsq.
//   screen, operands: v = hi + lo + r with |r| <= ub^2 |v|; the kernel contracts hi.hi + lo.hi + hi.lo, what it drops
//                    is bounded by ub^2 (3 + 5 ub) A.
//   screen, accumulation: the initial accumulator fl(B + wsq/2) and 3 * ceil(D/32) chained MFMAs, each charged KAPPA
//                    ulps of the largest magnitude the accumulator can take, Bm = 2.01 B + max wsq / 2 (the hardware's
//                    internal summation is not documented; tests/test_gpu_exact.py measures it at <= 1 ulp), plus the
//                    8 ulps the index bits packed into the key's low mantissa bits hide.
//   E (in units of d' = tau / 2 + B):  E32 / 2 ... spelled out in exact_bound() (host side, somhip.hip).
// The bound is deliberately loose (worst-case rounding everywhere): widening E only adds candidate groups, and a
// candidate group costs one 64-unit re-score.
#pragma once
#include "bmu_bf16_k16x3.hpp"
#include "bmu_f32_res.hpp"

namespace somhip {

constexpr int EX_GROUP = 64;          // units per group = one stage of the float32 stage image = two screen stages
constexpr int EX_CAND = 32;           // candidate groups kept per row; more -> the row goes to the float32 kernel
constexpr int EX_SCAN_SPLIT = 4;      // waves that share a row's groups in the scan

struct ExactBound {                   // E(n) = cA * |x_n| * wmax + cW * wmax^2 + cB * Bm   (d' units)
    float cA, cW, cB;
};

// gmin [n_groups][gm_stride] -> cand [N][EX_CAND], count [N].  Block = 64 rows x EX_SCAN_SPLIT group ranges.
__global__ __launch_bounds__(64 * EX_SCAN_SPLIT) void exact_scan_kernel(const uint32_t* __restrict__ gmin, long gm_stride,
                                                                       int n_groups, long N,
                                                                       const unsigned long long* __restrict__ best64,
                                                                       const float* __restrict__ xsq,
                                                                       const float* __restrict__ wmax2,
                                                                       const float* __restrict__ xmax2, ExactBound eb,
                                                                       int* __restrict__ cand, int* __restrict__ count) {
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * 64 + lane;
    const bool live = row < N;
    const long r = live ? row : 0;
    // thr: unsigned compare on the bit patterns (all d' are positive floats; a NaN pattern is above every threshold)
    const float wm = __builtin_sqrtf(*wmax2) * (1.0f + 1.0f / 1024.0f);
    const float big = __builtin_sqrtf(*wmax2) * __builtin_sqrtf(*xmax2) * (1.0f + 1.0f / 1024.0f);   // prep_wsqh_kernel's B
    const float bm = 2.01f * big + 0.5f * wm * wm;
    const float xn = __builtin_sqrtf(xsq[r]) * (1.0f + 1.0f / 1024.0f);
    const float m = __uint_as_float((uint32_t)(best64[r] >> 32));
    const float e = eb.cA * xn * wm + eb.cW * wm * wm + eb.cB * bm;
    const float thr_f = m + e;
    // a threshold that is not a finite positive number (NaN / infinite row or norms) selects nothing: fallback
    const bool ok = live && thr_f > 0.0f && thr_f < 3.0e38f;
    const uint32_t thr = ok ? __float_as_uint(thr_f) : 0u;
    const int g0 = (int)((long)n_groups * part / EX_SCAN_SPLIT), g1 = (int)((long)n_groups * (part + 1) / EX_SCAN_SPLIT);
    const uint32_t* src = gmin + r;
    for (int g = g0; g < g1; g += 8) {
        uint32_t v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = g + q < g1 ? src[(long)(g + q) * gm_stride] : 0xFFFFFFFFu;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (v[q] <= thr && ok) {
                const int slot = atomicAdd(count + r, 1);
                if (slot < EX_CAND) cand[r * EX_CAND + slot] = g + q;
            }
    }
}

// One wave per row.  Scores of the 64 units of each candidate group exactly as bmu_f32_res_kernel<SCORE_EUCLID_PART>
// forms them, first minimum in unit order over all candidates.  Rows it cannot settle are appended to fb_list.
template <int KG>
__global__ __launch_bounds__(256) void exact_rescore_kernel(const float* __restrict__ X, long N, int D,
                                                            const char* __restrict__ Wfst, int K,
                                                            const int* __restrict__ cand, const int* __restrict__ count,
                                                            int* __restrict__ out, int* __restrict__ fb_list,
                                                            int* __restrict__ fb_count) {
    constexpr int STAGE = fr_stage_bytes(KG);
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (row >= N) return;
    const int cnt = count[row];
    bool fallback = cnt <= 0 || cnt > EX_CAND;
    float best = __builtin_inff();
    int bunit = 0x7fffffff;
    if (!fallback) {
        const float* xrow = X + row * D;
        const int ut = lane >> 5, col = lane & 31;
        for (int ci = 0; ci < cnt; ++ci) {
            const int g = cand[row * EX_CAND + ci];
            const char* st = Wfst + (long)g * STAGE;
            float c = 0.0f;
#pragma unroll
            for (int kg = 0; kg < KG; ++kg) {
                // lane col of the MFMA image holds the even features 8 kg + 2 j, lane col + 32 the odd ones
                const f32x4 a0 = *(const f32x4*)(st + ((long)(ut * KG + kg) * 64 + col) * 16);
                const f32x4 a1 = *(const f32x4*)(st + ((long)(ut * KG + kg) * 64 + col + 32) * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = 8 * kg + 2 * j;
                    const float x0 = k < D ? xrow[k] : 0.0f, x1 = k + 1 < D ? xrow[k + 1] : 0.0f;
                    c = __builtin_fmaf(a0[j], x0, c);
                    c = __builtin_fmaf(a1[j], x1, c);
                }
            }
            const float wsq = ((const float*)(st + (long)FR_UT * KG * 1024))[lane];   // +inf behind the last unit
            float v = score_f32<SCORE_EUCLID_PART>(c, wsq, 0.0f);
            int u = g * EX_GROUP + lane;
            if (!(v == v) || u >= K) v = __builtin_inff();                          // a NaN never wins ('<' semantics)
            if (v < best || (v == best && u < bunit)) { best = v; bunit = u; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64);
            const int ou = __shfl_xor(bunit, o, 64);
            if (ov < best || (ov == best && ou < bunit)) { best = ov; bunit = ou; }
        }
        // +inf: no unit with a finite score among the candidates; -inf: leave such rows to the parity kernel too
        fallback = !(best > -3.0e38f && best < 3.0e38f);
    }
    if (lane == 0) {
        if (fallback) fb_list[atomicAdd(fb_count, 1)] = (int)row;
        else out[row] = bunit;
    }
}

// fallback rows -> a dense block for the float32 kernel, and its ids back
__global__ __launch_bounds__(256) void exact_gather_rows_kernel(const float* __restrict__ X, const int* __restrict__ list,
                                                                int n, int D, float* __restrict__ out) {
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= (long)n * D) return;
    const int i = (int)(id / D), d = (int)(id - (long)i * D);
    out[id] = X[(long)list[i] * D + d];
}
__global__ __launch_bounds__(256) void exact_scatter_ids_kernel(const int* __restrict__ ids, const int* __restrict__ list,
                                                                int n, int* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[list[i]] = ids[i];
}

// wn = wsq (the float32 kernel's own |w|^2) and its maximum: the screen's initial accumulator then carries the same
// norm the re-score adds
__global__ __launch_bounds__(256) void exact_copy_wsq_kernel(const float* __restrict__ wsq, int K, float* __restrict__ wn,
                                                             float* __restrict__ wmax2) {
    const long u = (long)blockIdx.x * 256 + threadIdx.x;
    float s = 0.0f;
    if (u < K) { s = wsq[u]; wn[u] = s; }
    float m = (s == s) ? s : 0.0f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomic_max_pos_f32(wmax2, m);
}

}  // namespace somhip
