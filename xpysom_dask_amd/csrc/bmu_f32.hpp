// Fused distance + BMU argmin, float32 parity mode.
//
// Replaces, per mini-batch, the reference chain
//   DistanceFunction.__call__      distances.py:184-191
//   euclidean_squared_distance_part distances.py:11-23   (-2 x.w^T + w_sq^T)
//   euclidean_distance             distances.py:33-43   (QE path, xpysom.py:640,670)
//   cosine_distance                distances.py:45-59
//   argmin(axis=1)                 xpysom.py:416
// without ever writing the (n, K) distance matrix.
//
// Arithmetic: the cross term runs on v_mfma_f32_32x32x2_f32, which is bit-for-bit a
// k-ordered float32 fmaf chain, so identical codebook rows give identical distances
// and exact ties resolve to the lowest raveled unit id exactly as numpy.argmin does.
//
// Mapping: A operand = codebook rows (units), B operand = samples, so a lane owns one
// sample (column) and its 16 accumulator registers are 16 different units: the running
// argmin is pure per-lane VALU; one __shfl_xor(…,32) at the very end joins the two lane
// halves.  A workgroup = 4 waves x 32 samples and scans the whole codebook in tiles of
// 128 units staged through LDS (coalesced rows, +1 padded so fragment reads are
// conflict-free).
#pragma once
#include "som_common.hpp"

namespace somhip {

constexpr int F32_SB = 128;   // samples per workgroup
constexpr int F32_UB = 128;   // units per codebook tile
constexpr int F32_KC = 32;    // features per LDS chunk

enum { SCORE_EUCLID_PART = 0, SCORE_EUCLID_SQRT = 1, SCORE_COSINE = 2, SCORE_EUCLID_SQ = 3 };

template <int MODE>
__device__ __forceinline__ float score_f32(float cross, float wsq, float xsq) {
    if (MODE == SCORE_EUCLID_PART) {
        return __builtin_fmaf(-2.0f, cross, wsq);                 // -2*cross exact, one rounding: == numpy
    } else if (MODE == SCORE_EUCLID_SQ) {
        return __builtin_fmaf(-2.0f, cross, wsq) + xsq;           // distances.py:30-31
    } else if (MODE == SCORE_EUCLID_SQRT) {
        float t = __builtin_fmaf(-2.0f, cross, wsq) + xsq;        // distances.py:38-43
        return nan_to_num_f32(__builtin_sqrtf(t));
    } else {
        float den = __builtin_sqrtf(xsq * wsq);                   // distances.py:56-59
        return 1.0f - nan_to_num_f32(cross / den);
    }
}

// Running first-minimum over one 32x32 MFMA output tile, without a compare-and-branch per value.
// A lane holds 16 scores of its sample (acc register r <-> unit tile*32 + mfma32_row(r, half), ascending
// in r).  The 16 scores are formed, reduced with fminf (v_min3_f32; NaN never wins, as with '<'), and
// only lanes whose tile minimum beats their running best recover WHICH register held it -- the lowest
// r that compares equal, i.e. exactly the unit a sequential `if (v < best)` scan would have kept.
// key = tile << 4 | r; f32_key_unit() turns it back into the unit id.
// TAIL: the tile may hold padding units (>= K), which must not compete (only the last stage).
template <int MODE, bool TAIL>
__device__ __forceinline__ void f32_tile_argmin(const f32x16& acc, const f32x4 (&wv)[4], float xsq, int tile, int half,
                                                int K, float& best, int& bkey) {
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        v[r] = score_f32<MODE>(acc[r], wv[r >> 2][r & 3], xsq);
        if (TAIL && tile * 32 + mfma32_row(r, half) >= K) v[r] = __builtin_inff();
    }
    float m = __builtin_fminf(v[0], v[1]);
#pragma unroll
    for (int r = 2; r < 16; ++r) m = __builtin_fminf(m, v[r]);
    if (m < best) {
        int code = 15;
#pragma unroll
        for (int r = 14; r >= 0; --r) code = (v[r] == m) ? r : code;
        best = m;
        bkey = (tile << 4) | code;
    }
}

__device__ __forceinline__ int f32_key_unit(int key, int half) { return (key >> 4) * 32 + mfma32_row(key & 15, half); }

// The exact mode's groups whose 64 positions are an 8 x 8 patch of the map in FOUR-BY-FOUR blocks (som_common.hpp, the 16-unit
// sub-blocks of exact_skip.hpp): position w of the group = block (w >> 4) -- (0,0), (0,4), (4,0), (4,4) -- row (w >> 2) & 3 and
// column w & 3 inside it; the unit's RANK among the group's 64 units in ascending unit id = 8 (patch row) + patch column.
__device__ __forceinline__ int ex_rank44(int w) { return 8 * (4 * ((w >> 5) & 1) + ((w >> 2) & 3)) + 4 * ((w >> 4) & 1) + (w & 3); }

// f32_tile_argmin for such a group: positions no longer ascend with the unit ids, so "first minimum" is decided by rank -- among
// equal scores (within the tile, against the running best) the lowest RANK wins, which is the lowest unit id.  brank: the
// running best's rank (any value while best = +inf).  `tile` odd <-> the group's second 32 positions.
template <int MODE>
__device__ __forceinline__ void f32_tile_argmin_ranked(const f32x16& acc, const f32x4 (&wv)[4], float xsq, int tile, int half,
                                                       float& best, int& bkey, int& brank) {
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = score_f32<MODE>(acc[r], wv[r >> 2][r & 3], xsq);
    float m = __builtin_fminf(v[0], v[1]);
#pragma unroll
    for (int r = 2; r < 16; ++r) m = __builtin_fminf(m, v[r]);
    if (m <= best) {                                     // (a NaN minimum: neither; '<' as in the sequential scan, '==' by rank)
        int code = -1, rk = m < best ? 64 : brank;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q = ex_rank44(((tile & 1) << 5) | mfma32_row(r, half));
            if (v[r] == m && q < rk) { rk = q; code = r; }
        }
        if (code >= 0) { best = m; bkey = (tile << 4) | code; brank = rk; }
    }
}

// The (n, K) distance matrix itself, for the analysis calls that return it: XPySom.activate
// (xpysom.py:323-354, configured distance) and distance_from_weights (:647-671, sqrt'd Euclidean).
// Same tiling and the same bit-exact arithmetic as bmu_f32_kernel; the epilogue stores instead of
// reducing.  Not on the training path (nothing there materialises (n, K)).
template <int MODE>
__global__ __launch_bounds__(256) void dist_matrix_f32_kernel(const float* __restrict__ X, long N, int D, int Dp,
                                                              const float* __restrict__ W,
                                                              const float* __restrict__ wsq, int K,
                                                              const float* __restrict__ xsq,
                                                              float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    float* Ws = smem_f;
    float* wq = Ws + F32_UB * (F32_KC + 1);
    float* Xs = wq + F32_UB;
    const int xstride = F32_KC + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, col = lane & 31;
    const long s0 = (long)blockIdx.x * F32_SB;
    const long my_sample = s0 + wave * 32 + col;
    const int u0 = blockIdx.y * F32_UB;
    float xs = 0.0f;
    if (MODE != SCORE_EUCLID_PART) xs = (my_sample < N) ? xsq[my_sample] : 0.0f;
    f32x16 acc[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][r] = 0.0f;
    for (int kc = 0; kc < Dp; kc += F32_KC) {
        __syncthreads();
        for (int idx = tid; idx < F32_UB * F32_KC; idx += 256) {
            int r = idx >> 5, k = idx & 31;
            Ws[r * (F32_KC + 1) + k] = (u0 + r < K && kc + k < D) ? W[(long)(u0 + r) * D + kc + k] : 0.0f;
            Xs[r * xstride + k] = (s0 + r < N && kc + k < D) ? X[(s0 + r) * (long)D + kc + k] : 0.0f;
        }
        if (kc == 0 && tid < F32_UB) wq[tid] = (u0 + tid < K) ? wsq[u0 + tid] : 0.0f;
        __syncthreads();
        const float* xrow = Xs + (wave * 32 + col) * xstride + half;
        const float* wrow = Ws + col * (F32_KC + 1) + half;
#pragma unroll
        for (int k = 0; k < F32_KC; k += 2) {
            float b = xrow[k];
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
                acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wrow[rb * 32 * (F32_KC + 1) + k], b, acc[rb], 0, 0, 0);
        }
    }
    if (my_sample < N) {
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = rb * 32 + mfma32_row(r, half);
                if (u0 + row < K) out[my_sample * K + u0 + row] = score_f32<MODE>(acc[rb][r], wq[row], xs);
            }
    }
}

// sum of squares of each row, float32: xp.power(a, 2).sum(axis=1) -- xpysom.py:529-537 (w_sq),
// distances.py:30,53 (x_sq).  The squares are rounded to float32 first and then added in
// NumPy's pairwise order (n < 8 sequential; n <= 128 eight strided accumulators joined as a
// tree plus a sequential tail; larger n split in halves), so the result is bit-identical to
// NumPy's and, with the k-ordered fma chain of the MFMA (== OpenBLAS' sgemm micro-kernel for
// one K block), so is every distance and therefore every BMU, near-ties included.
// __fmul_rn/__fadd_rn keep hipcc from contracting the square into the add.
__device__ float np_pairwise_sq_sum(const float* __restrict__ a, int n) {
    if (n < 8) {
        float res = 0.0f;
        for (int i = 0; i < n; ++i) res = __fadd_rn(res, __fmul_rn(a[i], a[i]));
        return res;
    }
    if (n <= 128) {
        float r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = __fmul_rn(a[j], a[j]);
        int i = 8;
        for (; i < n - (n % 8); i += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = __fadd_rn(r[j], __fmul_rn(a[i + j], a[i + j]));
        }
        float res = __fadd_rn(__fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3])),
                              __fadd_rn(__fadd_rn(r[4], r[5]), __fadd_rn(r[6], r[7])));
        for (; i < n; ++i) res = __fadd_rn(res, __fmul_rn(a[i], a[i]));
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return __fadd_rn(np_pairwise_sq_sum(a, n2), np_pairwise_sq_sum(a + n2, n - n2));
}

// (out_p, inv: the codebook's |w|^2 a second time in the exact mode's patch order, inv[unit] = position: som_common.hpp)
__global__ __launch_bounds__(256) void row_sq_f32_kernel(const float* __restrict__ A, long rows, int D,
                                                         float* __restrict__ out, float* __restrict__ out_p = nullptr,
                                                         const int* __restrict__ inv = nullptr) {
    long row = (long)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    const float q = np_pairwise_sq_sum(A + row * (long)D, D);
    out[row] = q;
    if (out_p != nullptr) {
        out_p[inv[row]] = q;
    }
}

}  // namespace somhip
