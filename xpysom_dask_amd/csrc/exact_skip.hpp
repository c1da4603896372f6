// precision 'exact', resident rows from their second epoch on, euclidean, input_len <= 128: BLOCK SKIPPING in the screen.
//
// The screen (bmu_bf16_k16_kernel<.., GM>) is the exact epoch; what is left is not to run it where no BMU can be.  Two facts:
//   * last epoch's BMU u of a row x gives an upper bound on the distance to this epoch's BMU k*: the float32 kernel picks
//     k* with fl(tau(k*)) <= fl(tau(u)), so  |x - w_k*|^2 <= U(x) := |x|^2 + t + 1.5 share,  t the float32 score of u under
//     the CURRENT codebook (exact_seed_kernel evaluates it anyway), share the float32 kernel's error window in tau units;
//   * a group of 64 units (a patch of the map) with centroid c and radius r = max |w - c| has  |x - w| >= |x - c| - r  for
//     each of its units.
// So group g holds no candidate of row x if  |x - c_g| > sqrt(U(x)) + r_g.  Rows are visited in the order of their last
// BMU's patch (a sort of row ids per pass; the pass runs in that order: operand image, norms, seeds, merge keys are
// permuted copies, the re-score gathers rows and the ids are written through the permutation), so the 256 rows of a
// workgroup tile lie in one region of the map and share most of their groups.  The PLAN kernel below is the screen's
// MFMA loop on the centroids (1/64 of the units) with this test as its epilogue and an OR over the tile's rows; the screen
// then walks, per tile, the list of groups some row of it needs.  A skipped (row, group) stores no minimum: to the select
// kernel it is a group outside the window, which is what the test proved.  The row minimum m(n) is over the groups run:
// never below the true one, and the float32 winner's group is among them, so the selection argument of bmu_exact.hpp
// holds unchanged.  Rigour: |x - c|^2 = |x|^2 + tau_c is taken from the plan's screen value d'_c = S'(B' + tau_c / 2)
// less the screen's error bound for the centroid image (E_c / 2 of ex_row_bound on the centroids' scales), norms and
// radii are rounded outwards; a row whose quantities are not finite needs every group.
#pragma once
#include "bmu_bf16_k16.hpp"

namespace somhip {

constexpr int SK_TILE = K16_WG_SAMPLES;   // rows per plan / screen workgroup tile
static_assert(K16_STAGE_UNITS == 64, "block skipping: a stage of the resident screen is one 64-unit group, a plan word one stage of centroids");

// centroid and radius of every group of 64 consecutive units of W (patch order: a patch of the map).  One block (two waves)
// per group, thread d <-> feature d in both passes (rows read whole: coalesced); a unit's |w - c|^2 is a wave reduction.
__global__ __launch_bounds__(128) void exact_centroid_kernel(const float* __restrict__ W, int K, int D, float* __restrict__ C,
                                                             float* __restrict__ rg) {
    __shared__ float part[2][64];
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int u0 = g * 64, cnt = min(64, K - u0);
    float c = 0.0f;
    if (tid < D) {
        for (int k = 0; k < cnt; ++k) c += W[(long)(u0 + k) * D + tid];
        c /= (float)cnt;
        C[(long)g * D + tid] = c;
    }
    for (int k = 0; k < cnt; ++k) {
        float q = 0.0f;
        if (tid < D) { const float df = W[(long)(u0 + k) * D + tid] - c; q = df * df; }
        q = wave_sum(q);                                      // (a NaN anywhere in the unit: NaN)
        if (lane == 0) part[wave][k] = q;
    }
    __syncthreads();
    if (tid == 0) {
        float m = 0.0f;
        for (int k = 0; k < cnt; ++k) {
            const float d2 = part[0][k] + part[1][k];
            m = (d2 > m || !(d2 == d2)) ? d2 : m;             // (a NaN unit: a NaN radius, the group is never skipped)
        }
        // (the sum of squares in float32, any order: relative error <= 128 * 2^-24; the margin covers it many times over)
        rg[g] = __builtin_sqrtf(m) * (1.0f + 1.0f / 512.0f) + 1.0e-30f;
    }
}

// sort key of row n: the group (patch) of its last BMU; value: n
__global__ __launch_bounds__(256) void exact_sortkey_kernel(const int* __restrict__ prev, const int* __restrict__ inv, long n,
                                                            int K, int* __restrict__ keys, int* __restrict__ vals) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int u = prev[i];
    u = u < 0 ? 0 : u >= K ? K - 1 : u;
    keys[i] = (inv != nullptr ? inv[u] : u) >> 6;
    vals[i] = (int)i;
}

// the pass's operands in sorted order: image rows (DP halves), |x|^2, rounding error, seed, the float32 score of the last BMU.
// Positions behind the pass's rows (up to the tile multiple) get zero rows and NaN norms (they keep nothing, need nothing).
__global__ __launch_bounds__(256) void exact_gather_sorted_kernel(const int* __restrict__ order, long n, long np, int dp,
                                                                  const __bf16* __restrict__ Xb, const float* __restrict__ xsq,
                                                                  const float* __restrict__ xerr, const float* __restrict__ seed,
                                                                  const float* __restrict__ tq, __bf16* __restrict__ Xb_s,
                                                                  float* __restrict__ xsq_s, float* __restrict__ xerr_s,
                                                                  float* __restrict__ seed_s, float* __restrict__ tq_s) {
    const int per = dp / 8;                                   // 16-byte pieces per row
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= np * per) return;
    const long p = id / per;
    const int c = (int)(id - p * per);
    u32x4 v = {0u, 0u, 0u, 0u};
    long r = -1;
    if (p < n) { r = order[p]; v = *(const u32x4*)((const char*)Xb + (r * dp + c * 8) * 2); }
    *(u32x4*)((char*)Xb_s + (p * dp + c * 8) * 2) = v;
    if (c == 0) {
        const float nanv = __builtin_nanf("");
        xsq_s[p] = r >= 0 ? xsq[r] : nanv;
        xerr_s[p] = r >= 0 ? xerr[r] : nanv;
        seed_s[p] = r >= 0 ? seed[r] : nanv;
        tq_s[p] = r >= 0 ? tq[r] : nanv;
    }
}

// The plan: which groups does a tile of SK_TILE (sorted) rows need?  The resident kernel's MFMA loop over the centroid
// stage image (one centroid per group; 64 centroids per stage), epilogue: need(row, g) = not (d'_c > A(row) + (S'/2) (sU(row)
// + r_g)^2), OR over the tile's rows into need[tile][stage] (bit j <-> group 64 stage + j).  eb / scales: the centroid
// image's (cmax2 = {max |c|^2, max rounding error^2}); wmax2 / werr2: the codebook's (for the float32 share).
template <int KS32, class EL>
__global__ __launch_bounds__(64 * K16_NW, 2) void exact_plan_kernel(const __bf16* __restrict__ Xb, long N,
                                                                    const char* __restrict__ Cst, int n_cstages,
                                                                    const float* __restrict__ rg, int n_groups,
                                                                    const float* __restrict__ xsq, const float* __restrict__ xerr,
                                                                    const float* __restrict__ tq,
                                                                    const float* __restrict__ xmax2, const float* __restrict__ cmax2,
                                                                    const float* __restrict__ wmax2, const float* __restrict__ werr2,
                                                                    ExactBound eb, unsigned long long* __restrict__ need,
                                                                    int force_all = 0) {
    using E = typename EL::T;
    using bf16x8 = typename V8<E>::t;
    constexpr int DP = 32 * KS32;
    constexpr int STAGE = k16_stage_bytes(KS32);
    constexpr int PIECES = K16_T * KS32 + 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ unsigned long long wneed[K16_NW];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int quad = lane >> 4, col = lane & 15;
    const long wave_s0 = (long)blockIdx.x * K16_WG_SAMPLES + wave * (16 * K16_SB);

    bf16x8 xf[K16_SB][KS32];
    float A[K16_SB], sU[K16_SB];
    const ExactScales sc = ex_scales(xmax2, cmax2, cmax2 + 1);
    const ExactScales sw = ex_scales(xmax2, wmax2, werr2);
    const float S = sc.sx * sc.sw;
#pragma unroll
    for (int sb = 0; sb < K16_SB; ++sb) {
        const long row = wave_s0 + sb * 16 + col;
#pragma unroll
        for (int ks = 0; ks < KS32; ++ks) xf[sb][ks] = *(const bf16x8*)(Xb + row * DP + ks * 32 + quad * 8);
        // rows behind the pass need nothing (A = -inf: d' > rhs always); a row whose numbers are not finite needs everything
        A[sb] = -__builtin_inff(); sU[sb] = 0.0f;
        if (row < N) {
            const float q = xsq[row], t = tq[row];
            const float ec = 0.5f * ex_row_bound(eb, sc, q, xerr[row]) * (1.0f + 1.0f / 1024.0f);
            const float xn = __builtin_sqrtf(q) * (1.0f + 1.0f / 1024.0f);
            const float share = 2.0f * (eb.cA * xn * sw.wm + eb.cW * sw.wm * sw.wm);          // tau units, one float32 window
            // (no fmax here: it would swallow the NaN of a row whose last BMU is a NaN unit, and bound that row by zero)
            const float U0 = q * (1.0f + 1.0f / 1024.0f) + t + 1.5f * share * (1.0f + 1.0f / 1024.0f) + 0x1p-18f * (q + __builtin_fabsf(t));
            const float U = U0 < 0.0f ? 0.0f : U0;
            sU[sb] = __builtin_sqrtf(U) * (1.0f + 1.0f / 1024.0f);
            // d'_c > S' (B' + ((sU + r)^2 - |x|^2_lo) / 2) + e_c   <=>   skip
            // (+ margins: the float32 rounding of this line, and |c|^2 as float32 summed it against the real |c|^2)
            A[sb] = S * sc.big + ec - 0.5f * S * q * (1.0f - 1.0f / 1024.0f) + 0x1p-12f * S * (sc.big + q) + 0x1p-16f * S * sc.wm * sc.wm;
            if (force_all || !(A[sb] == A[sb]) || !(sU[sb] == sU[sb]) || !(A[sb] < 3.0e38f) || !(sU[sb] < 3.0e38f)) {
                A[sb] = __builtin_inff(); sU[sb] = 0.0f;                                         // need everything
            }
        }
    }
    const float hS = 0.5f * S * (1.0f + 1.0f / 1024.0f);

    // (gridDim.y workgroups share a tile's centroid stages: few tiles -- a batch of 65 536 rows is 256 -- would otherwise be
    //  one workgroup per CU walking all the stages alone)
    const int s_begin = (int)((long)n_cstages * blockIdx.y / gridDim.y);
    const int s_end = (int)((long)n_cstages * (blockIdx.y + 1) / gridDim.y);
    if (s_begin < s_end)
        for (int p = wave; p < PIECES; p += K16_NW)
            lds_dma_16(Cst + (long)s_begin * STAGE + (long)p * 1024 + lane * 16, smem + p * 1024);
    for (int s = s_begin; s < s_end; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + 1 < s_end) {
            const char* src = Cst + (long)(s + 1) * STAGE;
            char* dst = smem + ((s + 1 - s_begin) & 1) * STAGE;
            for (int p = wave; p < PIECES; p += K16_NW) lds_dma_16(src + (long)p * 1024 + lane * 16, dst + p * 1024);
        }
        const char* st = smem + ((s - s_begin) & 1) * STAGE;
        const float* wq = (const float*)(st + K16_T * KS32 * 1024);
        unsigned long long mine = 0ull;                       // bit (16 t16 + 4 quad + r) <-> centroid of that place in the stage
#pragma unroll
        for (int t16 = 0; t16 < K16_T; ++t16) {
            const f32x4 wv = *(const f32x4*)(wq + t16 * 16 + 4 * quad);
            f32x4 acc[K16_SB];
#pragma unroll
            for (int sb = 0; sb < K16_SB; ++sb) acc[sb] = wv;
#pragma unroll
            for (int ks = 0; ks < KS32; ++ks) {
                const bf16x8 a = *(const bf16x8*)(st + ((t16 * KS32 + ks) * 64 + lane) * 16);
#pragma unroll
                for (int sb = 0; sb < K16_SB; ++sb) acc[sb] = mfma16(a, xf[sb][ks], acc[sb]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int g = s * K16_STAGE_UNITS + t16 * 16 + 4 * quad + r;
                const float rad = g < n_groups ? rg[g] : 0.0f;
                bool nd = false;
#pragma unroll
                for (int sb = 0; sb < K16_SB; ++sb) {
                    const float q = sU[sb] + rad;
                    const float rhs = __builtin_fmaf(q * q, hS, A[sb]);
                    nd = nd || !(acc[sb][r] > rhs);
                }
                const unsigned long long b = __ballot(nd && g < n_groups);
                // lanes of quad qd vote for centroid 4 qd + r of the tile
#pragma unroll
                for (int qd = 0; qd < 4; ++qd)
                    if ((b >> (16 * qd)) & 0xFFFFull) mine |= 1ull << (t16 * 16 + 4 * qd + r);
            }
        }
        if (lane == 0) wneed[wave] = mine;
        __builtin_amdgcn_s_barrier();
        if (tid == 0) {
            unsigned long long all = 0ull;
#pragma unroll
            for (int w = 0; w < K16_NW; ++w) all |= wneed[w];
            need[(long)blockIdx.x * n_cstages + s] = all;
        }
    }
}

// need bitmap -> per tile the ascending list of groups to run, and its length.  One wave per tile.
__global__ __launch_bounds__(64) void exact_lists_kernel(const unsigned long long* __restrict__ need, int n_cstages, int n_groups,
                                                         int* __restrict__ glist, int* __restrict__ gcnt,
                                                         int* __restrict__ blocks_run) {
    const long tile = blockIdx.x;
    const int lane = threadIdx.x;
    int base = 0;
    for (int s = 0; s < n_cstages; ++s) {
        const unsigned long long w = need[tile * n_cstages + s];
        if ((w >> lane) & 1ull) glist[tile * n_groups + base + __popcll(w & ((1ull << lane) - 1ull))] = s * 64 + lane;
        base += __popcll(w);
    }
    if (lane == 0) { gcnt[tile] = base; atomicAdd(blocks_run, base); }
}

}  // namespace somhip
