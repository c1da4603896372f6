// Fused distance + BMU argmin, bf16, on v_mfma_f32_16x16x32_bf16.
//
// The arithmetic of bmu_bf16.hpp (d' = B + |w~|^2/2 - x~.w~, the norm term as the MFMA's initial accumulator) on a
// stage image in MFMA fragment order.  On random data MI355X holds a higher clock on the 16x16x32 shape than on
// 32x32x16 at equal cycles per flop (MI355X_MICROARCH.md, DVFS give-back item 7), and this kernel is
// power/clock-limited, so the shape is a throughput lever by itself (round 1's 32x32x16 form took 13.99 ms where this one took 11.88).
//
// Geometry: A = 16 units x 32 features (lane l: unit l&15, features 8*(l>>4)+j), B = 32 features x
// 16 samples (lane l: sample l&15), C/D: lane holds sample l&15, units 4*(l>>4)+reg, reg 0..3.
// A wave keeps 4 x 16 samples' B fragments in registers; one A fragment read from LDS feeds 4 MFMAs.
// Stage image (K16_STAGE_UNITS = 64 units): [t16 0..3][kstep32][lane 0..63][8 bf16 of -w~] + [64 x f32 B+|w~|^2/2]
// (+ pad to 1 KiB).
// Key = (bits & ~mask) | (t16<<2 | reg).
//
// Grid = (sample blocks) x (codebook parts): a workgroup scans only its part of the stages and
// merges its winner into out64[n] = (value bits << 32 | unit) with a 64-bit atomicMin.  The host
// picks the number of parts so that the grid fills whole rounds of the resident workgroup slots
// (4096 blocks on 768 slots would idle 11 % in the last round; 4096 x 3 parts is exactly 16).
#pragma once
#include "bmu_bf16.hpp"

namespace somhip {

constexpr int K16_T = 4;              // 16-unit tiles per stage
constexpr int K16_STAGE_UNITS = 16 * K16_T;
constexpr int K16_SB = 4;             // 16-sample blocks per wave
constexpr int K16_NW = 4;             // waves per workgroup (they share one LDS ring)
constexpr int K16_WG_SAMPLES = K16_NW * 16 * K16_SB;

__host__ __device__ constexpr int k16_stage_bytes(int ks32) { return (K16_T * ks32 + 1) * 1024; }

template <int KS32, class EL = Bf16>
__global__ __launch_bounds__(256) void prep_w_bf16_k16_kernel(const float* __restrict__ W, int K, int D,
                                                              char* __restrict__ Wst, int n_stages,
                                                              const float* __restrict__ unit_wsq,
                                                              const float* __restrict__ scale_max2 = nullptr) {
    using E = typename EL::T;
    using bf16x8 = typename V8<E>::t;
    long id = (long)blockIdx.x * 256 + threadIdx.x;
    long total = (long)n_stages * K16_T * KS32 * 64;
    if (id >= total) return;
    int lane = id & 63;
    long t = id >> 6;
    int ks = t % KS32; t /= KS32;
    int t16 = t % K16_T;
    long stage = t / K16_T;
    long u = stage * K16_STAGE_UNITS + t16 * 16 + (lane & 15);
    int k0 = ks * 32 + (lane >> 4) * 8;
    // cosine distance (distances.py:45-59): rows are scaled to unit length, so that
    // argmin_k (B - x~ . w^~_k) == argmin_k (1 - x.w_k/(|x||w_k|)); a zero row stays zero
    // (its similarity is nan_to_num(0/0) = 0 in the reference, and x~ . 0 = 0 here).
    float scale = 1.0f;
    if (unit_wsq != nullptr && u < K) { float q = unit_wsq[u]; scale = q > 0.0f ? 1.0f / __builtin_sqrtf(q) : 0.0f; }
    if (scale_max2 != nullptr) scale = ex_scale(*scale_max2);   // exact mode on half operands: a power of two
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float f = (u < K && k0 + j < D) ? W[u * D + k0 + j] * scale : 0.0f;
        v[j] = cvt<E>(-f);
    }
    *(bf16x8*)(Wst + stage * k16_stage_bytes(KS32) + ((long)(t16 * KS32 + ks) * 64 + lane) * 16) = v;
}

// The exact mode's codebook preparation up to 128 features (bmu_exact.hpp): the stage image of the units scaled by the power
// of two ex_scale(*scale_max2) AND max_k |w^_k - w~_k|^2, what the MFMA will read instead of the scaled float32 units
// (half_operand_error), in one pass -- prep_w_bf16_k16_kernel + exact_werr_kernel.  One wave per 16-unit tile walks the
// tile's KS32 feature chunks (lane = (unit, 8 features) of each: one 16-byte fragment chunk of the image per chunk), so a
// unit's error is four lanes of one wave.
// Wst_lo (or null): the same fragments of the units' SECOND half -- lo = half(2^11 ((-w^) - hi)), so that -w^ = hi + 2^-11 lo up
// to 2^-11 of lo's own magnitude: what the exact mode's refinement pass (bmu_exact.hpp) multiplies beside hi.
template <int KS32, class EL = Bf16>
__global__ __launch_bounds__(256) void prep_w_exact_k16_kernel(const float* __restrict__ W, int K, int D,
                                                               char* __restrict__ Wst, int n_stages,
                                                               const float* __restrict__ scale_max2,
                                                               float* __restrict__ werr2, char* __restrict__ Wst_lo = nullptr) {
    using E = typename EL::T;
    using bf16x8 = typename V8<E>::t;
    const int lane = threadIdx.x & 63;
    const long tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6);          // (stage, t16)
    if (tile >= (long)n_stages * K16_T) return;
    const long stage = tile / K16_T;
    const int t16 = (int)(tile - stage * K16_T);
    const long u = stage * K16_STAGE_UNITS + t16 * 16 + (lane & 15);
    const float scale = ex_scale(*scale_max2);
    float er = 0.0f;
#pragma unroll
    for (int ks = 0; ks < KS32; ++ks) {
        const int k0 = ks * 32 + (lane >> 4) * 8;
        bf16x8 v, vl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float f = (u < K && k0 + j < D) ? W[u * D + k0 + j] * scale : 0.0f;
            const E hb = cvt<E>(-f);
            v[j] = hb;
            const float e = half_operand_error(-f, (float)hb);
            er = __builtin_fmaf(e, e, er);
            const float r = ((-f) - (float)hb) * 2048.0f;     // (exact: a float32 difference of neighbours, a power of two)
            vl[j] = cvt<E>(r == r && __builtin_fabsf(r) < 3.0e38f ? r : 0.0f);
        }
        *(bf16x8*)(Wst + stage * k16_stage_bytes(KS32) + ((long)(t16 * KS32 + ks) * 64 + lane) * 16) = v;
        if (Wst_lo != nullptr) *(bf16x8*)(Wst_lo + stage * k16_stage_bytes(KS32) + ((long)(t16 * KS32 + ks) * 64 + lane) * 16) = vl;
    }
    er += __shfl_xor(er, 16, 64);                            // the unit's four feature quarters
    er += __shfl_xor(er, 32, 64);
    float m = (er == er) ? er : 0.0f;                        // (a NaN unit is left out, as in exact_werr_kernel)
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) atomic_max_pos_f32(werr2, m);
}

// _merge_updates (xpysom.py:446-455) fused with the NEXT epoch's operand preparation: one pass over the fused
// accumulator writes the merged float32 codebook, the bf16 stage image of -w~ and |w~_k|^2 (+ its maximum),
// i.e. merge_kernel + prep_w_bf16_k16_kernel + prep_wnorm_kernel (euclidean) in one launch and one read of
// the codebook.  Workgroup = MP_TILES 16-unit tiles, KS32 waves; thread = (unit, 8 features) of each tile, exactly one
// 16-byte fragment chunk of the image.  W = where(den != 0, num / den, W) as merge_kernel computes it.
constexpr int MP_TILES = 4;           // 16-unit tiles per workgroup: their loads are all issued before the first is used
template <int KS32, class EL = Bf16>
__global__ __launch_bounds__(64 * KS32) void merge_prep_k16_kernel(float* __restrict__ W, const float* __restrict__ ACC,
                                                                 int K, int D, int D1p, char* __restrict__ Wst,
                                                                 float* __restrict__ wn, float* __restrict__ wmax2,
                                                                 long n_tiles) {
    using E = typename EL::T;
    using bf16x8 = typename V8<E>::t;
    __shared__ float red[MP_TILES][KS32][16];
    const int lane = threadIdx.x & 63, ks = threadIdx.x >> 6;
    const int k0 = ks * 32 + (lane >> 4) * 8;
    const bool full = k0 + 8 <= D && (D & 3) == 0;         // 16-byte aligned rows on both sides (D1p is a multiple of 4)
    float w[MP_TILES][8], den[MP_TILES];
    long unit[MP_TILES];
#pragma unroll
    for (int i = 0; i < MP_TILES; ++i) {                   // phase 1: every load of the workgroup's tiles
        const long tile = (long)blockIdx.x * MP_TILES + i;
        const long stage = tile / K16_T;
        const long u = stage * K16_STAGE_UNITS + (tile - stage * K16_T) * 16 + (lane & 15);
        unit[i] = (tile < n_tiles && u < K) ? u : -1;
        den[i] = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) w[i][j] = 0.0f;
        if (unit[i] >= 0) {
            den[i] = ACC[u * D1p + D];
            if (full) {
                const f32x4 a = *(const f32x4*)(ACC + u * D1p + k0), c = *(const f32x4*)(ACC + u * D1p + k0 + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { w[i][j] = a[j]; w[i][4 + j] = c[j]; }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (k0 + j < D) w[i][j] = ACC[u * D1p + k0 + j];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MP_TILES; ++i) {                   // phase 2: merge, codebook row, stage image, norm partials
        const long tile = (long)blockIdx.x * MP_TILES + i;
        const long stage = tile / K16_T;
        const int t16 = (int)(tile - stage * K16_T);
        const long u = unit[i];
        if (u >= 0) {
            if (den[i] != 0.0f) {
#pragma unroll
                for (int j = 0; j < 8; ++j) w[i][j] = w[i][j] / den[i];
                if (full) {
                    f32x4 q0, q1;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { q0[j] = w[i][j]; q1[j] = w[i][4 + j]; }
                    *(f32x4*)(W + u * D + k0) = q0;
                    *(f32x4*)(W + u * D + k0 + 4) = q1;
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (k0 + j < D) W[u * D + k0 + j] = w[i][j];
                }
            } else {                                       // no row in reach of this unit: the old weights stay
#pragma unroll
                for (int j = 0; j < 8; ++j) w[i][j] = (k0 + j < D) ? W[u * D + k0 + j] : 0.0f;
            }
        }
        if (tile < n_tiles) {
            bf16x8 v;
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const E b = cvt<E>(w[i][j]);
                v[j] = cvt<E>(-w[i][j]);                 // (rounding is sign-symmetric: -bf16(w) == bf16(-w))
                const float f = (float)b;
                s = __builtin_fmaf(f, f, s);
            }
            *(bf16x8*)(Wst + stage * k16_stage_bytes(KS32) + ((long)(t16 * KS32 + ks) * 64 + lane) * 16) = v;
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            if (lane < 16) red[i][ks][lane] = s;
        }
    }
    __syncthreads();
    if (ks == 0 && lane < 16) {
        float m = 0.0f;
#pragma unroll
        for (int i = 0; i < MP_TILES; ++i) {
            float t = 0.0f;
#pragma unroll
            for (int q = 0; q < KS32; ++q) t += red[i][q][lane];
            if (unit[i] >= 0) { wn[unit[i]] = t; m = fmaxf(m, t); }
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (lane == 0) atomic_max_pos_f32(wmax2, m);
    }
}

// GM (precision 'exact', bmu_exact.hpp): the kernel keeps VALUES only -- the row minimum (out64's upper half; no unit
// indices exist in this instance, the float32 re-score names the unit) and, per stage (= one GROUP of 64 units) and row,
// the group's minimum.  A lane holds its quad's minimum for each of the wave's 4 sample blocks; three v_permlane*_swap +
// v_min steps transpose and reduce so that quad q ends with the full minimum of sample block q -- lane l then owns row
// wave_s0 + l.  A group minimum can only matter to the scan if it is within the row's bound E of the FINAL row minimum,
// hence of the minimum so far: only those are stored (gmin[stage * gm_stride + row], exec-masked), and every wave and
// stage leaves the 64-bit mask of the lanes it stored in gflags[(row / 64) * n_stages + stage] -- on a random
// codebook ~2 % of the matrix is written and read, on the smoothest maps 10-25 %.
// TL (block skipping, exact_skip.hpp): the workgroup walks its tile's dense LIST of 16-unit tiles (group << 2 | sub-block)
// instead of all stages; only the listed tiles' fragments are staged and multiplied.  With GM: the exact mode's screen
// under a plan.  Without: the scout's pick of a pseudo last BMU among a tile's few listed groups (unit indices kept).
// (the kernel's body for one workgroup's worth of rows: tile bx of K16_WG_SAMPLES rows, part by of ny of the codebook stages -- or,
//  TL, of the tile's list.  The kernel below calls it once, or -- the exact mode's screen under a plan -- once per work item.)
template <int KS32, class EL, bool GM, bool TL>
__device__ __forceinline__ void bmu_bf16_k16_body(const __bf16* __restrict__ Xb, long N,
                                                  const char* __restrict__ Wst, int n_stages, int K,
                                                  unsigned long long* __restrict__ out64,
                                                  uint32_t* __restrict__ gmin, long gm_stride,
                                                  unsigned long long* __restrict__ gflags,
                                                  const float* __restrict__ xsq, const float* __restrict__ xerr,
                                                  const float* __restrict__ xmax2, const float* __restrict__ wmax2,
                                                  const float* __restrict__ werr2, const ExactBound& eb,
                                                  const float* __restrict__ seed, const int* __restrict__ glist,
                                                  const int* __restrict__ gcnt, const long bx, const int by, const int ny) {
    using E = typename EL::T;
    using bf16x8 = typename V8<E>::t;
    constexpr int DP = 32 * KS32;
    constexpr int STAGE = k16_stage_bytes(KS32);
    constexpr int PIECES = K16_T * KS32 + 1;
    constexpr uint32_t IDX_MASK = 4 * K16_T - 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int quad = lane >> 4, col = lane & 15;
    const long wave_s0 = bx * K16_WG_SAMPLES + wave * (16 * K16_SB);

    bf16x8 xf[K16_SB][KS32];
#pragma unroll
    for (int sb = 0; sb < K16_SB; ++sb) {
        const long row = wave_s0 + sb * 16 + col;
#pragma unroll
        for (int ks = 0; ks < KS32; ++ks) xf[sb][ks] = *(const bf16x8*)(Xb + row * DP + ks * 32 + quad * 8);
    }

    uint32_t gbest[K16_SB], cbest[K16_SB];   // unsigned: a NaN (either sign) is above every finite positive d'
    int gstage[K16_SB];
    f32x4 accP[K16_SB];
#pragma unroll
    for (int sb = 0; sb < K16_SB; ++sb) {
        gbest[sb] = 0xFFFFFFFFu; cbest[sb] = 0xFFFFFFFFu; gstage[sb] = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) accP[sb][r] = __builtin_inff();
    }

    float run_min = __builtin_inff(), row_e = 0.0f;       // GM: lane l <-> row wave_s0 + l: its minimum so far, its bound E
    int run_arg = 0;                                      //     ... and the group (stage) that holds it
    float run_cap = __builtin_inff();                     //     ... and the seed's cap on what can be selected at all
    if (GM) {
        const long r = wave_s0 + lane;
        row_e = r < N ? ex_row_bound(eb, ex_scales(xmax2, wmax2, werr2), xsq[r], xerr[r]) : __builtin_nanf("");
        if (!(row_e == row_e)) row_e = __builtin_inff();   // a row the bound does not cover: keep everything (the scan drops it)
        // seed (exact_seed_kernel): d'(t) + E/2 for the float32 score t of ONE unit of the row's choice (last epoch's BMU),
        // known before the scan.  A unit that beats that one in the float32 kernel has a screen value of at most seed +
        // the float32 share of E (exact_select_kernel applies the same cap): no group minimum above it is ever selected
        if (seed != nullptr && r < N) {
            const float sd = seed[r] + ex_f32_share(eb, ex_scales(xmax2, wmax2, werr2), xsq[r]);
            if (sd == sd) run_cap = sd;
        }
    }
    // this workgroup's share of the codebook stages: the loop index s walks [s_begin, s_end)  (TL: see the tile-list loop below)
    // (TL without GM: the scout of exact_skip.hpp -- the plain kernel, unit indices kept, over a tile's few listed groups)
    const int n_walk = n_stages;
    const int s_begin = TL ? 0 : (int)((long)n_walk * by / ny);
    const int s_end = TL ? 0 : (int)((long)n_walk * (by + 1) / ny);
    auto item_of = [&](int s) -> int { return s < s_end ? ((s << 4) | 15) : 15; };
    auto dma_item = [&](int it, char* dst) {
        const char* src = Wst + (long)(it >> 4) * STAGE;
        for (int p = wave; p < PIECES; p += K16_NW) lds_dma_16(src + (long)p * 1024 + lane * 16, dst + p * 1024);
    };
    int it_prev = 15, it_cur = item_of(s_begin), it_next = item_of(s_begin + 1);
    if (s_begin < s_end) dma_item(it_cur, smem);

    auto reduce_tile = [&](const f32x4 (&acc)[K16_SB], int t16) {
        if (GM) {
            // the exact mode's screen needs VALUES only (the row minimum and the groups' minima): which unit holds
            // them is decided by the float32 re-score.  Two v_min3 per tile and sample block, no key packing.
#pragma unroll
            for (int sb = 0; sb < K16_SB; ++sb) {
                cbest[sb] = min(min(cbest[sb], __float_as_uint(acc[sb][0])), __float_as_uint(acc[sb][1]));
                cbest[sb] = min(min(cbest[sb], __float_as_uint(acc[sb][2])), __float_as_uint(acc[sb][3]));
            }
            return;
        }
#pragma unroll
        for (int sb = 0; sb < K16_SB; ++sb) {
            uint32_t key[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float f = acc[sb][r];
                key[r] = (__float_as_uint(f) & ~IDX_MASK) | (uint32_t)(t16 * 4 + r);
            }
            cbest[sb] = min(min(cbest[sb], key[0]), key[1]);
            cbest[sb] = min(min(cbest[sb], key[2]), key[3]);
        }
    };
    auto fold_stage = [&](int stage, bool live) {
        if (GM && live) {
            static_assert(!GM || K16_SB == 4, "the group-minimum transpose pairs four 16-sample blocks with four lane quads");
            // rows of 16 lanes (quads) q0..q3 each hold (v0, v1, v2, v3): after the three steps quad q holds min over quads of v_q
            auto a = __builtin_amdgcn_permlane16_swap(cbest[0], cbest[K16_SB > 1 ? 1 : 0], false, false);
            const uint32_t t01 = min(a[0], a[1]);         // quads: (v0 q01, v1 q01, v0 q23, v1 q23)
            auto b = __builtin_amdgcn_permlane16_swap(cbest[K16_SB > 2 ? 2 : 0], cbest[K16_SB > 3 ? 3 : 0], false, false);
            const uint32_t t23 = min(b[0], b[1]);         // quads: (v2 q01, v3 q01, v2 q23, v3 q23)
            auto c = __builtin_amdgcn_permlane32_swap(t01, t23, false, false);
            const uint32_t full = min(c[0], c[1]);         // quads: (v0, v1, v2, v3), each over all four quads
            const float f = __uint_as_float(full);         // (positive, or a NaN pattern: compares false, never kept)
            const bool keep = f <= __builtin_fminf(run_min + row_e, run_cap);
            if (f < run_min) run_arg = stage;              // (the group that holds the row minimum: the first re-score round)
            run_min = __builtin_fminf(run_min, f);
            if (keep) gmin[(long)stage * gm_stride + wave_s0 + lane] = full;
            const unsigned long long mask = __ballot(keep);
            if (lane == 0) gflags[ex_flag_index(wave_s0 >> 6, stage, n_stages, gm_stride)] = mask;
        }
#pragma unroll
        for (int sb = 0; sb < K16_SB; ++sb) {
            if (!GM && cbest[sb] < gbest[sb]) { gbest[sb] = cbest[sb]; gstage[sb] = stage; }   // (GM: run_min is the row minimum)
            cbest[sb] = 0xFFFFFFFFu;
        }
    };

    SOM_STAMP_BEGIN();
    if (TL) {
        // TILE LISTS (block skipping, exact_skip.hpp): the workgroup walks its tile's dense list of 16-unit tiles
        // (group << 2 | sub-block, ascending), FOUR to a barrier whatever groups they belong to -- late in a schedule a
        // group keeps one or two of its four tiles, and a barrier per group left the pipe waiting.  A slot of the ring holds
        // the four tiles' fragments (by LDS-DMA) and their 16 initial accumulators each (a wave per tile: a plain load one
        // chunk ahead, written into the slot's tail before the next barrier).  A group's minimum is folded when its last
        // tile is in (its tiles are consecutive in the list, across chunks too).  The parts of a tile's list are cut at
        // group boundaries: one part stores a group's minima.
        const int* tl = glist + bx * (4 * n_stages);
        const int n_t = gcnt[bx];
        auto bound = [&](int y) -> int {
            int b = (int)((long)n_t * y / ny);
            while (b > 0 && b < n_t && (tl[b] >> 2) == (tl[b - 1] >> 2)) ++b;
            return __builtin_amdgcn_readfirstlane(b);
        };
        const int b0 = bound(by), e0 = bound(by + 1);
        auto ent = [&](int i) -> int { return i < e0 ? __builtin_amdgcn_readfirstlane(tl[i]) : -1; };
        constexpr int TQ = K16_T * KS32 * 1024;              // the slot's tail: [tile j][16] initial accumulators
        auto dma_chunk = [&](const int (&en)[4], char* dst) {
            for (int p = wave; p < K16_T * KS32; p += K16_NW) {
                const int j = p / KS32, ks = p - j * KS32;
                if (en[j] >= 0)
                    lds_dma_16(Wst + (long)(en[j] >> 2) * STAGE + (long)(((en[j] & 3) * KS32 + ks) * 1024) + lane * 16, dst + p * 1024);
            }
        };
        // wave w <-> the chunk's tile w: its 16 initial accumulators (lanes 0..15)
        auto load_wq = [&](const int (&en)[4]) -> float {
            const int e = en[wave];
            return (e >= 0 && lane < 16) ? *(const float*)(Wst + (long)(e >> 2) * STAGE + TQ + ((e & 3) * 16 + lane) * 4) : 0.0f;
        };
        int cur[4], nxt[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { cur[j] = ent(b0 + j); nxt[j] = ent(b0 + 4 + j); }
        if (b0 < e0) {
            dma_chunk(cur, smem);
            const float w0 = load_wq(cur);
            if (lane < 16) *(float*)(smem + TQ + (wave * 16 + lane) * 4) = w0;
        }
        int k = 0;
        for (int c = b0; c < e0; c += 4, ++k) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            int nn[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) nn[j] = ent(c + 8 + j);
            char* sn = smem + ((k + 1) & 1) * STAGE;
            float wqn = 0.0f;
            if (c + 4 < e0) { dma_chunk(nxt, sn); wqn = load_wq(nxt); }
            const char* st = smem + (k & 1) * STAGE;
            const float* wq = (const float*)(st + TQ);
            auto load_tile = [&](int j, bf16x8 (&a)[KS32], f32x4& wv) {
#pragma unroll
                for (int ks = 0; ks < KS32; ++ks) a[ks] = *(const bf16x8*)(st + (j * KS32 + ks) * 1024 + lane * 16);
                wv = *(const f32x4*)(wq + j * 16 + 4 * quad);
            };
            // MFMAs of one tile, its reduction, and the group's fold when the next tile belongs to another group
            auto run_tile = [&](const bf16x8 (&a)[KS32], const f32x4& wv, int e, int e_next) {
                f32x4 acc[K16_SB];
#pragma unroll
                for (int sb = 0; sb < K16_SB; ++sb) acc[sb] = wv;
#pragma unroll
                for (int ks = 0; ks < KS32; ++ks)
#pragma unroll
                    for (int sb = 0; sb < K16_SB; ++sb) acc[sb] = mfma16(a[ks], xf[sb][ks], acc[sb]);
                reduce_tile(acc, e & 3);                       // (GM: values only, the tile's place in its group is not kept)
                if (e_next < 0 || (e_next >> 2) != (e >> 2)) fold_stage(e >> 2, true);
            };
            // (the tiles' operands alternate between two register sets: the next tile's are read under this tile's MFMAs)
            bf16x8 aA[KS32], aB[KS32];
            f32x4 wvA, wvB;
            load_tile(0, aA, wvA);
            if (cur[1] >= 0) load_tile(1, aB, wvB);
            run_tile(aA, wvA, cur[0], cur[1] >= 0 ? cur[1] : -1);
            if (cur[1] >= 0) {
                if (cur[2] >= 0) load_tile(2, aA, wvA);
                run_tile(aB, wvB, cur[1], cur[2] >= 0 ? cur[2] : -1);
                if (cur[2] >= 0) {
                    if (cur[3] >= 0) load_tile(3, aB, wvB);
                    run_tile(aA, wvA, cur[2], cur[3] >= 0 ? cur[3] : -1);
                    if (cur[3] >= 0) run_tile(aB, wvB, cur[3], nxt[0]);
                }
            }
            if (c + 4 < e0 && lane < 16) *(float*)(sn + TQ + (wave * 16 + lane) * 4) = wqn;
#pragma unroll
            for (int j = 0; j < 4; ++j) { cur[j] = nxt[j]; nxt[j] = nn[j]; }
        }
    }
    for (int s = s_begin; s < s_end; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int it_next2 = item_of(s + 2);
        if (s + 1 < s_end) dma_item(it_next, smem + ((s + 1 - s_begin) & 1) * STAGE);
        const char* st = smem + ((s - s_begin) & 1) * STAGE;
        const float* wq = (const float*)(st + K16_T * KS32 * 1024);

        f32x4 wv = *(const f32x4*)(wq + 4 * quad);
        bf16x8 a[KS32];
#pragma unroll
        for (int ks = 0; ks < KS32; ++ks) a[ks] = *(const bf16x8*)(st + ks * 1024 + lane * 16);

#pragma unroll
        for (int t16 = 0; t16 < K16_T; ++t16) {
            f32x4 wvN;
            bf16x8 aN[KS32];
            if (t16 + 1 < K16_T) {
                wvN = *(const f32x4*)(wq + (t16 + 1) * 16 + 4 * quad);
#pragma unroll
                for (int ks = 0; ks < KS32; ++ks)
                    aN[ks] = *(const bf16x8*)(st + ((t16 + 1) * KS32 + ks) * 1024 + lane * 16);
            }
            f32x4 accT[K16_SB];
#pragma unroll
            for (int sb = 0; sb < K16_SB; ++sb) accT[sb] = wv;
#pragma unroll
            for (int ks = 0; ks < KS32; ++ks)
#pragma unroll
                for (int sb = 0; sb < K16_SB; ++sb)
                    accT[sb] = mfma16(a[ks], xf[sb][ks], accT[sb]);
            reduce_tile(accP, (t16 + K16_T - 1) % K16_T);
            if (t16 == 0) fold_stage(it_prev >> 4, s > s_begin);
#pragma unroll
            for (int sb = 0; sb < K16_SB; ++sb) accP[sb] = accT[sb];
            if (t16 + 1 < K16_T) {
                wv = wvN;
#pragma unroll
                for (int ks = 0; ks < KS32; ++ks) a[ks] = aN[ks];
            }
        }
        // (explicit sched_group_barrier / iglp_opt interleave requests were measured: equal or worse than
        //  hipcc's own schedule of this block -- DESIGN.md 3.4)
        it_prev = it_cur; it_cur = it_next; it_next = it_next2;
    }
    if (!TL) {
        reduce_tile(accP, K16_T - 1);
        fold_stage(it_prev >> 4, s_begin < s_end);
    }
    SOM_STAMP_END();

    if (GM) {
        // the row minimum IS the minimum so far after the last group (lane l <-> row wave_s0 + l): the plain value, every
        // bit of it, and the group that holds it (no unit indices exist here); parts merge by value, then lower group
        const long row = wave_s0 + lane;
        if (row < N) atomicMin(out64 + row, ((unsigned long long)__float_as_uint(run_min) << 32) | (uint32_t)run_arg);
        return;
    }

#pragma unroll
    for (int sb = 0; sb < K16_SB; ++sb) {
        uint32_t code = gbest[sb] & IDX_MASK;
        uint32_t unit = (uint32_t)gstage[sb] * K16_STAGE_UNITS + (code >> 2) * 16 + quad * 4 + (code & 3);
        // all distances are positive floats: (value bits, unit) orders as one unsigned 64-bit key
        unsigned long long comp = ((unsigned long long)(gbest[sb] & ~IDX_MASK) << 32) | unit;
        unsigned long long o = __shfl_xor(comp, 16, 64);
        if (o < comp) comp = o;
        o = __shfl_xor(comp, 32, 64);
        if (o < comp) comp = o;
        const long row = wave_s0 + sb * 16 + col;
        if (quad == 0 && row < N) atomicMin(out64 + row, comp);
    }
}

// items (TL; nullptr: the grid is (tiles, parts)): the screen under a plan as a WORK QUEUE.  The tiles' lists are uneven -- mid-schedule
// the median tile lists ~100 blocks, a few list 700 -- and a grid of one workgroup per tile ended in a tail of a few long walks
// that held a fifth to two fifths of the launch (tools/wg_timeline.py).  exact_items_kernel cuts every list into parts of about
// equal length (item = (tile, part | parts << 16)); a workgroup per slot of the chip takes items off one counter until none is left
// (every workgroup reaches `it >= *n_items`: the grid drains).  A part re-reads its tile's rows: only the long lists are cut.
template <int KS32, class EL = Bf16, bool GM = false, bool TL = false>
__global__ __launch_bounds__(64 * K16_NW, (TL && GM) ? 3 : 2) void bmu_bf16_k16_kernel(const __bf16* __restrict__ Xb, long N,
                                                              const char* __restrict__ Wst, int n_stages, int K,
                                                              unsigned long long* __restrict__ out64,
                                                              uint32_t* __restrict__ gmin = nullptr, long gm_stride = 0,
                                                              unsigned long long* __restrict__ gflags = nullptr,
                                                              const float* __restrict__ xsq = nullptr,
                                                              const float* __restrict__ xerr = nullptr,
                                                              const float* __restrict__ xmax2 = nullptr,
                                                              const float* __restrict__ wmax2 = nullptr,
                                                              const float* __restrict__ werr2 = nullptr,
                                                              ExactBound eb = ExactBound(),
                                                              const float* __restrict__ seed = nullptr,
                                                              const int* __restrict__ glist = nullptr,
                                                              const int* __restrict__ gcnt = nullptr,
                                                              const int2* __restrict__ items = nullptr,
                                                              const int* __restrict__ n_items = nullptr,
                                                              int* __restrict__ item_ctr = nullptr) {
    if (TL && items != nullptr) {
        __shared__ int s_item;
        const int n = *n_items;
        for (;;) {
            if (threadIdx.x == 0) s_item = atomicAdd(item_ctr, 1);
            __syncthreads();                                 // (everyone is done with the previous item's LDS ring too)
            const int it = s_item;
            __syncthreads();
            if (it >= n) return;
            const int2 iv = items[it];
            bmu_bf16_k16_body<KS32, EL, GM, TL>(Xb, N, Wst, n_stages, K, out64, gmin, gm_stride, gflags, xsq, xerr, xmax2, wmax2, werr2, eb,
                                                seed, glist, gcnt, (long)iv.x, iv.y & 0xFFFF, iv.y >> 16);
        }
    }
    bmu_bf16_k16_body<KS32, EL, GM, TL>(Xb, N, Wst, n_stages, K, out64, gmin, gm_stride, gflags, xsq, xerr, xmax2, wmax2, werr2, eb, seed,
                                        glist, gcnt, (long)blockIdx.x, (int)blockIdx.y, (int)gridDim.y);
}

// out64 -> raveled ids (a padding unit can only win on a NaN row: numpy's argmin gives 0 there)
__global__ __launch_bounds__(256) void bmu_finalize_kernel(const unsigned long long* __restrict__ out64, long N, int K,
                                                           int* __restrict__ out) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    uint32_t u = (uint32_t)out64[i];
    out[i] = u < (uint32_t)K ? (int)u : 0;
}

}  // namespace somhip
