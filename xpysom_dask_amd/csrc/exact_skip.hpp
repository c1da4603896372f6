// precision 'exact', euclidean, input_len <= 128, maps of >= 4096 units: BLOCK SKIPPING in the screen -- resident rows from their
// second epoch on (bound: last epoch's BMU) and, through the SCOUT at the end of this file, every other large row set.
//
// The screen (bmu_bf16_k16_kernel<.., GM>) is the exact epoch; what is left is not to run it where no BMU can be.  Two facts:
//   * last epoch's BMU u of a row x gives an upper bound on the distance to this epoch's BMU k*: the float32 kernel picks
//     k* with fl(tau(k*)) <= fl(tau(u)), so  |x - w_k*|^2 <= U(x) := |x|^2 + tau_up(u) + share,  tau_up an upper bound of
//     u's real score tau(u) = |w_u|^2 - 2 x.w_u under the CURRENT codebook, share the float32 kernel's error window;
//   * a block of units (a patch of the map) with centroid c and radius r = max |w - c| has  |x - w| >= |x - c| - r  for
//     each of its units.
// So a block holds no candidate of row x if  |x - c| > sqrt(U(x)) + r.
//
// ROWS stay resident in the order of their last BMU's patch (somhip.hip: a sorted copy of the operand image, the float32
// rows, the norms; re-sorted only when the order has gone stale -- most rows move to a neighbouring patch at most from
// one epoch to the next, and a 256-row tile of neighbours shares its blocks either way), so the 256 rows of a workgroup
// tile lie in one region of the map.
// BLOCKS come at two levels: the 64-unit GROUPS of the screen (one stage of its image = an 8 x 8 patch of the map) and
// their four 16-unit SUB-BLOCKS (the stage's four MFMA tiles = 4 x 4 squares of the patch where both map sides are multiples
// of 8 -- som_patch_order --, 2 x 8 strips elsewhere).  Late in a schedule the units of a patch spread out (its radius grows
// to the scale of the data) and the group bound alone keeps a quarter of all blocks; the sub-blocks' own centroids and radii
// cut that to a third.  The PLAN kernel below is the screen's MFMA loop
// on centroids with the test as its epilogue and an OR over the tile's rows: level 1 over the group centroids (1/64 of
// the units), level 2 over the sub-block centroids of the groups level 1 kept.  The screen then walks, per tile, the list
// of (group, 4-bit sub-block mask) items some row of it needs.
// A skipped (row, block) stores no minimum: to the select kernel it is a block outside the window, which is what the test
// proved.  The row minimum m(n) is over the blocks run: never below the true one, and the float32 winner's block is
// among them, so the selection argument of bmu_exact.hpp holds unchanged (the re-score evaluates whole groups).
// SEED: tau_up(u) comes from the operands the plan holds anyway -- the row's half image and unit u's fragments of the
// screen's stage image, a vector-ALU dot product in the plan's prologue: v_u = S (B + tau(u) / 2) up to the screen's
// one-unit error (measured operand rounding + accumulation), which is charged.  The same v_u caps what the screen
// stores at all (bmu_exact.hpp, seed).
// Rigour: |x - c|^2 = |x|^2 + tau_c is taken from the plan's screen value d'_c = S'(B' + tau_c / 2) less the screen's
// error bound for the centroid image (E_c / 2 of ex_row_bound on the centroids' scales), norms and radii are rounded
// outwards; a row whose quantities are not finite needs every block.
#pragma once
#include "bmu_bf16_k16.hpp"

namespace somhip {

constexpr int SK_TILE = K16_WG_SAMPLES;   // rows per plan / screen workgroup tile
static_assert(K16_STAGE_UNITS == 64 && K16_T == 4, "block skipping: a stage of the resident screen is one 64-unit group of four 16-unit tiles");

// Centroids and radii of the blocks of W (patch order), both levels in one launch: workgroup g holds group g's 64
// units in registers (thread <-> a feature of a sub-block; rows read whole: coalesced) and forms the centroid of the group and of each of
// its four 16-unit sub-blocks, their |c|^2, and the radii r = max |w - c| (a unit's |w - c|^2 is a wave reduction).
// Level 1: slot g = group g.  Level 2: slot 16 (g >> 2) + 4 (g & 3) + b = sub-block b of group g -- sixteen consecutive slots are
// the sub-blocks of four consecutive groups, one 16-row MFMA tile of the level-2 centroid image.  A slot without units:
// centroid 0, radius -1 (never needed).  The grid covers whole level-2 tiles (n_groups rounded up to four).
// cmax2 (per level {max |c|^2, max rounding error^2}): a centroid is a mean of units, |c| <= max |w|: the images take the
// codebook's own scale (no reduction over the centroids); the error slot is reset here and filled by the image kernel.
struct CentroidLevel { float* C; float* rg; float* csq; float* cmax2; int n_slots; };
// (eight waves: thread (feature d, quarter q) holds the sixteen units of sub-block q -- a group's 32 KB are in flight at once)
__global__ __launch_bounds__(512) void exact_centroids_kernel(const float* __restrict__ W, int K, int D, int n_groups,
                                                              CentroidLevel l1, CentroidLevel l2, const float* __restrict__ wmax2) {
    __shared__ float qsum[4][128];                            // the sub-blocks' feature sums
    __shared__ float part[2][64][2];                          // [feature half][unit][group / sub-block]: partial |w - c|^2
    __shared__ float csq_s[2][5];
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int d = tid & 127, q = tid >> 7, fh = (tid >> 6) & 1;   // feature, sub-block, which 64 features of the 128
    const long u0 = (long)g * 64 + 16 * q;
    const int cnt = (int)max(0L, min(64L, (long)K - (long)g * 64));
    const int cb = max(0, min(16, cnt - 16 * q));
    if (g == 0 && tid == 0) { l1.cmax2[0] = *wmax2; l1.cmax2[1] = 0.0f; l2.cmax2[0] = *wmax2; l2.cmax2[1] = 0.0f; }
    float w[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) w[k] = (d < D && k < cb) ? W[(u0 + k) * D + d] : 0.0f;
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) sum += w[k];
    qsum[q][d] = sum;
    __syncthreads();
    const float cs = cb > 0 ? sum / (float)cb : 0.0f;
    const float cg = cnt > 0 ? ((qsum[0][d] + qsum[1][d]) + (qsum[2][d] + qsum[3][d])) / (float)cnt : 0.0f;
    if (d < D) {
        if (q == 0 && g < l1.n_slots) l1.C[(long)g * D + d] = cg;
        const int j = 16 * (g >> 2) + 4 * (g & 3) + q;
        if (j < l2.n_slots) l2.C[(long)j * D + d] = cs;
    }
    // |c|^2 of the five centroids (float32, any order: the plan's margin covers it)
    {
        const float a = wave_sum(cs * cs);
        if (lane == 0) csq_s[fh][1 + q] = a;
        if (q == 0) { const float bq = wave_sum(cg * cg); if (lane == 0) csq_s[fh][0] = bq; }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (k < cb) {                                         // (uniform over the wave)
            const float dg = w[k] - cg, ds = w[k] - cs;
            const float qg = wave_sum(d < D ? dg * dg : 0.0f);    // (a NaN anywhere in the unit: NaN)
            const float qs = wave_sum(d < D ? ds * ds : 0.0f);
            if (lane == 0) { part[fh][16 * q + k][0] = qg; part[fh][16 * q + k][1] = qs; }
        }
    }
    __syncthreads();
    if (tid < 5) {
        // thread 0: the group; threads 1..4: its sub-blocks
        const int b = tid - 1;
        const int k0 = tid == 0 ? 0 : 16 * b, k1 = tid == 0 ? cnt : min(cnt, 16 * b + 16);
        float m = 0.0f;
        for (int k = k0; k < k1; ++k) {
            const float d2 = part[0][k][tid == 0 ? 0 : 1] + part[1][k][tid == 0 ? 0 : 1];
            m = (d2 > m || !(d2 == d2)) ? d2 : m;             // (a NaN unit: a NaN radius, the block is never skipped)
        }
        // (the sum of squares in float32, any order: relative error <= 128 * 2^-24; the margin covers it many times over)
        const float r = k1 > k0 ? __builtin_sqrtf(m) * (1.0f + 1.0f / 512.0f) + 1.0e-30f : -1.0f;
        const float sq = csq_s[0][tid] + csq_s[1][tid];
        if (tid == 0) { if (g < l1.n_slots) { l1.rg[g] = r; l1.csq[g] = sq; } }
        else {
            const int j = 16 * (g >> 2) + 4 * (g & 3) + b;
            if (j < l2.n_slots) { l2.rg[j] = r; l2.csq[j] = sq; }
        }
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

// position (patch order) of every sorted row's last BMU: what the plan's prologue reads instead of three dependent loads
__global__ __launch_bounds__(256) void exact_lastpos_kernel(const int* __restrict__ prev, const int* __restrict__ order,
                                                            const int* __restrict__ inv, long n, int K, int* __restrict__ lastpos) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    int u = prev[order[p]];
    u = u < 0 ? 0 : u >= K ? K - 1 : u;                      // (any unit gives a valid bound)
    lastpos[p] = inv != nullptr ? inv[u] : u;
}

// The pass in sorted order: image rows (DP halves), the float32 rows themselves (what the re-score reads: a candidate
// group's rows are then neighbours in memory; input_len a multiple of 8: even features first, then the odd ones), |x|^2, rounding error -- and the rows' SECOND half image for the refinement
// pass (bmu_exact.hpp): lo = half(2^11 (x^ - hi)), x^ = sx x.  Positions behind the pass's rows (up to the tile multiple) get
// zero rows and NaN norms (they keep nothing, need nothing).  Sixteen lanes per row.
template <class EL>
__global__ __launch_bounds__(256) void exact_gather_sorted_kernel(const int* __restrict__ order, long n, long np, int dp, int D,
                                                                  const __bf16* __restrict__ Xb, const float* __restrict__ X,
                                                                  const float* __restrict__ xsq, const float* __restrict__ xerr,
                                                                  const float* __restrict__ xmax2,
                                                                  __bf16* __restrict__ Xb_s, __bf16* __restrict__ Xl_s,
                                                                  float* __restrict__ Xf_s,
                                                                  float* __restrict__ xsq_s, float* __restrict__ xerr_s) {
    using E = typename EL::T;
    using bf16x8 = typename V8<E>::t;
    // (sixteen lanes per row -- dp <= 128 halves = at most sixteen 16-byte pieces --, four rows per wave: a wave per row left
    //  three quarters of its lanes idle and the kernel at a third of the fabric's rate)
    const int lane = threadIdx.x & 15;
    const long p = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (p >= np) return;
    const long r = p < n ? (long)order[p] : -1;
    const float sx = ex_scale(*xmax2);
    const bool vec = (D & 7) == 0;
    for (int c = lane; c < dp / 8; c += 16) {                 // 16-byte pieces of the half images = 8 features
        bf16x8 v, vl;
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[j] = (E)0.0f; vl[j] = (E)0.0f; }
        if (r >= 0) {
            v = *(const bf16x8*)((const char*)Xb + (r * dp + c * 8) * 2);
            float f[8];
            if (vec && c * 8 < D) {                           // (one read of the float32 row: its copy and its second half)
                // (the float32 copy DE-INTERLEAVED: the row's even features, then its odd ones -- a lane of the re-score's 32x32x2 MFMA
                //  feeds one parity of the features, and reads its half of the row in whole 16-byte pieces: exact_rescore_mfma_kernel)
                const f32x4 a = *(const f32x4*)(X + r * D + c * 8), b = *(const f32x4*)(X + r * D + c * 8 + 4);
                const f32x4 ev = {a[0], a[2], b[0], b[2]}, od = {a[1], a[3], b[1], b[3]};
                *(f32x4*)(Xf_s + p * D + c * 4) = ev;
                *(f32x4*)(Xf_s + p * D + D / 2 + c * 4) = od;
#pragma unroll
                for (int j = 0; j < 4; ++j) { f[j] = a[j]; f[4 + j] = b[j]; }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    f[j] = c * 8 + j < D ? X[r * D + c * 8 + j] : 0.0f;
                    if (c * 8 + j < D) Xf_s[p * D + c * 8 + j] = f[j];
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = (f[j] * sx - (float)v[j]) * 2048.0f;
                vl[j] = cvt<E>(d == d && __builtin_fabsf(d) < 3.0e38f ? d : 0.0f);
            }
        }
        *(bf16x8*)((char*)Xb_s + (p * dp + c * 8) * 2) = v;
        if (Xl_s != nullptr) *(bf16x8*)((char*)Xl_s + (p * dp + c * 8) * 2) = vl;   // (allocated when the refinement first engages)
    }
    if (lane == 0) {
        const float nanv = __builtin_nanf("");
        xsq_s[p] = r >= 0 ? xsq[r] : nanv;
        xerr_s[p] = r >= 0 ? xerr[r] : nanv;
    }
}

// a positive float32 rounded UP to a NORMAL IEEE half (what one MFMA operand can carry of it without ever reading less):
// at least 2^-14 (an MFMA that flushes subnormal inputs would read 0 below that), +inf beyond 65504, NaN stays NaN
__device__ __forceinline__ float up_to_half(float v) {
    if (!(v == v)) return v;
    if (v > 65504.0f) return __builtin_inff();
    if (v < 0x1p-14f) return 0x1p-14f;
    const _Float16 hv = (_Float16)v;                         // round to nearest even
    float f = (float)hv;
    if (f < v) f = (float)__builtin_bit_cast(_Float16, (unsigned short)(__builtin_bit_cast(unsigned short, hv) + 1));   // next half up
    return f;
}

// OR of a value over the sixteen lanes of its DPP row (a quad of the MFMA layout: lanes 16 q .. 16 q + 15), in every lane of the
// row: four rotations inside the row, no LDS, no scalar unit
__device__ __forceinline__ uint32_t row16_or(uint32_t v) {
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xF, 0xF, false);   // row_ror:1
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xF, 0xF, false);   // row_ror:2
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xF, 0xF, false);   // row_ror:4
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, false);   // row_ror:8
    return v;
}

// The test of the plan, arranged so that the MFMA evaluates all of it but one comparison per (row, centroid):
//   skip  <=>  d'_c > A(row) + hS (sU(row) + r_c)^2  =  [A + hS sU^2] + hS r_c^2 + 2 hS sU r_c
//         <=>  (d'_c - hS r_c^2) - (sx sU (1 + 2^-10)) (sw r_c)  >  P(row) := A + hS sU^2,        2 hS = S (1 + 2^-10), S = sx sw.
// hS r_c^2 leaves the initial accumulator, the cross term is ONE more feature: the rows carry up_to_half(sx sU (1 + 2^-10)), the
// centroids -up_to_half(sw r_c) -- both rounded up, so the product the MFMA forms is never below the real one (the test errs
// towards "needed") -- in the first slot of an extra 32-feature step.
// The centroid stage images of both levels in one launch (a wave per 16-slot MFMA tile; the tiles of level 1 first): the
// fragments of -c~ as prep_w_exact_k16_kernel lays out the units' (scaled by the codebook's power of two; the measured rounding
// error's maximum into cmax2[1]), and the stage's tail: [0, 64) floats: initial accumulators S'(B' + |c|^2 / 2) - hS r^2,
// [64, 128): -up_to_half(sw r) (an empty slot: +inf and 0: never needed; a NaN radius: NaN: always needed).
template <int KS32, class EL>
__global__ __launch_bounds__(256) void exact_centroid_image_kernel(CentroidLevel l1, char* __restrict__ Cst1, int n_cstages1,
                                                                   CentroidLevel l2, char* __restrict__ Cst2, int n_cstages2, int D,
                                                                   const float* __restrict__ xmax2, const float* __restrict__ wmax2,
                                                                   char* __restrict__ Cst1_plain = nullptr) {
    using E = typename EL::T;
    using bf16x8 = typename V8<E>::t;
    constexpr int STAGE = k16_stage_bytes(KS32);
    const int lane = threadIdx.x & 63;
    long tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long t1 = (long)n_cstages1 * K16_T;
    const bool second = tile >= t1;
    if (second) tile -= t1;
    if (tile >= (long)(second ? n_cstages2 : n_cstages1) * K16_T) return;
    const CentroidLevel& lv = second ? l2 : l1;
    char* Cst = second ? Cst2 : Cst1;
    const long stage = tile / K16_T;
    const int t16 = (int)(tile - stage * K16_T);
    const long slot = stage * K16_STAGE_UNITS + t16 * 16 + (lane & 15);
    const float sw = ex_scale(*wmax2), sx = ex_scale(*xmax2);
    float er = 0.0f;
#pragma unroll
    for (int ks = 0; ks < KS32; ++ks) {
        const int k0 = ks * 32 + (lane >> 4) * 8;
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float f = (slot < lv.n_slots && k0 + j < D) ? lv.C[slot * D + k0 + j] * sw : 0.0f;
            const E hb = cvt<E>(-f);
            v[j] = hb;
            const float e = half_operand_error(-f, (float)hb);
            er = __builtin_fmaf(e, e, er);
        }
        *(bf16x8*)(Cst + stage * STAGE + ((long)(t16 * KS32 + ks) * 64 + lane) * 16) = v;
        if (!second && Cst1_plain != nullptr) *(bf16x8*)(Cst1_plain + stage * STAGE + ((long)(t16 * KS32 + ks) * 64 + lane) * 16) = v;
    }
    er += __shfl_xor(er, 16, 64);                            // the centroid's four feature quarters
    er += __shfl_xor(er, 32, 64);
    float m = (er == er) ? er : 0.0f;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0) atomic_max_pos_f32(lv.cmax2 + 1, m);
    if (lane < 16) {
        float* tail = (float*)(Cst + (stage + 1) * (long)STAGE - 1024);
        const int within = t16 * 16 + lane;
        const float rad = slot < lv.n_slots ? lv.rg[slot] : -1.0f;
        // (the scout's copy of the level-1 image: the plain screen value of the centroid, S'(B' + |c|^2 / 2) -- nearest centroid)
        float* ptail = (!second && Cst1_plain != nullptr) ? (float*)(Cst1_plain + (stage + 1) * (long)STAGE - 1024) : nullptr;
        if (rad < 0.0f) { tail[within] = __builtin_inff(); tail[64 + within] = 0.0f; if (ptail) ptail[within] = __builtin_inff(); }
        else {
            const float big = __builtin_sqrtf(*wmax2) * __builtin_sqrtf(*xmax2) * (1.0f + 1.0f / 1024.0f);   // (ex_scales: B')
            const float S = sx * sw, hS = 0.5f * S * (1.0f + 1.0f / 1024.0f);
            const float s0 = __builtin_fmaf(0.5f * S, lv.csq[slot], S * big);
            tail[within] = s0 - hS * rad * rad * (1.0f + 0x1p-20f);
            tail[64 + within] = -up_to_half(sw * rad);
            if (ptail) ptail[within] = (s0 == s0 && s0 < 3.0e38f) ? s0 : __builtin_inff();
        }
    }
}

// The plan: which blocks does a tile of SK_TILE (sorted) rows need?  The resident kernel's MFMA loop over a centroid
// stage image (64 centroids per stage), epilogue: need(row, j) = not (d'_c > A(row) + (S'/2) (sU(row) + r_j)^2), OR over
// the tile's rows into need[tile][stage] (bit i <-> centroid slot 64 stage + i).
//   LEVEL2 == false: the centroids of the groups.  The prologue also forms, per row, the screen value v_u of the row's last
//     BMU u (vector ALU, from the row's half image and u's fragments of the screen's stage image Wst), from it the seed of
//     the screen (seed_s) and sqrt(U) (sU_s: kept for level 2; +inf: the row needs everything).
//   LEVEL2 == true: the centroids of the 16-unit sub-blocks, slot order as exact_centroid_kernel's (a 16-slot MFMA tile =
//     the sub-blocks of four consecutive groups); only the tiles whose groups level 1 kept (need1) are loaded and run.
// eb / scales: the centroid image's (cmax2 = {max |c|^2, max rounding error^2}); wmax2 / werr2: the codebook's.
template <int KS32, class EL, bool LEVEL2>
__global__ __launch_bounds__(64 * K16_NW, LEVEL2 ? 3 : 2) void exact_plan_kernel(const __bf16* __restrict__ Xb, long N,
                                                                    const char* __restrict__ Cst, int n_cstages,
                                                                    const float* __restrict__ rg, int n_slots,
                                                                    const float* __restrict__ xsq, const float* __restrict__ xerr,
                                                                    float* __restrict__ sU_s,
                                                                    const float* __restrict__ xmax2, const float* __restrict__ cmax2,
                                                                    const float* __restrict__ wmax2, const float* __restrict__ werr2,
                                                                    ExactBound eb, unsigned long long* __restrict__ need,
                                                                    const int* __restrict__ lastpos, const char* __restrict__ Wst,
                                                                    float* __restrict__ seed_s,
                                                                    const unsigned long long* __restrict__ need1, int n_cstages1,
                                                                    int force_all, const int* __restrict__ lastpos2 = nullptr,
                                                                    int* __restrict__ scout_wins = nullptr) {
    using E = typename EL::T;
    using bf16x8 = typename V8<E>::t;
    constexpr int DP = 32 * KS32;
    constexpr int STAGE = k16_stage_bytes(KS32);
    constexpr int PIECES = K16_T * KS32 + 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int act_n;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int quad = lane >> 4, col = lane & 15;
    const long wave_s0 = (long)blockIdx.x * K16_WG_SAMPLES + wave * (16 * K16_SB);

    bf16x8 xf[K16_SB][KS32];
    bf16x8 xe[K16_SB];                                       // the extra feature step: slot 0 = up_to_half(sx sU (1 + 2^-10)), else 0
    float A[K16_SB], sU[K16_SB], P[K16_SB];
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
            const float q = xsq[row], xe = xerr[row];
            float su;
            if (!LEVEL2) {
                // v_u: the screen's value of unit u = the row's last BMU on the operands the screen reads (-w~ fragments of
                // the stage image, the row's x~), a float32 fma chain: every product of two halves is exact, the chain and
                // the two cross-quad adds err by < 40 ulps of the largest accumulator magnitude (charged: e_valu)
                // (lastpos2: a second unit per row -- the scout's pick beside the real last BMU --: the smaller value bounds as well)
                auto value_of = [&](int pos) -> float {           // (a position below K: exact_lastpos_kernel, exact_scout_pos_kernel)
                    const char* stg = Wst + (long)(pos >> 6) * STAGE;
                    const int t16 = (pos >> 4) & 3, c = pos & 15;
                    float d = 0.0f;
#pragma unroll
                    for (int ks = 0; ks < KS32; ++ks) {
                        const bf16x8 a = *(const bf16x8*)(stg + ((t16 * KS32 + ks) * 64 + quad * 16 + c) * 16);
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) d = __builtin_fmaf((float)a[jj], (float)xf[sb][ks][jj], d);
                    }
                    d += __shfl_xor(d, 16, 64);
                    d += __shfl_xor(d, 32, 64);
                    return *(const float*)(stg + K16_T * KS32 * 1024 + (t16 * 16 + c) * 4) + d;
                };
                float vu = value_of(lastpos[row]);
                if (lastpos2 != nullptr) {
                    // (lastpos: the scout's pick, lastpos2: the real last BMU.  A WIN of the scout: its unit's squared distance --
                    //  |x|^2 + tau(v), tau(v) = 2 (v / S - B) -- is a tenth or more below the last BMU's: counted, so that the host
                    //  knows whether the scout is still worth its launches next epoch)
                    const float v2 = value_of(lastpos2[row]);
                    const float Sq = sw.sx * sw.sw;
                    const float ua = q + 2.0f * (vu / Sq - sw.big), ub = q + 2.0f * (v2 / Sq - sw.big);
                    const bool win = ua < 0.9f * ub;
                    const unsigned long long wm = __ballot(win && quad == 0);
                    if (scout_wins != nullptr && lane == __builtin_ctzll(__ballot(true)) && wm != 0ull) atomicAdd(scout_wins, (int)__popcll(wm));
                    vu = (v2 < vu || !(vu == vu)) ? v2 : vu;      // (a NaN value -- a NaN unit -- gives way to the other unit's)
                }
                const float Sw = sw.sx * sw.sw;
                const float e = ex_row_bound(eb, sw, q, xe);          // two-unit window E of the screen (d' units)
                const float f32s = ex_f32_share(eb, sw, q);           // its float32 share (two evaluations)
                const float e_valu = 48.0f * 0x1p-24f * Sw * sw.bmag;
                // seed: whatever beats u in the float32 kernel has a screen value of at most v_u + E + e_valu = seed + f32s
                const float sd = vu + (e - f32s) * (1.0f + 1.0f / 1024.0f) + e_valu + Sw * sw.bmag * 0x1p-21f;
                if (quad == 0) seed_s[row] = (sd > 0.0f && sd < 3.0e38f) ? sd : __builtin_inff();
                // tau_up >= u's real score: d'_real(u) <= v_u + (one-unit screen error) + e_valu,  tau = 2 (d' / S - B)
                const float e_one = 0.5f * (e - f32s) * (1.0f + 1.0f / 1024.0f) + e_valu;
                const float tau_up = 2.0f * (((vu + e_one) - Sw * sw.big) / Sw) + 0x1p-20f * (sw.big + __builtin_fabsf(vu) / Sw);
                const float xn = __builtin_sqrtf(q) * (1.0f + 1.0f / 1024.0f);
                const float share = 2.0f * (eb.cA * xn * sw.wm + eb.cW * sw.wm * sw.wm);      // tau units, one float32 window
                // (no fmax here: it would swallow the NaN of a row whose last BMU is a NaN unit, and bound that row by zero)
                const float U0 = q * (1.0f + 1.0f / 1024.0f) + tau_up + share * (1.0f + 1.0f / 1024.0f) + 0x1p-18f * (q + __builtin_fabsf(tau_up));
                const float U = U0 < 0.0f ? 0.0f : U0;
                su = __builtin_sqrtf(U) * (1.0f + 1.0f / 1024.0f);
                if (force_all || !(su == su) || !(su < 3.0e38f) || !(e == e)) su = __builtin_inff();
                if (quad == 0) sU_s[row] = su;
            } else {
                su = sU_s[row];
            }
            const float ec = 0.5f * ex_row_bound(eb, sc, q, xe) * (1.0f + 1.0f / 1024.0f);
            sU[sb] = su;
            // d'_c > S' (B' + ((sU + r)^2 - |x|^2_lo) / 2) + e_c   <=>   skip
            // (+ margins: the float32 rounding of this line, and |c|^2 as float32 summed it against the real |c|^2)
            A[sb] = S * sc.big + ec - 0.5f * S * q * (1.0f - 1.0f / 1024.0f) + 0x1p-12f * S * (sc.big + q) + 0x1p-16f * S * sc.wm * sc.wm;
            if (!(A[sb] == A[sb]) || !(A[sb] < 3.0e38f) || !(su < 3.0e38f)) {
                A[sb] = __builtin_inff(); sU[sb] = 0.0f;                                         // need everything
            }
        }
        // P = A + hS sU^2 (+ the extra feature step's and the larger accumulators' share of the MFMA rounding: the cross term
        // can double the accumulator's magnitude, and there is one more MFMA in the chain: 4 (KS32 + 1) x 6 ulps of S' Bm')
        const float bx = up_to_half(sc.sx * sU[sb] * (1.0f + 1.0f / 1024.0f));
        P[sb] = A[sb] + 0.5f * S * (1.0f + 1.0f / 1024.0f) * sU[sb] * sU[sb] * (1.0f + 0x1p-20f) + (float)(24 * (KS32 + 1)) * 0x1p-23f * S * sc.bmag;
        if (!(bx < 3.0e38f) || !(P[sb] == P[sb])) P[sb] = __builtin_inff();                      // (sU beyond the half range: need everything)
        if (A[sb] == -__builtin_inff()) P[sb] = -__builtin_inff();                               // (rows behind the pass)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) xe[sb][jj] = (E)0.0f;
        if (quad == 0 && bx < 3.0e38f) xe[sb][0] = (E)bx;
    }

    // the words this workgroup produces (one per stage), gathered in LDS with LDS atomics (a global atomic per wave and stage
    // sat in front of the next stage's barrier)
    int* act = (int*)(smem + 2 * STAGE);
    unsigned long long* nl = (unsigned long long*)(smem + 2 * STAGE + (LEVEL2 ? (size_t)n_cstages1 * 64 * sizeof(int) : 0));
    for (int i = tid; i < n_cstages; i += 64 * K16_NW) nl[i] = 0ull;
    if constexpr (LEVEL2) {
        // LEVEL 2 walks a DENSE list of the groups level 1 kept, sixteen to a barrier: four MFMA tiles of four groups' sub-blocks
        // each, GATHERED -- late in a schedule a quarter of the groups is kept, spread so evenly that nearly every tile of
        // four consecutive groups holds one: walking image tiles ran four times level 1's work to test a quarter of it.
        // A tile's fragments come by LDS-DMA with per-lane addresses (lane (quad, c) <-> sub-block c & 3 of the tile's
        // (c >> 2)-th group), its initial accumulators and radii by plain loads one chunk ahead (a wave per tile), written
        // into the slot's tail (laid out like a stage's) before the next barrier.
        static_assert(K16_NW == 4, "a wave per tile fills the slot's tail");
        if (wave == 0) {
            int cnt = 0;
            for (int s1 = 0; s1 < n_cstages1; ++s1) {
                const unsigned long long w = need1[(long)blockIdx.x * n_cstages1 + s1];
                const int g = s1 * 64 + lane;
                const bool on = ((w >> lane) & 1ull) && 4 * (g >> 2) * 4 < n_slots;
                const unsigned long long mk = __ballot(on);
                if (on) act[cnt + __popcll(mk & ((1ull << lane) - 1ull))] = g;
                cnt += __popcll(mk);
            }
            if (lane == 0) act_n = cnt;
        }
        __syncthreads();
        const int n_act = act_n;
        const int c_all = (n_act + 15) / 16;
        const int b0 = 16 * (int)((long)c_all * blockIdx.y / gridDim.y);
        const int e0 = min(n_act, 16 * (int)((long)c_all * (blockIdx.y + 1) / gridDim.y));
        constexpr int TQ = K16_T * KS32 * 1024;              // the slot's tail: [0, 64) initial accumulators, [64, 128) -up_to_half(sw r)
        // where sub-block `sub` of group g sits in the level-2 image: stage g >> 4, tile (g >> 2) & 3, row 4 (g & 3) + sub
        auto frag_of = [&](int g, int sub, int ks, int q) -> const char* {
            return Cst + (long)(g >> 4) * STAGE + (((((g >> 2) & 3) * KS32 + ks) * 64 + q * 16 + 4 * (g & 3) + sub) * 16);
        };
        auto dma_chunk = [&](int c0, char* dst) {
            for (int p = wave; p < K16_T * KS32; p += K16_NW) {
                const int j = p / KS32, ks = p - j * KS32;
                if (c0 + 4 * j < e0) {                        // (wave-uniform)
                    const int gi = c0 + 4 * j + (col >> 2);
                    const int g = gi < e0 ? act[gi] : act[c0 + 4 * j];   // (a slot behind the list: any valid address; its bits are dropped)
                    lds_dma_16(frag_of(g, col & 3, ks, quad), dst + p * 1024);
                }
            }
        };
        // wave w <-> the chunk's tile w: lanes 0..15 its initial accumulators, lanes 16..31 its radii (row i <-> sub-block i & 3 of
        // the tile's (i >> 2)-th group)
        auto load_tail = [&](int c0) -> float {
            const int i = lane & 15, gi = c0 + 4 * wave + (i >> 2);
            if (lane >= 32 || gi >= e0) return lane < 16 ? __builtin_inff() : 0.0f;     // (an empty slot: never needed)
            const int g = act[gi];
            return *(const float*)(Cst + (long)(g >> 4) * STAGE + TQ + (((lane >> 4) * 64) + ((g >> 2) & 3) * 16 + 4 * (g & 3) + (i & 3)) * 4);
        };
        auto store_tail = [&](char* dst, float v) {
            if (lane < 32) *(float*)(dst + TQ + (((lane >> 4) * 64) + wave * 16 + (lane & 15)) * 4) = v;
        };
        if (b0 < e0) { dma_chunk(b0, smem); store_tail(smem, load_tail(b0)); }
        int k = 0;
        for (int c0 = b0; c0 < e0; c0 += 16, ++k) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            char* sn = smem + ((k + 1) & 1) * STAGE;
            float tail_n = 0.0f;
            if (c0 + 16 < e0) { dma_chunk(c0 + 16, sn); tail_n = load_tail(c0 + 16); }
            const char* st = smem + (k & 1) * STAGE;
            const float* wq = (const float*)(st + TQ);
            auto load_tile = [&](int j, bf16x8 (&a)[KS32], f32x4& wv, float& rneg) {
#pragma unroll
                for (int ks = 0; ks < KS32; ++ks) a[ks] = *(const bf16x8*)(st + ((j * KS32 + ks) * 64 + lane) * 16);
                wv = *(const f32x4*)(wq + j * 16 + 4 * quad);
                rneg = wq[64 + j * 16 + col];
            };
            uint32_t lane_bits = 0u;                           // bit 4 j + r: tile j of the chunk, accumulator register r
            auto run_tile = [&](int j, const bf16x8 (&a)[KS32], const f32x4& wv, float rneg) {
                bf16x8 ae;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) ae[jj] = (E)0.0f;
                if (quad == 0) ae[0] = (E)rneg;
                f32x4 acc[K16_SB];
#pragma unroll
                for (int sb = 0; sb < K16_SB; ++sb) acc[sb] = wv;
#pragma unroll
                for (int ks = 0; ks < KS32; ++ks)
#pragma unroll
                    for (int sb = 0; sb < K16_SB; ++sb) acc[sb] = mfma16(a[ks], xf[sb][ks], acc[sb]);
#pragma unroll
                for (int sb = 0; sb < K16_SB; ++sb) acc[sb] = mfma16(ae, xe[sb], acc[sb]);
                // (lane (quad, col): bit r <-> its row needs sub-block r of the tile's quad-th group; gathered over the rows
                //  once per chunk, below -- a ballot and four scalar tests per accumulator register kept the scalar unit busier
                //  than the matrix pipe)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    bool nd = false;
#pragma unroll
                    for (int sb = 0; sb < K16_SB; ++sb) nd = nd || !(acc[sb][r] > P[sb]);
                    lane_bits |= nd ? (1u << (4 * j + r)) : 0u;
                }
            };
            bf16x8 aA[KS32], aB[KS32];
            f32x4 wvA, wvB;
            float rnA, rnB;
            const bool h1 = c0 + 4 < e0, h2 = c0 + 8 < e0, h3 = c0 + 12 < e0;
            load_tile(0, aA, wvA, rnA);
            if (h1) load_tile(1, aB, wvB, rnB);
            run_tile(0, aA, wvA, rnA);
            if (h1) {
                if (h2) load_tile(2, aA, wvA, rnA);
                run_tile(1, aB, wvB, rnB);
                if (h2) {
                    if (h3) load_tile(3, aB, wvB, rnB);
                    run_tile(2, aA, wvA, rnA);
                    if (h3) run_tile(3, aB, wvB, rnB);
                }
            }
            // the chunk's sixteen groups: OR over the sixteen rows of every quad (lanes of one DPP row), then the quad's first lane
            // files tile j's nibble under the tile's quad-th group
            const uint32_t any_row = row16_or(lane_bits);
            if (col == 0 && any_row != 0u) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t nib = (any_row >> (4 * j)) & 15u;
                    const int gi = c0 + 4 * j + quad;
                    if (nib != 0u && gi < e0) {
                        const int g = act[gi];
                        atomicOr(nl + (g >> 4), (unsigned long long)nib << (4 * (g & 15)));
                    }
                }
            }
            if (c0 + 16 < e0) store_tail(sn, tail_n);
        }
        __syncthreads();
        // (two parts of a tile's walk may share a stage: OR into the words the host cleared)
        for (int i = tid; i < n_cstages; i += 64 * K16_NW)
            if (nl[i] != 0ull) atomicOr(need + (long)blockIdx.x * n_cstages + i, nl[i]);
        return;
    }
    const int n_walk = n_cstages;
    // (gridDim.y workgroups share a tile's centroid stages: few tiles -- a batch of 65 536 rows is 256 -- would otherwise be
    //  one workgroup per CU walking all the stages alone)
    const int s_begin = (int)((long)n_walk * blockIdx.y / gridDim.y);
    const int s_end = (int)((long)n_walk * (blockIdx.y + 1) / gridDim.y);
    auto item_of = [&](int i) -> int { return (i << 4) | 15; };
    auto dma_stage = [&](int s, uint32_t tm, char* dst) {
        const char* src = Cst + (long)s * STAGE;
        for (int p = wave; p < PIECES; p += K16_NW)
            if (p == PIECES - 1 || ((tm >> (p / KS32)) & 1u)) lds_dma_16(src + (long)p * 1024 + lane * 16, dst + p * 1024);
    };
    int st_cur = 0, st_next = 0;
    uint32_t tm_cur = 0, tm_next = 0;
    if (s_begin < s_end) {
        const int it = item_of(s_begin);
        st_cur = it >> 4; tm_cur = (uint32_t)it & 15u;
        dma_stage(st_cur, tm_cur, smem);
    }
    for (int i = s_begin; i < s_end; ++i) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (i + 1 < s_end) {
            const int it = item_of(i + 1);
            st_next = it >> 4; tm_next = (uint32_t)it & 15u;
            dma_stage(st_next, tm_next, smem + ((i + 1 - s_begin) & 1) * STAGE);
        }
        const char* st = smem + ((i - s_begin) & 1) * STAGE;
        const float* wq = (const float*)(st + K16_T * KS32 * 1024);
        uint32_t lane_bits = 0u;                              // (the word's bit 16 t16 + 4 quad + r <-> centroid slot of that place in the stage)
        // a tile's operands: its KS32 fragments, its initial accumulators, the centroids' side of the extra step (slot 0 =
        // -up_to_half(sw r)); the NEXT tile's are read under this tile's MFMAs and epilogue
        auto load_tile = [&](int t, bf16x8 (&a)[KS32], f32x4& wv, float& rneg) {
#pragma unroll
            for (int ks = 0; ks < KS32; ++ks) a[ks] = *(const bf16x8*)(st + ((t * KS32 + ks) * 64 + lane) * 16);
            wv = *(const f32x4*)(wq + t * 16 + 4 * quad);
            rneg = wq[64 + t * 16 + col];
        };
        // MFMAs + test of one tile; its operands in (a, wv, rneg)
        auto run_tile = [&](int t16, const bf16x8 (&a)[KS32], const f32x4& wv, float rneg) {
            bf16x8 ae;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) ae[jj] = (E)0.0f;
            if (quad == 0) ae[0] = (E)rneg;
            f32x4 acc[K16_SB];
#pragma unroll
            for (int sb = 0; sb < K16_SB; ++sb) acc[sb] = wv;
#pragma unroll
            for (int ks = 0; ks < KS32; ++ks)
#pragma unroll
                for (int sb = 0; sb < K16_SB; ++sb) acc[sb] = mfma16(a[ks], xf[sb][ks], acc[sb]);
#pragma unroll
            for (int sb = 0; sb < K16_SB; ++sb) acc[sb] = mfma16(ae, xe[sb], acc[sb]);
            // lane (quad, col): bit 4 t16 + r <-> its row needs centroid 4 quad + r of tile t16 (gathered over the rows once per stage)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                bool nd = false;
#pragma unroll
                for (int sb = 0; sb < K16_SB; ++sb) nd = nd || !(acc[sb][r] > P[sb]);
                lane_bits |= nd ? (1u << (4 * t16 + r)) : 0u;
            }
        };
        // up to four tiles, their operands alternating between two register sets (no copies): the next tile's are read
        // under this tile's MFMAs and test
        const uint32_t m0 = 15u, m1 = m0 & (m0 - 1u), m2 = m1 & (m1 - 1u), m3 = m2 & (m2 - 1u);
        const int t0 = __builtin_ctz(m0), t1 = m1 ? __builtin_ctz(m1) : -1, t2 = m2 ? __builtin_ctz(m2) : -1, t3 = m3 ? __builtin_ctz(m3) : -1;
        bf16x8 aA[KS32], aB[KS32];
        f32x4 wvA, wvB;
        float rnA, rnB;
        load_tile(t0, aA, wvA, rnA);
        if (t1 >= 0) load_tile(t1, aB, wvB, rnB);
        run_tile(t0, aA, wvA, rnA);
        if (t1 >= 0) {
            if (t2 >= 0) load_tile(t2, aA, wvA, rnA);
            run_tile(t1, aB, wvB, rnB);
            if (t2 >= 0) {
                if (t3 >= 0) load_tile(t3, aB, wvB, rnB);
                run_tile(t2, aA, wvA, rnA);
                if (t3 >= 0) run_tile(t3, aB, wvB, rnB);
            }
        }
        // OR over the sixteen rows of every quad; the quad's first lane spreads its four nibbles (one per tile) to their places
        const uint32_t any_row = row16_or(lane_bits);
        if (col == 0 && any_row != 0u) {
            const uint32_t lo = ((any_row & 15u) | (((any_row >> 4) & 15u) << 16)) << (4 * quad);
            const uint32_t hi = (((any_row >> 8) & 15u) | (((any_row >> 12) & 15u) << 16)) << (4 * quad);
            atomicOr(nl + (i - s_begin), ((unsigned long long)hi << 32) | lo);
        }
        st_cur = st_next; tm_cur = tm_next;
    }
    __syncthreads();
    for (int i = s_begin + tid; i < s_end; i += 64 * K16_NW) need[(long)blockIdx.x * n_cstages + i] = nl[i - s_begin];
}

// need bitmaps -> per tile the ascending list of items (group << 4 | mask of the group's 16-unit sub-blocks to run), and its
// length.  need1: bit j of word s <-> group 64 s + j (level 1).  need2 (or null: every sub-block of a kept group runs):
// word 4 s + k holds, for the groups 64 s + 16 k .. + 15, a nibble each (bit 4 (g & 15) + sub <-> sub-block `sub` of group g).
// ... and the same blocks as the tile's dense list of 16-unit tiles (tlist / tcnt: what the screen walks; glist / gcnt: what
// the select kernel walks).  One wave per tile; tile_counts[tile] = (16-unit blocks listed, groups level 1 kept).
__global__ __launch_bounds__(64) void exact_lists_kernel(const unsigned long long* __restrict__ need1, int n_cstages,
                                                         const unsigned long long* __restrict__ need2, int n_groups,
                                                         int* __restrict__ glist, int* __restrict__ gcnt,
                                                         int2* __restrict__ tile_counts, int* __restrict__ tlist,
                                                         int* __restrict__ tcnt) {
    const long tile = blockIdx.x;
    const int lane = threadIdx.x;
    const unsigned long long below = (1ull << lane) - 1ull;
    int base = 0, blk = 0, kept = 0;
    for (int s0 = 0; s0 < n_cstages; s0 += 64) {
        // (the tile's level-1 words in one load, lane <-> stage; then stage by stage out of registers)
        const unsigned long long mine = s0 + lane < n_cstages ? need1[tile * n_cstages + s0 + lane] : 0ull;
        const uint32_t lo = (uint32_t)mine, hi = (uint32_t)(mine >> 32);
        unsigned long long todo = __ballot(mine != 0ull);
        for (; todo != 0ull; todo &= todo - 1ull) {
            const int sl = (int)__builtin_ctzll(todo), s = s0 + sl;
            const unsigned long long w = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)hi, sl) << 32) |
                                         (uint32_t)__builtin_amdgcn_readlane((int)lo, sl);
            uint32_t nib = 15u;
            if (need2 != nullptr) nib = (uint32_t)(need2[tile * 4 * n_cstages + 4 * s + (lane >> 4)] >> (4 * (lane & 15))) & 15u;
            const int g = s * 64 + lane;
            const bool l1 = ((w >> lane) & 1ull) && g < n_groups;
            const bool on = l1 && nib != 0u;
            const unsigned long long mk = __ballot(on);
            if (on) glist[tile * n_groups + base + __popcll(mk & ((1ull << lane) - 1ull))] = (g << 4) | (int)nib;
            base += __popcll(mk);
            kept += __popcll(__ballot(l1));
            // the same blocks as a DENSE list of 16-unit tiles (group << 2 | sub-block), ascending: what the screen walks, four
            // to a barrier, whatever group they belong to
            int before = 0, total = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const unsigned long long mb = __ballot(on && ((nib >> b) & 1u));
                before += __popcll(mb & below);
                total += __popcll(mb);
            }
            if (on) {
                int o = blk + before;
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if ((nib >> b) & 1u) tlist[tile * 4 * n_groups + o++] = (g << 2) | b;
            }
            blk += total;
        }
    }
    // (per tile: 4 096 waves adding into two words of one cache line took as long as the rest of this kernel)
    if (lane == 0) { gcnt[tile] = base; tcnt[tile] = blk; tile_counts[tile] = make_int2(blk, kept); }
}

// sum of the tiles' (16-unit blocks listed, groups level 1 kept) into the pass's counters.  One workgroup.
// items != nullptr: ... and the listed screen's WORK ITEMS (bmu_bf16_k16_kernel<.., GM, TL>): every tile's list cut into parts of
// about L blocks -- L = len_pct % of the mean list (1.25 x), or what gives every slot of the chip two items where the tiles are few; never below
// 32 blocks (a part re-reads its tile's 64 KB of rows) --, item = (tile, part | parts << 16) in tile order; the queue's counter reset.
// At most tiles + max(100 / len_pct tiles, 2.1 slots) items (the three cases of L): with len_pct >= 25, within 5 tiles + 4 slots.
__global__ __launch_bounds__(1024) void exact_list_totals_kernel(const int2* __restrict__ tile_counts, long tiles, int* __restrict__ blocks_run,
                                                                 int* __restrict__ groups_run, int slots = 0, int2* __restrict__ items = nullptr,
                                                                 int* __restrict__ n_items = nullptr, int* __restrict__ item_ctr = nullptr,
                                                                 int len_pct = 125) {
    __shared__ int sb[16], sk[16];
    __shared__ int tot_b;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int b = 0, k = 0;
    for (long t = tid; t < tiles; t += 1024) { const int2 c = tile_counts[t]; b += c.x; k += c.y; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { b += __shfl_xor(b, o, 64); k += __shfl_xor(k, o, 64); }
    if (lane == 0) { sb[wave] = b; sk[wave] = k; }
    __syncthreads();
    if (tid == 0) {
        b = 0; k = 0;
        for (int w = 0; w < 16; ++w) { b += sb[w]; k += sk[w]; }
        *blocks_run = b; *groups_run = k;
        tot_b = b;
    }
    if (items == nullptr) return;
    __syncthreads();
    const long total = tot_b;
    const long by_mean = (len_pct * total) / (100 * (tiles > 0 ? tiles : 1)) + 1, by_slots = total / (2 * (long)(slots > 0 ? slots : 1));
    const int L = (int)max(32L, min(by_mean, by_slots));
    auto parts_of = [&](int cnt) -> int { return cnt <= L ? 1 : (cnt + L - 1) / L; };
    // a contiguous run of tiles per thread; exclusive scan of the runs' item counts over the 1 024 threads
    const long per = (tiles + 1023) / 1024, t_b = min((long)tid * per, tiles), t_e = min(t_b + per, tiles);
    int mine = 0;
    for (long t = t_b; t < t_e; ++t) mine += parts_of(tile_counts[t].x);
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
    __syncthreads();                                         // (sb is read above by thread 0)
    if (lane == 63) sb[wave] = incl;
    __syncthreads();
    int before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { if (w < wave) before += sb[w]; all += sb[w]; }
    int o = before + incl - mine;
    for (long t = t_b; t < t_e; ++t) {
        const int np = parts_of(tile_counts[t].x);
        for (int q = 0; q < np; ++q) items[o++] = make_int2((int)t, q | (np << 16));
    }
    if (tid == 0) { *n_items = all; *item_ctr = 0; }
}

// ---- the SCOUT: a bound for rows WITHOUT a last BMU (query rows, streamed chunks, a row set's first epoch) and for the
// epochs in which last epoch's BMU says little (a schedule's first epochs: rows still travel across the map).
//   1. every row's nearest GROUP centroid g*: the plain resident kernel (bmu_bf16_k16_kernel<.., false, false>) on the plain
//      copy of the level-1 centroid image -- 1/64 of a full scan.
//   2. the rows sorted by g* (the radix sort of update.hpp), their operands gathered in that order: the 256 rows of a tile
//      then share one or two groups.
//   3. per tile the set of its rows' groups (+, where last epoch's BMUs exist, those units' groups) as a tile list
//      (exact_scout_lists_kernel), and the resident kernel once more over those few groups, unit indices kept
//      (bmu_bf16_k16_kernel<.., GM = false, TL = true>): the best unit among them is the row's PSEUDO last BMU.
// Every unit of g* is within r of the centroid, so that unit is no farther than |x - c_g*| + r_g*: the centroid-only bound
// U_c of tools/ucent_probe.py, and usually much nearer.  Any unit gives a valid bound: the plan's prologue (exact_plan_kernel)
// evaluates the unit it is handed rigorously, whatever picked it -- the scout runs in half precision and owes nothing.
__global__ __launch_bounds__(256) void exact_groupkey_kernel(const int* __restrict__ g, long n, int n_groups, int* __restrict__ keys,
                                                             int* __restrict__ vals) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int v = g[i];
    keys[i] = v < 0 ? 0 : v >= n_groups ? n_groups - 1 : v;
    vals[i] = (int)i;
}

// per 256-row tile of the sorted pass: the distinct groups of its rows' keys (gkey[p]: ascending over the pass) and, where
// lastpos is given, of their last BMUs (lastpos[p] >> 6) -- a bitmap of the groups in LDS -- as the tile's dense list of
// 16-unit tiles (all four of every group, ascending: what bmu_bf16_k16_kernel<.., TL> walks).  One wave per tile.
// group_items: the list as (group << 4 | 15) items, one per group (what the wide kernel walks), tcnt = groups listed.
__global__ __launch_bounds__(64) void exact_scout_lists_kernel(const int* __restrict__ gkey, const int* __restrict__ lastpos, long n,
                                                               int n_groups, int* __restrict__ tlist, int* __restrict__ tcnt,
                                                               int group_items = 0) {
    extern __shared__ unsigned long long sbm[];
    const long tile = blockIdx.x;
    const int lane = threadIdx.x;
    const int nw = (n_groups + 63) / 64;
    for (int w = lane; w < nw; w += 64) sbm[w] = 0ull;
    __syncthreads();
    for (int i = 0; i < SK_TILE / 64; ++i) {
        const long p = tile * SK_TILE + i * 64 + lane;
        if (p < n) {
            int g = gkey[p];
            g = g < 0 ? 0 : g >= n_groups ? n_groups - 1 : g;
            atomicOr(sbm + (g >> 6), 1ull << (g & 63));
            if (lastpos != nullptr) {
                int g2 = lastpos[p] >> 6;
                g2 = g2 < 0 ? 0 : g2 >= n_groups ? n_groups - 1 : g2;
                atomicOr(sbm + (g2 >> 6), 1ull << (g2 & 63));
            }
        }
    }
    __syncthreads();
    int base = 0;
    int* out = tlist + tile * (group_items ? 1 : 4) * (long)n_groups;
    for (int w0 = 0; w0 < nw; w0 += 64) {
        unsigned long long word = w0 + lane < nw ? sbm[w0 + lane] : 0ull;
        const int c = __popcll(word);
        int incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        int at = base + incl - c;
        for (; word != 0ull; word &= word - 1ull) {
            const int g = (w0 + lane) * 64 + (int)__builtin_ctzll(word);
            if (group_items) out[at] = (g << 4) | 15;
            else {
#pragma unroll
                for (int b = 0; b < 4; ++b) out[4 * at + b] = (g << 2) | b;
            }
            ++at;
        }
        base += __shfl(incl, 63, 64);
    }
    if (lane == 0) tcnt[tile] = group_items ? base : 4 * base;
}

// Is a plan from the scout worth its launches at all?  First the cheap question: how much of the map does a ROW need?  A sample of
// the rows (one workgroup each, strided through the row set) against every group centroid in float32: with sqrt(U) = min_g (|x -
// c_g| + r_g) -- the centroid-only bound -- how many groups pass level 1's test |x - c_g| - r_g <= sqrt(U)?  counts[0] += groups
// needed, counts[1] += rows sampled.  A tile needs the union over its 256 rows: where a row alone needs nearly every group (a
// random codebook, rows without structure) there is nothing to plan, and the launch is spared the scout altogether.
__global__ __launch_bounds__(256) void exact_scout_rowneed_kernel(const float* __restrict__ X, long N, int D, int n_samples,
                                                                  const float* __restrict__ Cc, const float* __restrict__ rg,
                                                                  int n_groups, int* __restrict__ counts) {
    extern __shared__ float xs[];                            // the row, then four wave minima / counts
    __shared__ float wmin[4];
    __shared__ int wcnt[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long step = N / n_samples;
    const float* x = X + ((long)blockIdx.x * step + step / 2) * D;
    for (int k = tid; k < D; k += 256) xs[k] = x[k];
    __syncthreads();
    auto dist = [&](int g) -> float {
        const float* c = Cc + (long)g * D;
        float q = 0.0f;
        if ((D & 3) == 0) {
            for (int k = 0; k < D; k += 4) {
                const f32x4 v = *(const f32x4*)(c + k);
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float t = xs[k + j] - v[j]; q = __builtin_fmaf(t, t, q); }
            }
        } else {
            for (int k = 0; k < D; ++k) { const float t = xs[k] - c[k]; q = __builtin_fmaf(t, t, q); }
        }
        return __builtin_sqrtf(q);
    };
    float m = __builtin_inff();
    for (int g = tid; g < n_groups; g += 256) {
        const float r = rg[g];
        if (r >= 0.0f) m = __builtin_fminf(m, dist(g) + r);  // (a slot without units: r < 0)
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = __builtin_fminf(m, __shfl_xor(m, o, 64));
    if (lane == 0) wmin[wave] = m;
    __syncthreads();
    m = __builtin_fminf(__builtin_fminf(wmin[0], wmin[1]), __builtin_fminf(wmin[2], wmin[3]));
    int need = 0;
    for (int g = tid; g < n_groups; g += 256) {
        const float r = rg[g];
        if (r >= 0.0f) need += !(dist(g) - r > m) ? 1 : 0;   // (NaN distances: needed)
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) need += __shfl_xor(need, o, 64);
    if (lane == 0) wcnt[wave] = need;
    __syncthreads();
    if (tid == 0) { atomicAdd(counts, wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3]); atomicAdd(counts + 1, 1); }
}

// ... then the exact one, on SAMPLE TILES: every stride-th 256-row tile of the pass in its sorted
// order -- the very tiles the plan would see -- copied out (their rows' ids and sort keys) as a small pass of its own; the host
// gathers, picks and plans that sample exactly as it would the pass and reads the executed share back before committing the
// whole pass to the gather, the pick and the plan (somhip.hip, launch_bmu_exact).
__global__ __launch_bounds__(256) void exact_sample_tiles_kernel(const int* __restrict__ order, const int* __restrict__ keys, long stride_tiles,
                                                                 int* __restrict__ order_out, int* __restrict__ keys_out) {
    const long src = (long)blockIdx.x * stride_tiles * SK_TILE + threadIdx.x;
    const long dst = (long)blockIdx.x * SK_TILE + threadIdx.x;
    static_assert(SK_TILE == 256, "one thread per row of a tile");
    order_out[dst] = order[src];
    keys_out[dst] = keys[src];
}

// the scout's pick per sorted position -> lastpos (a position below K: what the plan's prologue evaluates), and the merge
// keys back to all ones for the screen proper
// (perm: positions -> UNIT ids, where the consumer wants units -- the wide plan's float32 seed)
__global__ __launch_bounds__(256) void exact_scout_pos_kernel(unsigned long long* __restrict__ best64, long n, int K,
                                                              int* __restrict__ lastpos, const int* __restrict__ perm = nullptr) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const uint32_t u = (uint32_t)best64[p];
    const int pos = u < (uint32_t)K ? (int)u : 0;            // (a NaN row: any unit gives a valid bound)
    lastpos[p] = perm != nullptr ? perm[pos] : pos;
    best64[p] = ~0ull;
}

}  // namespace somhip
