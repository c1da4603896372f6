// precision 'exact': float32-exact BMUs at MFMA-half speed.
//
// What it returns is, row for row and bit for bit, what the float32 parity kernel returns (bmu_f32_res.hpp:
// argmin_k fl(|w_k|^2 - 2 c_k), c_k the k-ordered float32 fma chain of x . w_k -- the reference's
// euclidean_squared_distance_part + argmin, distances.py:11-23, xpysom.py:416), near-ties and exact ties included.
// How: a cheap SCREEN that may be wrong by a bounded amount, then the float32 chain itself on the few units the screen
// cannot rule out.
//
//   1. screen    bmu_bf16_k16_kernel<.., F16, GM = true> (bmu_bf16_k16.hpp): one pass of the 16x16x32 MFMA kernel on
//                IEEE-half operands computes d'(n,k) = S (B + |w_k|^2/2 - x_n . w_k) approximately and keeps VALUES
//                only: the row minimum m(n) and, per GROUP of 64 units (one stage) and row, the group's minimum --
//                stored (gmin[group][row]) only where it is within the row's bound E(n) of the minimum so far, with the
//                64-bit mask of the rows stored per stage and wave (gflags).  No unit indices exist in that instance.
//   2. candidates  exact_select_kernel: every group with gmin <= m(n) + E(n) is a candidate of row n.  E(n) bounds
//                (float32 kernel's own rounding) + (screen's error), both relative to tau = |w|^2 - 2 x.w in real
//                arithmetic -- derivation below.  The float32 winner k* has s(k*) <= s(k) for every k, so its screen
//                value cannot exceed the screen's minimum by more than E: its group is a candidate (and was stored:
//                the minimum so far is never below the final one).
//   3. re-score  the (row, group) pairs are bucketed by group as they are found (every group owns a list with room for a
//                whole pass; exact_tiles_kernel cuts the lists into tiles) and exact_rescore_mfma_kernel forms, tile by tile (up to 128 rows of a group's list x its 64 units), the
//                float32 scores by the SAME instruction stream as the parity kernel (v_mfma_f32_32x32x2_f32 on the
//                parity kernel's own stage image, same k order, same epilogue) and keeps the first minimum in unit
//                order per row.  Because k* is among the re-scored units and is the global first minimum, it is the
//                re-scored first minimum.
//   4. fallback  rows the scheme cannot vouch for -- no candidate at all (NaN / infinite rows or norms), a minimum that
//                is not finite, a pass with more pairs than re-scoring is worth (a quarter of the groups per row on average: a
//                degenerate codebook) -- go to the float32 kernel itself (exact_finalize_kernel lists them).
//   order        which 64 units form a group is free.  The images hold the units
//                PATCH by patch (8 x 8 units of the map per group where the sides allow: som_common.hpp) -- the units near a
//                row's best one are a blob of the map, and a blob touches a third to a half as many patches as 64-unit
//                strips of map rows.  Steps 1-3 work on positions; within a group positions ascend with unit ids, and
//                the re-score kernels translate a tile's winner to its UNIT id before the merge, so the first minimum
//                in unit order still wins.  Step 4's float32 kernel wants the units' own order and rebuilds its image.
//   skipping     resident rows, from their second epoch on: exact_skip.hpp plans, per 256-row tile of rows sorted by their
//                last BMU's patch, which groups step 1 runs at all (centroid / radius bound against the distance to last
//                epoch's BMU); a group not run is a group outside every row's window.
//
// Error bound (euclidean, input_len <= 128).  u = 2^-24, A(n,k) = sum_d |x_d w_kd| <= |x_n| max_k|w_k|.
//   float32 kernel:  |c - x.w| <= gamma_D A (fma chain of D terms), s = fl(wsq - 2c):
//                    |s - tau| <= (2 gamma_D + 2u)(1+u) A + u wsq.
//   screen, operands: rows and units are first scaled by powers of two (ex_scale: the longest norm lands in
//                    [2^13, 2^14), exact), then rounded to float16.  The operand error is MEASURED, not assumed: the
//                    preparation kernels know both x^ (scaled float32) and x~ (what the MFMA reads), so |x^_n - x~_n|
//                    per row and max_k |w^_k - w~_k| are numbers (a subnormal half counts with its whole value: covers
//                    an MFMA that flushes it), and  |x~.w~ - x^.w^| = |dx.w^ + x~.dw| <= |dx| |w^| + (|x^| + |dx|) |dw|
//                    by Cauchy-Schwarz -- about 0.4 of the worst-case 2^-11 relative rounding on Gaussian-like rows.
//   screen, accumulation: the initial accumulator fl(S (B + wsq/2)) and ONE chain of ceil(D/32) MFMAs, each charged
//                    KAPPA = 6 ulps of the largest magnitude the accumulator can take, Bm = 2.01 B + max wsq / 2 (the
//                    hardware's internal summation is not documented; tests/test_gpu_exact.py measures it through
//                    som_debug_mfma16 at <= 2.4 ulps and fails above 3).
//   E(n) (in units of d' = S (tau / 2 + B)):  ex_row_bound() in bmu_bf16.hpp, constants in exact_bound() (somhip.hip):
//                    E32_tau + 2 (operand term + accumulation term).
// The bound is deliberately loose (worst-case rounding everywhere it is not measured): widening E only adds candidate
// groups, and a candidate group costs one 64-unit re-score.
#pragma once
#include "bmu_bf16_k16.hpp"
#include "bmu_bf16_wide.hpp"
#include "bmu_f32_res.hpp"
#include "bmu_f32_tiled.hpp"

namespace somhip {

constexpr int EX_GROUP = 64;          // units per group = one stage of the float32 stage image = one stage of the screen
constexpr int EX_PAIRS = 64;          // least capacity of a pass in (row, group) pairs per row ON AVERAGE (exact_reserve sizes it)
constexpr int EX_SCAN_SPLIT = 4;      // waves that share a row's groups in the scan
constexpr int EX_WERR_UNITS = 16;     // units per wave in exact_werr_kernel
constexpr int EX_TR = 128;            // rows per re-score tile (4 waves x 32 rows against one 64-unit group)

// Small per-pass counters, one allocation, zeroed by one memset before the scan:
//   [0, n_groups)             gcount: (row, group) pairs per group = fill cursor of the group's row list
//   [n_groups, 2 n_groups)    gstart: the lists' lengths after round 1 (two-round scheme)
//   [2 n_groups]              fb_count: rows for the float32 fallback kernel
//   [2 n_groups + 1]          n_tiles:  re-score tiles
//   [2 n_groups + 2]          overflow: the pass has more pairs than the lists hold (a degenerate codebook: identical
//                             units everywhere) -- every row goes to the float32 kernel

// (gmin [n_groups][gm_stride], gflags [gm_stride / 64][n_groups]) -> the groups' row lists plist [n_groups][gm_stride] (every
// group owns room for a whole pass: nothing to size, nothing to prefix-sum), gcount [n_groups] and the rows' candidate
// counts.  Block = 64 consecutive rows x EX_SCAN_SPLIT ranges of groups; lane = row.
// The screen stored only the group minima within E of the row minimum SO FAR and left, per group and 64 rows, the mask
// of the rows it stored.  A wave reads the masks of 64 groups with one coalesced load (lane = group), walks the groups
// whose mask is not empty -- eight at a time, so that their value loads are in flight together --, loads only the stored
// rows' values and keeps the ones within E of the FINAL row minimum.  The 64 lanes test the SAME group at the same time:
// a group's hits take consecutive list positions from ONE returning atomic per wave and 64 groups (lane j adds group
// j's population and hands out the base).  The order of a list depends on the order of the atomics; the re-score's
// result does not (every (row, group) pair is scored on its own and merged by atomicMin).
// The per-row merge keys of the re-score start from all ones: reset here, once every wave has read the screen's minimum.
// ROUND2 (the two-round scheme, exact_first_kernel below): best64 already holds the float32 best t*(n) of the round-1
// re-score (the group with the screen's minimum).  A unit k can only beat or tie t* if its screen value is within the
// ONE-unit error E/2 of d'(t*) -- half the two-unit window m + E of the one-round scheme, from a reference point that is
// the true minimum's rather than the screen's: 20-32 % fewer candidate pairs on smooth maps (measured).  rowarg[n] = the group
// round 1 scored (skipped here); the lists and gcount continue behind round 1's entries; best64 is left alone.
template <bool ROUND2>
__global__ __launch_bounds__(64 * EX_SCAN_SPLIT) void exact_select_kernel(const uint32_t* __restrict__ gmin,
                                                                         const unsigned long long* __restrict__ gflags,
                                                                         long gm_stride, int n_groups, long N,
                                                                         unsigned long long* __restrict__ best64,
                                                                         const float* __restrict__ xsq,
                                                                         const float* __restrict__ wmax2,
                                                                         const float* __restrict__ xmax2, ExactBound eb,
                                                                         const float* __restrict__ xerr,
                                                                         const float* __restrict__ werr2,
                                                                         int* __restrict__ plist, int* __restrict__ gcount,
                                                                         int* __restrict__ rowcnt,
                                                                         const int* __restrict__ rowarg = nullptr,
                                                                         const float* __restrict__ seed = nullptr,
                                                                         const int* __restrict__ glist = nullptr,
                                                                         const int* __restrict__ gcnt = nullptr,
                                                                         int rows_per_list = 0) {
    __shared__ int cnt_s[EX_SCAN_SPLIT][64];
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const long row0 = (long)blockIdx.x * 64;
    const long row = row0 + lane;
    const bool live = row < N;
    const long r = live ? row : 0;
    const ExactScales sc = ex_scales(xmax2, wmax2, werr2);
    const float e = ex_row_bound(eb, sc, xsq[r], xerr[r]);
    float thr_f;
    int arg = -1;
    if (ROUND2) {
        const unsigned long long k64 = best64[r];
        const uint32_t key = (uint32_t)(k64 >> 32);
        const uint32_t bits = (key & 0x80000000u) ? (key & 0x7FFFFFFFu) : ~key;   // the merge key's float (exact_finalize_kernel)
        const float t = __uint_as_float(bits);
        const float S = sc.sx * sc.sw;
        // d'(t): what the screen would hold for a unit whose float32 score is t -- S (B + t / 2), cosine: S (B - 1 + t) --,
        // + the one-unit bound E / 2 (+ margins for evaluating both in float32: 2^-10 of E, 4 ulps of the largest d')
        const float dp = eb.unit ? __builtin_fmaf(S, t - 1.0f, S * sc.big) : __builtin_fmaf(0.5f * S, t, S * sc.big);
        thr_f = dp + 0.5f * e * (1.0f + 1.0f / 1024.0f) + S * sc.bmag * 0x1p-21f;
        if (k64 == ~0ull || (bits & 0x7F800000u) == 0x7F800000u) thr_f = __builtin_nanf("");   // nothing scored, or not finite
        arg = rowarg[r];
    } else {
        thr_f = __uint_as_float((uint32_t)(best64[r] >> 32)) + e;
        // seed (exact_seed_kernel / the plan's prologue): an upper bound, less the float32 share, on the screen value of
        // whatever beats one given unit in the float32 kernel
        if (seed != nullptr) thr_f = __builtin_fminf(thr_f, seed[r] + ex_f32_share(eb, sc, xsq[r]));
    }
    // a threshold that is not a finite positive number (a row the bound does not cover, a NaN minimum) selects nothing
    const bool ok = live && thr_f > 0.0f && thr_f < 3.0e38f;
    // unsigned compare on the bit patterns: every d' is a positive float, a NaN pattern is above every threshold
    const uint32_t thr = ok ? __float_as_uint(thr_f) : 0u;
    if (!ROUND2) {
        __syncthreads();
        if (part == 0 && live) best64[row] = ~0ull;
    }
    // the groups this wave walks: a quarter of all groups -- or (block skipping: the screen ran a LIST of groups per
    // rows_per_list-row tile and wrote masks for those only) of the tile's list
    const int* my_list = glist != nullptr ? glist + (row0 / rows_per_list) * (long)n_groups : nullptr;
    const int n_walk = my_list != nullptr ? gcnt[row0 / rows_per_list] : n_groups;
    const int g_begin = (int)((long)n_walk * part / EX_SCAN_SPLIT), g_end = (int)((long)n_walk * (part + 1) / EX_SCAN_SPLIT);
    const uint32_t* src = gmin + r;
    const unsigned long long below = (1ull << lane) - 1;
    int mine = 0;
    for (int gb = g_begin; gb < g_end; gb += 64) {
        // lane j <-> the j-th group of this chunk
        const int gid = gb + lane < g_end ? (my_list != nullptr ? (my_list[gb + lane] >> 4) : gb + lane) : -1;
        const unsigned long long fw = gid >= 0 ? gflags[ex_flag_index(blockIdx.x, gid, n_groups, gm_stride)] : 0ull;
        const unsigned long long any = __ballot(fw != 0ull);
        if (any == 0) continue;
        const uint32_t fw_lo = (uint32_t)fw, fw_hi = (uint32_t)(fw >> 32);
        unsigned long long hits = 0;                       // bit j: the chunk's j-th group is a candidate of this lane's row
        unsigned long long todo = any;
        while (todo != 0) {
            int j[8];
            uint32_t v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                j[q] = todo != 0 ? (int)__builtin_ctzll(todo) : -1;
                if (todo != 0) todo &= todo - 1;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                v[q] = 0xFFFFFFFFu;
                if (j[q] >= 0) {                           // (wave-uniform)
                    const unsigned long long stored = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)fw_hi, j[q]) << 32) |
                                                      (uint32_t)__builtin_amdgcn_readlane((int)fw_lo, j[q]);
                    const int g = __builtin_amdgcn_readlane(gid, j[q]);
                    if ((stored >> lane) & 1ull) v[q] = src[(long)g * gm_stride];   // (only the rows the screen stored)
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (j[q] >= 0) {
                    // (ROUND2: the group round 1 scored for this row is not selected again -- by its id: under a tile list
                    //  the chunk's j-th group is not group gb + j)
                    const int g = __builtin_amdgcn_readlane(gid, j[q]);
                    hits |= (unsigned long long)(v[q] <= thr && !(ROUND2 && g == arg)) << j[q];
                }
        }
        if (!ok) hits = 0;
        if (__ballot(hits != 0ull) == 0) continue;
        int c = 0;                                         // lane j: the wave's hits in the chunk's j-th group
        for (todo = any; todo != 0; todo &= todo - 1) {
            const int jj = (int)__builtin_ctzll(todo);
            const unsigned long long mk = __ballot((hits >> jj) & 1ull);
            if (lane == jj) c = (int)__builtin_popcountll(mk);
        }
        int base = 0;
        if (c > 0) base = atomicAdd(gcount + gid, c);
        for (todo = any; todo != 0; todo &= todo - 1) {
            const int jj = (int)__builtin_ctzll(todo);
            const unsigned long long mk = __ballot((hits >> jj) & 1ull);
            if (mk == 0) continue;
            const int o = __builtin_amdgcn_readlane(base, jj);
            const int g = __builtin_amdgcn_readlane(gid, jj);
            if ((hits >> jj) & 1ull) plist[(long)g * gm_stride + o + (int)__builtin_popcountll(mk & below)] = (int)row;
        }
        mine += (int)__builtin_popcountll(hits);
    }
    cnt_s[part][lane] = mine;
    __syncthreads();
    if (part == 0 && live) {
        int t = 0;
#pragma unroll
        for (int p = 0; p < EX_SCAN_SPLIT; ++p) t += cnt_s[p][lane];
        rowcnt[row] = t + (ROUND2 && arg >= 0 ? 1 : 0);
    }
}

// seed[n] = an upper bound on the screen value of unit u = prev[n] (any unit: last epoch's BMU is the useful choice) for row
// n, from the float32 operands: t = |w_u|^2 - 2 x.w_u in float32 (a 16-lane fma tree: its error is within the float32
// kernel's share of the bound), d'(t) + E/2 as in the second re-score round.  Whatever beats unit u in the float32 kernel
// has a screen value of at most seed + the float32 share of E: the screen stores no group minimum above that and the
// select kernel selects none -- on a trained map the seed IS about the row minimum, and the groups passed on the way down
// to it (most of what a smooth map makes the screen store and the select kernel read) are never written, and the
// two-unit window m + E shrinks towards the one-unit E/2 around a float32 reference.
// Euclidean, input_len <= 128; 16 lanes per row.
__global__ __launch_bounds__(256) void exact_seed_kernel(const float* __restrict__ X, long N, int D,
                                                         const float* __restrict__ W, const float* __restrict__ wsq, int K,
                                                         const int* __restrict__ prev, const float* __restrict__ xsq,
                                                         const float* __restrict__ xerr, const float* __restrict__ wmax2,
                                                         const float* __restrict__ xmax2, const float* __restrict__ werr2,
                                                         ExactBound eb, float* __restrict__ seed,
                                                         float* __restrict__ tq = nullptr) {
    const int sub = threadIdx.x & 15;
    const long row = ((long)blockIdx.x * 256 + threadIdx.x) >> 4;
    const bool live = row < N;
    const long r = live ? row : 0;
    int u = prev[r];
    u = u < 0 ? 0 : u >= K ? K - 1 : u;                      // (any unit gives a valid bound)
    float c = 0.0f;
    for (int k = sub; k < D; k += 16) c = __builtin_fmaf(X[r * D + k], W[(long)u * D + k], c);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if (sub != 0 || !live) return;
    const ExactScales sc = ex_scales(xmax2, wmax2, werr2);
    const float e = ex_row_bound(eb, sc, xsq[r], xerr[r]);
    const float t = __builtin_fmaf(-2.0f, c, wsq[u]);
    const float S = sc.sx * sc.sw;
    const float dp = __builtin_fmaf(0.5f * S, t, S * sc.big);
    const float sd = dp + 0.5f * e * (1.0f + 1.0f / 1024.0f) + S * sc.bmag * 0x1p-21f;
    seed[r] = (sd > 0.0f && sd < 3.0e38f) ? sd : __builtin_inff();
    if (tq != nullptr) tq[r] = t;                            // (exact_skip.hpp: the bound on the distance to this epoch's BMU)
}

// Round 1 of the two-round scheme: every row goes to the list of the group that holds its screen minimum (the screen
// leaves that group in the low half of the row's merge key).  rowarg[n] = that group, or -1 for a row the bound does not
// cover (it selects nothing in round 2 either and ends in the fallback list).  One returning atomic per wave and distinct
// group (rows of a wave mostly share a few groups on smooth maps).  The merge keys start from all ones.
__global__ __launch_bounds__(256) void exact_first_kernel(unsigned long long* __restrict__ best64, long N, int n_groups,
                                                          long gm_stride, const float* __restrict__ xsq,
                                                          const float* __restrict__ wmax2, const float* __restrict__ xmax2,
                                                          ExactBound eb, const float* __restrict__ xerr,
                                                          const float* __restrict__ werr2, int* __restrict__ plist,
                                                          int* __restrict__ gcount, int* __restrict__ rowarg) {
    const long row = (long)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool live = row < N;
    const long r = live ? row : 0;
    const unsigned long long k64 = best64[r];
    const float e = ex_row_bound(eb, ex_scales(xmax2, wmax2, werr2), xsq[r], xerr[r]);
    const float thr_f = __uint_as_float((uint32_t)(k64 >> 32)) + e;
    const uint32_t g = (uint32_t)k64;
    const bool ok = live && thr_f > 0.0f && thr_f < 3.0e38f && g < (uint32_t)n_groups;
    if (live) { best64[row] = ~0ull; rowarg[row] = ok ? (int)g : -1; }
    const int myg = ok ? (int)g : -1;
    const unsigned long long below = (1ull << lane) - 1;
    unsigned long long todo = __ballot(ok);
    while (todo != 0) {
        const int leader = (int)__builtin_ctzll(todo);
        const int g0 = __builtin_amdgcn_readlane(myg, leader);
        const unsigned long long mk = __ballot(myg == g0);
        int base = 0;
        if (lane == leader) base = atomicAdd(gcount + g0, (int)__builtin_popcountll(mk));
        base = __builtin_amdgcn_readlane(base, leader);
        if (myg == g0) plist[(long)g0 * gm_stride + base + (int)__builtin_popcountll(mk & below)] = (int)row;
        todo &= ~mk;
    }
}

// gcount -> the tile table (group, first list entry of the tile in plist, its rows) and n_tiles; more pairs than
// `capacity` (a degenerate codebook: the float32 kernel over all units costs less than re-scoring that many pairs) ->
// overflow flag, no tiles.  One block.
// gstart (nullptr: zeros): the lists' entries before it were tiled by an earlier round; gstart_out (nullptr: none) takes
// the lists' lengths as the next round's gstart.
__global__ __launch_bounds__(1024) void exact_tiles_kernel(const int* __restrict__ gcount, int n_groups, long gm_stride,
                                                           long capacity, int4* __restrict__ tile_tab,
                                                           int* __restrict__ n_tiles_out, int* __restrict__ overflow,
                                                           const int* __restrict__ gstart = nullptr,
                                                           int* __restrict__ gstart_out = nullptr,
                                                           int* __restrict__ pairs_out = nullptr) {
    __shared__ long wsum[16];
    __shared__ int wtsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (n_groups + 1023) / 1024;
    const int b = tid * per, e = min(b + per, n_groups);
    long s = 0;
    int ts = 0;
    for (int g = b; g < e; ++g) {
        const int c = gcount[g] - (gstart ? gstart[g] : 0);
        s += c; ts += (c + EX_TR - 1) / EX_TR;
    }
    // inclusive scan of both over the 1 024 threads: inside each wave by shuffles, then over the sixteen wave totals
    long si = s;
    int ti = ts;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const long a = (long)__shfl_up((long long)si, o, 64);
        const int c = __shfl_up(ti, o, 64);
        if (lane >= o) { si += a; ti += c; }
    }
    if (lane == 63) { wsum[wave] = si; wtsum[wave] = ti; }
    __syncthreads();
    long total = 0;
    int ttotal = 0, tbefore = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        if (w < wave) tbefore += wtsum[w];
        total += wsum[w]; ttotal += wtsum[w];
    }
    if (tid == 0 && pairs_out != nullptr) *pairs_out = (int)min(total, 0x7fffffffL);
    if (total > capacity) {
        if (tid == 0) { *overflow = 1; *n_tiles_out = 0; }
        return;
    }
    // (a group's tiles are written by its whole wave, sixty-four at a time: a popular group has a hundred and more, and one
    //  thread writing them one by one set this launch's time)
    int toff = tbefore + ti - ts;
    for (int k = 0; k < per; ++k) {
        const int g = b + k;
        const bool on = g < e;
        const int first = (on && gstart) ? gstart[g] : 0;
        const int c = on ? gcount[g] - first : 0;
        const int off = on ? (int)((long)g * gm_stride) + first : 0;
        const int mine = toff;
        toff += (c + EX_TR - 1) / EX_TR;
        if (on && gstart_out) gstart_out[g] = first + c;
        for (int src = 0; src < 64; ++src) {
            const int c_s = __builtin_amdgcn_readlane(c, src);
            if (c_s <= 0) continue;                           // (wave-uniform)
            const int g_s = __builtin_amdgcn_readlane(g, src), off_s = __builtin_amdgcn_readlane(off, src);
            const int t_s = __builtin_amdgcn_readlane(mine, src);
            for (int i = lane; i * EX_TR < c_s; i += 64)
                tile_tab[t_s + i] = make_int4(g_s, off_s + i * EX_TR, min(EX_TR, c_s - i * EX_TR), 0);
        }
    }
    if (tid == 1023) *n_tiles_out = ttotal;
}

// ---- REFINEMENT (resident sorted rows, input_len <= 128): the candidate (row, group) pairs once more on the half-precision
// pipe, with BOTH operands' second halves -- x^ = hi + 2^-11 lo, -w^ = hi + 2^-11 lo (exact_gather_sorted_kernel,
// prep_w_exact_k16_kernel) --:   V = [wq + (-wh).xh] + 2^-11 [(-wh).xl + (-wl).xh],   three MFMAs where the float32 re-score
// spends sixteen times that.  What V misses of the real score is the product of the two FIRST-half rounding errors
// (|dw| |dx|, both measured), the second halves' own rounding (2^-11 of them) and the accumulation of the first chain (the
// screen's own charge): a window E2 some twenty times narrower than the screen's E (ex_refine_bound).  exact_refine_kernel leaves
// every pair's group minimum of V (rmin, indexed like plist: the screen's spent minima) and the row's minimum over its
// pairs (rowmin2); exact_select2_kernel keeps, in place, the pairs within E2 of that minimum.  On the smooth maps of a
// schedule's middle a row has five candidate groups by the screen's window and one or two by this one: the float32
// re-score, the exact mode's largest item there, shrinks accordingly.  The float32 winner survives: its V is within
// E2 / 2 + (float32 share) / 2 of its real score, and so is the V that defines rowmin2.
// Two-unit window of the refined values, in d' units (scales: the codebook's).
__device__ __forceinline__ float ex_refine_bound(const ExactBound& eb, const ExactScales& s, float xsq, float xerr) {
    const float xn = __builtin_sqrtf(xsq) * (1.0f + 1.0f / 1024.0f);
    const float S = s.sx * s.sw;
    // float32 kernel's share + accumulation of the first chain (as in E) + 2 x [ |dw| |dx| + 2^-10 (|w^| |dx| + |x^| |dw|)
    // (the second halves' own rounding, subnormal second halves included: < 2^-25 an element) ] + the final additions
    const float e = S * (eb.cA * xn * s.wm + eb.cW * s.wm * s.wm + eb.cB * s.bmag) +
                    eb.cM * (xerr * s.we + 0x1p-10f * (xerr * s.sw * s.wm + (s.sx * xn + xerr) * s.we)) + 0x1p-21f * S * s.bmag +
                    eb.cM * 0x1p-20f * (s.sw * s.wm + s.sx * xn);
    const bool ok = e > 0.0f && e < 3.0e38f && !eb.unit;
    return ok ? e : __builtin_inff();                        // (no bound: keep every pair)
}

// One tile (up to EX_TR rows of a group's list x the group's 64 units) per step, persistent workgroups over the tile table
// as exact_rescore_mfma_kernel; the group's two stages (first and second halves) stay in LDS across its tiles.  Wave w takes
// the tile's rows 32 w .. 32 w + 31 (two 16-row MFMA blocks): lane (quad, col) loads the row's 8-feature pieces of both
// images, as the screen's B operand.
template <int KS32, class EL>
__global__ __launch_bounds__(256, 3) void exact_refine_kernel(const __bf16* __restrict__ Xh, const __bf16* __restrict__ Xl,
                                                              const char* __restrict__ Wst, const char* __restrict__ Wst_lo,
                                                              const int4* __restrict__ tile_tab, const int* __restrict__ n_tiles_dev,
                                                              const int* __restrict__ plist, uint32_t* __restrict__ rmin,
                                                              uint32_t* __restrict__ rowmin2) {
    using E = typename EL::T;
    using bf16x8 = typename V8<E>::t;
    constexpr int DP = 32 * KS32;
    constexpr int STAGE = k16_stage_bytes(KS32);
    constexpr int FR = K16_T * KS32;                         // fragment pieces (1 KB) of a stage
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [first-half stage incl. its tail][second-half fragments]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int quad = lane >> 4, col = lane & 15;
    const int n_tiles = *n_tiles_dev;
    const int per = (n_tiles + (int)gridDim.x - 1) / (int)gridDim.x;
    // (workgroups go to the XCDs round robin: consecutive runs of tiles -- neighbouring groups, whose candidate rows are
    //  largely the same -- to workgroups of ONE XCD, so that a row read for one group is in that L2 for the next)
    const int nx = (int)gridDim.x / 8;
    const int run = ((int)gridDim.x % 8 == 0) ? ((int)blockIdx.x % 8) * nx + (int)blockIdx.x / 8 : (int)blockIdx.x;
    const int t_begin = run * per, t_end = min(t_begin + per, n_tiles);
    int g_have = -1;
    for (int t = t_begin; t < t_end; ++t) {
        const int4 tt = tile_tab[t];
        const int g = tt.x;
        if (g != g_have) {                                   // (uniform over the workgroup)
            __builtin_amdgcn_s_barrier();                    // everyone is done with the previous group's stages
            for (int p = wave; p < FR + 1; p += 4) lds_dma_16(Wst + (long)g * STAGE + (long)p * 1024 + lane * 16, smem + p * 1024);
            for (int p = wave; p < FR; p += 4) lds_dma_16(Wst_lo + (long)g * STAGE + (long)p * 1024 + lane * 16, smem + STAGE + p * 1024);
        }
        int row[2];
        bf16x8 xh[2][KS32], xl[2][KS32];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const int i = wave * 32 + rb * 16 + col;
            row[rb] = i < tt.z ? plist[tt.y + i] : -1;
#pragma unroll
            for (int ks = 0; ks < KS32; ++ks) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { xh[rb][ks][j] = (E)0.0f; xl[rb][ks][j] = (E)0.0f; }
                if (row[rb] >= 0) {
                    xh[rb][ks] = *(const bf16x8*)(Xh + (long)row[rb] * DP + ks * 32 + quad * 8);
                    xl[rb][ks] = *(const bf16x8*)(Xl + (long)row[rb] * DP + ks * 32 + quad * 8);
                }
            }
        }
        if (g != g_have) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            g_have = g;
        }
        if (__ballot(row[0] >= 0 || row[1] >= 0) == 0ull) continue;   // (wave-uniform: this wave's 32 list entries are empty)
        const float* wq = (const float*)(smem + FR * 1024);
        uint32_t best[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};       // unsigned: positive floats order as their bits, NaN patterns are above
#pragma unroll
        for (int ut = 0; ut < K16_T; ++ut) {
            const f32x4 wv = *(const f32x4*)(wq + ut * 16 + 4 * quad);
            f32x4 ah[2] = {wv, wv}, al[2];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) al[rb][r] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < KS32; ++ks) {
                const bf16x8 wh = *(const bf16x8*)(smem + ((ut * KS32 + ks) * 64 + lane) * 16);
                const bf16x8 wl = *(const bf16x8*)(smem + STAGE + ((ut * KS32 + ks) * 64 + lane) * 16);
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    ah[rb] = mfma16(wh, xh[rb][ks], ah[rb]);
                    al[rb] = mfma16(wh, xl[rb][ks], al[rb]);
                    al[rb] = mfma16(wl, xh[rb][ks], al[rb]);
                }
            }
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) best[rb] = min(best[rb], __float_as_uint(__builtin_fmaf(al[rb][r], 0x1p-11f, ah[rb][r])));
        }
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            uint32_t m = best[rb];
            m = min(m, (uint32_t)__shfl_xor((int)m, 16, 64));
            m = min(m, (uint32_t)__shfl_xor((int)m, 32, 64));
            if (quad == 0 && row[rb] >= 0) {
                rmin[tt.y + wave * 32 + rb * 16 + col] = m;
                atomicMin(rowmin2 + row[rb], m);
            }
        }
    }
}

// rowmin2[row] (the row's refined minimum) -> the bits of its threshold rowmin2 + E2, in place; all ones where the threshold is not
// a finite positive number (no bound, a NaN minimum: every pair is kept).  One pass over the rows, so that exact_select2_kernel
// gathers one word per list entry instead of three (the minimum and the two norms the bound is made of).
__global__ __launch_bounds__(256) void exact_thr2_kernel(uint32_t* __restrict__ rowmin2, long n, const float* __restrict__ xsq,
                                                         const float* __restrict__ xerr, const float* __restrict__ wmax2,
                                                         const float* __restrict__ xmax2, const float* __restrict__ werr2, ExactBound eb) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const ExactScales sc = ex_scales(xmax2, wmax2, werr2);
    const float thr = __uint_as_float(rowmin2[i]) + ex_refine_bound(eb, sc, xsq[i], xerr[i]);
    rowmin2[i] = (thr > 0.0f && thr < 3.0e38f) ? __float_as_uint(thr) : 0xFFFFFFFFu;
}

// One workgroup per group: its list compacted IN PLACE to the pairs whose refined minimum is within E2 of the row's
// (chunks of 256 entries in order: a chunk is read whole before the survivors are written, never beyond where it was read).
__global__ __launch_bounds__(256) void exact_select2_kernel(int* __restrict__ plist, const uint32_t* __restrict__ rmin, long gm_stride,
                                                            int* __restrict__ gcount, const uint32_t* __restrict__ thr2,
                                                            int* __restrict__ kept_total) {
    __shared__ int wsum[4];
    __shared__ int out_s;
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cnt = gcount[g];
    if (cnt <= 0) return;
    int* list = plist + (long)g * gm_stride;
    const uint32_t* rm = rmin + (long)g * gm_stride;
    if (tid == 0) out_s = 0;
    __syncthreads();
    for (int c0 = 0; c0 < cnt; c0 += 256) {
        const int i = c0 + tid;
        int row = -1;
        bool keep = false;
        if (i < cnt) {
            row = list[i];
            keep = rm[i] <= thr2[row];                       // (exact_thr2_kernel: all ones where nothing bounds the row)
        }
        const unsigned long long mk = __ballot(keep);
        if (lane == 0) wsum[wave] = __popcll(mk);
        __syncthreads();                                     // (every entry of the chunk is in registers)
        int before = out_s;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        if (keep) list[before + __popcll(mk & ((1ull << lane) - 1ull))] = row;
        __syncthreads();
        if (tid == 0) out_s += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    if (tid == 0) { gcount[g] = out_s; atomicAdd(kept_total, out_s); }
}

// The float32 scores of one tile -- up to EX_TR rows of a group's list against the group's 64 units -- on
// v_mfma_f32_32x32x2_f32, exactly as bmu_f32_res_kernel<SCORE_EUCLID_PART> forms them (same stage image, same k order,
// same epilogue, same first-minimum rule); a row's best (value, unit) over its groups merges through the same
// order-preserving 64-bit atomicMin the parity kernel's codebook parts use.  Persistent: a workgroup takes a contiguous
// run of the tile table -- consecutive tiles mostly belong to one group, whose stage is fetched once and stays in LDS
// -- and between two stage changes its four waves run free of each other (no barrier per tile).
template <int KG>
__global__ __launch_bounds__(256, 3) void exact_rescore_mfma_kernel(const float* __restrict__ X, int D,
                                                                    const char* __restrict__ Wfst, int K,
                                                                    const int4* __restrict__ tile_tab,
                                                                    const int* __restrict__ n_tiles_dev,
                                                                    const int* __restrict__ plist,
                                                                    unsigned long long* __restrict__ best64,
        const int* __restrict__ perm, const int* __restrict__ order = nullptr, int sub44 = 0, int deint = 0) {
    constexpr int STAGE = fr_stage_bytes(KG);
    constexpr int PIECES = FR_UT * KG + 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // ONE stage: occupancy (three workgroups per CU), not a
                                                                  // ring, hides a tile's load latency under other tiles' MFMAs
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, col = lane & 31;
    const int n_tiles = *n_tiles_dev;
    int g_have = -1;                                       // group whose stage the LDS holds
    // a workgroup takes a contiguous run of tiles: consecutive tiles mostly belong to ONE group, whose stage stays put
    const int per = (n_tiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int t_begin = blockIdx.x * per, t_end = min(t_begin + per, n_tiles);
    // this lane's list entry (a row id, or -1) of tile t; read one tile ahead of its use
    auto entry = [&](int t) -> int {
        if (t >= t_end) return -1;
        const int4 tt = tile_tab[t];
        const int i = wave * 32 + col;
        return i < tt.z ? plist[tt.y + i] : -1;
    };
    int row_next = entry(t_begin);
    for (int t = t_begin; t < t_end; ++t) {
        const int g = tile_tab[t].x;
        const int row = row_next;
        const bool wave_live = __ballot(row >= 0) != 0;    // (wave-uniform)
        if (g != g_have) {                                 // (uniform over the workgroup)
            __builtin_amdgcn_s_barrier();                  // everyone is done with the previous group's stage
            for (int p = wave; p < PIECES; p += 4) lds_dma_16(Wfst + (long)g * STAGE + (long)p * 1024 + lane * 16, smem + p * 1024);
        }
        // (block skipping, exact_skip.hpp: the pass runs in sorted order, list entries are sorted positions: the row itself
        //  sits at order[entry]; the merge key stays at the entry)
        const int xrow = (order != nullptr && row >= 0) ? order[row] : row;
        float xf[4 * KG];
        if (deint) {
            // (the sorted copy of exact_gather_sorted_kernel: a row's even features, then its odd ones: this lane's parity in
            //  whole 16-byte pieces -- half the loads of the interleaved row, nothing read to be dropped)
#pragma unroll
            for (int c = 0; c < KG; ++c) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (row >= 0 && 8 * c < D) v = *(const f32x4*)(X + (long)xrow * D + half * (D >> 1) + 4 * c);
#pragma unroll
                for (int j = 0; j < 4; ++j) xf[4 * c + j] = v[j];
            }
        } else if ((D & 3) == 0) {
#pragma unroll
            for (int c = 0; c < 2 * KG; ++c) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (row >= 0 && 4 * c < D) v = *(const f32x4*)(X + (long)xrow * D + 4 * c);
                xf[2 * c] = half ? v[1] : v[0];
                xf[2 * c + 1] = half ? v[3] : v[2];
            }
        } else {
#pragma unroll
            for (int s = 0; s < 4 * KG; ++s) {
                const int k = 2 * s + half;
                xf[s] = (row >= 0 && k < D) ? X[(long)xrow * D + k] : 0.0f;
            }
        }
        row_next = entry(t + 1);                           // (in flight under this tile's MFMAs)
        if (g != g_have) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the stage
            __builtin_amdgcn_s_barrier();                      // ... and everybody else's
            asm volatile("" ::: "memory");
            g_have = g;
        }
        if (!wave_live) continue;
        const char* st = smem;
        const float* wq = (const float*)(st + FR_UT * KG * 1024);
        float best = __builtin_inff();
        int bkey = 0, brank = 0;
#pragma unroll
        for (int ut = 0; ut < FR_UT; ++ut) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
            for (int kg = 0; kg < KG; ++kg) {
                const f32x4 a4 = *(const f32x4*)(st + ((ut * KG + kg) * 64 + lane) * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j], xf[4 * kg + j], acc, 0, 0, 0);
            }
            f32x4 wv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) wv[q] = *(const f32x4*)(wq + ut * 32 + 8 * q + 4 * half);
            const int tile = g * FR_UT + ut;
            static_assert(FR_UT == 2, "ex_rank44: a group is two 32-unit MFMA tiles");
            if (sub44) f32_tile_argmin_ranked<SCORE_EUCLID_PART>(acc, wv, 0.0f, tile, half, best, bkey, brank);   // (whole 8 x 8 patches: no tail)
            else if ((tile + 1) * 32 > K) f32_tile_argmin<SCORE_EUCLID_PART, true>(acc, wv, 0.0f, tile, half, K, best, bkey);
            else f32_tile_argmin<SCORE_EUCLID_PART, false>(acc, wv, 0.0f, tile, half, K, best, bkey);
        }
        int bidx = f32_key_unit(bkey, half);
        if (!sub44) brank = bidx;                         // (positions ascend with the unit ids inside the group)
        const float ob = __shfl_xor(best, 32, 64);
        const int oi = __shfl_xor(bidx, 32, 64), ork = __shfl_xor(brank, 32, 64);
        if (ob < best || (ob == best && ork < brank)) { best = ob; bidx = oi; }
        if (half == 0 && row >= 0) {                      // (best == +inf: no unit of this group scored below it)
            const uint32_t bits = __float_as_uint(best);
            const uint32_t key = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
            // (patch order: inside a group the lowest unit id among equal scores was kept; across groups the merge compares UNITS)
            atomicMin(best64 + row, ((unsigned long long)key << 32) | (uint32_t)(perm != nullptr ? perm[bidx] : bidx));
        }
    }
}

// The same re-score beyond 128 features, on the float32 TILE image (bmu_f32_tiled.hpp: [128-unit block][32-feature chunk]
// of 16 KB fragments + the block's |w|^2): a tile is up to 128 rows of a group's list against the group's 64 units (one
// half of a unit block), the features go by in chunks of 32 -- the group's 8 KB of fragments by LDS-DMA into a two-slot
// ring, the rows' 128 bytes gathered into registers a chunk ahead -- and the accumulators live across the chunks: the
// k-ordered chain of bmu_f32_tiled_kernel, its epilogues (score_f32<MODE>: euclidean or cosine), its first-minimum rule.
template <int MODE>
__global__ __launch_bounds__(256, 3) void exact_rescore_tiled_kernel(const float* __restrict__ X, int D,
                                                                     const float* __restrict__ xsq,
                                                                     const char* __restrict__ Wfimg, int n_kchunks, int K,
                                                                     const int4* __restrict__ tile_tab,
                                                                     const int* __restrict__ n_tiles_dev,
                                                                     const int* __restrict__ plist,
                                                                     unsigned long long* __restrict__ best64,
        const int* __restrict__ perm, const int* __restrict__ order = nullptr, int sub44 = 0) {
    __shared__ __attribute__((aligned(16))) char ring[2][8192];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, col = lane & 31;
    const int n_tiles = *n_tiles_dev;
    const int per = (n_tiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int t_begin = blockIdx.x * per, t_end = min(t_begin + per, n_tiles);
    const bool vec = (D & 3) == 0;
    auto entry = [&](int t) -> int {
        if (t >= t_end) return -1;
        const int4 tt = tile_tab[t];
        const int i = wave * 32 + col;
        return i < tt.z ? plist[tt.y + i] : -1;
    };
    // this lane's 16 operands of feature chunk kc: features 32 kc + 8 g + 2 j + half, g, j = 0..3
    auto gather = [&](int row, int kc, float (&b)[16]) {
        const int k0 = 32 * kc;
        if (vec) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (row >= 0 && k0 + 4 * c < D) v = *(const f32x4*)(X + (long)row * D + k0 + 4 * c);
                b[2 * c] = half ? v[1] : v[0];
                b[2 * c + 1] = half ? v[3] : v[2];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int k = k0 + 2 * i + half;
                b[i] = (row >= 0 && k < D) ? X[(long)row * D + k] : 0.0f;
            }
        }
    };
    int row_next = entry(t_begin);
    for (int t = t_begin; t < t_end; ++t) {
        const int g = tile_tab[t].x;
        const int ub = g >> 1, uh = g & 1;                 // 128-unit block of the tile image, which half of it
        const int row = row_next;
        const bool wave_live = __ballot(row >= 0) != 0;    // (wave-uniform)
        const float xs = (MODE != SCORE_EUCLID_PART && row >= 0) ? xsq[row] : 0.0f;
        const char* wtile = Wfimg + (long)ub * n_kchunks * FT_WTILE;
        auto stage_dma = [&](int kc, int slot) {           // the half block's 8 KB of fragments of chunk kc
            const char* src = wtile + (long)kc * FT_WTILE + uh * 8192;
            for (int p = wave; p < 8; p += 4) lds_dma_16(src + p * 1024 + lane * 16, ring[slot] + p * 1024);
        };
        __builtin_amdgcn_s_barrier();                      // everyone is done with the previous tile's ring
        stage_dma(0, 0);
        float b[16], bn[16];
        gather(row, 0, b);
        row_next = entry(t + 1);
        f32x16 acc[2];
#pragma unroll
        for (int tu = 0; tu < 2; ++tu)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tu][r] = 0.0f;
        for (int kc = 0; kc < n_kchunks; ++kc) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // chunk kc: this wave's ring pieces and row operands
            __builtin_amdgcn_s_barrier();                      // ... everybody's pieces; the other slot is free
            asm volatile("" ::: "memory");
            if (kc + 1 < n_kchunks) { stage_dma(kc + 1, (kc + 1) & 1); gather(row, kc + 1, bn); }
            if (wave_live) {
                const char* st = ring[kc & 1];
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    f32x4 a[2];
#pragma unroll
                    for (int tu = 0; tu < 2; ++tu) a[tu] = *(const f32x4*)(st + ((tu * 4 + g4) * 64 + lane) * 16);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int tu = 0; tu < 2; ++tu)
                            acc[tu] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tu][j], b[4 * g4 + j], acc[tu], 0, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) b[i] = bn[i];
        }
        if (!wave_live) continue;
        const float* wq = (const float*)(wtile + FT_TILE);     // the block's 128 |w|^2 (+inf behind the last unit)
        float best = __builtin_inff();
        int bkey = 0, brank = 0;
#pragma unroll
        for (int tu = 0; tu < 2; ++tu) {
            f32x4 wv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) wv[q] = *(const f32x4*)(wq + uh * 64 + tu * 32 + 8 * q + 4 * half);
            const int tile = ub * (FT_BN / 32) + uh * 2 + tu;
            if (sub44) f32_tile_argmin_ranked<MODE>(acc[tu], wv, xs, tile, half, best, bkey, brank);
            else if ((tile + 1) * 32 > K) f32_tile_argmin<MODE, true>(acc[tu], wv, xs, tile, half, K, best, bkey);
            else f32_tile_argmin<MODE, false>(acc[tu], wv, xs, tile, half, K, best, bkey);
        }
        int bidx = f32_key_unit(bkey, half);
        if (!sub44) brank = bidx;
        const float ob = __shfl_xor(best, 32, 64);
        const int oi = __shfl_xor(bidx, 32, 64), ork = __shfl_xor(brank, 32, 64);
        if (ob < best || (ob == best && ork < brank)) { best = ob; bidx = oi; }
        if (half == 0 && row >= 0) {
            const uint32_t bits = __float_as_uint(best);
            const uint32_t key = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
            atomicMin(best64 + row, ((unsigned long long)key << 32) | (uint32_t)(perm != nullptr ? perm[bidx] : bidx));
        }
    }
}

// merge key -> id; rows the scheme cannot vouch for (no candidate was scored: the key is still all ones; a best score
// that is not finite; an overflowed pass) -> the fallback list
__global__ __launch_bounds__(256) void exact_finalize_kernel(const unsigned long long* __restrict__ best64, long N, int K,
                                                             const int* __restrict__ overflow, int* __restrict__ out,
                                                             int* __restrict__ fb_list, int* __restrict__ fb_count,
                                                             const int* __restrict__ order = nullptr) {
    const long row = (long)blockIdx.x * 256 + threadIdx.x;
    if (row >= N) return;
    const long orow = order != nullptr ? order[row] : row;   // (sorted pass: the key at position row belongs to row order[row])
    const unsigned long long k64 = best64[row];
    const uint32_t key = (uint32_t)(k64 >> 32), unit = (uint32_t)k64;
    const uint32_t bits = (key & 0x80000000u) ? (key & 0x7FFFFFFFu) : ~key;
    const bool finite = (bits & 0x7F800000u) != 0x7F800000u;
    if (*overflow || k64 == ~0ull || !finite || unit >= (uint32_t)K) fb_list[atomicAdd(fb_count, 1)] = (int)orow;
    else out[orow] = (int)unit;
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

// max_n v[n] (positive floats; NaN left out) into *out, which the caller zeroed.  Grid-stride: a few hundred workgroups hand in
// one maximum per wave (a workgroup per 256 values put 16 384 atomics of a million-row set on one address: 0.1 ms for 4 MB)
__global__ __launch_bounds__(256) void exact_max_kernel(const float* __restrict__ v, long n, float* __restrict__ out) {
    float m = 0.0f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) { const float s = v[i]; if (s == s) m = fmaxf(m, s); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomic_max_pos_f32(out, m);
}

// |x^_n - x~_n| of every row for the tile-image screens (beyond 128 features): x^ = x * (1/|x| when unit_sq) * the power
// of two of ex_scale(*scale_max2), exactly as prep_tiles_bf16_kernel forms it; one wave per row
template <class EL>
__global__ __launch_bounds__(256) void exact_rowerr_kernel(const float* __restrict__ X, long N, int D,
                                                           const float* __restrict__ unit_sq,
                                                           const float* __restrict__ scale_max2, float* __restrict__ xerr) {
    using E = typename EL::T;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= N) return;
    float scale = 1.0f;
    if (unit_sq != nullptr) { const float q = unit_sq[row]; scale = q > 0.0f ? 1.0f / __builtin_sqrtf(q) : 0.0f; }
    const float pow2 = ex_scale(*scale_max2);
    float er = 0.0f;
    for (int k = lane; k < D; k += 64) {
        const float f = X[row * D + k] * scale * pow2;
        const float e = half_operand_error(f, (float)cvt<E>(f));
        er = __builtin_fmaf(e, e, er);
    }
    er = wave_sum(er);
    if (lane == 0) xerr[row] = __builtin_sqrtf(er) * (1.0f + 1.0f / 1024.0f);
}

// max_k |w^_k - w~_k|^2 over the units (w^ = sw w in float32 -- of the unit-length w / |w| when unit_wsq --, w~ = what the
// MFMA reads: half_operand_error) into *out
template <class EL>
__global__ __launch_bounds__(256) void exact_werr_kernel(const float* __restrict__ W, int K, int D,
                                                         const float* __restrict__ wmax2, float* __restrict__ out,
                                                         const float* __restrict__ unit_wsq = nullptr) {
    using E = typename EL::T;
    // a wave takes EX_WERR_UNITS consecutive units, one at a time (coalesced reads of the unit's row), and hands in ONE maximum
    const int lane = threadIdx.x & 63;
    const long u_begin = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * EX_WERR_UNITS;
    const float pow2 = ex_scale(*wmax2);
    float m = 0.0f;
    for (long u = u_begin; u < u_begin + EX_WERR_UNITS && u < K; ++u) {
        float scale = 1.0f;
        bool poison = false;
        if (unit_wsq != nullptr) {
            const float q = unit_wsq[u];
            scale = q > 0.0f ? 1.0f / __builtin_sqrtf(q) : 0.0f;
            // cosine: |x|^2 |w|^2 must stay a normal float32 for the float32 kernel's division to be what the bound models
            // (ex_row_bound keeps |x|^2 in the same window); a zero, tiny, huge or NaN unit: no bound -> every row falls back
            if (!(q > 0x1p-60f && q < 0x1p60f)) poison = true;
        }
        scale *= pow2;
        float er = 0.0f;
        for (int k = lane; k < D; k += 64) {
            const float f = W[u * D + k] * scale;
            const float e = half_operand_error(f, (float)cvt<E>(f));
            er = __builtin_fmaf(e, e, er);
        }
        er = wave_sum(er);
        if (poison) er = __builtin_inff();
        if (er == er) m = fmaxf(m, er);
    }
    if (lane == 0 && u_begin < K) atomic_max_pos_f32(out, m);
}

// ---- the canary (som_set_verify / SOM_VERIFY=n): n strided rows of every BMU launch are scored again by the float32
// kernel, and the launch's own pick must be the float32 pick or within the precision mode's bound of it.
// picks[i], best[i]: the launch's / the float32 kernel's unit for gathered row i.  tol_rel: the mode's bound on the
// score gap relative to |x| max|w| (euclidean: scores |w|^2 - 2 x.w) or absolute (cosine: scores 1 - cos); 0 = the
// ids must be equal.  bad[0] counts offenders, bad[1..3] = (row, pick, float32 pick) of one of them.
__global__ __launch_bounds__(256) void verify_picks_kernel(const float* __restrict__ Xv, int n, int D,
                                                           const float* __restrict__ W, const float* __restrict__ wsq,
                                                           const float* __restrict__ wmax2, const int* __restrict__ rows,
                                                           const int* __restrict__ picks, const int* __restrict__ best,
                                                           int cosine, float tol_rel, int* __restrict__ bad) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int p = picks[i], b = best[i];
    if (p == b) return;
    bool ok = false;
    if (tol_rel > 0.0f) {
        const float* x = Xv + (long)i * D;
        double cp = 0.0, cb = 0.0, xs = 0.0;
        for (int k = 0; k < D; ++k) {
            const double xv = x[k];
            cp += xv * W[(long)p * D + k]; cb += xv * W[(long)b * D + k]; xs += xv * xv;
        }
        if (cosine) {
            const double sp = 1.0 - cp / sqrt(xs * wsq[p]), sb = 1.0 - cb / sqrt(xs * wsq[b]);
            ok = sp <= sb + tol_rel || !(xs > 0.0);
        } else {
            const double sp = wsq[p] - 2.0 * cp, sb = wsq[b] - 2.0 * cb;
            ok = sp <= sb + (double)tol_rel * sqrt(xs) * sqrt((double)*wmax2);
        }
    }
    if (!ok && atomicAdd(bad, 1) == 0) { bad[1] = rows[i]; bad[2] = p; bad[3] = b; }
}

// The float32 first-minimum of one gathered row over ALL units by the plainest possible route: one workgroup per row,
// a thread per unit stride, the k-ordered fmaf chain on the vector ALU straight from the float32 codebook and |w|^2 --
// no MFMA, no stage image, no LDS-DMA, no codebook parts.  Same arithmetic as the parity kernels (score_f32), so its
// answer is theirs; sharing nothing with them is the point.
template <int MODE>
__global__ __launch_bounds__(256) void verify_best_kernel(const float* __restrict__ Xv, int D, const float* __restrict__ W,
                                                          const float* __restrict__ wsq, int K, int* __restrict__ best_out) {
    extern __shared__ float vsm[];                         // D floats of the row, then the reduction scratch
    __shared__ float rv[256];
    __shared__ int ru[256];
    __shared__ float xsq_s;
    const int tid = threadIdx.x;
    const float* x = Xv + (long)blockIdx.x * D;
    for (int k = tid; k < D; k += 256) vsm[k] = x[k];
    if (tid == 0) xsq_s = np_pairwise_sq_sum(x, D);
    __syncthreads();
    const float xsq = xsq_s;
    float best = __builtin_inff();
    int bu = 0;
    for (int u = tid; u < K; u += 256) {
        const float* w = W + (long)u * D;
        float c = 0.0f;
        for (int k = 0; k < D; ++k) c = __builtin_fmaf(w[k], vsm[k], c);
        const float v = score_f32<MODE>(c, wsq[u], xsq);
        if (v < best) { best = v; bu = u; }               // (ascending u per thread: the first minimum of its stride)
    }
    rv[tid] = best; ru[tid] = bu;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            const float ov = rv[tid + o];
            const int ou = ru[tid + o];
            if (ov < rv[tid] || (ov == rv[tid] && ou < ru[tid])) { rv[tid] = ov; ru[tid] = ou; }
        }
        __syncthreads();
    }
    if (tid == 0) best_out[blockIdx.x] = rv[0] < __builtin_inff() ? ru[0] : 0;   // nothing below +inf: numpy.argmin's 0
}

__global__ __launch_bounds__(256) void verify_pick_rows_kernel(long N, int n, const int* __restrict__ ids, int* __restrict__ rows,
                                                               int* __restrict__ picks) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long r = (long)i * (N / n) + (N / n) / 2;        // n strided rows, mid-stride
    rows[i] = (int)r;
    picks[i] = ids[r];
}

// ---- measurement hook (som_debug_mfma16): ONE v_mfma_f32_16x16x32 on caller-supplied operands, so that a test can
// put a number on what exact_bound() charges per MFMA (KAPPA ulps of the accumulator's magnitude): the hardware's
// internal summation of the 32 products and the accumulator is not documented.
// A [16][32] (row = output row), B [32][16] (column = output column), C / D [16][16], all row-major; one wave.
template <class EL>
__global__ __launch_bounds__(64) void debug_mfma16_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B,
                                                          const float* __restrict__ Cin, float* __restrict__ Dout) {
    using E = typename EL::T;
    using v8 = typename V8<E>::t;
    const int lane = threadIdx.x, col = lane & 15, quad = lane >> 4;
    v8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint16_t ua = A[col * 32 + quad * 8 + j], ub = B[(quad * 8 + j) * 16 + col];
        a[j] = __builtin_bit_cast(E, ua);
        b[j] = __builtin_bit_cast(E, ub);
    }
    f32x4 c;
#pragma unroll
    for (int r = 0; r < 4; ++r) c[r] = Cin[(quad * 4 + r) * 16 + col];
    const f32x4 d = mfma16(a, b, c);
#pragma unroll
    for (int r = 0; r < 4; ++r) Dout[(quad * 4 + r) * 16 + col] = d[r];
}

// the codebook and its |w|^2 in patch order (perm[position] = unit): what the exact mode's operand images are prepared from
__global__ __launch_bounds__(256) void exact_permute_kernel(const float* __restrict__ W, const float* __restrict__ wsq, int K,
                                                            int D, const int* __restrict__ perm, float* __restrict__ Wp,
                                                            float* __restrict__ wsq_p) {
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    if ((D & 3) == 0) {
        const int q = D >> 2;
        if (id >= (long)K * q) return;
        const int pos = (int)(id / q), c = (int)(id - (long)pos * q);
        const int u = perm[pos];
        *(f32x4*)(Wp + (long)pos * D + 4 * c) = *(const f32x4*)(W + (long)u * D + 4 * c);
        if (c == 0) wsq_p[pos] = wsq[u];
    } else {
        if (id >= (long)K * D) return;
        const int pos = (int)(id / D), d = (int)(id - (long)pos * D);
        const int u = perm[pos];
        Wp[id] = W[(long)u * D + d];
        if (d == 0) wsq_p[pos] = wsq[u];
    }
}

// wn = wsq (the float32 kernel's own |w|^2) and its maximum: the screen's initial accumulator then carries the same
// norm the re-score adds
// (blocks of 1 024 units hand in ONE maximum each: a wave per atomic -- 1 024 of them polling one address at 256 x 256 units --
//  took 17 us for a 256 KB copy)
__global__ __launch_bounds__(1024) void exact_copy_wsq_kernel(const float* __restrict__ wsq, int K, float* __restrict__ wn,
                                                              float* __restrict__ wmax2) {
    __shared__ float wave_max[16];
    const long u = (long)blockIdx.x * 1024 + threadIdx.x;
    float s = 0.0f;
    if (u < K) { s = wsq[u]; wn[u] = s; }
    float m = (s == s) ? s : 0.0f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x < 16) {
        m = wave_max[threadIdx.x];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (threadIdx.x == 0) atomic_max_pos_f32(wmax2, m);
    }
}

}  // namespace somhip
