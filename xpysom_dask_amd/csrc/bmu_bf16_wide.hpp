// Fused distance + BMU argmin, bf16, 128 < input_len <= 800 with the SAMPLES resident in registers.
//
// The two-sided tiling of bmu_bf16_tiled.hpp stages both operands for every (unit block, feature chunk) and its
// sample tiles miss the L2 once per pass (18 % of 845 GB staged per launch at 512x512x784: 12 % of the launch,
// DESIGN.md 3.1a).  Here a workgroup keeps its 256 samples' B fragments in registers for the whole codebook scan, as
// bmu_bf16_k16.hpp does up to 128 features, and only the codebook streams:
//   * 8 waves (two per SIMD, 256 registers each); a wave holds 2 x 16 samples x KS32 feature chunks = 8 KS32 VGPRs of
//     B fragments (200 at 784 features) and 2 unit tiles x 2 sample blocks of accumulators;
//   * stage = 32 units in fragment order, [tile 0..1][chunk 0..KS32-1][lane][8 bf16 of -w~] + 32 initial accumulators
//     (B + |w~|^2/2) in the stage's last KiB: (2 KS32 + 1) KiB, 51 KiB at 784 features; three ring slots;
//   * one A fragment read from LDS feeds two MFMAs (128 B/clk/CU of LDS reads at full rate);
//   * stage s+2 is issued (LDS-DMA) right after the barrier of stage s, into the slot stage s-1 was read from, and
//     awaited (vmcnt(0)) before the barrier of stage s+1: a whole stage of MFMAs (3 200 cycles per SIMD) covers it;
//   * every workgroup resident on the chip scans the same codebook part from the same stage at the same pace, so a
//     stage misses the L2 once per XCD and the other 31 workgroups hit; the sample image is read once per part.
// Arithmetic, offset B, key packing ((bits & ~mask) | (tile << 2 | reg)), part split and the 64-bit atomicMin merge
// are those of bmu_bf16_k16.hpp; the sample operand is the tile image of bmu_bf16_tiled.hpp (256-row blocks).
#pragma once
#include <type_traits>
#include "bmu_bf16.hpp"
#include "bmu_bf16_tiled.hpp"

namespace somhip {

constexpr int WD_T = 2;                          // 16-unit tiles per stage
constexpr int WD_STAGE_UNITS = 16 * WD_T;
constexpr int WD_SB = 2;                         // 16-sample blocks per wave
constexpr int WD_NW = 8;                         // waves per workgroup
constexpr int WD_WG_SAMPLES = WD_NW * WD_SB * 16;
constexpr int WD_SLOTS = 3;
constexpr int WD_XTILE = (WD_WG_SAMPLES / 16) * 1024;   // bytes of one (256-row block, 32-feature chunk) sample tile
static_assert(WD_WG_SAMPLES == TileCfg<8, 2, 4>::BM && WD_XTILE == TileCfg<8, 2, 4>::XTILE,
              "the sample operand is the 256-row tile image of the tiled kernel");

__host__ __device__ constexpr int wd_stage_bytes(int ks32) { return (WD_T * ks32 + 1) * 1024; }

// codebook -> stage image (the initial accumulators are written by prep_wsqh_kernel: they depend on the row set).
template <class EL = Bf16>
__global__ __launch_bounds__(256) void prep_w_bf16_wide_kernel(const float* __restrict__ W, int K, int D, int ks32,
                                                               char* __restrict__ Wst, int n_stages,
                                                               const float* __restrict__ unit_wsq,
                                                               const float* __restrict__ scale_max2 = nullptr) {
    using E = typename EL::T;
    using bf16x8 = typename V8<E>::t;
    long id = (long)blockIdx.x * 256 + threadIdx.x;
    long total = (long)n_stages * WD_T * ks32 * 64;
    if (id >= total) return;
    int lane = id & 63;
    long t = id >> 6;
    int ks = t % ks32; t /= ks32;
    int t16 = t % WD_T;
    long stage = t / WD_T;
    long u = stage * WD_STAGE_UNITS + t16 * 16 + (lane & 15);
    int k0 = ks * 32 + (lane >> 4) * 8;
    float scale = 1.0f;                          // cosine: unit-length rows (prep_w_bf16_k16_kernel)
    if (unit_wsq != nullptr && u < K) { float q = unit_wsq[u]; scale = q > 0.0f ? 1.0f / __builtin_sqrtf(q) : 0.0f; }
    const float pow2 = scale_max2 != nullptr ? ex_scale(*scale_max2) : 1.0f;   // exact mode: a power of two on top (exact)
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float f = (u < K && k0 + j < D) ? W[u * D + k0 + j] * scale * pow2 : 0.0f;
        v[j] = cvt<E>(-f);
    }
    *(bf16x8*)(Wst + stage * wd_stage_bytes(ks32) + ((long)(t16 * ks32 + ks) * 64 + lane) * 16) = v;
}

// _merge_updates (xpysom.py:446-455) fused with the NEXT epoch's operand preparation on the wide path, as
// merge_prep_k16_kernel is on the 128-feature path: one pass over the fused accumulator writes the merged float32
// codebook, the bf16 stage image and |w~_k|^2 (+ its maximum) -- merge_kernel + row_sq_f32_kernel (cosine) +
// prep_w_bf16_wide_kernel + rownorm_bf16_kernel in one launch that reads the accumulator once and the old codebook
// only where no row was in reach.  Workgroup = one 16-unit tile, 4 waves; wave w takes the feature chunks w, w+4, ...;
// thread = (unit, 8 features) of each of its chunks = one 16-byte fragment chunk of the image.  cosine != 0: the
// image holds the unit-length rows (scale 1/|w|, a zero row stays zero) and |w~|^2 is reported as 0 (rownorm_bf16_kernel).
constexpr int WD_MP_ITERS = 7;                   // chunks per wave: up to 28 feature chunks (the wide kernel stops at 25)
template <class EL = Bf16>
__global__ __launch_bounds__(256) void merge_prep_wide_kernel(float* __restrict__ W, const float* __restrict__ ACC, int K,
                                                              int D, int D1p, int ks32, char* __restrict__ Wst,
                                                              float* __restrict__ wn, float* __restrict__ wmax2,
                                                              int cosine) {
    using E = typename EL::T;
    using bf16x8 = typename V8<E>::t;
    __shared__ float red[2][4][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, quad = lane >> 4;
    const long tile = blockIdx.x;
    const long stage = tile / WD_T;
    const int t16 = (int)(tile - stage * WD_T);
    const long u = tile * 16 + col;
    const bool live = u < K;
    const bool aligned = (D & 3) == 0;                      // 16-byte aligned rows on both sides (D1p is a multiple of 4)
    const float den = live ? ACC[u * D1p + D] : 0.0f;
    float w[WD_MP_ITERS][8];
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < WD_MP_ITERS; ++i) {
        const int ks = wave + 4 * i;
        const int k0 = ks * 32 + quad * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) w[i][j] = 0.0f;
        if (ks >= ks32 || !live || k0 >= D) continue;
        const bool full = aligned && k0 + 8 <= D;
        const float* src = den != 0.0f ? ACC + u * D1p + k0 : W + u * D + k0;   // no row in reach: the old weights stay
        if (full) {
            const f32x4 a = *(const f32x4*)src, c = *(const f32x4*)(src + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { w[i][j] = a[j]; w[i][4 + j] = c[j]; }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (k0 + j < D) w[i][j] = src[j];
        }
        if (den != 0.0f) {
#pragma unroll
            for (int j = 0; j < 8; ++j) w[i][j] = w[i][j] / den;
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
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) q = __builtin_fmaf(w[i][j], w[i][j], q);
    }
    float scale = 1.0f;
    if (cosine) {                                            // |w|^2 of the merged row: over the quads, then over the waves
        q += __shfl_xor(q, 16, 64);
        q += __shfl_xor(q, 32, 64);
        if (lane < 16) red[0][wave][lane] = q;
        __syncthreads();
        const float qq = (red[0][0][col] + red[0][1][col]) + (red[0][2][col] + red[0][3][col]);
        scale = qq > 0.0f ? 1.0f / __builtin_sqrtf(qq) : 0.0f;
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < WD_MP_ITERS; ++i) {
        const int ks = wave + 4 * i;
        if (ks >= ks32) continue;
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float f = w[i][j] * scale;
            const E b = cvt<E>(f);
            v[j] = cvt<E>(-f);                             // (rounding is sign-symmetric: -bf16(f) == bf16(-f))
            const float r = (float)b;
            s = __builtin_fmaf(r, r, s);
        }
        *(bf16x8*)(Wst + stage * wd_stage_bytes(ks32) + ((long)(t16 * ks32 + ks) * 64 + lane) * 16) = v;
    }
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (lane < 16) red[1][wave][lane] = s;
    __syncthreads();
    if (wave == 0 && lane < 16) {
        float t = (red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane]);
        if (!live) t = 0.0f;
        if (live) wn[u] = cosine ? 0.0f : t;
        float m = t;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (lane == 0) atomic_max_pos_f32(wmax2, m);
    }
}

// GM (precision 'exact' beyond 128 features, bmu_exact.hpp): as in bmu_bf16_k16_kernel<.., GM = true> -- values only,
// the minimum of every GROUP of 64 units (two stages) per row stored where it is within the row's bound of the
// minimum so far, and the mask of the rows stored (32 per wave) per group; the parts split on group boundaries.
// TL (exact_skip_wide.hpp): the workgroup walks its tile's list of GROUPS (glist: group << 4 | 15 items, gcnt of them) -- both
// stages of each -- instead of every stage.  With GM: the screen under a plan; without: the scout's pick of a pseudo last BMU.
// PLAN (exact_skip_wide.hpp): the same scan over a stage image of group CENTROIDS (32 to a stage; the stage's tail holds,
// behind the 32 initial accumulators S'(B' + |c|^2 / 2) - hS r^2, the 32 radii sw r) with the plan's test as its epilogue:
// need(row, c) = not (acc - (sx sqrt(U))(sw r) > P(row)), rows' P and sx sqrt(U) in planP / planXs; OR over the tile's 256
// rows into need[tile][word] (bit g & 63 of word g >> 6 <-> group g): what exact_lists_kernel turns into the tile's list.
// (the kernel's body for one tile of WD_WG_SAMPLES rows: tile bx, part by of ny of the stages -- TL: of the tile's listed groups)
template <int KS32, class EL, bool GM, bool TL, bool PLAN>
__device__ __forceinline__ void bmu_bf16_wide_body(const char* __restrict__ Ximg, long N, const char* __restrict__ Wst, int n_stages,
                                                   unsigned long long* __restrict__ out64, uint32_t* __restrict__ gmin, long gm_stride,
                                                   uint32_t* __restrict__ gflags32, const float* __restrict__ xsq,
                                                   const float* __restrict__ xerr, const float* __restrict__ xmax2,
                                                   const float* __restrict__ wmax2, const float* __restrict__ werr2, const ExactBound& eb,
                                                   const int* __restrict__ glist, const int* __restrict__ gcnt, int n_groups_all,
                                                   const float* __restrict__ planP, const float* __restrict__ planXs,
                                                   unsigned long long* __restrict__ need, int n_words, const long bx, const int by,
                                                   const int ny) {
    // (TL without GM: the scout of exact_skip_wide.hpp -- the plain kernel, unit indices kept, over a tile's few listed groups)
    static_assert(!PLAN || (!GM && !TL), "the plan is a mode of its own");
    using E = typename EL::T;
    using bf16x8 = typename V8<E>::t;
    constexpr int STAGE = wd_stage_bytes(KS32);
    constexpr int PIECES = WD_T * KS32;                      // whole 1 KiB pieces; the C-in row follows them
    constexpr int CIN_LANES = (PLAN ? 2 : 1) * WD_STAGE_UNITS * 4 / 16;   // (PLAN: the 32 radii behind the 32 initial accumulators)
    constexpr uint32_t IDX_MASK = 4 * WD_T - 1;              // (tile << 2 | register) in the low mantissa bits
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int quad = lane >> 4, col = lane & 15;

    // this workgroup's share of the codebook stages: POSITIONS [s_begin, s_end) of its walk; sid(position) = the stage there
    int s_begin = (int)((long)n_stages * by / ny);
    int s_end = (int)((long)n_stages * (by + 1) / ny);
    if (GM) {                                                // parts of whole groups (pairs of stages)
        const int n_groups = (n_stages + 1) / 2;
        s_begin = 2 * (int)((long)n_groups * by / ny);
        s_end = min(2 * (int)((long)n_groups * (by + 1) / ny), n_stages);
    }
    const int* my_list = nullptr;
    if (TL) {                                                // the tile's listed groups, two stages each (the host: K % 64 == 0)
        my_list = glist + bx * n_groups_all;
        const int n_g = gcnt[bx];
        s_begin = 2 * (int)((long)n_g * by / ny);
        s_end = 2 * (int)((long)n_g * (by + 1) / ny);
    }
    auto sid = [&](int i) -> int { return TL ? 2 * (__builtin_amdgcn_readfirstlane(my_list[i >> 1]) >> 4) + (i & 1) : i; };
    if (s_begin >= s_end) return;                            // (whole workgroup: no barrier is left behind)
    // GM: lane l < 32 <-> row (block, wave, l): its minimum so far, its bound E; pmin: this lane's minimum over the group
    const long wave_row0 = bx * WD_WG_SAMPLES + wave * (WD_SB * 16);
    float run_min = __builtin_inff(), row_e = __builtin_inff();
    int run_arg = 0;
    uint32_t pmin[WD_SB];
#pragma unroll
    for (int sb = 0; sb < WD_SB; ++sb) pmin[sb] = 0xFFFFFFFFu;
    if (GM) {
        const long r = wave_row0 + (lane & 31);
        const float e = r < N ? ex_row_bound(eb, ex_scales(xmax2, wmax2, werr2), xsq[r], xerr[r]) : __builtin_nanf("");
        if (e == e) row_e = e;                               // (a row the bound does not cover keeps everything: the scan drops it)
    }

    auto issue = [&](int i, int slot) {
        // (uniform base + a 32-bit lane offset: the scalar-base form of the load, no 64-bit address kept in registers)
        const char* src = Wst + (long)sid(i) * STAGE;
        const uint32_t lane16 = (uint32_t)lane * 16u;
        char* dst = smem + slot * STAGE;
        for (int p = wave; p < PIECES; p += WD_NW) lds_dma_16(src + p * 1024 + lane16, dst + p * 1024);
        if (wave == PIECES % WD_NW && lane < CIN_LANES) lds_dma_16(src + PIECES * 1024 + lane16, dst + PIECES * 1024);
    };
    issue(s_begin, 0);
    if (s_begin + 1 < s_end) issue(s_begin + 1, 1);

    // the wave's samples: B fragments of 2 x 16 rows, every feature chunk
    bf16x8 xf[WD_SB][KS32];
    {
        const char* xb = Ximg + bx * KS32 * WD_XTILE + ((wave * WD_SB) * 64 + lane) * 16;
#pragma unroll
        for (int ks = 0; ks < KS32; ++ks)
#pragma unroll
            for (int sb = 0; sb < WD_SB; ++sb) xf[sb][ks] = *(const bf16x8*)(xb + (long)ks * WD_XTILE + sb * 1024);
    }

    uint32_t gbest[WD_SB];                                   // unsigned keys: a NaN of either sign never beats a finite d'
    int gstage[WD_SB];
#pragma unroll
    for (int sb = 0; sb < WD_SB; ++sb) { gbest[sb] = 0xFFFFFFFFu; gstage[sb] = 0; }

    f32x4 acc[WD_T][WD_SB];
    auto slot_of = [&](int s) { return smem + ((s - s_begin) % WD_SLOTS) * STAGE; };
    auto begin_stage = [&](const char* st) {
        const float* wq = (const float*)(st + PIECES * 1024);
#pragma unroll
        for (int t = 0; t < WD_T; ++t) {
            const f32x4 c = *(const f32x4*)(wq + t * 16 + 4 * quad);
#pragma unroll
            for (int sb = 0; sb < WD_SB; ++sb) acc[t][sb] = c;
        }
    };
    // The fragments of chunk ks + 1 are requested BEFORE the MFMAs of chunk ks, by hand: ds_read_b128 and the counted
    // lgkmcnt wait are inline assembly.  Left to the compiler every read sinks to just before its use (register pressure:
    // 200 of the 256 VGPRs hold samples) and, pinned with sched_barrier, its wait-count pass still answers with
    // lgkmcnt(0) every second chunk; either way a wave waits out the LDS latency with only its partner's four MFMAs to
    // cover it -- the matrix pipe idled 41 % of the launch (profiles/r03_c5_pmc_traffic: SQ_WAIT_INST_ANY 0.35 of the
    // wave cycles, LDS array 31 % busy, no bank conflicts).  (Waits the compiler adds for its own LDS reads -- the initial
    // accumulators -- see fewer reads in flight than there are: stricter than needed, never too weak.)
    auto chunks = [&](const char* st, auto k0, auto k1) {
        constexpr int K0 = decltype(k0)::value, K1 = decltype(k1)::value;
        const uint32_t base = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)st + (uint32_t)lane * 16u;
        // SOM_WIDE_PLAIN_READS (build.py sets it for a hipcc other than the one this schedule was validated with; the test
        // suite builds such a library and compares): plain C++ reads, the compiler's own waits.  The hand-placed form relies
        // on the register allocator renaming an[] into a[] (no copy of a register whose read is still in flight) and on
        // no spill of them: true of the validated compiler's output, checked per build by SOM_VERIFY in smoke().
#ifdef SOM_WIDE_PLAIN_READS
        auto read16 = [&](int piece) { return *(const f32x4*)(st + lane * 16 + piece * 1024); };
#else
        auto read16 = [&](int piece) {
            f32x4 v;
            asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(base + (uint32_t)piece * 1024u));
            return v;
        };
#endif
        f32x4 a[WD_T], an[WD_T];
#pragma unroll
        for (int t = 0; t < WD_T; ++t) a[t] = read16(t * KS32 + K0);
#pragma unroll
        for (int ks = K0; ks < K1; ++ks) {
            static_assert(WD_T == 2, "the counted wait below leaves exactly the next chunk's two reads in flight");
            if (ks + 1 < K1) {
#pragma unroll
                for (int t = 0; t < WD_T; ++t) an[t] = read16(t * KS32 + ks + 1);
#ifndef SOM_WIDE_PLAIN_READS
                asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a[0]), "+v"(a[1]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]));
#endif
            }
#pragma unroll
            for (int t = 0; t < WD_T; ++t)
#pragma unroll
                for (int sb = 0; sb < WD_SB; ++sb)
                    acc[t][sb] = mfma16(__builtin_bit_cast(bf16x8, a[t]), xf[sb][ks], acc[t][sb]);
            if (ks + 1 < K1) {
#pragma unroll
                for (int t = 0; t < WD_T; ++t) a[t] = an[t];
            }
        }
    };
    uint32_t pend_full = 0;                                  // GM: the group minimum waiting to be stored, whether to, its group
    bool pend_keep = false;
    int pend_group = -1;
    auto flush_group = [&]() {
        if (!GM || pend_group < 0) return;                   // (wave-uniform)
        // (uniform base + 32-bit lane offset: no 64-bit address kept in registers across the stages)
        uint32_t* grow = gmin + ((long)pend_group * gm_stride + wave_row0);
        uint32_t off = (uint32_t)lane;
        asm volatile("" : "+v"(off));                         // (opaque: the compiler would hoist gmin + row as a 64-bit VGPR pair and spill it)
        if (pend_keep) grow[off] = pend_full;
        const unsigned long long mask = __ballot(pend_keep);
        if (lane == 0) gflags32[ex_flag_index(wave_row0 >> 6, pend_group, (n_stages + 1) >> 1, gm_stride) * 2 + ((wave_row0 >> 5) & 1)] = (uint32_t)mask;
        pend_group = -1;
    };
    float pP[WD_SB], pXs[WD_SB];                             // PLAN: the rows' thresholds and sx sqrt(U) (1 + 2^-10)
    unsigned long long* nl = (unsigned long long*)(smem + WD_SLOTS * STAGE);
    if (PLAN) {
#pragma unroll
        for (int sb = 0; sb < WD_SB; ++sb) {
            const long r = wave_row0 + sb * 16 + col;
            pP[sb] = r < N ? planP[r] : -__builtin_inff();   // (rows behind the pass need nothing)
            pXs[sb] = r < N ? planXs[r] : 0.0f;
        }
        for (int i = tid; i < n_words; i += 64 * WD_NW) nl[i] = 0ull;
    }
    auto finish_stage = [&](int s, bool last, const char* st) {
        if (PLAN) {
            const float* rq = (const float*)(st + PIECES * 1024) + WD_STAGE_UNITS;
            uint32_t mine = 0u;                              // bit (16 t + 4 quad + r) <-> centroid of that place in the stage
#pragma unroll
            for (int t = 0; t < WD_T; ++t) {
                const f32x4 swr = *(const f32x4*)(rq + t * 16 + 4 * quad);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    bool nd = false;
#pragma unroll
                    for (int sb = 0; sb < WD_SB; ++sb) nd = nd || !(__builtin_fmaf(-pXs[sb], swr[r], acc[t][sb][r]) > pP[sb]);
                    const unsigned long long b = __ballot(nd);
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd)
                        if ((b >> (16 * qd)) & 0xFFFFull) mine |= 1u << (t * 16 + 4 * qd + r);
                }
            }
            if (lane == 0 && mine != 0u) atomicOr(nl + (s >> 1), (unsigned long long)mine << (32 * (s & 1)));
            return;
        }
        if (GM) {
        static_assert(!GM || WD_SB == 2, "the group-minimum store pairs two 16-sample blocks per wave");
#pragma unroll
        for (int sb = 0; sb < WD_SB; ++sb) {
            uint32_t c = 0xFFFFFFFFu;
#pragma unroll
            for (int t = 0; t < WD_T; ++t) {
                c = min(min(c, __float_as_uint(acc[t][sb][0])), __float_as_uint(acc[t][sb][1]));
                c = min(min(c, __float_as_uint(acc[t][sb][2])), __float_as_uint(acc[t][sb][3]));
            }
            pmin[sb] = min(pmin[sb], c);
        }
        if ((s & 1) || last) {                           // the group is complete: join the four lane quads, store
            uint32_t v[WD_SB];
#pragma unroll
            for (int sb = 0; sb < WD_SB; ++sb) {
                const auto a = __builtin_amdgcn_permlane32_swap(pmin[sb], pmin[sb], false, false);
                const uint32_t m2 = min(a[0], a[1]);
                const auto b = __builtin_amdgcn_permlane16_swap(m2, m2, false, false);
                v[sb] = min(b[0], b[1]);
                pmin[sb] = 0xFFFFFFFFu;
            }
            // lanes 0..15: sample block 0, lanes 16..31: sample block 1 = 32 consecutive rows
            const uint32_t full = (lane & 16) ? v[WD_SB - 1] : v[0];
            const float f = __uint_as_float(full);
            const bool keep = lane < 32 && f <= run_min + row_e;
            if (f < run_min) run_arg = s >> 1;               // (the group that holds the row minimum: the first re-score round)
            run_min = __builtin_fminf(run_min, f);
            // the stores wait for the next stage (flush_group): issued here they would be the youngest vector-memory
            // operations at the loop's vmcnt(0), and every wave would sit out a store's round trip before the barrier
            pend_full = full; pend_keep = keep; pend_group = s >> 1;
        }
    } else {
#pragma unroll
    for (int sb = 0; sb < WD_SB; ++sb) {
        uint32_t c0 = 0xFFFFFFFFu, c1 = 0xFFFFFFFFu;
#pragma unroll
        for (int t = 0; t < WD_T; ++t) {
            const uint32_t k0 = (__float_as_uint(acc[t][sb][0]) & ~IDX_MASK) | (uint32_t)(t * 4 + 0);
            const uint32_t k1 = (__float_as_uint(acc[t][sb][1]) & ~IDX_MASK) | (uint32_t)(t * 4 + 1);
            const uint32_t k2 = (__float_as_uint(acc[t][sb][2]) & ~IDX_MASK) | (uint32_t)(t * 4 + 2);
            const uint32_t k3 = (__float_as_uint(acc[t][sb][3]) & ~IDX_MASK) | (uint32_t)(t * 4 + 3);
            c0 = min(min(c0, k0), k1);
            c1 = min(min(c1, k2), k3);
        }
        const uint32_t c = min(c0, c1);
        if (c < gbest[sb]) { gbest[sb] = c; gstage[sb] = s; }
    }
    }
    };
    // the samples are in their registers before the loop: an (empty) use of each here makes the compiler's wait-count pass
    // place its vmcnt wait HERE -- with their first use inside a branch of the loop it would wait for vmcnt(0) there, behind
    // the next stage's DMA
#pragma unroll
    for (int ks = 0; ks < KS32; ++ks)
#pragma unroll
        for (int sb = 0; sb < WD_SB; ++sb) asm volatile("" ::"v"(xf[sb][ks]) : "memory");
    using KC0 = std::integral_constant<int, 0>;
    using KCN = std::integral_constant<int, KS32>;
    // (measured, DESIGN.md 3.4: a counted vmcnt that leaves stage s+2's pieces in flight across the barrier of s+1 -- two
    //  stage times to land instead of one -- is 6 % SLOWER at 784 features: the workgroups of an XCD drift apart and stop
    //  sharing their L2 misses; waves 4..7 run half a stage behind their SIMD partners (stagger): +1 % at 784, -5 % at 256)
    SOM_STAMP_BEGIN();
    for (int s = s_begin; s < s_end; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of stage s+1 (first: of s, and its samples)
        __builtin_amdgcn_s_barrier();                         // ... everybody's; and nobody reads stage s-1 any more
        asm volatile("" ::: "memory");
        if (s + 2 < s_end) issue(s + 2, (s + 2 - s_begin) % WD_SLOTS);   // -> the slot stage s-1 was read from
        flush_group();                                        // the previous group's stores: a whole stage to complete
        const char* st = slot_of(s);
        begin_stage(st);
        chunks(st, KC0(), KCN());
        finish_stage(sid(s), s == s_end - 1, st);
    }
    SOM_STAMP_END();
    flush_group();
    if (PLAN) {
        __syncthreads();
        // (the parts of a tile's walk share words: OR into the words the host cleared)
        for (int i = tid; i < n_words; i += 64 * WD_NW)
            if (nl[i] != 0ull) atomicOr(need + bx * n_words + i, nl[i]);
        return;
    }

    if (GM) {
        // the row minimum IS the minimum so far after the last group (lanes 0..31 <-> rows wave_row0 + lane): the plain
        // value, every bit of it, and the group that holds it (no unit indices kept)
        const long row = wave_row0 + lane;
        if (lane < 32 && row < N) atomicMin(out64 + row, ((unsigned long long)__float_as_uint(run_min) << 32) | (uint32_t)run_arg);
        return;
    }
#pragma unroll
    for (int sb = 0; sb < WD_SB; ++sb) {
        const uint32_t code = gbest[sb] & IDX_MASK;
        const uint32_t unit = (uint32_t)gstage[sb] * WD_STAGE_UNITS + (code >> 2) * 16 + quad * 4 + (code & 3);
        unsigned long long comp = ((unsigned long long)(gbest[sb] & ~IDX_MASK) << 32) | unit;
        unsigned long long o = __shfl_xor(comp, 16, 64);
        if (o < comp) comp = o;
        o = __shfl_xor(comp, 32, 64);
        if (o < comp) comp = o;
        const long row = bx * WD_WG_SAMPLES + (wave * WD_SB + sb) * 16 + col;
        if (quad == 0 && row < N) atomicMin(out64 + row, comp);
    }
}

// items (GM + TL; nullptr: the grid is (tiles, parts)): the listed screen as a work queue, as bmu_bf16_k16_kernel's -- the tiles'
// lists are as uneven here (tools/wg_timeline.py with WT_SIDE=512 WT_D=784: a third of a launch was a tail of a few long walks).
template <int KS32, class EL = Bf16, bool GM = false, bool TL = false, bool PLAN = false>
__global__ __launch_bounds__(64 * WD_NW) void bmu_bf16_wide_kernel(const char* __restrict__ Ximg, long N,
                                                                   const char* __restrict__ Wst, int n_stages,
                                                                   unsigned long long* __restrict__ out64,
                                                                   uint32_t* __restrict__ gmin = nullptr, long gm_stride = 0,
                                                                   uint32_t* __restrict__ gflags32 = nullptr,
                                                                   const float* __restrict__ xsq = nullptr,
                                                                   const float* __restrict__ xerr = nullptr,
                                                                   const float* __restrict__ xmax2 = nullptr,
                                                                   const float* __restrict__ wmax2 = nullptr,
                                                                   const float* __restrict__ werr2 = nullptr,
                                                                   ExactBound eb = ExactBound(),
                                                                   const int* __restrict__ glist = nullptr,
                                                                   const int* __restrict__ gcnt = nullptr, int n_groups_all = 0,
                                                                   const float* __restrict__ planP = nullptr,
                                                                   const float* __restrict__ planXs = nullptr,
                                                                   unsigned long long* __restrict__ need = nullptr, int n_words = 0,
                                                                   const int2* __restrict__ items = nullptr,
                                                                   const int* __restrict__ n_items = nullptr,
                                                                   int* __restrict__ item_ctr = nullptr) {
    if (GM && TL && items != nullptr) {
        __shared__ int s_item;
        const int n = *n_items;
        for (;;) {
            if (threadIdx.x == 0) s_item = atomicAdd(item_ctr, 1);
            __syncthreads();                                 // (everyone is done with the previous item's LDS ring too)
            const int it = s_item;
            __syncthreads();
            if (it >= n) return;
            const int2 iv = items[it];
            bmu_bf16_wide_body<KS32, EL, GM, TL, PLAN>(Ximg, N, Wst, n_stages, out64, gmin, gm_stride, gflags32, xsq, xerr, xmax2, wmax2, werr2, eb,
                                                       glist, gcnt, n_groups_all, planP, planXs, need, n_words, (long)iv.x, iv.y & 0xFFFF,
                                                       iv.y >> 16);
        }
    }
    bmu_bf16_wide_body<KS32, EL, GM, TL, PLAN>(Ximg, N, Wst, n_stages, out64, gmin, gm_stride, gflags32, xsq, xerr, xmax2, wmax2, werr2, eb, glist,
                                               gcnt, n_groups_all, planP, planXs, need, n_words, (long)blockIdx.x, (int)blockIdx.y,
                                               (int)gridDim.y);
}

}  // namespace somhip
