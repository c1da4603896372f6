// Fused distance + BMU argmin, bf16, for input_len > 128 (BASELINE configs[4]: 784 features).
//
// With more than 128 features a wave can no longer keep its samples' B operands in registers for
// the whole codebook scan, so this kernel is a two-sided GEMM tiling with the same fused argmin
// epilogue instead of a C write:
//   * workgroup tile = 256 samples x 256 units, 8 waves as 2 (sample halves) x 4 (unit quarters), wave
//     tile 128 x 64 (32 v_mfma_f32_16x16x32_bf16 accumulators = 128 VGPRs); 128 x 128 with 4 waves for
//     small maps; K-loop over the features in stages of 32 (one MFMA k-step);
//   * BOTH operands are pre-arranged in HBM in MFMA fragment order, per (block of rows, 32-feature
//     chunk) one contiguous tile [t16][lane 0..63][8 bf16]; the W tiles carry their initial
//     accumulators (B + |w~|^2/2) behind them.  A stage is therefore linear 1 KiB LDS-DMA bursts, and
//     every fragment read is a conflict-free lane-linear ds_read_b128;
//   * 4-slot LDS ring; the two waves of a SIMD run one barrier apart (one in its MFMA segment, the
//     other fetching fragments / issuing DMA / reducing) -- see the schedule comment in the kernel;
//   * arithmetic, offset B, key packing, part split and 64-bit atomicMin merge exactly as in
//     bmu_bf16_k16.hpp (which stays the kernel for input_len <= 128: it reads X once, this one
//     re-reads the sample tile for every unit block, from L2 / Infinity Cache).
#pragma once
#include <type_traits>
#include "bmu_bf16.hpp"

namespace somhip {

constexpr int TL_BK = 32;                      // features per stage = one MFMA k-step
constexpr int TL_KS = TL_BK / 32;
constexpr int TL_SLOTS = 4;                    // ring depth in stages (power of two)
// Tile geometry: a wave owns WS sample blocks of 16 x 64 units; the workgroup is NWR x NWC waves.
//   <4,2,2>: 128 samples x 128 units, 4 waves (small maps / few rows)
//   <8,2,4>: 256 samples x 256 units, 8 waves: twice the flops per staged byte -- the staged operands come
//            from L2 / Infinity Cache, and at 128 x 128 that traffic (12 TB/s at 0.8 PFLOP/s) is the limit.
template <int WS, int NWR, int NWC>
struct TileCfg {
    static constexpr int BM = NWR * WS * 16, BN = NWC * 64, WAVES = NWR * NWC;
    static constexpr int XTILE = (BM / 16) * TL_KS * 1024;               // sample fragments per stage
    static constexpr int WFRAG = (BN / 16) * TL_KS * 1024;               // unit fragments per stage
    static constexpr int WTILE = WFRAG + ((BN * 4 + 1023) / 1024) * 1024; // + BN initial accumulators (padded)
    static constexpr int XPIECES = XTILE / 1024, WPIECES = WTILE / 1024;
    static constexpr int LDS_BYTES = TL_SLOTS * (XTILE + WTILE);
};

// rows (samples or units) -> fragment-ordered tiles.  One thread per 16-byte chunk.
// img layout: [block of `brows` rows][kchunk][t16][ks 0..1][lane][8 bf16]  (tile_bytes per (block,kchunk))
template <class EL = Bf16>
__global__ __launch_bounds__(256) void prep_tiles_bf16_kernel(const float* __restrict__ A, long rows, int D,
                                                              int n_kchunks, long n_blocks, int brows, int tile_bytes,
                                                              float sign, const float* __restrict__ unit_sq,
                                                              char* __restrict__ img,
                                                              const float* __restrict__ scale_max2 = nullptr) {
    using E = typename EL::T;
    using bf16x8 = typename V8<E>::t;
    long id = (long)blockIdx.x * 256 + threadIdx.x;
    const int nt16 = brows / 16;
    const long per_block = (long)n_kchunks * nt16 * TL_KS * 64;
    if (id >= n_blocks * per_block) return;
    long blk = id / per_block;
    int r = (int)(id - blk * per_block);
    int lane = r & 63;
    int t = r >> 6;
    int ks = t % TL_KS; t /= TL_KS;
    int t16 = t % nt16;
    int kc = t / nt16;
    long row = blk * brows + t16 * 16 + (lane & 15);
    int k0 = kc * TL_BK + ks * 32 + (lane >> 4) * 8;
    float scale = sign;
    if (unit_sq != nullptr && row < rows) { float q = unit_sq[row]; scale = q > 0.0f ? sign / __builtin_sqrtf(q) : 0.0f; }
    const float pow2 = scale_max2 != nullptr ? ex_scale(*scale_max2) : 1.0f;   // exact mode: a power of two on top (exact)
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float f = (row < rows && k0 + j < D) ? A[row * D + k0 + j] * scale * pow2 : 0.0f;
        v[j] = cvt<E>(f);
    }
    *(bf16x8*)(img + (blk * n_kchunks + kc) * (long)tile_bytes + ((long)(t16 * TL_KS + ks) * 64 + lane) * 16) = v;
}

// |a~_row|^2 of bf16-rounded (optionally unit-scaled) rows and their maximum.  One wave per row.
template <class EL = Bf16>
__global__ __launch_bounds__(256) void rownorm_bf16_kernel(const float* __restrict__ A, long rows, int D,
                                                           const float* __restrict__ unit_sq, int zero_norm,
                                                           float* __restrict__ norm2, float* __restrict__ max2) {
    using E = typename EL::T;
    long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float scale = 1.0f;
    if (unit_sq != nullptr) { float q = unit_sq[row]; scale = q > 0.0f ? 1.0f / __builtin_sqrtf(q) : 0.0f; }
    float s = 0.0f;
    for (int k = lane; k < D; k += 64) {
        const float f = (float)cvt<E>(A[row * D + k] * scale);
        s = __builtin_fmaf(f, f, s);
    }
    s = wave_sum(s);
    if (lane == 0) {
        if (norm2) norm2[row] = zero_norm ? 0.0f : s;
        atomic_max_pos_f32(max2, s);
    }
}

// initial accumulators B + |w~|^2/2 behind every W tile of a unit block
__global__ __launch_bounds__(256) void prep_tiles_cin_kernel(const float* __restrict__ wn, int K,
                                                             const float* __restrict__ wmax2,
                                                             const float* __restrict__ xmax2, int n_kchunks,
                                                             long n_ublocks, int bn, int wfrag, int wtile,
                                                             char* __restrict__ Wimg) {
    long id = (long)blockIdx.x * 256 + threadIdx.x;
    if (id >= n_ublocks * n_kchunks * bn) return;
    int within = (int)(id % bn);
    long tile = id / bn;                                   // (ublock * n_kchunks + kc)
    long u = (tile / n_kchunks) * bn + within;
    const float big = __builtin_sqrtf(*wmax2) * __builtin_sqrtf(*xmax2) * (1.0f + 1.0f / 1024.0f);
    ((float*)(Wimg + tile * (long)wtile + wfrag))[within] = u < K ? __builtin_fmaf(0.5f, wn[u], big) : BF_PAD_NORM;
}

template <int WS, int NWR, int NWC, class EL = Bf16>
__global__ __launch_bounds__(64 * NWR * NWC, 2) void bmu_bf16_tiled_kernel(const char* __restrict__ Ximg, long N,
                                                                           const char* __restrict__ Wimg,
                                                                           int n_ublocks, int n_kchunks, int K,
                                                                           unsigned long long* __restrict__ out64,
                                                                           int n_sblocks, int n_parts) {
    using E = typename EL::T;
    using C = TileCfg<WS, NWR, NWC>;
    using bf16x8 = typename V8<E>::t;
    static_assert(NWR == 2 && TL_KS == 1, "two wave groups (sample halves), one MFMA k-step per stage");
    constexpr int TL_BM = C::BM, TL_BN = C::BN, TL_TILE = C::XTILE, TL_WFRAG = C::WFRAG, TL_WTILE = C::WTILE;
    constexpr int NG = NWC;                                                // waves per group
    constexpr int XPW = C::XPIECES / NG, WPW = (TL_WFRAG / 1024) / NG;     // 1 KiB DMA pieces per wave and stage
    constexpr int CIN_BYTES = TL_BN * 4 / NG, CIN_LANES = CIN_BYTES / 16;  // the wave's share of the C-in row
    static_assert(XPW * NG * 1024 == TL_TILE && WPW * NG * 1024 == TL_WFRAG && CIN_LANES >= 1 && CIN_LANES <= 64,
                  "stage pieces must divide evenly over a group's waves");
    constexpr int LOADS_X = XPW, LOADS_W = WPW + 1;
    constexpr uint32_t IDX_MASK = 15u;                   // (tile16 << 2 | reg) in the low mantissa bits
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const xring = smem;                            // TL_SLOTS sample tiles
    char* const wring = smem + TL_SLOTS * TL_TILE;       // TL_SLOTS unit tiles (+ their C-in rows)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / NWC, wc = wave % NWC;          // sample half = wave group, unit part
    const int quad = lane >> 4, col = lane & 15;
    // XCD-aware block order (speed only).  Workgroups are dealt round-robin over the 8 XCDs and each
    // XCD has its own L2: give the ~32 workgroups resident on one XCD a (few sample blocks) x (all
    // codebook parts) patch, so a sample tile is shared by n_parts workgroups through that L2 and a unit
    // tile by the workgroups of the patch that run the same part -- instead of every workgroup of the
    // XCD streaming a private sample block.
    const int b = blockIdx.x;
    const int xcd = b & 7, li = b >> 3;
    const int part = li % n_parts;
    const long sblock = (long)(li / n_parts) * 8 + xcd;
    if (sblock >= n_sblocks) return;                      // whole workgroup: no barrier is left behind

    const int ub_begin = (int)((long)n_ublocks * part / n_parts);
    const int ub_end = (int)((long)n_ublocks * (part + 1) / n_parts);
    const long n_stages = (long)(ub_end - ub_begin) * n_kchunks;
    if (n_stages <= 0) return;

    // ---- schedule ---------------------------------------------------------------------------------
    // Stage p = (unit block, 32-feature chunk): a sample tile X(p) and a unit tile W(p), ring slot p & 3.
    // The two waves of a SIMD (wave i and i + NWC: the two sample halves) alternate roles, one barrier
    // apart: while group 0 runs the 32 MFMAs of stage p, group 1 fetches its fragments of stage p from
    // LDS, issues LDS-DMA for stage p + 3 and reduces a finished unit block -- then they swap.  Group 1
    // passes one extra barrier up front (group 0 one at the end), which is all that skews them; every
    // segment is bounded by barriers, so the alternation holds for the whole scan.
    //   interval I(2p)   : group 0 LOAD(p)   | group 1 MFMA(p-1)
    //   interval I(2p+1) : group 0 MFMA(p)   | group 1 LOAD(p)
    // Group 0 streams the sample tiles (L2 hits), group 1 the unit tiles (+ C-in rows).  VMEM retires in
    // order, so a counted wait leaves the two youngest stages of a wave's own stream in flight:
    //   group 0, end of MFMA(p): vmcnt(2 LOADS_X) -> X(p+1) landed, one barrier before anyone reads it;
    //   group 1, end of LOAD(p): vmcnt(2 LOADS_W) -> W(p+1) landed, likewise.
    // A slot is re-filled (stage p+3 -> the slot of stage p-1) only after both groups' reads of it have
    // retired behind an lgkmcnt(0) and a barrier.  Past the last stage the issue pointers stop advancing
    // (the last tile is re-fetched into a slot nobody reads), so the counts stay uniform.
    const char* xsrc = Ximg + sblock * (long)n_kchunks * TL_TILE + lane * 16;
    const char* wsrc = Wimg + (long)ub_begin * n_kchunks * TL_WTILE + lane * 16;
    int kc_i = 0, slot_i = 0;                            // next stage to issue: its chunk (samples) and slot
    long left_i = n_stages;
    auto issue_x = [&]() {
        const char* src = xsrc + (long)kc_i * TL_TILE;
        char* dst = xring + slot_i * TL_TILE;
#pragma unroll
        for (int i = 0; i < XPW; ++i) lds_dma_16(src + (wc + NG * i) * 1024, dst + (wc + NG * i) * 1024);
        if (left_i > 1) { --left_i; if (++kc_i == n_kchunks) kc_i = 0; }
        slot_i = (slot_i + 1) & (TL_SLOTS - 1);
    };
    auto issue_w = [&]() {
        char* dst = wring + slot_i * TL_WTILE;
#pragma unroll
        for (int i = 0; i < WPW; ++i) lds_dma_16(wsrc + (wc + NG * i) * 1024, dst + (wc + NG * i) * 1024);
        if (lane < CIN_LANES) lds_dma_16(wsrc + TL_WFRAG + wc * CIN_BYTES, dst + TL_WFRAG + wc * CIN_BYTES);
        if (left_i > 1) { --left_i; wsrc += TL_WTILE; }
        slot_i = (slot_i + 1) & (TL_SLOTS - 1);
    };

    uint32_t gbest[WS];                                 // unsigned keys: a NaN of either sign never beats a finite d'
    int gblock[WS];
#pragma unroll
    for (int sb = 0; sb < WS; ++sb) { gbest[sb] = 0xFFFFFFFFu; gblock[sb] = 0; }
    f32x4 acc[4][WS];                                    // [unit tile16][sample block16]
    bf16x8 fa[4], fb[WS];                                // fragments of the stage about to be multiplied
    f32x4 cin[4];                                        // C-in rows of the unit block that starts with it

    // per-lane fragment offsets inside a tile (constant over the whole scan)
    const int a_off = (wc * 4 * 64 + lane) * 16;                     // + tu * 1024
    const int b_off = (wr * WS * 64 + lane) * 16;                    // + sb * 1024
    const int c_off = TL_WFRAG + (wc * 64 + 4 * quad) * 4;           // + tu * 64

    auto load_frags = [&](int slot, int kc) {
        const char* xs = xring + slot * TL_TILE;
        const char* ws = wring + slot * TL_WTILE;
#pragma unroll
        for (int tu = 0; tu < 4; ++tu) fa[tu] = *(const bf16x8*)(ws + a_off + tu * 1024);
#pragma unroll
        for (int sb = 0; sb < WS; ++sb) fb[sb] = *(const bf16x8*)(xs + b_off + sb * 1024);
        if (kc == 0) {
#pragma unroll
            for (int tu = 0; tu < 4; ++tu) cin[tu] = *(const f32x4*)(ws + c_off + tu * 64);
        }
    };
    auto reduce_block = [&](int ub) {                    // the 64 x 64 blocks of distances are complete
#pragma unroll
        for (int sb = 0; sb < WS; ++sb) {
            uint32_t c0 = 0xFFFFFFFFu, c1 = 0xFFFFFFFFu;
#pragma unroll
            for (int tu = 0; tu < 4; ++tu) {
                const float f0 = acc[tu][sb][0], f1 = acc[tu][sb][1], f2 = acc[tu][sb][2], f3 = acc[tu][sb][3];
                const uint32_t k0 = (__float_as_uint(f0) & ~IDX_MASK) | (uint32_t)(tu * 4 + 0);
                const uint32_t k1 = (__float_as_uint(f1) & ~IDX_MASK) | (uint32_t)(tu * 4 + 1);
                const uint32_t k2 = (__float_as_uint(f2) & ~IDX_MASK) | (uint32_t)(tu * 4 + 2);
                const uint32_t k3 = (__float_as_uint(f3) & ~IDX_MASK) | (uint32_t)(tu * 4 + 3);
                c0 = min(min(c0, k0), k1);
                c1 = min(min(c1, k2), k3);
            }
            const uint32_t c = min(c0, c1);
            if (c < gbest[sb]) { gbest[sb] = c; gblock[sb] = ub; }
        }
    };

    auto run = [&](auto group_tag) {
        constexpr bool G1 = decltype(group_tag)::value;  // group 1: unit-tile stream, one barrier behind
#pragma unroll
        for (int i = 0; i < TL_SLOTS - 1; ++i) { if (G1) issue_w(); else issue_x(); }
        if (G1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LOADS_W) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LOADS_X) : "memory");
        __builtin_amdgcn_s_barrier();                    // stage 0 is in LDS
        if (G1) __builtin_amdgcn_s_barrier();            // group 1 idles through group 0's LOAD(0)
        asm volatile("" ::: "memory");
        int ub = ub_begin, kc = 0, slot = 0, done_ub = -1;
        for (long p = 0; p < n_stages; ++p) {
            // ---- LOAD(p): fragments to registers, next DMA, reduction of a finished unit block ----
            load_frags(slot, kc);
            if (G1) issue_w(); else issue_x();           // stage p+3 -> the slot stage p-1 was read from
            if (done_ub >= 0) { reduce_block(done_ub); done_ub = -1; }
            if (G1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * LOADS_W) : "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // ---- MFMA(p) ---------------------------------------------------------------------------
            if (kc == 0) {                               // C-in = B + |w~|^2/2 of this wave's 64 units
#pragma unroll
                for (int tu = 0; tu < 4; ++tu)
#pragma unroll
                    for (int sb = 0; sb < WS; ++sb) acc[tu][sb] = cin[tu];
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int tu = 0; tu < 4; ++tu)
#pragma unroll
                for (int sb = 0; sb < WS; ++sb)
                    acc[tu][sb] = mfma16(fa[tu], fb[sb], acc[tu][sb]);
            __builtin_amdgcn_s_setprio(0);
            if (kc == n_kchunks - 1) done_ub = ub;
            if (!G1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LOADS_X) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (++kc == n_kchunks) { kc = 0; ++ub; }
            slot = (slot + 1) & (TL_SLOTS - 1);
        }
        if (done_ub >= 0) reduce_block(done_ub);
        if (!G1) __builtin_amdgcn_s_barrier();           // pairs with group 1's last barrier
    };
    if (wr == 0) run(std::false_type{}); else run(std::true_type{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // no LDS-DMA may land after the workgroup's LDS is released

#pragma unroll
    for (int sb = 0; sb < WS; ++sb) {
        const uint32_t code = gbest[sb] & IDX_MASK;
        const uint32_t unit = (uint32_t)gblock[sb] * TL_BN + wc * 64 + (code >> 2) * 16 + quad * 4 + (code & 3);
        unsigned long long comp = ((unsigned long long)(gbest[sb] & ~IDX_MASK) << 32) | unit;
        unsigned long long o = __shfl_xor(comp, 16, 64);
        if (o < comp) comp = o;
        o = __shfl_xor(comp, 32, 64);
        if (o < comp) comp = o;
        const long row = sblock * TL_BM + wr * (WS * 16) + sb * 16 + col;
        if (quad == 0 && row < N) atomicMin(out64 + row, comp);
    }
}

}  // namespace somhip
