// Fused distance + BMU argmin, bf16, for input_len > 128 (BASELINE configs[4]: 784 features).
//
// With more than 128 features a wave can no longer keep its samples' B operands in registers for
// the whole codebook scan, so this kernel is a classic two-sided GEMM tiling with the same fused
// argmin epilogue instead of a C write:
//   * workgroup tile = 128 samples x 128 units, 4 waves as 2 x 2, wave tile 64 x 64
//     (16 v_mfma_f32_16x16x32_bf16 accumulators = 64 VGPRs), K-loop over the features in chunks of 64;
//   * BOTH operands are pre-arranged in HBM in MFMA fragment order, per (128-row block, 64-feature
//     chunk) one contiguous 16 KiB tile [t16 0..7][kstep 0..1][lane 0..63][8 bf16]; the W tiles carry
//     their 128 initial accumulators (B + |w~|^2/2) behind them.  A stage (X tile + W tile, 33 KiB)
//     is therefore two linear LDS-DMA bursts, and every fragment read is a lane-linear ds_read_b128;
//   * 2-slot LDS ring, one barrier per stage, next stage in flight under the 32 MFMAs per wave;
//   * arithmetic, offset B, key packing, part split and 64-bit atomicMin merge exactly as in
//     bmu_bf16_k16.hpp (which stays the kernel for input_len <= 128: it reads X once, this one
//     re-reads the sample tile for every 128-unit block, from L2 / Infinity Cache).
#pragma once
#include "bmu_bf16.hpp"

namespace somhip {

constexpr int TL_BK = 64;
// Tile geometry: a wave owns WS sample blocks of 16 x 64 units; the workgroup is NWR x NWC waves.
//   <4,2,2>: 128 samples x 128 units, 4 waves (small maps / few rows)
//   <8,2,4>: 256 samples x 256 units, 8 waves: twice the flops per staged byte -- the staged operands come
//            from L2 / Infinity Cache, and at 128 x 128 that traffic (12 TB/s at 0.8 PFLOP/s) is the limit.
template <int WS, int NWR, int NWC>
struct TileCfg {
    static constexpr int BM = NWR * WS * 16, BN = NWC * 64, WAVES = NWR * NWC;
    static constexpr int XTILE = (BM / 16) * (TL_BK / 32) * 1024;        // sample fragments per stage
    static constexpr int WFRAG = (BN / 16) * (TL_BK / 32) * 1024;        // unit fragments per stage
    static constexpr int WTILE = WFRAG + ((BN * 4 + 1023) / 1024) * 1024; // + BN initial accumulators (padded)
    static constexpr int STAGE = XTILE + WTILE;
    static constexpr int XPIECES = XTILE / 1024, WPIECES = WTILE / 1024;
};

// rows (samples or units) -> fragment-ordered tiles.  One thread per 16-byte chunk.
// img layout: [block of `brows` rows][kchunk][t16][ks 0..1][lane][8 bf16]  (tile_bytes per (block,kchunk))
//
// split (precision 'bf16x3'): every value is split into hi = bf16(v), lo = bf16(v - hi) and the
// feature axis is tripled -- samples carry [hi | hi | lo], units [hi | lo | hi] -- so the same MFMA
// contraction yields x_hi.w_hi + x_hi.w_lo + x_lo.w_hi, i.e. x.w to ~2^-16 relative (only lo.lo is
// dropped) at three times the bf16 work.  split: 0 = plain bf16, 1 = sample pattern, 2 = unit pattern.
__global__ __launch_bounds__(256) void prep_tiles_bf16_kernel(const float* __restrict__ A, long rows, int D,
                                                              int n_kchunks, long n_blocks, int brows, int tile_bytes,
                                                              float sign, const float* __restrict__ unit_sq,
                                                              char* __restrict__ img, int split) {
    long id = (long)blockIdx.x * 256 + threadIdx.x;
    const int nt16 = brows / 16;
    const long per_block = (long)n_kchunks * nt16 * 2 * 64;
    if (id >= n_blocks * per_block) return;
    long blk = id / per_block;
    int r = (int)(id - blk * per_block);
    int lane = r & 63;
    int t = r >> 6;
    int ks = t & 1; t >>= 1;
    int t16 = t % nt16;
    int kc = t / nt16;
    long row = blk * brows + t16 * 16 + (lane & 15);
    int k0 = kc * TL_BK + ks * 32 + (lane >> 4) * 8;
    float scale = sign;
    if (unit_sq != nullptr && row < rows) { float q = unit_sq[row]; scale = q > 0.0f ? sign / __builtin_sqrtf(q) : 0.0f; }
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (split == 0) {
            float f = (row < rows && k0 + j < D) ? A[row * D + k0 + j] * scale : 0.0f;
            v[j] = (__bf16)f;
        } else {
            const int kv = k0 + j, seg = kv / D, k = kv - seg * D;
            float f = (row < rows && seg < 3) ? A[row * D + k] * scale : 0.0f;
            const __bf16 hi = (__bf16)f;
            const bool want_lo = split == 1 ? seg == 2 : seg == 1;
            v[j] = want_lo ? (__bf16)(f - (float)hi) : hi;
        }
    }
    *(bf16x8*)(img + (blk * n_kchunks + kc) * (long)tile_bytes + ((long)(t16 * 2 + ks) * 64 + lane) * 16) = v;
}

// |a~_row|^2 of bf16-rounded (optionally unit-scaled) rows and their maximum.  One wave per row.
// exact != 0 ('bf16x3'): the float32 rows themselves, not their bf16 roundings.
__global__ __launch_bounds__(256) void rownorm_bf16_kernel(const float* __restrict__ A, long rows, int D,
                                                           const float* __restrict__ unit_sq, int zero_norm,
                                                           float* __restrict__ norm2, float* __restrict__ max2,
                                                           int exact) {
    long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float scale = 1.0f;
    if (unit_sq != nullptr) { float q = unit_sq[row]; scale = q > 0.0f ? 1.0f / __builtin_sqrtf(q) : 0.0f; }
    float s = 0.0f;
    for (int k = lane; k < D; k += 64) {
        float f = A[row * D + k] * scale;
        if (!exact) f = (float)(__bf16)f;
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

template <int WS, int NWR, int NWC>
__global__ __launch_bounds__(64 * NWR * NWC, 2) void bmu_bf16_tiled_kernel(const char* __restrict__ Ximg, long N,
                                                                           const char* __restrict__ Wimg,
                                                                           int n_ublocks, int n_kchunks, int K,
                                                                           unsigned long long* __restrict__ out64,
                                                                           int n_sblocks, int n_parts) {
    using C = TileCfg<WS, NWR, NWC>;
    constexpr int TL_BM = C::BM, TL_BN = C::BN, TL_TILE = C::XTILE, TL_WFRAG = C::WFRAG, TL_WTILE = C::WTILE;
    constexpr int TL_STAGE = C::STAGE, TL_XPIECES = C::XPIECES, TL_WPIECES = C::WPIECES, NW = C::WAVES;
    constexpr uint32_t IDX_MASK = 15u;                   // (tile16 << 2 | reg) in the low mantissa bits
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / NWC, wc = wave % NWC;          // sample part, unit part
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

    // Stage pointers advance incrementally (no division in the loop): the sample tile of k-chunk kc
    // and the unit tile of (ublock, kc).  Stages are issued one ahead of their use.
    const char* xbase = Ximg + sblock * (long)n_kchunks * TL_TILE;
    const char* wnext = Wimg + (long)ub_begin * n_kchunks * TL_WTILE;   // W tile of the next stage to issue
    int kc_issue = 0;
    const int lane16 = lane * 16;
    auto issue = [&](int slot) {
        const char* xs = xbase + (long)kc_issue * TL_TILE + lane16;
        const char* ws = wnext + lane16;
        char* dst = smem + slot * TL_STAGE;
#pragma unroll
        for (int i = 0; i < (TL_XPIECES + TL_WPIECES + NW - 1) / NW; ++i) {
            const int p = wave + NW * i;
            if (p < TL_XPIECES) lds_dma_16(xs + p * 1024, dst + p * 1024);
            else if (p < TL_XPIECES + TL_WPIECES) lds_dma_16(ws + (p - TL_XPIECES) * 1024, dst + p * 1024);
        }
        wnext += TL_WTILE;
        if (++kc_issue == n_kchunks) kc_issue = 0;
    };

    int32_t gbest[WS];
    int gblock[WS];
#pragma unroll
    for (int sb = 0; sb < WS; ++sb) { gbest[sb] = 0x7FFFFFFF; gblock[sb] = 0; }
    f32x4 acc[4][WS];                                    // [unit tile16][sample block16]

    // per-lane fragment offsets inside a stage (constant over the whole scan)
    const int a_off = TL_TILE + (wc * 4 * 2 * 64 + lane) * 16;         // + (tu*2 + ks) * 1024
    const int b_off = (wr * WS * 2 * 64 + lane) * 16;                  // + (sb*2 + ks) * 1024
    const int c_off = TL_TILE + TL_WFRAG + (wc * 64 + 4 * quad) * 4;   // + tu * 64

    auto compute = [&](const char* st, int ub, int kc) {
        if (kc == 0) {                                   // C-in = B + |w~|^2/2 of this wave's 64 units
#pragma unroll
            for (int tu = 0; tu < 4; ++tu) {
                const f32x4 wv = *(const f32x4*)(st + c_off + tu * 64);
#pragma unroll
                for (int sb = 0; sb < WS; ++sb) acc[tu][sb] = wv;
            }
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[4], b[WS];
#pragma unroll
            for (int tu = 0; tu < 4; ++tu) a[tu] = *(const bf16x8*)(st + a_off + (tu * 2 + ks) * 1024);
#pragma unroll
            for (int sb = 0; sb < WS; ++sb) b[sb] = *(const bf16x8*)(st + b_off + (sb * 2 + ks) * 1024);
#pragma unroll
            for (int tu = 0; tu < 4; ++tu)
#pragma unroll
                for (int sb = 0; sb < WS; ++sb)
                    acc[tu][sb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[tu], b[sb], acc[tu][sb], 0, 0, 0);
        }
        if (kc == n_kchunks - 1) {                       // the 64 x 64 block of distances is complete: reduce it
#pragma unroll
            for (int sb = 0; sb < WS; ++sb) {
                int32_t c0 = 0x7FFFFFFF, c1 = 0x7FFFFFFF;
#pragma unroll
                for (int tu = 0; tu < 4; ++tu) {
                    const float f0 = acc[tu][sb][0], f1 = acc[tu][sb][1], f2 = acc[tu][sb][2], f3 = acc[tu][sb][3];
                    const int32_t k0 = (int32_t)((__float_as_uint(f0) & ~IDX_MASK) | (uint32_t)(tu * 4 + 0));
                    const int32_t k1 = (int32_t)((__float_as_uint(f1) & ~IDX_MASK) | (uint32_t)(tu * 4 + 1));
                    const int32_t k2 = (int32_t)((__float_as_uint(f2) & ~IDX_MASK) | (uint32_t)(tu * 4 + 2));
                    const int32_t k3 = (int32_t)((__float_as_uint(f3) & ~IDX_MASK) | (uint32_t)(tu * 4 + 3));
                    c0 = min(min(c0, k0), k1);
                    c1 = min(min(c1, k2), k3);
                }
                const int32_t c = min(c0, c1);
                if (c < gbest[sb]) { gbest[sb] = c; gblock[sb] = ub; }
            }
        }
    };

    if (n_stages > 0) issue(0);
    int ub = ub_begin, kc = 0;
    for (long q = 0; q < n_stages; q += 2) {             // two stages per trip: ring slots are compile-time
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (q + 1 < n_stages) issue(1);
        compute(smem, ub, kc);
        if (++kc == n_kchunks) { kc = 0; ++ub; }
        if (q + 1 >= n_stages) break;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (q + 2 < n_stages) issue(0);
        compute(smem + TL_STAGE, ub, kc);
        if (++kc == n_kchunks) { kc = 0; ++ub; }
    }

#pragma unroll
    for (int sb = 0; sb < WS; ++sb) {
        const uint32_t code = (uint32_t)gbest[sb] & IDX_MASK;
        const uint32_t unit = (uint32_t)gblock[sb] * TL_BN + wc * 64 + (code >> 2) * 16 + quad * 4 + (code & 3);
        unsigned long long comp = ((unsigned long long)((uint32_t)gbest[sb] & ~IDX_MASK) << 32) | unit;
        unsigned long long o = __shfl_xor(comp, 16, 64);
        if (o < comp) comp = o;
        o = __shfl_xor(comp, 32, 64);
        if (o < comp) comp = o;
        const long row = sblock * TL_BM + wr * (WS * 16) + sb * 16 + col;
        if (quad == 0 && row < N && n_stages > 0) atomicMin(out64 + row, comp);
    }
}

}  // namespace somhip
