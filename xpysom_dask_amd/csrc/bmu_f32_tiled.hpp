// Fused distance + BMU argmin, float32 parity mode, for input_len > 128.
//
// The float32 counterpart of bmu_bf16_tiled.hpp: two-sided tiling (128 samples x 128 units per
// workgroup, 4 waves as 2 x 2, wave tile 64 x 64 = four v_mfma_f32_32x32x2_f32 accumulators), K-loop
// over the features in chunks of 32, both operands pre-arranged per (128-row block, 32-feature chunk)
// as [t32 0..3][group 0..3][lane 0..63][4 floats = the lane's operand of k-steps 4g..4g+3], so a stage
// is two linear LDS-DMA bursts and every fragment read a lane-linear ds_read_b128.
// Arithmetic is the same k-ordered fmaf chain from 0 as bmu_f32_kernel / bmu_f32_res_kernel (the
// accumulators simply live across the chunks), the epilogues are score_f32<MODE>, '<' keeps the
// first minimum, so BMUs stay bit-identical to those kernels.  Replaces the LDS-chunked
// bmu_f32_kernel on this shape range (12 -> ~100 TFLOP/s at D = 784).
#pragma once
#include "bmu_bf16.hpp"
#include "bmu_f32.hpp"

namespace somhip {

constexpr int FT_BM = 128, FT_BN = 128, FT_BK = 32;
constexpr int FT_TILE = (FT_BM / 32) * (FT_BK / 8) * 1024;      // 16 KiB of fragments
constexpr int FT_WTILE = FT_TILE + 1024;                        // + 128 |w|^2 (padded)
constexpr int FT_STAGE = FT_TILE + FT_WTILE;
constexpr int FT_XPIECES = FT_TILE / 1024, FT_WPIECES = FT_WTILE / 1024;

// rows -> fragment tiles; one thread per 16-byte chunk.  tail != nullptr: the 128 per-row values
// (|w|^2, +inf for padding rows) are written behind every tile of the block.
__global__ __launch_bounds__(256) void prep_tiles_f32_kernel(const float* __restrict__ A, long rows, int D,
                                                             int n_kchunks, long n_blocks, int tile_bytes,
                                                             const float* __restrict__ tail,
                                                             char* __restrict__ img) {
    long id = (long)blockIdx.x * 256 + threadIdx.x;
    const long per_tile = 4 * 4 * 64 + (tail ? 128 : 0);
    const long per_block = (long)n_kchunks * per_tile;
    if (id >= n_blocks * per_block) return;
    long blk = id / per_block;
    int r = (int)(id - blk * per_block);
    int kc = r / (int)per_tile;
    r -= kc * (int)per_tile;
    char* base = img + (blk * n_kchunks + kc) * (long)tile_bytes;
    if (r >= 4 * 4 * 64) {
        int within = r - 4 * 4 * 64;
        long u = blk * 128 + within;
        ((float*)(base + FT_TILE))[within] = u < rows ? tail[u] : __builtin_inff();
        return;
    }
    int lane = r & 63;
    int t = r >> 6;
    int g = t & 3, t32 = t >> 2;
    long row = blk * 128 + t32 * 32 + (lane & 31);
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int k = kc * FT_BK + 8 * g + 2 * j + (lane >> 5);
        v[j] = (row < rows && k < D) ? A[row * D + k] : 0.0f;
    }
    *(f32x4*)(base + ((long)(t32 * 4 + g) * 64 + lane) * 16) = v;
}

template <int MODE, bool TOP2>
__global__ __launch_bounds__(256, 2) void bmu_f32_tiled_kernel(const char* __restrict__ Ximg, long N,
                                                               const float* __restrict__ xsq,
                                                               const char* __restrict__ Wimg, int n_ublocks,
                                                               int n_kchunks, int K, int* __restrict__ out,
                                                               int* __restrict__ out2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;             // sample half, unit half
    const int half = lane >> 5, col = lane & 31;
    const long sblock = blockIdx.x;
    const long n_stages = (long)n_ublocks * n_kchunks;

    const char* xbase = Ximg + sblock * (long)n_kchunks * FT_TILE;
    const char* wnext = Wimg;
    int kc_issue = 0;
    const int lane16 = lane * 16;
    auto issue = [&](int slot) {
        const char* xs = xbase + (long)kc_issue * FT_TILE + lane16;
        const char* ws = wnext + lane16;
        char* dst = smem + slot * FT_STAGE;
#pragma unroll
        for (int i = 0; i < (FT_XPIECES + FT_WPIECES + 3) / 4; ++i) {
            const int p = wave + 4 * i;
            if (p < FT_XPIECES) lds_dma_16(xs + p * 1024, dst + p * 1024);
            else if (p < FT_XPIECES + FT_WPIECES) lds_dma_16(ws + (p - FT_XPIECES) * 1024, dst + p * 1024);
        }
        wnext += FT_WTILE;
        if (++kc_issue == n_kchunks) kc_issue = 0;
    };

    float xs[2];
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
        const long row = sblock * FT_BM + wr * 64 + sb * 32 + col;
        xs[sb] = (MODE != SCORE_EUCLID_PART && row < N) ? xsq[row] : 0.0f;
    }
    float best[2], sec[2];
    int bidx[2], sidx[2];
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) { best[sb] = sec[sb] = __builtin_inff(); bidx[sb] = sidx[sb] = 0; }
    f32x16 acc[2][2];                                    // [unit tile32][sample block32]

    const int a_off = FT_TILE + (wc * 2 * 4 * 64 + lane) * 16;        // + (tu*4 + g) * 1024
    const int b_off = (wr * 2 * 4 * 64 + lane) * 16;                  // + (sb*4 + g) * 1024
    const int c_off = FT_TILE + FT_TILE + (wc * 64 + 4 * half) * 4;   // + (tu*32 + 8*q) * 4

    auto compute = [&](const char* st, int ub, int kc) {
        if (kc == 0) {
#pragma unroll
            for (int tu = 0; tu < 2; ++tu)
#pragma unroll
                for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[tu][sb][r] = 0.0f;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 a[2], b[2];
#pragma unroll
            for (int tu = 0; tu < 2; ++tu) a[tu] = *(const f32x4*)(st + a_off + (tu * 4 + g) * 1024);
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) b[sb] = *(const f32x4*)(st + b_off + (sb * 4 + g) * 1024);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int tu = 0; tu < 2; ++tu)
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb)
                        acc[tu][sb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tu][j], b[sb][j], acc[tu][sb], 0, 0, 0);
        }
        if (kc == n_kchunks - 1) {
#pragma unroll
            for (int tu = 0; tu < 2; ++tu) {
                f32x4 wv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) wv[q] = *(const f32x4*)(st + c_off + (tu * 32 + 8 * q) * 4);
                if (!TOP2) {                             // bidx holds f32_tile_argmin keys until the end
                    const int tile = ub * (FT_BN / 32) + wc * 2 + tu;
                    if ((tile + 1) * 32 > K) {
#pragma unroll
                        for (int sb = 0; sb < 2; ++sb)
                            f32_tile_argmin<MODE, true>(acc[tu][sb], wv, xs[sb], tile, half, K, best[sb], bidx[sb]);
                    } else {
#pragma unroll
                        for (int sb = 0; sb < 2; ++sb)
                            f32_tile_argmin<MODE, false>(acc[tu][sb], wv, xs[sb], tile, half, K, best[sb], bidx[sb]);
                    }
                    continue;
                }
#pragma unroll
                for (int sb = 0; sb < 2; ++sb) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {       // units ascend with (ub, tu, r) for a fixed lane half
                        const int u = ub * FT_BN + wc * 64 + tu * 32 + mfma32_row(r, half);
                        const float v = score_f32<MODE>(acc[tu][sb][r], wv[r >> 2][r & 3], xs[sb]);
                        if (u < K && v < sec[sb]) {
                            if (v < best[sb]) { sec[sb] = best[sb]; sidx[sb] = bidx[sb]; best[sb] = v; bidx[sb] = u; }
                            else { sec[sb] = v; sidx[sb] = u; }
                        }
                    }
                }
            }
        }
    };

    if (n_stages > 0) issue(0);
    int ub = 0, kc = 0;
    for (long q = 0; q < n_stages; q += 2) {
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
        compute(smem + FT_STAGE, ub, kc);
        if (++kc == n_kchunks) { kc = 0; ++ub; }
    }

    // merge (value, then lower id): first the two lane halves, then the two unit-half waves via LDS
    auto merge2 = [](float& b1, int& i1, float& s1, int& j1, float ob, int oi, float os, int oj) {
        const bool other_first = ob < b1 || (ob == b1 && oi < i1);
        const float c1 = other_first ? b1 : ob;  const int k1 = other_first ? i1 : oi;   // loser of the firsts
        const float c2 = other_first ? os : s1;  const int k2 = other_first ? oj : j1;   // winner's own second
        if (other_first) { b1 = ob; i1 = oi; }
        const bool take_c1 = c1 < c2 || (c1 == c2 && k1 < k2);
        s1 = take_c1 ? c1 : c2;
        j1 = take_c1 ? k1 : k2;
    };
    if (!TOP2) {
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) bidx[sb] = f32_key_unit(bidx[sb], half);
    }
    __syncthreads();                                     // the ring is no longer read
    float* sv = (float*)smem;                            // [wr][sb][col]{best, sec}, then ids
    int* si = (int*)(smem + 2 * 2 * 32 * 2 * sizeof(float));
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
        merge2(best[sb], bidx[sb], sec[sb], sidx[sb], __shfl_xor(best[sb], 32, 64), __shfl_xor(bidx[sb], 32, 64),
               __shfl_xor(sec[sb], 32, 64), __shfl_xor(sidx[sb], 32, 64));
        const int slot = ((wr * 2 + sb) * 32 + col) * 2;
        if (wc == 1 && half == 0) { sv[slot] = best[sb]; sv[slot + 1] = sec[sb]; si[slot] = bidx[sb]; si[slot + 1] = sidx[sb]; }
    }
    __syncthreads();
    if (wc == 0 && half == 0) {
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
            const int slot = ((wr * 2 + sb) * 32 + col) * 2;
            merge2(best[sb], bidx[sb], sec[sb], sidx[sb], sv[slot], si[slot], sv[slot + 1], si[slot + 1]);
            const long row = sblock * FT_BM + wr * 64 + sb * 32 + col;
            if (row < N) {
                out[row] = bidx[sb];
                if (TOP2) out2[row] = sidx[sb];
            }
        }
    }
}

}  // namespace somhip
