// Fused distance + BMU argmin, float32 parity mode, register-resident samples (input_len <= 128).
//
// Same arithmetic, bit for bit, as bmu_f32_kernel (bmu_f32.hpp): the cross term is the k-ordered
// fmaf chain of v_mfma_f32_32x32x2_f32 starting from 0, the epilogues restate the reference's
// formulas (distances.py:11-59) literally, '<' keeps the first minimum.  What changes is the
// data movement, borrowed from the bf16 kernel:
//   * a wave keeps its 64 samples' B operands in registers for the whole scan
//     (lane l: sample l&31, feature 2*s + (l>>5) for k-step s);
//   * the codebook is pre-arranged per epoch into stages of 64 units in exactly the LDS image the
//     MFMA wants -- [tile 0..1][group g][lane 0..63][4 floats = the lane's A operand of k-steps
//     4g..4g+3] followed by the 64 |w|^2 -- so a stage is one linear LDS-DMA burst and every
//     fragment read is a conflict-free lane-linear ds_read_b128;
//   * 2-slot ring, one barrier per stage, the next stage's DMA in flight under this stage's MFMAs;
//   * grid = (sample blocks) x (codebook parts), as in the bf16 kernel: few rows (winner(),
//     quantization_error()) spread the scan itself over the chip.  Parts merge through a 64-bit
//     atomicMin of (order-preserving float key << 32 | unit): smaller value first, then the lower unit
//     id -- exactly the '<' / first-minimum rule.  (The top-2 variant runs with one part.)
#pragma once
#include "bmu_bf16.hpp"
#include "bmu_f32.hpp"

namespace somhip {

constexpr int FR_UT = 2;                       // 32-unit tiles per stage
constexpr int FR_STAGE_UNITS = 32 * FR_UT;
constexpr int FR_SBW = 2;                      // 32-sample blocks per wave
constexpr int FR_WG_SAMPLES = 4 * 32 * FR_SBW;

__host__ __device__ constexpr int fr_stage_bytes(int kg) { return (FR_UT * kg + 1) * 1024; }

// float32 codebook -> stage image.  One thread per 16-byte chunk.
__global__ __launch_bounds__(256) void prep_w_f32_res_kernel(const float* __restrict__ W, const float* __restrict__ wsq,
                                                             int K, int D, int kg, char* __restrict__ Wst,
                                                             int n_stages) {
    long id = (long)blockIdx.x * 256 + threadIdx.x;
    const long per_stage = (long)FR_UT * kg * 64 + 64;       // fragment chunks + one thread per |w|^2
    if (id >= (long)n_stages * per_stage) return;
    long stage = id / per_stage;
    int r = (int)(id - stage * per_stage);
    char* base = Wst + stage * fr_stage_bytes(kg);
    if (r >= FR_UT * kg * 64) {
        int within = r - FR_UT * kg * 64;
        long u = stage * FR_STAGE_UNITS + within;
        ((float*)(base + (long)FR_UT * kg * 1024))[within] = u < K ? wsq[u] : __builtin_inff();
        return;
    }
    int lane = r & 63;
    int t = r >> 6;
    int g = t % kg, ut = t / kg;
    long u = stage * FR_STAGE_UNITS + ut * 32 + (lane & 31);
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int k = 8 * g + 2 * j + (lane >> 5);
        v[j] = (u < K && k < D) ? W[u * D + k] : 0.0f;
    }
    *(f32x4*)(base + ((long)(ut * kg + g) * 64 + lane) * 16) = v;
}

// TOP2: also report the second-best unit (out2) -- the pair XPySom.topographic_error needs
// (xpysom.py:709-746 takes it from a full argsort of the (n, K) distance matrix).
template <int MODE, int KG, bool TOP2 = false>
__global__ __launch_bounds__(256, 2) void bmu_f32_res_kernel(const float* __restrict__ X, long N, int D,
                                                             const float* __restrict__ xsq,
                                                             const char* __restrict__ Wst, int n_stages, int K,
                                                             int* __restrict__ out, int* __restrict__ out2,
                                                             unsigned long long* __restrict__ out64) {
    constexpr int STAGE = fr_stage_bytes(KG);
    constexpr int PIECES = FR_UT * KG + 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, col = lane & 31;
    const long wave_s0 = (long)blockIdx.x * FR_WG_SAMPLES + wave * (32 * FR_SBW);

    // B operands: xf[sb][s] = X[sample][2*s + half]
    float xf[FR_SBW][4 * KG];
    float xs[FR_SBW];
#pragma unroll
    for (int sb = 0; sb < FR_SBW; ++sb) {
        const long row = wave_s0 + sb * 32 + col;
        const bool live = row < N;
        xs[sb] = (MODE != SCORE_EUCLID_PART && live) ? xsq[row] : 0.0f;
        if ((D & 3) == 0) {                               // rows are 16-byte aligned: vector loads
#pragma unroll
            for (int c = 0; c < 2 * KG; ++c) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (live && 4 * c < D) v = *(const f32x4*)(X + row * D + 4 * c);
                xf[sb][2 * c] = half ? v[1] : v[0];
                xf[sb][2 * c + 1] = half ? v[3] : v[2];
            }
        } else {
#pragma unroll
            for (int s = 0; s < 4 * KG; ++s) {
                const int k = 2 * s + half;
                xf[sb][s] = (live && k < D) ? X[row * D + k] : 0.0f;
            }
        }
    }

    float best[FR_SBW], sec[FR_SBW];
    int bidx[FR_SBW], sidx[FR_SBW];
#pragma unroll
    for (int sb = 0; sb < FR_SBW; ++sb) { best[sb] = sec[sb] = __builtin_inff(); bidx[sb] = sidx[sb] = 0; }

    const int s_begin = (int)((long)n_stages * blockIdx.y / gridDim.y);
    const int s_end = (int)((long)n_stages * (blockIdx.y + 1) / gridDim.y);
    for (int p = wave; p < PIECES; p += 4)
        lds_dma_16(Wst + (long)s_begin * STAGE + (long)p * 1024 + lane * 16, smem + p * 1024);

    for (int s = s_begin; s < s_end; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + 1 < s_end) {
            const char* src = Wst + (long)(s + 1) * STAGE;
            char* dst = smem + ((s + 1 - s_begin) & 1) * STAGE;
            for (int p = wave; p < PIECES; p += 4) lds_dma_16(src + (long)p * 1024 + lane * 16, dst + p * 1024);
        }
        const char* st = smem + ((s - s_begin) & 1) * STAGE;
        const float* wq = (const float*)(st + FR_UT * KG * 1024);

#pragma unroll
        for (int ut = 0; ut < FR_UT; ++ut) {
            f32x16 acc[FR_SBW];
#pragma unroll
            for (int sb = 0; sb < FR_SBW; ++sb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[sb][r] = 0.0f;
#pragma unroll
            for (int g = 0; g < KG; ++g) {
                const f32x4 a4 = *(const f32x4*)(st + ((ut * KG + g) * 64 + lane) * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int sb = 0; sb < FR_SBW; ++sb)
                        acc[sb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j], xf[sb][4 * g + j], acc[sb], 0, 0, 0);
            }
            f32x4 wv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) wv[q] = *(const f32x4*)(wq + ut * 32 + 8 * q + 4 * half);
            if (!TOP2) {                                 // bidx holds f32_tile_argmin keys until the end
                const int tile = s * FR_UT + ut;
                if ((tile + 1) * 32 > K) {
#pragma unroll
                    for (int sb = 0; sb < FR_SBW; ++sb)
                        f32_tile_argmin<MODE, true>(acc[sb], wv, xs[sb], tile, half, K, best[sb], bidx[sb]);
                } else {
#pragma unroll
                    for (int sb = 0; sb < FR_SBW; ++sb)
                        f32_tile_argmin<MODE, false>(acc[sb], wv, xs[sb], tile, half, K, best[sb], bidx[sb]);
                }
                continue;
            }
#pragma unroll
            for (int sb = 0; sb < FR_SBW; ++sb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {           // units ascend with r for a fixed lane half
                    const int u = s * FR_STAGE_UNITS + ut * 32 + mfma32_row(r, half);
                    const float v = score_f32<MODE>(acc[sb][r], wv[r >> 2][r & 3], xs[sb]);
                    if (u < K && v < sec[sb]) {
                        if (v < best[sb]) { sec[sb] = best[sb]; sidx[sb] = bidx[sb]; best[sb] = v; bidx[sb] = u; }
                        else { sec[sb] = v; sidx[sb] = u; }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int sb = 0; sb < FR_SBW; ++sb) {
        if (!TOP2) bidx[sb] = f32_key_unit(bidx[sb], half);
        float ob = __shfl_xor(best[sb], 32, 64);
        int oi = __shfl_xor(bidx[sb], 32, 64);
        const long row = wave_s0 + sb * 32 + col;
        if (TOP2) {   // merge two sorted pairs (value, then lower id): keep the two smallest
            float os = __shfl_xor(sec[sb], 32, 64);
            int osi = __shfl_xor(sidx[sb], 32, 64);
            const bool other_first = ob < best[sb] || (ob == best[sb] && oi < bidx[sb]);
            float b1 = other_first ? ob : best[sb];  int i1 = other_first ? oi : bidx[sb];
            float c1 = other_first ? best[sb] : ob;  int j1 = other_first ? bidx[sb] : oi;   // loser of the firsts
            float c2 = other_first ? os : sec[sb];   int j2 = other_first ? osi : sidx[sb];  // winner's own second
            const bool take_c1 = c1 < c2 || (c1 == c2 && j1 < j2);
            if (half == 0 && row < N) { out[row] = i1; out2[row] = take_c1 ? j1 : j2; }
            (void)b1;
        } else {
            if (ob < best[sb] || (ob == best[sb] && oi < bidx[sb])) { best[sb] = ob; bidx[sb] = oi; }
            if (half == 0 && row < N) {
                if (out64 == nullptr) out[row] = bidx[sb];
                else {   // order-preserving key: negative floats reversed, positive above them
                    const uint32_t bits = __float_as_uint(best[sb]);
                    const uint32_t key = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
                    atomicMin(out64 + row, ((unsigned long long)key << 32) | (uint32_t)bidx[sb]);
                }
            }
        }
    }
}

}  // namespace somhip
