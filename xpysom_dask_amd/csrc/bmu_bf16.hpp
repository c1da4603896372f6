// Fused distance + BMU argmin, bf16 MFMA throughput mode (euclidean).
//
// Same reference chain as bmu_f32.hpp (distances.py:11-23 + xpysom.py:416), computed as
//     d'(n,k) = |x~_n|^2/2 + |w~_k|^2/2 - x~_n . w~_k  =  |x~_n - w~_k|^2 / 2  >= 0
// on bf16-rounded x~, w~ with float32 accumulation in v_mfma_f32_32x32x16_bf16.  Rounding
// BOTH norms from the same bf16 values keeps d' a true distance in the rounded space (its
// error shrinks with the distance, which is what matters for picking the nearest unit) and
// keeps it non-negative up to float32 rounding noise around an exact match, so its bits
// order like SIGNED integers (a slightly negative value is such a match and rightly wins).
//
// Layout in HBM (built once per epoch by prep_w_bf16 / prep_wsqh):
//   the codebook as a sequence of STAGES of 128 units; each stage is one contiguous block
//     [ut=0..3][kstep][lane 0..63][8 bf16]   A-operand fragments of -w~, 1 KiB per (ut,kstep)
//     [128 x f32 |w~|^2/2]  (+ pad to 1 KiB)
//   i.e. exactly the image the kernel wants in LDS, so staging is a linear LDS-DMA copy
//   (global_load_lds_dwordx4) and every ds_read_b128 is lane-linear (conflict free).
//
// Kernel: workgroup = 4 waves; each wave keeps 64 samples' B-operand fragments in
// registers for the whole scan (samples on the MFMA lane/column axis, units on the
// register/row axis), streams the codebook stages through a 2-deep LDS ring (one
// barrier per stage, DMA of stage s+1 in flight under the MFMAs of stage s) and reduces
// each 32x32 tile with 16 v_and_or + 8 v_min3_i32: the low 6 mantissa bits of d' are
// replaced by (ut, register) so one integer min carries value and index together.
#pragma once
#include "som_common.hpp"

namespace somhip {

constexpr int BF_UT = 4;                 // 32-unit tiles per stage
constexpr int BF_STAGE_UNITS = 32 * BF_UT;
constexpr int BF_SBW = 2;                // 32-sample blocks per wave
constexpr int BF_WG_SAMPLES = 4 * 32 * BF_SBW;
constexpr float BF_PAD_NORM = 1.0e30f;                  // |w|^2/2 of padding units: never wins

__host__ __device__ constexpr int bf_stage_bytes(int ksteps) { return (BF_UT * ksteps + 1) * 1024; }

__device__ __forceinline__ void lds_dma_16(const void* gsrc, void* ldst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)ldst, 16, 0, 0);
}

// --------------------------------------------------------------------------------------
// prep: float32 codebook -> bf16 stage image.  One thread per 16-byte fragment chunk.
template <int KSTEPS>
__global__ __launch_bounds__(256) void prep_w_bf16_kernel(const float* __restrict__ W, int K, int D,
                                                          char* __restrict__ Wst, int n_stages) {
    long id = (long)blockIdx.x * 256 + threadIdx.x;
    long total = (long)n_stages * BF_UT * KSTEPS * 64;
    if (id >= total) return;
    int lane = id & 63;
    long t = id >> 6;
    int ks = t % KSTEPS; t /= KSTEPS;
    int ut = t % BF_UT;
    long stage = t / BF_UT;
    long u = stage * BF_STAGE_UNITS + ut * 32 + (lane & 31);
    int k0 = ks * 16 + (lane >> 5) * 8;
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float f = (u < K && k0 + j < D) ? W[u * D + k0 + j] : 0.0f;
        v[j] = (__bf16)(-f);          // fragments carry -w~ so the MFMA accumulates norms - x.w
    }
    *(bf16x8*)(Wst + stage * bf_stage_bytes(KSTEPS) + ((long)(ut * KSTEPS + ks) * 64 + lane) * 16) = v;
}

// |w~|^2/2 (of the bf16-rounded values) per unit, written behind each stage's fragments.
__global__ __launch_bounds__(256) void prep_wsqh_kernel(const float* __restrict__ W, int K, int D,
                                                        char* __restrict__ Wst, int n_stages, int ksteps) {
    long u = (long)blockIdx.x * 256 + threadIdx.x;
    if (u >= (long)n_stages * BF_STAGE_UNITS) return;
    float s = BF_PAD_NORM;
    if (u < K) {
        s = 0.0f;
        for (int k = 0; k < D; ++k) { float f = (float)(__bf16)W[u * D + k]; s = __builtin_fmaf(f, f, s); }
        s = 0.5f * s;
    }
    long stage = u / BF_STAGE_UNITS;
    int within = u % BF_STAGE_UNITS;
    float* dst = (float*)(Wst + stage * bf_stage_bytes(ksteps) + (long)BF_UT * ksteps * 1024);
    dst[within] = s;
}

// samples -> bf16 rows [Np][Dp] (zero padded) and |x~|^2/2 per row.  One wave per row.
__global__ __launch_bounds__(256) void prep_x_bf16_kernel(const float* __restrict__ X, long N, int D, int Dp,
                                                          long Np, __bf16* __restrict__ Xb,
                                                          float* __restrict__ xsqh) {
    long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (row >= Np) return;
    float s = 0.0f;
    for (int k = lane; k < Dp; k += 64) {
        float f = (row < N && k < D) ? X[row * D + k] : 0.0f;
        __bf16 b = (__bf16)f;
        Xb[row * Dp + k] = b;
        float fb = (float)b;
        s = __builtin_fmaf(fb, fb, s);
    }
    s = wave_sum(s);
    if (lane == 0) xsqh[row] = 0.5f * s;
}

// --------------------------------------------------------------------------------------
template <int KSTEPS>
__global__ __launch_bounds__(256, 2) void bmu_bf16_kernel(const __bf16* __restrict__ Xb,
                                                          const float* __restrict__ xsqh, long N,
                                                          const char* __restrict__ Wst, int n_stages, int K,
                                                          int* __restrict__ out) {
    constexpr int DP = 16 * KSTEPS;
    constexpr int STAGE = bf_stage_bytes(KSTEPS);
    constexpr int PIECES = BF_UT * KSTEPS + 1;          // 1-KiB DMA pieces per stage
    constexpr uint32_t IDX_MASK = 63u;                  // (ut<<4 | reg) in the low 6 bits
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, col = lane & 31;
    const long wave_s0 = (long)blockIdx.x * BF_WG_SAMPLES + wave * (32 * BF_SBW);

    // B-operand fragments of this wave's samples: lane holds X~[sample col][16*ks + 8*half + j]
    bf16x8 xf[BF_SBW][KSTEPS];
    float xq[BF_SBW];
#pragma unroll
    for (int sb = 0; sb < BF_SBW; ++sb) {
        const long row = wave_s0 + sb * 32 + col;          // rows are padded to a multiple of the WG size
        xq[sb] = xsqh[row];
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) xf[sb][ks] = *(const bf16x8*)(Xb + row * DP + ks * 16 + half * 8);
    }

    int32_t gbest[BF_SBW];
    int gstage[BF_SBW];
#pragma unroll
    for (int sb = 0; sb < BF_SBW; ++sb) { gbest[sb] = 0x7FFFFFFF; gstage[sb] = 0; }

    // stage 0 -> ring slot 0
    for (int p = wave; p < PIECES; p += 4) lds_dma_16(Wst + (long)p * 1024 + lane * 16, smem + p * 1024);

    for (int s = 0; s < n_stages; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my pieces of stage s have landed
        __builtin_amdgcn_s_barrier();                       // everyone's have; slot (s+1)&1 is free again
        asm volatile("" ::: "memory");                      // keep the LDS reads below the barrier
        if (s + 1 < n_stages) {
            const char* src = Wst + (long)(s + 1) * STAGE;
            char* dst = smem + ((s + 1) & 1) * STAGE;
            for (int p = wave; p < PIECES; p += 4) lds_dma_16(src + (long)p * 1024 + lane * 16, dst + p * 1024);
        }
        const char* st = smem + (s & 1) * STAGE;
        const float* wq = (const float*)(st + BF_UT * KSTEPS * 1024);

        int32_t cbest[BF_SBW];
#pragma unroll
        for (int sb = 0; sb < BF_SBW; ++sb) cbest[sb] = 0x7FFFFFFF;

#pragma unroll
        for (int ut = 0; ut < BF_UT; ++ut) {
            f32x4 wv[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) wv[g] = *(const f32x4*)(wq + ut * 32 + 8 * g + 4 * half);
            bf16x8 a[KSTEPS];
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) a[ks] = *(const bf16x8*)(st + (ut * KSTEPS + ks) * 1024 + lane * 16);
#pragma unroll
            for (int sb = 0; sb < BF_SBW; ++sb) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = wv[r >> 2][r & 3] + xq[sb];
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ++ks)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks], xf[sb][ks], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const float f0 = acc[r], f1 = acc[r + 1];   // (bit_cast straight from a vector element reads lane 0)
                    int32_t k0 = (int32_t)((__float_as_uint(f0) & ~IDX_MASK) | (uint32_t)(ut * 16 + r));
                    int32_t k1 = (int32_t)((__float_as_uint(f1) & ~IDX_MASK) | (uint32_t)(ut * 16 + r + 1));
                    int32_t m = k0 < k1 ? k0 : k1;
                    cbest[sb] = cbest[sb] < m ? cbest[sb] : m;
                }
            }
        }
#pragma unroll
        for (int sb = 0; sb < BF_SBW; ++sb)
            if (cbest[sb] < gbest[sb]) { gbest[sb] = cbest[sb]; gstage[sb] = s; }
    }

#pragma unroll
    for (int sb = 0; sb < BF_SBW; ++sb) {
        uint32_t code = (uint32_t)gbest[sb] & IDX_MASK;
        uint32_t unit = (uint32_t)gstage[sb] * BF_STAGE_UNITS + (code >> 4) * 32 + mfma32_row(code & 15, half);
        long long comp = (long long)(((unsigned long long)((uint32_t)gbest[sb] & ~IDX_MASK) << 32) | unit);
        long long other = __shfl_xor(comp, 32, 64);
        if (other < comp) comp = other;                   // signed: value first, then the lower unit id
        uint32_t u = (uint32_t)comp;
        if (u >= (uint32_t)K) u = 0;                      // only a NaN row can pick a padding unit
        const long row = wave_s0 + sb * 32 + col;
        if (half == 0 && row < N) out[row] = (int)u;
    }
}

}  // namespace somhip
