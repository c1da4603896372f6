// precision 'bf16x3' for input_len <= 128: the register-resident kernel of bmu_bf16_k16.hpp on hi/lo-split operands.
//
// Every value is split into hi = bf16(v), lo = bf16(v - hi); the kernel contracts
//     x_hi . w_hi  +  x_lo . w_hi  +  x_hi . w_lo                 (only lo . lo, ~2^-18 relative, is dropped)
// on v_mfma_f32_16x16x32_bf16 with the exact float32 |w|^2 / 2 + B as the initial accumulator -- the arithmetic of the
// tiled kernel's split mode (bmu_bf16_tiled.hpp), which stays the bf16x3 kernel for input_len > 128.  What changes is
// the data movement: a wave keeps the hi AND lo fragments of 2 x 16 samples in registers for the whole scan, so a stage
// of 32 units (hi and lo fragments + 32 initial accumulators = 17 KiB) feeds 48 MFMAs per wave and the argmin costs
// 0.5 vector instructions per MFMA (one key pack per THREE MFMAs).  The tiled kernel on the tripled feature axis
// stages 33 KiB per 32 MFMAs per wave and is bound by issuing those LDS-DMA pieces (0.17 of the bf16 peak,
// algorithmic); this form streams a third of the bytes per MFMA.
//
// Geometry, key packing, part split and the 64-bit atomicMin merge as in bmu_bf16_k16.hpp.
// Stage image (K3_STAGE_UNITS = 32 units): [t16 0..1][kstep32][lane][8 bf16 of -w_hi] | the same of -w_lo |
// [32 x f32 B + |w|^2/2] (+ pad to 1 KiB).  Row image: [row][hi: DP bf16 | lo: DP bf16].
#pragma once
#include "bmu_bf16_k16.hpp"

namespace somhip {

constexpr int K3_T = 2;               // 16-unit tiles per stage
constexpr int K3_STAGE_UNITS = 16 * K3_T;
constexpr int K3_SB = 2;              // 16-sample blocks per wave (hi + lo fragments: 64 VGPRs at 128 features)
constexpr int K3_NW = 4;              // waves per workgroup
constexpr int K3_WG_SAMPLES = K3_NW * 16 * K3_SB;

__host__ __device__ constexpr int k3_stage_bytes(int ks32) { return (2 * K3_T * ks32 + 1) * 1024; }

// float32 codebook -> hi / lo stage image.  One thread per 16-byte fragment chunk; unit_wsq != nullptr (cosine): rows
// scaled to unit length first.
template <int KS32, class EL = Bf16>
__global__ __launch_bounds__(256) void prep_w_bf16_k16x3_kernel(const float* __restrict__ W, int K, int D,
                                                                char* __restrict__ Wst, int n_stages,
                                                                const float* __restrict__ unit_wsq) {
    using E = typename EL::T;
    using bf16x8 = typename V8<E>::t;
    long id = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)n_stages * 2 * K3_T * KS32 * 64;
    if (id >= total) return;
    const int lane = id & 63;
    long t = id >> 6;
    const int ks = t % KS32; t /= KS32;
    const int t16 = t % K3_T; t /= K3_T;
    const int part = t % 2;                       // 0 = hi, 1 = lo
    const long stage = t / 2;
    const long u = stage * K3_STAGE_UNITS + t16 * 16 + (lane & 15);
    const int k0 = ks * 32 + (lane >> 4) * 8;
    float scale = 1.0f;
    if (unit_wsq != nullptr && u < K) { float q = unit_wsq[u]; scale = q > 0.0f ? 1.0f / __builtin_sqrtf(q) : 0.0f; }
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float f = (u < K && k0 + j < D) ? -(W[u * D + k0 + j] * scale) : 0.0f;
        const E hi = cvt<E>(f);
        v[j] = part == 0 ? hi : cvt<E>(f - (float)hi);
    }
    *(bf16x8*)(Wst + stage * k3_stage_bytes(KS32) + ((long)((part * K3_T + t16) * KS32 + ks) * 64 + lane) * 16) = v;
}

// rows -> [hi | lo] bf16 images (zero padded to Np rows of 2 * Dp) and the maximum of the EXACT float32 |x|^2.
// One wave per row; unit != 0 (cosine): the row is scaled to unit length first (a zero row stays zero).
template <class EL = Bf16>
__global__ __launch_bounds__(256) void prep_x_bf16x3_kernel(const float* __restrict__ X, long N, int D, int Dp, long Np,
                                                            __bf16* __restrict__ Xb, float* __restrict__ xmax2, int unit) {
    using E = typename EL::T;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= Np) return;
    float scale = 1.0f;
    if (unit) {
        float q = 0.0f;
        for (int k = lane; k < D; k += 64) { float f = row < N ? X[row * D + k] : 0.0f; q = __builtin_fmaf(f, f, q); }
        q = wave_sum(q);
        scale = q > 0.0f ? 1.0f / __builtin_sqrtf(q) : 0.0f;
    }
    float s = 0.0f;
    for (int k = lane; k < Dp; k += 64) {
        const float f = (row < N && k < D) ? X[row * D + k] * scale : 0.0f;
        const E hi = cvt<E>(f);
        ((E*)Xb)[row * 2 * Dp + k] = hi;
        ((E*)Xb)[row * 2 * Dp + Dp + k] = cvt<E>(f - (float)hi);
        s = __builtin_fmaf(f, f, s);
    }
    s = wave_sum(s);
    if (lane == 0) atomic_max_pos_f32(xmax2, s);
}

template <int KS32, class EL = Bf16>
__global__ __launch_bounds__(64 * K3_NW, 2) void bmu_bf16_k16x3_kernel(const __bf16* __restrict__ Xb, long N,
                                                                       const char* __restrict__ Wst, int n_stages, int K,
                                                                       unsigned long long* __restrict__ out64) {
    using E = typename EL::T;
    using bf16x8 = typename V8<E>::t;
    constexpr int DP = 32 * KS32;
    constexpr int STAGE = k3_stage_bytes(KS32);
    constexpr int PIECES = 2 * K3_T * KS32 + 1;
    constexpr int LO = K3_T * KS32 * 1024;               // byte offset of the lo fragments inside a stage
    constexpr uint32_t IDX_MASK = K3_T <= 2 ? 7u : 15u;   // (tile16 << 2 | reg) in the low mantissa bits
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int quad = lane >> 4, col = lane & 15;
    const long wave_s0 = (long)blockIdx.x * K3_WG_SAMPLES + wave * (16 * K3_SB);

    bf16x8 xh[K3_SB][KS32], xl[K3_SB][KS32];
#pragma unroll
    for (int sb = 0; sb < K3_SB; ++sb) {
        const long row = wave_s0 + sb * 16 + col;
#pragma unroll
        for (int ks = 0; ks < KS32; ++ks) {
            xh[sb][ks] = *(const bf16x8*)(Xb + row * 2 * DP + ks * 32 + quad * 8);
            xl[sb][ks] = *(const bf16x8*)(Xb + row * 2 * DP + DP + ks * 32 + quad * 8);
        }
    }

    uint32_t gbest[K3_SB], cbest[K3_SB];                 // unsigned keys: a NaN never beats a finite positive d'
    int gstage[K3_SB];
    f32x4 accP[K3_SB];
#pragma unroll
    for (int sb = 0; sb < K3_SB; ++sb) {
        gbest[sb] = 0xFFFFFFFFu; cbest[sb] = 0xFFFFFFFFu; gstage[sb] = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) accP[sb][r] = __builtin_inff();
    }

    const int s_begin = (int)((long)n_stages * blockIdx.y / gridDim.y);
    const int s_end = (int)((long)n_stages * (blockIdx.y + 1) / gridDim.y);
    for (int p = wave; p < PIECES; p += K3_NW)
        lds_dma_16(Wst + (long)s_begin * STAGE + (long)p * 1024 + lane * 16, smem + p * 1024);

    auto reduce_tile = [&](const f32x4 (&acc)[K3_SB], int t16) {
#pragma unroll
        for (int sb = 0; sb < K3_SB; ++sb) {
            uint32_t key[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) key[r] = (__float_as_uint(acc[sb][r]) & ~IDX_MASK) | (uint32_t)(t16 * 4 + r);
            cbest[sb] = min(min(cbest[sb], key[0]), key[1]);
            cbest[sb] = min(min(cbest[sb], key[2]), key[3]);
        }
    };
    auto fold_stage = [&](int stage) {
#pragma unroll
        for (int sb = 0; sb < K3_SB; ++sb) {
            if (cbest[sb] < gbest[sb]) { gbest[sb] = cbest[sb]; gstage[sb] = stage; }
            cbest[sb] = 0xFFFFFFFFu;
        }
    };

    for (int s = s_begin; s < s_end; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + 1 < s_end) {
            const char* src = Wst + (long)(s + 1) * STAGE;
            char* dst = smem + ((s + 1 - s_begin) & 1) * STAGE;
            for (int p = wave; p < PIECES; p += K3_NW) lds_dma_16(src + (long)p * 1024 + lane * 16, dst + p * 1024);
        }
        const char* st = smem + ((s - s_begin) & 1) * STAGE;
        const float* wq = (const float*)(st + 2 * LO);

#pragma unroll
        for (int t16 = 0; t16 < K3_T; ++t16) {
            const f32x4 wv = *(const f32x4*)(wq + t16 * 16 + 4 * quad);
            bf16x8 ah[KS32], al[KS32];
#pragma unroll
            for (int ks = 0; ks < KS32; ++ks) {
                ah[ks] = *(const bf16x8*)(st + (t16 * KS32 + ks) * 1024 + lane * 16);
                al[ks] = *(const bf16x8*)(st + LO + (t16 * KS32 + ks) * 1024 + lane * 16);
            }
            f32x4 accT[K3_SB];
#pragma unroll
            for (int sb = 0; sb < K3_SB; ++sb) accT[sb] = wv;
            // the small terms first, the hi . hi term last: the float32 accumulator sees them in ascending magnitude
#pragma unroll
            for (int ks = 0; ks < KS32; ++ks)
#pragma unroll
                for (int sb = 0; sb < K3_SB; ++sb) {
                    accT[sb] = mfma16(al[ks], xh[sb][ks], accT[sb]);
                    accT[sb] = mfma16(ah[ks], xl[sb][ks], accT[sb]);
                }
#pragma unroll
            for (int ks = 0; ks < KS32; ++ks)
#pragma unroll
                for (int sb = 0; sb < K3_SB; ++sb)
                    accT[sb] = mfma16(ah[ks], xh[sb][ks], accT[sb]);
            reduce_tile(accP, (t16 + K3_T - 1) % K3_T);  // the previous tile's keys, under this tile's MFMAs
            if (t16 == 0) fold_stage(s - 1);
#pragma unroll
            for (int sb = 0; sb < K3_SB; ++sb) accP[sb] = accT[sb];
        }
    }
    reduce_tile(accP, K3_T - 1);
    fold_stage(s_end - 1);

#pragma unroll
    for (int sb = 0; sb < K3_SB; ++sb) {
        const uint32_t code = gbest[sb] & IDX_MASK;
        const uint32_t unit = (uint32_t)gstage[sb] * K3_STAGE_UNITS + (code >> 2) * 16 + quad * 4 + (code & 3);
        unsigned long long comp = ((unsigned long long)(gbest[sb] & ~IDX_MASK) << 32) | unit;
        unsigned long long o = __shfl_xor(comp, 16, 64);
        if (o < comp) comp = o;
        o = __shfl_xor(comp, 32, 64);
        if (o < comp) comp = o;
        const long row = wave_s0 + sb * 16 + col;
        if (quad == 0 && row < N) atomicMin(out64 + row, comp);
    }
}

}  // namespace somhip
