// Fused distance + BMU argmin, bf16 MFMA throughput mode (euclidean).
//
// Same reference chain as bmu_f32.hpp (distances.py:11-23 + xpysom.py:416), computed as
//     d'(n,k) = B + |w~_k|^2/2 - x~_n . w~_k          (argmin_k d' == argmin_k |x~_n - w~_k|^2)
// on bf16-rounded x~, w~ with float32 accumulation in v_mfma_f32_32x32x16_bf16.  The norm
// is taken from the SAME bf16 values as the products, so d' is a true distance in the rounded
// space.  B = max_n|x~_n| * max_k|w~_k| (Cauchy-Schwarz) makes every d' positive, so its bit
// pattern orders like an integer and B + |w~|^2/2 is simply the MFMA's initial accumulator,
// read from LDS straight into the C registers: no per-element VALU besides the argmin.
//
// Layout in HBM (built once per epoch by prep_w_bf16 / prep_wsqh):
//   the codebook as a sequence of STAGES of 128 units; each stage is one contiguous block
//     [ut=0..3][kstep][lane 0..63][8 bf16]   A-operand fragments of -w~, 1 KiB per (ut,kstep)
//     [128 x f32  B + |w~|^2/2]  (+ pad to 1 KiB)
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

// positive floats order like their bit patterns: a float max through an integer atomic
__device__ __forceinline__ void atomic_max_pos_f32(float* addr, float v) {
    // a million waves share one address: only a value that would raise the maximum pays for the atomic
    // (a NaN norm -- a NaN row or unit -- is left out: the offset B must stay finite for everybody else)
    if (v == v && __float_as_uint(v) > __hip_atomic_load((unsigned int*)addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax((unsigned int*)addr, __float_as_uint(v));
}

// |w~|^2 of every unit (bf16-rounded values) and its maximum over the codebook
__global__ __launch_bounds__(256) void prep_wnorm_kernel(const float* __restrict__ W, int K, int D,
                                                         float* __restrict__ wn, float* __restrict__ wmax2,
                                                         const float* __restrict__ unit_wsq) {
    long u = (long)blockIdx.x * 256 + threadIdx.x;
    float s = 0.0f;
    if (u < K) {
        float scale = 1.0f;       // cosine: the stage image holds unit-length rows and no norm term
        if (unit_wsq != nullptr) { float q = unit_wsq[u]; scale = q > 0.0f ? 1.0f / __builtin_sqrtf(q) : 0.0f; }
        for (int k = 0; k < D; ++k) { float f = (float)(__bf16)(W[u * D + k] * scale); s = __builtin_fmaf(f, f, s); }
        wn[u] = unit_wsq != nullptr ? 0.0f : s;
    }
    float m = s;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomic_max_pos_f32(wmax2, m);
}

// initial accumulators  B + |w~|^2/2  written behind each stage's fragments; B from the two maxima
// (+ the launch's per-row merge keys best64[0..n_rows) = all ones, when given: one launch instead of a memset and a kernel)
__global__ __launch_bounds__(256) void prep_wsqh_kernel(const float* __restrict__ wn, int K,
                                                        const float* __restrict__ wmax2,
                                                        const float* __restrict__ xmax2, char* __restrict__ Wst,
                                                        int n_stages, int stage_bytes, int stage_units,
                                                        unsigned long long* __restrict__ best64, long n_rows) {
    long u = (long)blockIdx.x * 256 + threadIdx.x;
    if (best64 != nullptr && u < n_rows) best64[u] = ~0ull;
    if (u >= (long)n_stages * stage_units) return;
    const float big = __builtin_sqrtf(*wmax2) * __builtin_sqrtf(*xmax2) * (1.0f + 1.0f / 1024.0f);
    float s = (u < K) ? __builtin_fmaf(0.5f, wn[u], big) : BF_PAD_NORM;
    long stage = u / stage_units;
    int within = u % stage_units;
    float* dst = (float*)(Wst + (stage + 1) * (long)stage_bytes - 1024);   // the stage's last KiB
    dst[within] = s;
}

// samples -> bf16 rows [Np][Dp] (zero padded) and max_n |x~_n|^2.  One wave per row.
// unit != 0 (cosine): the row is scaled to unit length first (a zero row stays zero).
__global__ __launch_bounds__(256) void prep_x_bf16_kernel(const float* __restrict__ X, long N, int D, int Dp,
                                                          long Np, __bf16* __restrict__ Xb,
                                                          float* __restrict__ xmax2, int unit) {
    long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
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
        float f = (row < N && k < D) ? X[row * D + k] * scale : 0.0f;
        __bf16 b = (__bf16)f;
        Xb[row * Dp + k] = b;
        float fb = (float)b;
        s = __builtin_fmaf(fb, fb, s);
    }
    s = wave_sum(s);
    if (lane == 0) atomic_max_pos_f32(xmax2, s);
}

// --------------------------------------------------------------------------------------
template <int KSTEPS>
__global__ __launch_bounds__(256, 2) void bmu_bf16_kernel(const __bf16* __restrict__ Xb, long N,
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
#pragma unroll
    for (int sb = 0; sb < BF_SBW; ++sb) {
        const long row = wave_s0 + sb * 32 + col;          // rows are padded to a multiple of the WG size
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) xf[sb][ks] = *(const bf16x8*)(Xb + row * DP + ks * 16 + half * 8);
    }

    int32_t gbest[BF_SBW];
    int gstage[BF_SBW];
#pragma unroll
    for (int sb = 0; sb < BF_SBW; ++sb) { gbest[sb] = 0x7FFFFFFF; gstage[sb] = 0; }

    // Software pipeline inside the wave: while the 16 MFMAs of tile t run, the VALU reduces tile
    // t-1 (accP) and the LDS fetches the fragments of tile t+1, so the matrix pipe never waits for
    // this wave's own epilogue.  accP starts as +inf: its keys lose against every real distance.
    f32x16 accP[BF_SBW];
#pragma unroll
    for (int sb = 0; sb < BF_SBW; ++sb)
#pragma unroll
        for (int r = 0; r < 16; ++r) accP[sb][r] = __builtin_inff();
    int32_t cbest[BF_SBW], cbest2[BF_SBW];
#pragma unroll
    for (int sb = 0; sb < BF_SBW; ++sb) { cbest[sb] = 0x7FFFFFFF; cbest2[sb] = 0x7FFFFFFF; }

    // stage 0 -> ring slot 0
    for (int p = wave; p < PIECES; p += 4) lds_dma_16(Wst + (long)p * 1024 + lane * 16, smem + p * 1024);

    auto reduce_tile = [&](const f32x16 (&acc)[BF_SBW], int ut) {
#pragma unroll
        for (int sb = 0; sb < BF_SBW; ++sb) {
            int32_t key[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float f = acc[sb][r];               // (bit_cast straight from a vector element reads lane 0)
                key[r] = (int32_t)((__float_as_uint(f) & ~IDX_MASK) | (uint32_t)(ut * 16 + r));
            }
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
                cbest[sb] = min(min(cbest[sb], key[r]), key[r + 1]);
                cbest2[sb] = min(min(cbest2[sb], key[r + 2]), key[r + 3]);
            }
        }
    };
    auto fold_stage = [&](int stage) {
#pragma unroll
        for (int sb = 0; sb < BF_SBW; ++sb) {
            const int32_t c = min(cbest[sb], cbest2[sb]);
            if (c < gbest[sb]) { gbest[sb] = c; gstage[sb] = stage; }
            cbest[sb] = 0x7FFFFFFF;
            cbest2[sb] = 0x7FFFFFFF;
        }
    };

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

        f32x4 wv[4];
        bf16x8 a[KSTEPS];
#pragma unroll
        for (int g = 0; g < 4; ++g) wv[g] = *(const f32x4*)(wq + 8 * g + 4 * half);
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) a[ks] = *(const bf16x8*)(st + ks * 1024 + lane * 16);

#pragma unroll
        for (int ut = 0; ut < BF_UT; ++ut) {
            f32x4 wvN[4];
            bf16x8 aN[KSTEPS];
            if (ut + 1 < BF_UT) {
#pragma unroll
                for (int g = 0; g < 4; ++g) wvN[g] = *(const f32x4*)(wq + (ut + 1) * 32 + 8 * g + 4 * half);
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ++ks)
                    aN[ks] = *(const bf16x8*)(st + ((ut + 1) * KSTEPS + ks) * 1024 + lane * 16);
            }
            f32x16 accT[BF_SBW];
#pragma unroll
            for (int sb = 0; sb < BF_SBW; ++sb) {          // C-in = B + |w~|^2/2, as read from LDS
#pragma unroll
                for (int r = 0; r < 16; ++r) accT[sb][r] = wv[r >> 2][r & 3];
            }
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
                for (int sb = 0; sb < BF_SBW; ++sb)
                    accT[sb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks], xf[sb][ks], accT[sb], 0, 0, 0);
            // the pending tile: tile ut-1 of this stage, or tile 3 of the previous stage
            reduce_tile(accP, (ut + BF_UT - 1) % BF_UT);
            if (ut == 0) fold_stage(s - 1);
#pragma unroll
            for (int sb = 0; sb < BF_SBW; ++sb) accP[sb] = accT[sb];
            if (ut + 1 < BF_UT) {
#pragma unroll
                for (int g = 0; g < 4; ++g) wv[g] = wvN[g];
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ++ks) a[ks] = aN[ks];
            }
        }
        // interleave request for the whole stage body: per MFMA, three epilogue VALU and one LDS read
#pragma unroll
        for (int i = 0; i < BF_UT * BF_SBW * KSTEPS; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);   // VALU
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
        }
    }
    reduce_tile(accP, BF_UT - 1);
    fold_stage(n_stages - 1);

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
