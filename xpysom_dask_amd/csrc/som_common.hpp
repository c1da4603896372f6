// Shared device-side types and helpers for libsomhip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define SOM_WAVE 64

// Diagnostic build only (-DSOM_STAMPS; tools/stamps.py): a workgroup's first wave stamps s_memtime (shader clock) and
// s_memrealtime (100 MHz) on entry and on exit of the BMU kernels' scan and leaves the two differences in a buffer of
// their own -- the clock the chip holds INSIDE the kernel is d(memtime) / d(memrealtime) * 100 MHz
// (MI355X_MICROARCH.md, DVFS give-back item 6).  No product code path reads the buffer; without the macro nothing exists.
#ifdef SOM_STAMPS
__device__ unsigned long long* g_som_stamps = nullptr;     // [2 * workgroups] or nullptr
struct SomStamp {
    unsigned long long t0, r0;
    __device__ __forceinline__ SomStamp() : t0(__builtin_amdgcn_s_memtime()), r0(__builtin_amdgcn_s_memrealtime()) {}
    __device__ __forceinline__ void done() const {
        if (g_som_stamps != nullptr && threadIdx.x == 0) {
            const unsigned long long wg = (unsigned long long)blockIdx.y * gridDim.x + blockIdx.x;
#if SOM_STAMPS == 2                                         // (tools/wg_timeline.py: when every workgroup's scan began and ended, 100 MHz)
            g_som_stamps[2 * wg] = r0;
            g_som_stamps[2 * wg + 1] = __builtin_amdgcn_s_memrealtime();
#else
            g_som_stamps[2 * wg] = __builtin_amdgcn_s_memtime() - t0;
            g_som_stamps[2 * wg + 1] = __builtin_amdgcn_s_memrealtime() - r0;
#endif
        }
    }
};
#define SOM_STAMP_BEGIN() const SomStamp som_stamp_
#define SOM_STAMP_END() som_stamp_.done()
#else
#define SOM_STAMP_BEGIN() do {} while (0)
#define SOM_STAMP_END() do {} while (0)
#endif

// The 16-bit operand type of the half-precision BMU kernels: __bf16 (precision 'bf16') or _Float16
// ('f16', and the exact mode's screen: three more mantissa bits at the same MFMA rate, range 6e-8 .. 65504).  The kernels are templates on
// its tag (`class EL`, `using E = typename EL::T`); their operand images are the same bytes either way.
// (kernels take the TAG, not the type: rocprofv3 cannot demangle a 16-bit float type in a kernel's template arguments)
struct Bf16 { typedef __bf16 T; };
struct F16 { typedef _Float16 T; };
template <class E> struct V8;
template <> struct V8<__bf16> { typedef bf16x8 t; };
template <> struct V8<_Float16> { typedef f16x8 t; };
// float -> operand type.  IEEE half saturates at +-65504 instead of overflowing to infinity (som_set_data / som_set_weights
// refuse rows and units beyond the range, but a mexican-hat update can overshoot its data: one infinite unit norm would
// make the offset B, and with it every distance of the launch, infinite); NaN stays NaN.
template <class E> __device__ __forceinline__ E cvt(float f) { return (E)f; }
template <> __device__ __forceinline__ _Float16 cvt<_Float16>(float f) {
    return (_Float16)(f != f ? f : __builtin_fminf(__builtin_fmaxf(f, -65504.0f), 65504.0f));
}
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// C/D register -> row of a 32x32 MFMA accumulator tile (dtype independent on gfx950):
// lane l holds column l&31, rows (r&3) + 8*(r>>2) + 4*(l>>5), r = 0..15.
__device__ __forceinline__ int mfma32_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// numpy.nan_to_num on float32: NaN -> 0, +inf -> FLT_MAX, -inf -> -FLT_MAX
__device__ __forceinline__ float nan_to_num_f32(float v) {
    if (v != v) return 0.0f;
    if (v == __builtin_inff()) return 3.402823466e+38f;
    if (v == -__builtin_inff()) return -3.402823466e+38f;
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// PATCH ORDER (exact mode): the operand images hold the units in an order in which a group of 64 consecutive positions
// is a compact PATCH of the map (8 x 8 units where the sides are multiples of 8) instead of 64 units of one map row.  On
// a smooth map the units near a row's best one form a blob around it; a blob touches a third to a half as many patches
// as strips (256 x 256 x 128, the smoothest states: 16.0 -> 4.3 and 18.9 -> 6.0 candidate groups per row; 512 x 512 x 784:
// 47 -> 26), and every candidate group is 64 x input_len of float32 re-score and one gather of the row.  The order is a
// table (som_create builds it: bands of 8 map rows, column by column, then every group's units sorted): perm[position] =
// unit, inv[unit] = position.  Within a group positions ascend with the unit ids (so the first-minimum rule holds
// inside a re-score tile); the re-score translates its winner back to a unit id before the merge, which therefore still
// prefers the lowest UNIT among equal scores.  A null table: the units' own order.
