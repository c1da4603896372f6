// The update half of XPySom._update / _merge_updates (xpysom.py:420-455), restructured.
//
// The reference materialises g[n,i,j] = h(bmu_n -> (i,j)) * eta for every sample and
// multiplies g^T (K x n) by x (n x D).  h depends on n only through bmu_n, and on the
// rectangular topology every neighbourhood of neighborhoods.py is a short sum of
// products of a function of the row offset and a function of the column offset:
//     g[n,i,j] = sum_t  Px_t[i, ci_n] * Py_t[j, cj_n]
//   gaussian     (:14-33)   t=1 : ex * ey                      ex = exp(-(i-a)^2/d) [* box mask]
//   mexican_hat  (:57-74)   t=2 : ex(1-2px/d) * ey  -  ex * (2py/d)ey
//   bubble       (:99-112)  t=1 : box * box
//   triangle     (:114-130) t=1 : tri * tri
// The hexagonal topology (gaussian_generic / mexican_hat_generic, :35-97; coordinates xpysom.py:201-206)
// shifts every second map column-index row j by -0.5 in x: xx[j,i] = i - s(j)/2, s(j) = 1 iff (Y-1-j) even.
// The x offset between unit (i,j) and BMU (ci,cj) is then (i - ci) + (s(cj) - s(j))/2: one of three values
// per (s(j), s(cj)) class, so the hexagonal neighbourhood is the rectangular one summed over 3 classes
//   {s(j)=s(cj): +0}, {s(j)=0,s(cj)=1: +1/2}, {s(j)=1,s(cj)=0: -1/2}
// with the class indicator folded into the column factor: nt = 3 (gaussian), 6 (mexican hat).
// Hence, exactly (only the float summation order differs):
//     S[b,:] = sum_{n: bmu_n = b} x_n,  c[b] = #{n: bmu_n = b}          (segment sum, HBM/atomic bound)
//     [num|den](i,j,:) = sum_t sum_a Px_t[i,a] sum_b Py_t[j,b] [S|c](a,b,:)   (two small exact-f32 MFMA GEMMs)
// which replaces the 2*N*K*D-flop GEMM of xpysom.py:437-438 by 2*K*(X+Y)*(D+1) flops.
#pragma once
#include "som_common.hpp"

namespace somhip {

// ---- segment sum: SC[b][0..D-1] = sum_{bmu_n = b} x_n ; SC[b][D] = #{bmu_n = b} ------------------
// The rows are first ordered by BMU (stable radix sort of (bmu_n, n) pairs, rocPRIM: the sorted order is a
// function of the BMUs alone).  Level 0: every wave walks a chunk of SEG_CHUNK consecutive sorted positions and
// keeps the running sum of the current unit in registers (lane = feature pair); row gathers are whole 4*D-byte
// rows, eight in flight per wave, so the pass is bound by reading X once (N*D*4 bytes).
//
// No atomics, fixed order.  A unit's rows are ONE run of the sorted order.  A run that lies inside one chunk
// is complete there: the wave adds it to SC[unit] with a plain read-modify-write -- it is the only writer of
// that unit in the launch.  A run cut by a chunk boundary leaves a PARTIAL instead: chunk w owns the two
// entries 2w (its first run, if that continues the previous chunk) and 2w+1 (its last run, if the next chunk
// continues it) of the next level's list (key, vector, count); unused entries carry key -1, and a chunk that is
// one run open on both sides fills 2w and gives 2w+1 the same key with a zero vector, so that the partials of
// one unit stay adjacent.  The next level is the same kernel over that list (SEG_CHUNK_UP entries per wave),
// and so on until one wave holds the whole list (1 Mi rows: 1 048 576 -> 65 536 -> 2 048 -> 64 entries).
// Every unit's sum is therefore formed in ONE order fixed by (N, the BMUs): two epochs from the same state
// are bitwise equal, and a unit that wins every row costs log-many tiny passes instead of N/32 serialised
// atomics on one address.
constexpr int SEG_CHUNK = 32;      // rows per wave, level 0
constexpr int SEG_CHUNK_UP = 64;   // partial entries per wave, upper levels

__global__ __launch_bounds__(256) void iota_kernel(int* __restrict__ v, long n) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) v[i] = (int)i;
}

// number of entries of the list a level with n entries and chunk c leaves behind
__host__ __device__ inline long seg_next_entries(long n, int c) { return 2 * ((n + c - 1) / c); }

// LEVEL0: keys = sorted BMUs, srow = sorted row ids, rows gathered from X (count 1 each).
// !LEVEL0: keys / vin = the previous level's partial list ([n] keys, [n][D1p] vectors, count in column D).
// VEC2 (D even): a lane owns feature pairs (8-byte accesses, one instruction per 128 features of a row).
// accumulate == 0: SC was zeroed and every unit is written at most once in the whole pass: plain stores.
template <bool LEVEL0, bool VEC2>
__global__ __launch_bounds__(256) void runsum_kernel(const float* __restrict__ X, const int* __restrict__ keys,
                                                     const int* __restrict__ srow, const float* __restrict__ vin,
                                                     long n, int D, int D1p, int accumulate, float* __restrict__ SC,
                                                     int* __restrict__ kout, float* __restrict__ vout) {
    constexpr int C = LEVEL0 ? SEG_CHUNK : SEG_CHUNK_UP;
    const int lane = threadIdx.x & 63;
    const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long p0 = wid * C;
    if (p0 >= n) return;
    const int cnt_here = (int)((n - p0 < C) ? (n - p0) : C);
    // lane i < cnt_here holds the i-th (key, row) of the chunk
    const int my_key = lane < cnt_here ? keys[p0 + lane] : -1;
    const int my_row = (LEVEL0 && lane < cnt_here) ? srow[p0 + lane] : 0;
    const float my_cnt = LEVEL0 ? 1.0f : ((lane < cnt_here && my_key >= 0) ? vin[(p0 + lane) * D1p + D] : 0.0f);
    const int first_key = __builtin_amdgcn_readfirstlane(my_key);
    const int last_key = __builtin_amdgcn_readlane(my_key, cnt_here - 1);
    const int prev_key = p0 > 0 ? keys[p0 - 1] : -2;
    const int next_key = p0 + cnt_here < n ? keys[p0 + cnt_here] : -2;
    const bool open_left = first_key >= 0 && __builtin_amdgcn_readfirstlane(prev_key) == first_key;
    const bool open_right = last_key >= 0 && __builtin_amdgcn_readfirstlane(next_key) == last_key;
    int slot_key[2] = {-1, -1};
    bool slot1_zero = false;

    for (int f0 = 0; f0 < D; f0 += 256) {                 // 4 features per lane per sweep
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        int cur = first_key;
        int run_begin = 0;
        float run = 0.f;
        // feature of acc[j]: VEC2: f0 + 128*(j>>1) + 2*lane + (j&1);  scalar: f0 + lane + 64*j
        auto feat = [&](int j) { return VEC2 ? f0 + 128 * (j >> 1) + 2 * lane + (j & 1) : f0 + lane + 64 * j; };
        auto put = [&](float* dst, bool add) {            // this lane's 4 sums (and the count) into one row
            if (VEC2) {
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int d = f0 + 128 * jj + 2 * lane;
                    if (d < D) {
                        float2 t = make_float2(acc[2 * jj], acc[2 * jj + 1]);
                        if (add) { const float2 o = *(const float2*)(dst + d); t.x += o.x; t.y += o.y; }
                        *(float2*)(dst + d) = t;
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int d = feat(j);
                    if (d < D) dst[d] = add ? dst[d] + acc[j] : acc[j];
                }
            }
            if (f0 == 0 && lane == 0) dst[D] = add ? dst[D] + run : run;
        };
        auto flush = [&](int end) {                       // the run [run_begin, end) of unit `cur` is complete in this chunk
            if (cur >= 0) {
                const bool first = run_begin == 0, last = end == cnt_here;
                if (first && open_left) {
                    put(vout + (2 * wid) * (long)D1p, false);
                    slot_key[0] = cur;
                    if (last && open_right) { slot_key[1] = cur; slot1_zero = true; }
                } else if (last && open_right) {
                    put(vout + (2 * wid + 1) * (long)D1p, false);
                    slot_key[1] = cur;
                } else {
                    put(SC + (long)cur * D1p, accumulate != 0);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = 0.f;
            run = 0.f;
        };
        auto entry_ptr = [&](int i) -> const float* {     // vector of the chunk's i-th entry (i uniform)
            if (LEVEL0) return X + (long)__builtin_amdgcn_readlane(my_row, i) * D;
            return vin + (p0 + i) * (long)D1p;
        };
        for (int i0 = 0; i0 < cnt_here; i0 += 8) {
            float v[8][4];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u;
                const int ii = i < cnt_here ? i : 0;
                const bool live = i < cnt_here && __builtin_amdgcn_readlane(my_key, ii) >= 0;
                const float* x = entry_ptr(ii);
                if (VEC2) {
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int d = f0 + 128 * jj + 2 * lane;
                        float2 t = make_float2(0.f, 0.f);
                        if (live && d < D) t = *(const float2*)(x + d);   // D even: d + 1 < D, 8-byte aligned
                        v[u][2 * jj] = t.x; v[u][2 * jj + 1] = t.y;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int d = f0 + lane + 64 * j;
                        v[u][j] = (live && d < D) ? x[d] : 0.f;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u;
                if (i < cnt_here) {
                    const int b = __builtin_amdgcn_readlane(my_key, i);
                    if (b != cur) { flush(i); cur = b; run_begin = i; }
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] += v[u][j];
                    run += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_cnt), i));
                }
            }
        }
        flush(cnt_here);
        if (slot1_zero) {                                 // one run, open on both sides: entry 2w+1 = (key, 0)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = 0.f;
            run = 0.f;
            put(vout + (2 * wid + 1) * (long)D1p, false);
        }
    }
    if (lane == 0 && kout != nullptr) { kout[2 * wid] = slot_key[0]; kout[2 * wid + 1] = slot_key[1]; }
}

// ---- neighbourhood factor tables ---------------------------------------------------------------
// P1: stage-1 matrices  [nt][Y][Y]   (Py_t[j][b])
// P2: stage-2 matrix    [X][nt*X]    (Px_t[i][a] at column t*X + a), carries eta.
// wide != 0: float64 evaluation rounded once to float32 (NumPy >= 2 with a numpy.float64 sigma);
// wide == 0: mimic the float32 evaluation: the exponent argument is rounded to float32
// (the dominant error term) before a correctly rounded exp.
struct NeighParams {
    double sigma, eta, d;   // d = 2*std_coeff^2*sigma^2 (host, double)
    int kind, compact, wide, X, Y, nt;
    int hex, base_nt;       // hexagonal topology: nt = 3 * base_nt
};

__device__ __forceinline__ double neigh_exp(double delta2, const NeighParams& p) {
    if (p.wide) return exp(-delta2 / p.d);
    float a = -(float)delta2 / (float)p.d;
    return (double)(float)exp((double)a);
}
// the reference's support mask, literally: n > c - sigma  and  n < c + sigma  in float64 (neighborhoods.py:29-31,
// 105-110).  c -/+ sigma is rounded before the compare, so a sigma within an ulp of an integer (asymptotic decay
// produces them: 5 / (1 + 2/3) = 3.0000000000000004) decides the boundary unit differently than |n - c| < sigma.
// (The hexagonal classes carry only the coordinate DIFFERENCE, half-unit shift included, so their mask compares
// that: the generic functions' mask, neighborhoods.py:50-54, up to the rounding of c -/+ sigma.)
__device__ __forceinline__ double neigh_box(double n, double c, double dl, const NeighParams& p) {
    if (p.hex) return (dl > -p.sigma && dl < p.sigma) ? 1.0 : 0.0;
    return (n > c - p.sigma && n < c + p.sigma) ? 1.0 : 0.0;
}
__device__ __forceinline__ double neigh_round(double v, const NeighParams& p) { return p.wide ? v : (double)(float)v; }

// value of factor `which` (0 = row factor Px, 1 = column factor Py) of base term t for unit coordinate n and
// BMU coordinate c (`shift`: the hexagonal classes' half-unit offset, added to n - c)
__device__ double neigh_factor(int which, int t, double n, double c, double shift, const NeighParams& p) {
    const double dl = (n - c) + shift;
    const double d2 = dl * dl;
    switch (p.kind) {
    case 0: {   // gaussian
        double e = neigh_exp(d2, p);
        if (p.compact) e *= neigh_box(n, c, dl, p);
        return e;
    }
    case 1: {   // mexican hat: (ex(1-2px/d)) * ey  -  ex * ((2py/d) ey)
        double e = neigh_exp(d2, p);
        double q = p.wide ? (2.0 / p.d) * d2 : (double)((float)(2.0 / p.d) * (float)d2);
        if (t == 0) return which == 0 ? neigh_round(e * (1.0 - q), p) : e;
        return which == 0 ? -e : neigh_round(q * e, p);
    }
    case 2:     // bubble
        return neigh_box(n, c, dl, p);
    default: {  // triangle
        double v = p.sigma - fabs(dl);
        if (v < 0.0) v = 0.0;
        if (p.compact) v *= neigh_box(n, c, dl, p);
        return neigh_round(v, p);
    }
    }
}

// p_dev != nullptr: the parameters are read from device memory instead (a captured hipGraph replays
// this launch with new sigma / eta every epoch; store_params_kernel refreshes them before the replay).
__global__ void store_params_kernel(NeighParams p, NeighParams* __restrict__ dst) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *dst = p;
}

__global__ __launch_bounds__(256) void neigh_tables_kernel(NeighParams p_val, const NeighParams* __restrict__ p_dev,
                                                           float* __restrict__ P1, float* __restrict__ P2) {
    const NeighParams p = p_dev ? *p_dev : p_val;
    long id = (long)blockIdx.x * 256 + threadIdx.x;
    const long n1 = (long)p.nt * p.Y * p.Y;
    const long n2 = (long)p.X * p.nt * p.X;
    if (id < n1) {
        int b = id % p.Y;
        long r = id / p.Y;
        int j = r % p.Y;
        int t = r / p.Y;
        double v = neigh_factor(1, t % p.base_nt, (double)j, (double)b, 0.0, p);
        if (p.hex) {                                   // class indicator on (s(j), s(cj))
            const int cls = t / p.base_nt;
            const int sj = ((p.Y - 1 - j) & 1) == 0, sb = ((p.Y - 1 - b) & 1) == 0;
            const bool in = cls == 0 ? sj == sb : cls == 1 ? (sj == 0 && sb == 1) : (sj == 1 && sb == 0);
            if (!in) v = 0.0;
        }
        P1[id] = (float)v;
    } else if (id < n1 + n2) {
        long q = id - n1;
        int col = q % ((long)p.nt * p.X);
        int i = q / ((long)p.nt * p.X);
        int t = col / p.X, a = col % p.X;
        const int cls = t / p.base_nt;
        const double shift = !p.hex ? 0.0 : cls == 1 ? 0.5 : cls == 2 ? -0.5 : 0.0;
        double v = neigh_factor(0, t % p.base_nt, (double)i, (double)a, shift, p);
        P2[q] = p.wide ? (float)(v * p.eta) : (float)v * (float)p.eta;
    }
}

// ---- OUT[b] = H (Ro x Ri) * M[b] (Ri x C), exact float32 on v_mfma_f32_32x32x2_f32 ---------------
// Workgroup = 4 waves as 2 (rows) x 2 (cols); wave tile 64 x 64 (2 x 2 MFMA tiles); block tile
// 128 x 128; k chunk 32.  The next chunk's global loads are issued into registers before the MFMAs of
// the current chunk and written to LDS after them (one LDS buffer, two barriers per chunk), so HBM/L2
// latency hides under the 64 MFMAs per wave.
//
// Bands.  Late in training sigma is small and the neighbourhood tables underflow to EXACT float32 zeros
// away from the diagonal (exp(-dx^2/d) < 2^-150 for |dx| > ~7 sigma at std_coeff 0.5; bubble / triangle
// are compact by definition).  band_ranges_kernel records, per 128-row block of H and per column
// segment (one per neighbourhood term), the smallest range of 32-column chunks that holds every nonzero;
// the GEMM then walks only those chunks.  Skipped chunks would have added 0 * m = 0 to the accumulators,
// so the result is bit-identical for finite data.
constexpr int LM_BM = 128, LM_BN = 128, LM_BK = 32;

// ranges[rb * nseg + s] = {segw - lo, hi}: columns [s*segw + lo, s*segw + hi) of rows [128 rb, 128 rb + 128) hold
// every nonzero of that segment; both fields only grow (atomicMax from a zeroed buffer: {0, 0} = all zero).
// One workgroup scans 128 rows x 64 columns (wave w: rows 32 w .. 32 w + 31, lane: one column); grid =
// (row blocks, segments * column chunks, batch), batch z: H + z * hz, ranges + z * rz.
__global__ __launch_bounds__(256) void band_ranges_kernel(const float* __restrict__ H, int Ro, int Ri, int nseg,
                                                          int segw, int2* __restrict__ ranges, long hz, long rz) {
    const int rb = blockIdx.x;
    const int cchunks = (segw + 63) / 64;
    const int seg = blockIdx.y / cchunks, cc = blockIdx.y - seg * cchunks;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* Hb = H + (long)blockIdx.z * hz;
    const int k = cc * 64 + lane;                      // column inside the segment
    const int r_begin = rb * LM_BM + wave * 32;
    bool nz = false;
    if (k < segw) {
#pragma unroll 8
        for (int i = 0; i < 32; ++i) {
            const int r = r_begin + i;
            if (r < Ro) nz |= Hb[(long)r * Ri + seg * segw + k] != 0.0f;
        }
    }
    const unsigned long long m = __ballot(nz);
    if (m != 0 && lane == 0) {
        const int first = cc * 64 + __builtin_ctzll(m), last = cc * 64 + 63 - __builtin_clzll(m);
        int* r = (int*)(ranges + (long)blockIdx.z * rz + (long)rb * nseg + seg);
        atomicMax(r, segw - first);
        atomicMax(r + 1, last + 1);
    }
}

// ranges == nullptr: the whole of H (one segment [0, Ri)).
__global__ __launch_bounds__(256) void leftmul_f32_kernel(const float* __restrict__ H, int Ro, int Ri,
                                                          const float* __restrict__ M, long m_batch_stride,
                                                          float* __restrict__ OUT, long o_batch_stride, long C,
                                                          const int2* __restrict__ ranges, int nseg, int segw) {
    __shared__ float Hs[LM_BM][LM_BK + 1];
    __shared__ float Ms[LM_BK][LM_BN + 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, col = lane & 31;
    const int wr = wave >> 1, wc = wave & 1;
    const long c0 = (long)blockIdx.x * LM_BN;
    const int i0 = blockIdx.y * LM_BM;
    const float* Mb = M + (long)blockIdx.z * m_batch_stride;
    float* Ob = OUT + (long)blockIdx.z * o_batch_stride;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][t][r] = 0.0f;

    float hreg[16], mreg[16];                          // this thread's share of the staged chunk
    auto fetch = [&](int r0, int kend) {               // chunk [r0, r0 + 32) clipped to columns < kend
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            int idx = tid + q * 256;
            int i = idx >> 5, k = idx & 31;            // H tile: 128 x 32
            hreg[q] = (i0 + i < Ro && r0 + k < kend) ? H[(long)(i0 + i) * Ri + r0 + k] : 0.0f;
            int kk = idx >> 7, c = idx & 127;          // M tile: 32 x 128
            mreg[q] = (r0 + kk < kend && c0 + c < C) ? Mb[(long)(r0 + kk) * C + c0 + c] : 0.0f;
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            int idx = tid + q * 256;
            Hs[idx >> 5][idx & 31] = hreg[q];
            Ms[idx >> 7][idx & 127] = mreg[q];
        }
    };

    // the chunks to walk (uniform over the workgroup): every segment is walked over the union [lo, hi) of
    // the segments' nonzero ranges, LM_BK columns at a time, clipped to hi
    int lo = 0, hi = Ri, ns = 1, sw = Ri;
    if (ranges != nullptr) {
        ns = nseg; sw = segw; lo = segw; hi = 0;
        for (int sg = 0; sg < nseg; ++sg) {
            const int2 v = ranges[blockIdx.y * nseg + sg];   // {segw - first nonzero, last nonzero + 1}
            if (v.y > 0) { lo = min(lo, ((segw - v.x) / LM_BK) * LM_BK); hi = max(hi, min(v.y, segw)); }
        }
        if (hi <= lo) lo = hi = 0;
    }
    const int per = (hi - lo + LM_BK - 1) / LM_BK;     // chunks per segment
    const int total = per * ns;
    auto chunk_at = [&](int c, int& r0, int& kend) {
        const int sg = c / per, j = c - sg * per;
        r0 = sg * sw + lo + j * LM_BK;
        kend = sg * sw + hi;
    };
    if (total > 0) { int r0, kend; chunk_at(0, r0, kend); fetch(r0, kend); }
    for (int c = 0; c < total; ++c) {
        __syncthreads();                                // everyone is done reading the previous chunk
        stash();
        __syncthreads();
        if (c + 1 < total) { int r0, kend; chunk_at(c + 1, r0, kend); fetch(r0, kend); }   // in flight under the MFMAs below
#pragma unroll
        for (int k = 0; k < LM_BK; k += 2) {
            float av[2], bv[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) av[a] = Hs[wr * 64 + a * 32 + col][k + half];
#pragma unroll
            for (int t = 0; t < 2; ++t) bv[t] = Ms[k + half][wc * 64 + t * 32 + col];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[t], acc[a][t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            long c = c0 + wc * 64 + t * 32 + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int i = i0 + wr * 64 + a * 32 + mfma32_row(r, half);
                if (i < Ro && c < C) Ob[(long)i * C + c] = acc[a][t][r];
            }
        }
}

// ---- merge: W = where(den != 0, num/den, W)  (xpysom.py:446-455) ---------------------------------
__global__ __launch_bounds__(256) void merge_kernel(float* __restrict__ W, const float* __restrict__ ACC,
                                                    long K, int D, int D1p) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= K * D) return;
    long k = i / D;
    int d = (int)(i - k * D);
    float den = ACC[k * D1p + D];
    if (den != 0.0f) W[i] = ACC[k * D1p + d] / den;
}

// ---- quantization error: sum_n |x_n - W[bmu_n]|  (xpysom.py:703-705) -----------------------------
__global__ __launch_bounds__(256) void qe_kernel(const float* __restrict__ X, const int* __restrict__ bmu,
                                                 const float* __restrict__ W, long N, int D,
                                                 double* __restrict__ sum_out) {
    __shared__ double part[4];
    long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float s = 0.0f;
    if (row < N) {
        const float* x = X + row * D;
        const float* w = W + (long)bmu[row] * D;
        for (int k = lane; k < D; k += 64) { float df = x[k] - w[k]; s = __builtin_fmaf(df, df, s); }
    }
    s = wave_sum(s);
    if (lane == 0) part[wave] = (row < N) ? (double)__builtin_sqrtf(s) : 0.0;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(sum_out, part[0] + part[1] + part[2] + part[3]);
}

}  // namespace somhip
