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
// function of the BMUs alone).  A unit's rows are then ONE run of the sorted order, and the sum of a run is
// formed by a tree whose shape depends on N only -- no atomics, one fixed order:
//   wave   walks a chunk of consecutive entries (level 0: SEG_CHUNK sorted rows, gathered as whole 4*D-byte rows,
//          eight in flight; upper levels: SEG_CHUNK_UP entries of a partial list) with the running sum of the
//          current unit in registers (lane = feature pair).  A run that begins and ends inside the chunk is
//          complete: the wave adds it to SC[unit] with a plain read-modify-write (it is that unit's only writer in
//          the launch).  A run cut by the chunk's edge leaves a PARTIAL (key, vector, count) in one of the wave's
//          two slots: slot 0 = its first run if that continues the previous chunk, slot 1 = its last run if the
//          next chunk continues it.  Unused slots carry key -1; a chunk that is one run open on both sides fills
//          slot 0 and gives slot 1 the same key with a zero vector, so one unit's partials stay adjacent.
//   block  the slots of a workgroup's waves live in LDS; after a barrier wave 0 walks them the same way.  Only
//          runs cut by the BLOCK's edges survive, as the two entries 2b, 2b+1 of the next level's list in HBM.
//   level  the next level is the same kernel over that list, until one block holds the whole list
//          (1 Mi rows, 16 waves per block: 1 048 576 rows -> 4 096 entries -> 32 entries -> done).
// So two epochs from the same state are bitwise equal, and a unit that wins every row costs a few tiny
// passes instead of N/32 serialised atomics on one address.
constexpr int SEG_CHUNK = 32;      // rows per wave, level 0
constexpr int SEG_CHUNK_UP = 16;   // partial entries per wave, upper levels: at least this (two groups of eight loads: a short
                                   // latency chain), up to 64 when that lets ONE workgroup finish the list (seg_chunk_up)
constexpr int SEG_MAX_WAVES = 16;  // waves per workgroup (fewer when 2 * waves * D1p floats would not fit in LDS)

__global__ __launch_bounds__(256) void iota_kernel(int* __restrict__ v, long n) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) v[i] = (int)i;
}

// ---- stable LSD radix sort of (key, value) pairs, 8 bits a pass: the rows by BMU for the segment sum (maps too large for the
// counting sort below), the rows by their last BMU's patch for the exact mode's resident order (exact_skip.hpp).  Hand-written
// for these two jobs (keys of 10 .. 20 bits, values = row indices, 10^5 .. 10^7 items): per pass
//   rs_hist_kernel     block = RS_BLOCK consecutive items: LDS histogram of the pass's digit -> table[digit * B + block]
//   rs_scan_kernel     a wave per digit: exclusive scan of its block counts in place, its total -> tot[digit]
//   rs_scatter_kernel  block = the same RS_BLOCK items, 256 at a time in item order, the waves one after the other: the first
//                      destination of a (digit, block) run = the totals of the lower digits + the block's place in its digit; an item's
//                      destination = its run's cursor + its rank among the wave's items of the same digit: the order of equal
//                      digits is the order of the items -- STABLE, and the same order on every run (no atomics decide a place)
// vin == nullptr: the values are the items' own indices (first pass of a sort of row ids).
constexpr int RS_BLOCK = 2048;

__global__ __launch_bounds__(256) void rs_hist_kernel(const int* __restrict__ keys, long n, int shift, int B, int* __restrict__ table) {
    __shared__ int hist[256];
    hist[threadIdx.x] = 0;
    __syncthreads();
    const long i0 = (long)blockIdx.x * RS_BLOCK;
#pragma unroll
    for (int q = 0; q < RS_BLOCK / 256; ++q) {
        const long i = i0 + q * 256 + threadIdx.x;
        if (i < n) atomicAdd(&hist[(keys[i] >> shift) & 255], 1);
    }
    __syncthreads();
    table[(long)threadIdx.x * B + blockIdx.x] = hist[threadIdx.x];
}

// one wave per digit: exclusive scan of the digit's B block counts in place, the digit's total -> tot[digit]
__global__ __launch_bounds__(256) void rs_scan_kernel(int* __restrict__ table, int B, int* __restrict__ tot) {
    const int d = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    int* t = table + (long)d * B;
    int carry = 0;
    for (int b0 = 0; b0 < B; b0 += 64) {
        const int b = b0 + lane;
        const int v = b < B ? t[b] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
        if (b < B) t[b] = carry + incl - v;
        carry += __shfl(incl, 63, 64);
    }
    if (lane == 0) tot[d] = carry;
}

__global__ __launch_bounds__(256) void rs_scatter_kernel(const int* __restrict__ kin, const int* __restrict__ vin, long n, int shift,
                                                         int B, const int* __restrict__ table, const int* __restrict__ tot,
                                                         int* __restrict__ kout, int* __restrict__ vout) {
    __shared__ int cur[256];                                 // next destination of every digit for this block
    __shared__ int wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {   // first destination of digit `tid`: the totals of the digits below it + this block's place inside the digit
        const int mine = tot[tid];
        int incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int before = 0;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        cur[tid] = before + incl - mine + table[(long)tid * B + blockIdx.x];
    }
    __syncthreads();
    const long i0 = (long)blockIdx.x * RS_BLOCK;
    for (int q = 0; q < RS_BLOCK / 256; ++q) {
        const long i = i0 + q * 256 + tid;
        const bool live = i < n;
        const int key = live ? kin[i] : 0;
        const int val = live ? (vin != nullptr ? vin[i] : (int)i) : 0;
        const int dg = (key >> shift) & 255;
        // the lanes of this wave with the same digit
        unsigned long long peers = __ballot(live);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long m = __ballot((dg >> b) & 1);
            peers &= ((dg >> b) & 1) ? m : ~m;
        }
        const int leader = live ? (int)__builtin_ctzll(peers) : lane;
        const int rank = (int)__builtin_popcountll(peers & ((1ull << lane) - 1ull));
        const int count = (int)__builtin_popcountll(peers);
        int dest = 0;
        for (int w = 0; w < 4; ++w) {                        // waves in item order: the sort is stable
            if (wave == w) {
                int old = 0;
                if (live && lane == leader) { old = cur[dg]; cur[dg] = old + count; }
                old = __shfl(old, leader, 64);
                dest = old + rank;
            }
            __syncthreads();
        }
        if (live) { kout[dest] = key; vout[dest] = val; }
    }
}

// ---- stable counting sort of the rows by BMU for maps of at most CS_MAX_K units ------------------------------------
// (sorted by unit, rows ascending inside a unit: exactly what the stable radix sort returns, so the run sum adds in the same
// fixed order.)  rocPRIM's merge sort takes six to seven launches at 100 000 rows (55 us of a 0.4 ms epoch at 64 x 64 x 32);
// with few units a histogram per 1 024-row block is small: table[unit][block], three launches.
//   cs_hist_kernel     block = 1 024 rows: LDS histogram of its units -> table[k * B + block]
//   cs_scan_kernel     wave per unit: exclusive scan of its B block counts in place, the unit's total -> tot[k]
//   cs_scatter_kernel  block = 1 024 rows: first destination of every (unit, block) = scan of tot over the units + the block's
//                      prefix; the block's 16 waves take their turns in order, a wave ranks its rows among equal units by
//                      ballots over the unit's bits (lower lane = lower row first)
constexpr int CS_MAX_K = 8192;        // units (LDS: 4 bytes per unit)
constexpr int CS_BLOCK = 1024;        // rows per block

__global__ __launch_bounds__(256) void cs_hist_kernel(const int* __restrict__ bmu, long N, int K, int B, int* __restrict__ table) {
    extern __shared__ int cs_lds[];
    for (int k = threadIdx.x; k < K; k += 256) cs_lds[k] = 0;
    __syncthreads();
    const long r0 = (long)blockIdx.x * CS_BLOCK;
#pragma unroll
    for (int q = 0; q < CS_BLOCK / 256; ++q) {
        const long r = r0 + q * 256 + threadIdx.x;
        if (r < N) atomicAdd(&cs_lds[bmu[r]], 1);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += 256) table[(long)k * B + blockIdx.x] = cs_lds[k];
}

__global__ __launch_bounds__(256) void cs_scan_kernel(int* __restrict__ table, int K, int B, int* __restrict__ tot) {
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (k >= K) return;
    int* t = table + (long)k * B;
    int carry = 0;
    for (int b0 = 0; b0 < B; b0 += 64) {
        const int b = b0 + lane;
        const int v = b < B ? t[b] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
        if (b < B) t[b] = carry + incl - v;
        carry += __shfl(incl, 63, 64);
    }
    if (lane == 0) tot[k] = carry;
}

__global__ __launch_bounds__(CS_BLOCK) void cs_scatter_kernel(const int* __restrict__ bmu, long N, int K, int B, int bits,
                                                             const int* __restrict__ table, const int* __restrict__ tot,
                                                             int* __restrict__ skey, int* __restrict__ srow) {
    extern __shared__ int cs_lds[];                        // base[K]: next destination of every unit for this block
    __shared__ int wsum[CS_BLOCK / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // exclusive scan of tot over the units (every block does it: K <= 8 192), + this block's prefix inside the unit
    const int per = (K + CS_BLOCK - 1) / CS_BLOCK;         // units per thread, consecutive
    const int k0 = tid * per;
    int mine = 0;
    for (int i = 0; i < per; ++i) if (k0 + i < K) mine += tot[k0 + i];
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wave; ++w) before += wsum[w];
    int run = before + incl - mine;
    for (int i = 0; i < per; ++i)
        if (k0 + i < K) { cs_lds[k0 + i] = run + table[(long)(k0 + i) * B + blockIdx.x]; run += tot[k0 + i]; }
    __syncthreads();
    const long r = (long)blockIdx.x * CS_BLOCK + tid;
    const bool live = r < N;
    const int key = live ? bmu[r] : 0;
    // the lanes of this wave with the same unit
    unsigned long long peers = __ballot(live);
    for (int b = 0; b < bits; ++b) {
        const unsigned long long m = __ballot((key >> b) & 1);
        peers &= ((key >> b) & 1) ? m : ~m;
    }
    const int leader = live ? (int)__builtin_ctzll(peers) : lane;
    const int rank = (int)__builtin_popcountll(peers & ((1ull << lane) - 1));
    const int count = (int)__builtin_popcountll(peers);
    int dest = 0;
    for (int w = 0; w < CS_BLOCK / 64; ++w) {              // waves in row order: the sort is stable
        if (wave == w) {
            int old = 0;
            if (live && lane == leader) { old = cs_lds[key]; cs_lds[key] = old + count; }
            old = __shfl(old, leader, 64);
            dest = old + rank;
        }
        __syncthreads();
    }
    if (live) { skey[dest] = key; srow[dest] = (int)r; }
}

__host__ __device__ inline int seg_waves_per_block(int D1p) {
    int nw = SEG_MAX_WAVES;
    while (nw > 1 && (long)2 * nw * (D1p + 1) * 4 > 64 * 1024) nw >>= 1;
    return nw;
}
// upper levels: entries per wave for a list of n entries -- the smallest multiple of 8 in [SEG_CHUNK_UP, 64] with
// which nw waves cover the list, else SEG_CHUNK_UP
__host__ __device__ inline int seg_chunk_up(long n, int nw) {
    long c = ((n + nw - 1) / nw + 7) / 8 * 8;
    return (c <= 64) ? (int)(c < SEG_CHUNK_UP ? SEG_CHUNK_UP : c) : SEG_CHUNK_UP;
}
// entries the list left behind by a level with n entries, chunk c and nw waves per block
__host__ __device__ inline long seg_next_entries(long n, int c, int nw) { return 2 * ((n + (long)c * nw - 1) / ((long)c * nw)); }

// One chunk walk.  Entry i (< cnt <= 64) has key readlane(my_key, i), count readlane(my_cnt, i) and its vector at
// ROWS ? X + readlane(my_row, i) * D : list + i * D1p.  Complete runs go to SC, the two possible partials to
// slot0 / slot1 (rows of D1p floats); key0 / key1 return the slots' keys (-1 = unused).
template <bool ROWS, bool VEC2>
__device__ __forceinline__ void seg_walk(const float* __restrict__ X, const float* list, int my_key, int my_row,
                                         float my_cnt, int cnt, bool open_left, bool open_right, int D, int D1p,
                                         int accumulate, float* __restrict__ SC, float* __restrict__ cnt_dense,
                                         float* slot0, float* slot1, int& key0, int& key1) {
    const int lane = threadIdx.x & 63;
    key0 = -1; key1 = -1;
    bool slot1_zero = false;
    for (int f0 = 0; f0 < D; f0 += 256) {                 // 4 features per lane per sweep
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        int cur = __builtin_amdgcn_readfirstlane(my_key);
        int run_begin = 0;
        float run = 0.f;
        // feature of acc[j]: VEC2: f0 + 128*(j>>1) + 2*lane + (j&1);  scalar: f0 + lane + 64*j
        auto put = [&](float* dst, bool add, long unit = -1) {   // this lane's 4 sums (and the count) into one row
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
                    const int d = f0 + lane + 64 * j;
                    if (d < D) dst[d] = add ? dst[d] + acc[j] : acc[j];
                }
            }
            if (f0 == 0 && lane == 0) {
                const float c = add ? dst[D] + run : run;
                dst[D] = c;
                if (unit >= 0) cnt_dense[unit] = c;          // the counts once more, densely (the transform's count path)
            }
        };
        auto flush = [&](int end) {                       // the run [run_begin, end) of unit `cur` ends here
            if (cur >= 0) {
                const bool first = run_begin == 0, last = end == cnt;
                if (first && open_left) {
                    put(slot0, false);
                    key0 = cur;
                    if (last && open_right) { key1 = cur; slot1_zero = true; }
                } else if (last && open_right) {
                    put(slot1, false);
                    key1 = cur;
                } else {
                    put(SC + (long)cur * D1p, accumulate != 0, cur);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = 0.f;
            run = 0.f;
        };
        for (int i0 = 0; i0 < cnt; i0 += 8) {
            float v[8][4];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u;
                const int ii = i < cnt ? i : 0;
                const bool live = i < cnt && __builtin_amdgcn_readlane(my_key, ii) >= 0;
                const float* x = ROWS ? X + (long)__builtin_amdgcn_readlane(my_row, ii) * D : list + (long)ii * D1p;
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
                if (i < cnt) {
                    const int b = __builtin_amdgcn_readlane(my_key, i);
                    if (b != cur) { flush(i); cur = b; run_begin = i; }
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] += v[u][j];
                    run += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_cnt), i));
                }
            }
        }
        flush(cnt);
        if (slot1_zero) {                                 // one run, open on both sides: slot 1 = (key, 0)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = 0.f;
            run = 0.f;
            put(slot1, false);
        }
    }
}

// LEVEL0: keys = sorted BMUs, srow = sorted row ids, rows gathered from X (count 1 each).
// !LEVEL0: keys / vin = the previous level's partial list ([n] keys, [n][D1p] vectors, count in column D).
// VEC2 (D even): a lane owns feature pairs (8-byte accesses, one instruction per 128 features of a row).
// accumulate == 0: SC was zeroed and every unit is written at most once in the whole pass: plain stores.
// Dynamic LDS: blockDim.x / 64 > 1 ? 2 * waves * (D1p + 1) floats : none (a one-wave block writes its slots
// straight to the next list).
template <bool LEVEL0, bool VEC2>
__global__ __launch_bounds__(64 * SEG_MAX_WAVES) void runsum_kernel(const float* __restrict__ X,
                                                                   const int* __restrict__ keys,
                                                                   const int* __restrict__ srow,
                                                                   const float* __restrict__ vin, long n, int chunk,
                                                                   int D, int D1p, int accumulate, float* __restrict__ SC,
                                                                   float* __restrict__ cnt_dense, int* __restrict__ kout,
                                                                   float* __restrict__ vout) {
    extern __shared__ __attribute__((aligned(16))) float seg_lds[];
    const int C = LEVEL0 ? SEG_CHUNK : chunk;             // <= 64: lane i holds entry i
    const int lane = threadIdx.x & 63;
    const int nw = blockDim.x >> 6, wave = threadIdx.x >> 6;
    const long blk = blockIdx.x;
    const long p0 = (blk * nw + wave) * C;
    float* slots = seg_lds;                               // [2 * nw][D1p]
    int* slot_keys = (int*)(seg_lds + (long)2 * nw * D1p);   // [2 * nw]
    const bool direct = nw == 1;
    int k0 = -1, k1 = -1;
    if (p0 < n) {
        const int cnt = (int)((n - p0 < C) ? (n - p0) : C);
        // lane i < cnt holds the i-th (key, row, count) of the chunk
        const int my_key = lane < cnt ? keys[p0 + lane] : -1;
        const int my_row = (LEVEL0 && lane < cnt) ? srow[p0 + lane] : 0;
        const float my_cnt = LEVEL0 ? 1.0f : ((lane < cnt && my_key >= 0) ? vin[(p0 + lane) * D1p + D] : 0.0f);
        const int first_key = __builtin_amdgcn_readfirstlane(my_key);
        const int last_key = __builtin_amdgcn_readlane(my_key, cnt - 1);
        const int prev_key = p0 > 0 ? keys[p0 - 1] : -2;
        const int next_key = p0 + cnt < n ? keys[p0 + cnt] : -2;
        const bool open_left = first_key >= 0 && __builtin_amdgcn_readfirstlane(prev_key) == first_key;
        const bool open_right = last_key >= 0 && __builtin_amdgcn_readfirstlane(next_key) == last_key;
        float* s0 = direct ? vout + (2 * blk) * (long)D1p : slots + (long)(2 * wave) * D1p;
        float* s1 = direct ? vout + (2 * blk + 1) * (long)D1p : slots + (long)(2 * wave + 1) * D1p;
        seg_walk<LEVEL0, VEC2>(X, LEVEL0 ? nullptr : vin + p0 * (long)D1p, my_key, my_row, my_cnt, cnt, open_left,
                               open_right, D, D1p, accumulate, SC, cnt_dense, s0, s1, k0, k1);
    }
    if (direct) {
        if (lane == 0 && p0 < n) { kout[2 * blk] = k0; kout[2 * blk + 1] = k1; }
        return;
    }
    if (lane == 0) { slot_keys[2 * wave] = k0; slot_keys[2 * wave + 1] = k1; }
    __syncthreads();
    if (wave != 0) return;
    // wave 0: the block's 2 * nw slots as one chunk.  A filled slot 0 of the first wave continues the previous
    // block by construction, a filled slot 1 of the last wave is continued by the next one.
    const int cnt2 = 2 * nw;
    const int my_key2 = lane < cnt2 ? slot_keys[lane] : -1;
    const float my_cnt2 = (lane < cnt2 && my_key2 >= 0) ? slots[(long)lane * D1p + D] : 0.0f;
    const bool open_left2 = __builtin_amdgcn_readfirstlane(my_key2) >= 0;
    const bool open_right2 = __builtin_amdgcn_readlane(my_key2, cnt2 - 1) >= 0;
    int g0 = -1, g1 = -1;
    seg_walk<false, VEC2>(nullptr, slots, my_key2, 0, my_cnt2, cnt2, open_left2, open_right2, D, D1p, accumulate, SC,
                          cnt_dense, vout + (2 * blk) * (long)D1p, vout + (2 * blk + 1) * (long)D1p, g0, g1);
    if (lane == 0) { kout[2 * blk] = g0; kout[2 * blk + 1] = g1; }
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
    int hex, base_nt, ncls; // hexagonal topology: nt = ncls * base_nt, one copy of the terms per parity class
    int swapped;            // mexican_hat + compact_support on the rectangular topology: row stage first (see below)
};

__device__ __forceinline__ double neigh_exp(double delta2, const NeighParams& p) {
    if (p.wide) return exp(-delta2 / p.d);
    float a = -(float)delta2 / (float)p.d;
    return (double)(float)exp((double)a);
}
// the reference's support mask, literally: n > c - sigma  and  n < c + sigma  in float64 (neighborhoods.py:29-31,
// 50-54, 105-110) on the units' coordinates.  c -/+ sigma is rounded before the compare, so a sigma within an ulp of
// a lattice distance (asymptotic decay produces them: 5 / (1 + 2/3) = 3.0000000000000004) decides the boundary unit
// differently than |n - c| < sigma would, and differently for different c.  n, c: absolute coordinates (hexagonal x:
// half-unit row offsets included -- which is why compact_support needs the fourth parity class).
__device__ __forceinline__ double neigh_box(double n, double c, const NeighParams& p) {
    return (n > c - p.sigma && n < c + p.sigma) ? 1.0 : 0.0;
}
__device__ __forceinline__ double neigh_round(double v, const NeighParams& p) { return p.wide ? v : (double)(float)v; }

// value of factor `which` (0 = row factor Px, 1 = column factor Py) of base term t for unit coordinate n and
// BMU coordinate c (absolute: on the hexagonal topology the x coordinates carry their rows' half-unit offsets)
__device__ double neigh_factor(int which, int t, double n, double c, const NeighParams& p) {
    const double dl = n - c;
    const double d2 = dl * dl;
    switch (p.kind) {
    case 0: {   // gaussian
        double e = neigh_exp(d2, p);
        if (p.compact) e *= neigh_box(n, c, p);
        return e;
    }
    case 1: {   // mexican hat: (ex(1-2px/d)) * ey  -  ex * ((2py/d) ey)
        double e = neigh_exp(d2, p);
        double q = p.wide ? (2.0 / p.d) * d2 : (double)((float)(2.0 / p.d) * (float)d2);
        if (!p.compact) {
            if (t == 0) return which == 0 ? neigh_round(e * (1.0 - q), p) : e;
            return which == 0 ? -e : neigh_round(q * e, p);
        }
        // compact_support as the reference computes it (neighborhoods.py:69-71, :91-93): px is multiplied by a mask M,
        // py is not, so with A = ex (1 - 2 px / d), Q = (2 py / d) ey:
        //     h = M (A ey - ex Q) + (1 - M)(ey - Q)
        // Hexagonal (generic): M = mx(i; ci) my(j; cj), four separable terms
        //     [mx A][my ey]  +  [-mx ex][my Q]  +  [-mx][my (ey - Q)]  +  [1][ey - Q].
        // Rectangular: M = m1(i; ci) m2(i; cj) -- the second mask compares the ROW index with the BMU's COLUMN --:
        // the same four terms with my = 1; m2 is applied between the row stage and the column stage (`swapped`).
        const double m = neigh_box(n, c, p);
        if (which == 0) {
            if (t == 0) return neigh_round(m * e * (1.0 - q), p);
            return t == 1 ? -m * e : t == 2 ? -m : 1.0;
        }
        const double my = p.hex ? m : 1.0;
        const double qe = neigh_round(q * e, p);
        if (t == 0) return my * e;
        if (t == 1) return my * qe;
        return (t == 2 ? my : 1.0) * neigh_round(e - qe, p);
    }
    case 2:     // bubble
        return neigh_box(n, c, p);
    default: {  // triangle
        double v = p.sigma - fabs(dl);
        if (v < 0.0) v = 0.0;
        if (p.compact) v *= neigh_box(n, c, p);
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
    if (p.swapped) {
        // row stage first (X == Y): P1 = the row factors [nt][X][X] (Fx_t[i][a], with eta), P2 = the column factors
        // [Y][nt*Y] (Gy_t[j][b] at column t*Y + b)
        if (id < n1) {
            int a = id % p.X;
            long r = id / p.X;
            int i = r % p.X;
            int t = r / p.X;
            double v = neigh_factor(0, t, (double)i, (double)a, p);
            P1[id] = p.wide ? (float)(v * p.eta) : (float)v * (float)p.eta;
        } else if (id < n1 + n2) {
            long q = id - n1;
            int col = q % ((long)p.nt * p.Y);
            int j = q / ((long)p.nt * p.Y);
            int t = col / p.Y, b = col % p.Y;
            P2[q] = (float)neigh_factor(1, t, (double)j, (double)b, p);
        }
        return;
    }
    if (id < n1) {
        int b = id % p.Y;
        long r = id / p.Y;
        int j = r % p.Y;
        int t = r / p.Y;
        double v = neigh_factor(1, t % p.base_nt, (double)j, (double)b, p);
        if (p.hex) {                                   // class indicator on (s(j), s(cj)): s = the row is shifted by -0.5
            const int cls = t / p.base_nt;
            const int sj = ((p.Y - 1 - j) & 1) == 0, sb = ((p.Y - 1 - b) & 1) == 0;
            bool in;
            if (p.ncls == 3) in = cls == 0 ? sj == sb : cls == 1 ? (sj == 0 && sb == 1) : (sj == 1 && sb == 0);
            else in = cls == 0 ? (sj == 0 && sb == 0) : cls == 1 ? (sj == 0 && sb == 1) : cls == 2 ? (sj == 1 && sb == 0)
                                                                                                  : (sj == 1 && sb == 1);
            if (!in) v = 0.0;
        }
        P1[id] = (float)v;
    } else if (id < n1 + n2) {
        long q = id - n1;
        int col = q % ((long)p.nt * p.X);
        int i = q / ((long)p.nt * p.X);
        int t = col / p.X, a = col % p.X;
        const int cls = t / p.base_nt;
        // x coordinates of the class: unit rows of classes 2, 3 and BMU rows of classes 1, 3 are shifted by -0.5
        // (three classes: only the difference matters, the both-shifted rows share class 0 with the unshifted ones)
        const double off_n = p.hex && (cls == 2 || cls == 3) ? -0.5 : 0.0;
        const double off_c = p.hex && (cls == 1 || cls == 3) ? -0.5 : 0.0;
        double v = neigh_factor(0, t % p.base_nt, (double)i + off_n, (double)a + off_c, p);
        P2[q] = p.wide ? (float)(v * p.eta) : (float)v * (float)p.eta;
    }
}

// `swapped` pipeline, between its two stages: V[i][t][b][:] *= m2(i, b) for the masked terms t < nt_masked, where
// m2 is the reference's second mask on px -- row index i against BMU column b (neighborhoods.py:70).
__global__ __launch_bounds__(256) void mask_rows_kernel(NeighParams p_val, const NeighParams* __restrict__ p_dev,
                                                        float* __restrict__ V, int nt_masked, int D1p) {
    const NeighParams p = p_dev ? *p_dev : p_val;
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    const long per_i = (long)p.nt * p.Y * D1p;
    const long total = (long)p.X * per_i;
    if (id >= total) return;
    const int i = (int)(id / per_i);
    const long r = id - (long)i * per_i;
    const int t = (int)(r / ((long)p.Y * D1p));
    const int b = (int)((r / D1p) % p.Y);
    if (t < nt_masked && neigh_box((double)i, (double)b, p) == 0.0) V[id] = 0.0f;
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
constexpr int LM_NARROW = 8;    // a last column tile this narrow goes to leftmul_narrow_f32_kernel (VALU) instead of a
                                // mostly empty 128-wide MFMA tile (64x64x32: 32 columns as a partial MFMA tile 11 us, as VALU 14)

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

// ranges == nullptr: the whole of H (one segment [0, Ri)).  C columns are computed; rows of M are ld, rows of OUT ldo floats apart.
// BM = rows of H per workgroup: 128 (waves 2 x 2, each 64 rows x 64 columns) or 64 (waves 1 x 4, each 64 rows x 32
// columns) for maps of at most 64 rows, whose 128-row tiles would be half empty.
template <int BM>
__global__ __launch_bounds__(256) void leftmul_f32_kernel(const float* __restrict__ H, int Ro, int Ri,
                                                          const float* __restrict__ M, long m_batch_stride,
                                                          float* __restrict__ OUT, long o_batch_stride, long C,
                                                          long ld, long ldo, const int2* __restrict__ ranges, int nseg,
                                                          int segw) {
    static_assert(BM == 128 || BM == 64, "leftmul: 128- or 64-row tiles");
    constexpr int TC = BM / 64;                          // 32-column tiles per wave: 2 (64 columns) or 1 (32 columns)
    constexpr int HQ = BM / 32;                          // 4-float pieces of the H tile per thread
    __shared__ float Hs[BM][LM_BK + 1];
    __shared__ __attribute__((aligned(16))) float Ms[LM_BK][LM_BN + 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, col = lane & 31;
    const int wr = BM == 128 ? wave >> 1 : 0;            // this wave's 64 rows ...
    const int wcol = BM == 128 ? (wave & 1) * 64 : wave * 32;   // ... and its first column inside the 128-column tile
    const long c0 = (long)blockIdx.x * LM_BN;
    const int i0 = blockIdx.y * BM;
    const float* Mb = M + (long)blockIdx.z * m_batch_stride;
    float* Ob = OUT + (long)blockIdx.z * o_batch_stride;

    f32x16 acc[2][TC];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int t = 0; t < TC; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][t][r] = 0.0f;

    // this thread's share of the staged chunk: four 4-float pieces of each operand tile -- piece q of the H tile
    // (128 x 32) is row (tid + 256 q) >> 3, columns 4 ((tid + 256 q) & 7) .. + 3; of the M tile (32 x 128) row
    // (tid + 256 q) >> 5, columns 4 ((tid + 256 q) & 31) .. + 3.  A piece that lies inside the operand and is
    // 16-byte aligned is ONE global_load_dwordx4 (the common case: 8 loads per thread and chunk where the scalar
    // form needed 32 loads with their 64-bit address arithmetic); otherwise its four floats are loaded one by one.
    f32x4 hreg[HQ], mreg[4];
    auto fetch = [&](int r0, int kend) {               // chunk [r0, r0 + 32) clipped to columns < kend
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = tid + q * 256;
            if (q < HQ) {
                const int i = idx >> 3, k = (idx & 7) * 4;
                const float* src = H + (long)(i0 + i) * Ri + r0 + k;
                // (the piece's own address decides: a banded chunk starts at sg * segw + lo, any float offset)
                if (((uintptr_t)src & 15) == 0 && i0 + i < Ro && r0 + k + 3 < kend) hreg[q] = *(const f32x4*)src;
                else
#pragma unroll
                    for (int e = 0; e < 4; ++e) hreg[q][e] = (i0 + i < Ro && r0 + k + e < kend) ? src[e] : 0.0f;
            }
            {
                const int kk = idx >> 5, c = (idx & 31) * 4;
                const float* src = Mb + (long)(r0 + kk) * ld + c0 + c;
                if (((uintptr_t)src & 15) == 0 && r0 + kk < kend && c0 + c + 3 < C) mreg[q] = *(const f32x4*)src;
                else
#pragma unroll
                    for (int e = 0; e < 4; ++e) mreg[q][e] = (r0 + kk < kend && c0 + c + e < C) ? src[e] : 0.0f;
            }
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = tid + q * 256;
            if (q < HQ) {
                const int i = idx >> 3, k = (idx & 7) * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) Hs[i][k + e] = hreg[q][e];      // (33-float rows: conflict-free reads, scalar writes)
            }
            *(f32x4*)&Ms[idx >> 5][(idx & 31) * 4] = mreg[q];               // 132-float rows: 16-byte aligned
        }
    };

    // the chunks to walk (uniform over the workgroup): every segment is walked over the union [lo, hi) of
    // the segments' nonzero ranges, LM_BK columns at a time, clipped to hi
    int lo = 0, hi = Ri, ns = 1, sw = Ri;
    if (ranges != nullptr) {
        ns = nseg; sw = segw; lo = segw; hi = 0;
        for (int sg = 0; sg < nseg; ++sg) {
            const int2 v = ranges[(blockIdx.y * BM / LM_BM) * nseg + sg];   // {segw - first nonzero, last nonzero + 1} of the 128-row block
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
            float av[2], bv[TC];
#pragma unroll
            for (int a = 0; a < 2; ++a) av[a] = Hs[wr * 64 + a * 32 + col][k + half];
#pragma unroll
            for (int t = 0; t < TC; ++t) bv[t] = Ms[k + half][wcol + t * 32 + col];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int t = 0; t < TC; ++t)
                    acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a], bv[t], acc[a][t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int t = 0; t < TC; ++t) {
            long c = c0 + wcol + t * 32 + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int i = i0 + wr * 64 + a * 32 + mfma32_row(r, half);
                if (i < Ro && c < C) Ob[(long)i * ldo + c] = acc[a][t][r];
            }
        }
}

// The columns [c_begin, C) of the same product when they are too few for a 128-wide MFMA tile (the count
// column and its padding behind input_len = 128: D1p = 132 = 128 + 4): a VALU kernel, thread = (row, column
// parity), k-ordered fmaf chains over the same chunks in the same order -- bit for bit what the MFMA tile
// computes (v_mfma_f32_32x32x2_f32 is that chain; skipped chunks add exact zeros) at a few percent of its time.
// Grid = (1, row blocks, batch).
template <int NQ>                                       // columns handled per thread: live <= 2 * NQ <= LM_NARROW
__global__ __launch_bounds__(256) void leftmul_narrow_f32_kernel(const float* __restrict__ H, int Ro, int Ri,
                                                                 const float* __restrict__ M, long m_batch_stride,
                                                                 float* __restrict__ OUT, long o_batch_stride, long C,
                                                                 long ld, long ldo, long c_begin,
                                                                 const int2* __restrict__ ranges, int nseg, int segw) {
    __shared__ float Hs[LM_BM][LM_BK + 1];
    __shared__ float Ms[LM_BK][2 * NQ];
    const int tid = threadIdx.x;
    const int i0 = blockIdx.y * LM_BM;
    const float* Mb = M + (long)blockIdx.z * m_batch_stride;
    float* Ob = OUT + (long)blockIdx.z * o_batch_stride;
    const int live = (int)(C - c_begin);                // <= LM_NARROW
    const int row = tid & (LM_BM - 1), par = tid >> 7;  // 256 threads = 128 rows x 2 column parities
    float acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = 0.0f;

    int lo = 0, hi = Ri, ns = 1, sw = Ri;
    if (ranges != nullptr) {
        ns = nseg; sw = segw; lo = segw; hi = 0;
        for (int sg = 0; sg < nseg; ++sg) {
            const int2 v = ranges[blockIdx.y * nseg + sg];
            if (v.y > 0) { lo = min(lo, ((segw - v.x) / LM_BK) * LM_BK); hi = max(hi, min(v.y, segw)); }
        }
        if (hi <= lo) lo = hi = 0;
    }
    const int per = (hi - lo + LM_BK - 1) / LM_BK;
    const int total = per * ns;
    for (int c = 0; c < total; ++c) {
        const int sg = c / per, j = c - sg * per;
        const int r0 = sg * sw + lo + j * LM_BK, kend = sg * sw + hi;
        __syncthreads();
        for (int idx = tid; idx < LM_BM * LM_BK; idx += 256) {
            const int i = idx >> 5, k = idx & 31;
            Hs[i][k] = (i0 + i < Ro && r0 + k < kend) ? H[(long)(i0 + i) * Ri + r0 + k] : 0.0f;
        }
        for (int idx = tid; idx < LM_BK * 2 * NQ; idx += 256) {
            const int k = idx / (2 * NQ), cc = idx - k * (2 * NQ);
            Ms[k][cc] = (r0 + k < kend && cc < live) ? Mb[(long)(r0 + k) * ld + c_begin + cc] : 0.0f;
        }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < LM_BK; ++k) {
            const float hv = Hs[row][k];
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc[q] = __builtin_fmaf(hv, Ms[k][2 * q + par], acc[q]);
        }
    }
    const int i = i0 + row;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
        if (i < Ro && 2 * q + par < live) Ob[(long)i * ldo + c_begin + 2 * q + par] = acc[q];
}

// ---- the update as the reference states it: num = g^T x, den = sum_n g  (xpysom.py:434-441) -----------------------
// The FAITHFUL form, kept beside the bucketed one for cross-checking and for the record: a K x N x D
// float32 MFMA GEMM whose A operand g[n, k] = sum_t Px_t[i_k, ci_n] Py_t[j_k, cj_n] is generated from the
// neighbourhood tables on the way into LDS and never exists in memory.  Same tiling as leftmul_f32_kernel (128 units
// x 128 features per workgroup, 32-row chunks of the samples as the k axis).  2 N K D flop: 17.6 TFLOP per epoch at
// 256 x 256 x 128 with 1 Mi rows, where the bucketed form needs 8.7 GFLOP.
__global__ __launch_bounds__(256) void faithful_update_f32_kernel(const float* __restrict__ X, const int* __restrict__ bmu,
                                                                  long N, int D, int D1p, int Xm, int Ym, int nt,
                                                                  const float* __restrict__ P1, const float* __restrict__ P2,
                                                                  float* __restrict__ ACC) {
    __shared__ float Gs[LM_BM][LM_BK + 1];               // g[unit][row of the chunk]
    __shared__ __attribute__((aligned(16))) float Xs[LM_BK][LM_BN + 4];
    __shared__ int ci_s[LM_BK], cj_s[LM_BK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, col = lane & 31;
    const int wr = wave >> 1, wc = wave & 1;
    const long c0 = (long)blockIdx.x * LM_BN;            // first feature column of the tile
    const int u0 = blockIdx.y * LM_BM;                   // first unit of the tile
    const int K = Xm * Ym;
    // this thread generates g for the 16 units (tid >> 5) + 8 q and the chunk row tid & 31
    int iu[16], ju[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int u = u0 + (tid >> 5) + 8 * q;
        iu[q] = u < K ? u / Ym : -1;
        ju[q] = u < K ? u % Ym : 0;
    }
    float den[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) den[q] = 0.0f;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][t][r] = 0.0f;

    for (long r0 = 0; r0 < N; r0 += LM_BK) {
        __syncthreads();                                 // everyone is done with the previous chunk
        if (tid < LM_BK) {
            const long n = r0 + tid;
            const int b = n < N ? bmu[n] : -1;
            ci_s[tid] = b >= 0 ? b / Ym : -1;
            cj_s[tid] = b >= 0 ? b % Ym : 0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {                    // the rows of the chunk: 32 x 128 features (16-byte pieces)
            const int idx = tid + q * 256;
            const int kk = idx >> 5, c = (idx & 31) * 4;
            const long n = r0 + kk;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (n < N && c0 + c + e < D) v[e] = X[n * D + c0 + c + e];
            *(f32x4*)&Xs[kk][c] = v;
        }
        __syncthreads();
        {
            const int k = tid & 31;
            const int ci = ci_s[k], cj = cj_s[k];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                float g = 0.0f;
                if (ci >= 0 && iu[q] >= 0)
                    for (int t = 0; t < nt; ++t)
                        g = __builtin_fmaf(P2[(long)iu[q] * nt * Xm + t * Xm + ci], P1[((long)t * Ym + ju[q]) * Ym + cj], g);
                Gs[(tid >> 5) + 8 * q][k] = g;
                den[q] += g;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < LM_BK; k += 2) {
            float av[2], bv[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) av[a] = Gs[wr * 64 + a * 32 + col][k + half];
#pragma unroll
            for (int t = 0; t < 2; ++t) bv[t] = Xs[k + half][wc * 64 + t * 32 + col];
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
            const long c = c0 + wc * 64 + t * 32 + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int u = u0 + wr * 64 + a * 32 + mfma32_row(r, half);
                if (u < K && c < D) ACC[(long)u * D1p + c] = acc[a][t][r];
            }
        }
    if (blockIdx.x == 0) {                               // den: the 32 threads of a half-wave hold one unit's partial sums
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            float s = den[q];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            const int u = u0 + (tid >> 5) + 8 * q;
            if ((tid & 31) == 0 && u < K) ACC[(long)u * D1p + D] = s;
        }
    }
}

// ---- the count column's own transform ---------------------------------------------------------------
// den = sum_t Px_t C Py_t^T with C[a][b] = the count of unit (a, b): two X x Y x {Y, nt X} products, 33 MFLOP each
// at 256 x 256.  Kept out of the batched MFMA transform, whose column tiles are 128 wide: input_len = 128 would
// give the count a second, empty tile.  OUT[i*oi + j*oj] = sum_k A[i*ai + k*ak] * B[k*bk + j*bj], k ascending,
// one fmaf chain per output: exactly the chain the MFMA tiles computed for this column (skipped zero bands
// add exact zeros), so the denominator is bit for bit what it was.  The operands are columns of wide arrays
// (one float per 528-byte row) or small dense matrices, so the kernel is latency-bound: a workgroup (16 x 16 outputs,
// one per thread: 256 workgroups at 256 x 256) issues the loads of SG_KC = 256 k-steps of both operands at once --
// one round trip per 256 k-steps -- before it multiplies.
constexpr int SG_KC = 256, SG_T = 16;   // k-steps staged at once; output tile side (one output per thread)
__global__ __launch_bounds__(256) void strided_gemm_f32_kernel(const float* __restrict__ A, long ai, long ak,
                                                               const float* __restrict__ B, long bk, long bj,
                                                               float* __restrict__ OUT, long oi, long oj, int M, int N,
                                                               int Kd, long a_batch, long b_batch, long o_batch) {
    __shared__ float As[SG_T][SG_KC + 1];
    __shared__ float Bs[SG_KC][SG_T + 1];
    const int tx = threadIdx.x & (SG_T - 1), ty = threadIdx.x / SG_T;   // 16 x 16
    const int i0 = blockIdx.y * SG_T, j0 = blockIdx.x * SG_T;
    A += (long)blockIdx.z * a_batch; B += (long)blockIdx.z * b_batch; OUT += (long)blockIdx.z * o_batch;
    // the unit-stride axis of each operand goes along consecutive threads (either k or the row / column index)
    const bool a_k_fast = ak <= ai, b_j_fast = bj <= bk;
    float acc = 0.f;
    for (int k0 = 0; k0 < Kd; k0 += SG_KC) {
        __syncthreads();
        constexpr int PER = SG_T * SG_KC / 256;                  // elements of each operand per thread
        float ra[PER], rb[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int e = q * 256 + threadIdx.x;
            const int ar = a_k_fast ? e / SG_KC : e % SG_T, akk = a_k_fast ? e % SG_KC : e / SG_T;
            ra[q] = (i0 + ar < M && k0 + akk < Kd) ? A[(long)(i0 + ar) * ai + (long)(k0 + akk) * ak] : 0.0f;
            const int bc = b_j_fast ? e % SG_T : e / SG_KC, bkk = b_j_fast ? e / SG_T : e % SG_KC;
            rb[q] = (k0 + bkk < Kd && j0 + bc < N) ? B[(long)(k0 + bkk) * bk + (long)(j0 + bc) * bj] : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int e = q * 256 + threadIdx.x;
            const int ar = a_k_fast ? e / SG_KC : e % SG_T, akk = a_k_fast ? e % SG_KC : e / SG_T;
            As[ar][akk] = ra[q];
            const int bc = b_j_fast ? e % SG_T : e / SG_KC, bkk = b_j_fast ? e / SG_T : e % SG_KC;
            Bs[bkk][bc] = rb[q];
        }
        __syncthreads();
        const int kn = Kd - k0 < SG_KC ? Kd - k0 : SG_KC;
#pragma unroll 8
        for (int k = 0; k < kn; ++k) acc = __builtin_fmaf(As[ty][k], Bs[k][tx], acc);
    }
    const int i = i0 + ty, j = j0 + tx;
    if (i < M && j < N) OUT[(long)i * oi + (long)j * oj] = acc;
}

// ---- merge: W = where(den != 0, num/den, W)  (xpysom.py:446-455) ---------------------------------
// (Wp, inv: the exact mode's copy of the codebook in patch order -- inv[unit] = position, som_common.hpp -- kept in step)
__global__ __launch_bounds__(256) void merge_kernel(float* __restrict__ W, const float* __restrict__ ACC,
                                                    long K, int D, int D1p, float* __restrict__ Wp = nullptr,
                                                    const int* __restrict__ inv = nullptr) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= K * D) return;
    long k = i / D;
    int d = (int)(i - k * D);
    float den = ACC[k * D1p + D];
    if (den != 0.0f) {
        const float v = ACC[k * D1p + d] / den;
        W[i] = v;
        if (Wp != nullptr) {
            Wp[(long)inv[k] * D + d] = v;
        }
    }
}

// ---- quantization error: sum_n |x_n - W[bmu_n]|  (xpysom.py:703-705) -----------------------------
// (persistent: a wave walks rows wave, wave + waves, ... and keeps its sum in a double; ONE atomic per workgroup -- a workgroup
//  per four rows put 262 144 double atomics of a million-row call on one address: 3.2 ms of a 0.3 ms kernel)
__global__ __launch_bounds__(256) void qe_kernel(const float* __restrict__ X, const int* __restrict__ bmu,
                                                 const float* __restrict__ W, long N, int D,
                                                 double* __restrict__ sum_out) {
    __shared__ double part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long waves = (long)gridDim.x * 4;
    double acc = 0.0;
    for (long row = (long)blockIdx.x * 4 + wave; row < N; row += waves) {
        const float* x = X + row * D;
        const float* w = W + (long)bmu[row] * D;
        float s = 0.0f;
        for (int k = lane; k < D; k += 64) { float df = x[k] - w[k]; s = __builtin_fmaf(df, df, s); }
        s = wave_sum(s);
        acc += (double)__builtin_sqrtf(s);
    }
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(sum_out, (part[0] + part[1]) + (part[2] + part[3]));
}

}  // namespace somhip
