// libsomhip.so -- C ABI over the gfx950 kernels (include/somhip.h).
// Host side only orchestrates: buffers, launch geometry checks, the per-epoch kernel
// sequence on one HIP stream, hipEvent timing.  No compute happens on the host and
// there is no CPU fallback: every path either launches the HIP kernels or fails.
#include <hip/hip_runtime.h>

#include <cstring>

#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "../../include/somhip.h"
#include "../../include/somhip_test.h"
#include "bmu_bf16_k16.hpp"
#include "bmu_bf16_tiled.hpp"
#include "bmu_bf16_wide.hpp"
#include "bmu_exact.hpp"
#include "exact_skip.hpp"
#include "exact_policy.hpp"
#include "exact_skip_wide.hpp"
#include "bmu_f32.hpp"
#include "bmu_f32_res.hpp"
#include "bmu_f32_tiled.hpp"
#include "bmu_pairwise.hpp"
#include "update.hpp"

using namespace somhip;

namespace {

thread_local std::string g_create_error;

// Environment switches.  Four are for users (SOM_VERIFY, SOM_DEBUG, SOM_EXACT_SKIP, SOM_GRAPH).  Everything
// else -- A/B switches of kernel variants, forced pass sizes, refused allocations -- belongs to the tests and the tools and is
// read ONLY when SOM_TEST_HOOKS=1 is set (tests/conftest.py sets it): a stray variable in a user's environment changes nothing.
const char* dev_env(const char* name) {
    static const bool on = [] { const char* v = std::getenv("SOM_TEST_HOOKS"); return v != nullptr && std::atoi(v) != 0; }();
    return on ? std::getenv(name) : nullptr;
}

struct EventPair { hipEvent_t a, b; int kernel; };

}  // namespace

// RCCL, bound at run time (dlopen): the library has no link-time dependency on it, and in a process that also
// runs torch the SAME copy torch loaded is used.  Only what the one exchange step of the path needs.
struct som_nccl_id { char internal[128]; };
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(som_nccl_id*) = nullptr;
    int (*CommInitRank)(void**, int, som_nccl_id, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
static RcclApi g_rccl;
static std::string g_rccl_error;

struct som_handle {
    som_config cfg{};
    void* comm = nullptr;        // ncclComm_t of som_comm_init (NULL: single GPU, or the host does the all-reduce)
    int comm_world = 1;
    hipStream_t comm_stream = nullptr;   // the blockwise all-reduce runs here, under the transform of the next block
    hipEvent_t ev_block = nullptr, ev_comm = nullptr;
    int X = 0, Y = 0, K = 0, D = 0, D1p = 0;
    int ks32 = 0;            // bf16, 16x16x32 shape: ceil(D/32)
    bool tiled = false;      // bf16, input_len > 128: two-sided tiling (bmu_bf16_tiled.hpp)
    bool f16 = false;        // precision f16: _Float16 operands instead of __bf16 (the same kernels, som_common.hpp)
    bool exact = false;      // precision 'exact': MFMA screen + float32 re-score of the candidates (bmu_exact.hpp)
    // exact mode: the operand images in PATCH ORDER (som_common.hpp; ex_perm[position] = unit, ex_inv[unit] = position) -- prepared
    // from a permuted copy of the codebook; wf_patch: the order the float32 image is in right now (the float32 kernels
    // proper -- fallback rows, top-2, analysis calls -- want the units' own order and rebuild it)
    bool ex_patch = false, wf_patch = false;
    bool ex_sub44 = false;                              // ... with every group an 8 x 8 patch in 4 x 4 blocks (patch_order)
    int* ex_perm = nullptr;
    int* ex_inv = nullptr;
    float* Wp = nullptr;     // [K][D] codebook in patch order
    float* wsq_p = nullptr;  // [K]   its |w|^2 (the float32 kernel's own values, permuted)
    struct ExactScratch {
        uint32_t* gmin = nullptr;            // [n_groups][stride] group minima of the chunk being screened (sparse: see gflags)
        unsigned long long* gflags = nullptr;   // [stride / 64][n_groups] which rows' minima the screen stored
        long stride = 0;                     //   rows per group line (a chunk of the row set, padded)
        int* rowcnt = nullptr;               // [stride] candidate groups of every row of the last pass (som_exact_last_counts)
        int* rowarg = nullptr;               // [stride] the group round 1 scored for the row (-1: none)
        float* seed = nullptr;               // [stride] exact_seed_kernel: the cap on the screen's keep threshold
        bool seed_on = true;                 // SOM_EXACT_SEED=0: no seed (A/B)
        bool seed_live = false;              // this pass's screen reads the seed
        int two_round = -1;                  // SOM_EXACT_TWO_ROUND=0|1 forces the one- / two-round re-score (default: two rounds beyond 128 features)
        int *fb_list = nullptr, *fb_ids = nullptr;
        int* ctr = nullptr;                  // gcount | gstart of round 2 | fb_count | n_tiles | overflow (zeroed per pass)
        int* plist = nullptr;                // [n_groups][stride] rows bucketed by candidate group
        int4* tile_tab = nullptr;            // re-score tiles: (group, first list entry, rows)
        long max_tiles = 0;
        float* fbX = nullptr;                // fallback rows, dense, for the float32 kernel
        long fb_cap = 0;
        int* fb_count_host = nullptr;        // pinned
        hipEvent_t fb_ready = nullptr;       // recorded behind the counter's copy
        int64_t rows_total = 0, rows_fallback = 0, chunks = 0;   // som_exact_stats
        int64_t blocks_run = 0, blocks_total = 0;                // som_exact_skip_stats: (256-row tile, 16-unit block) blocks of the screens
        long pass_rows_override = 0;
        long hook_refuse_above = 0, hook_pairs = 0; bool hook_refuse_skip = false;   // test hooks, read once in som_create (SOM_TEST_HOOKS=1)
        long stride_cap = 0;              // rows per pass the device had memory for (0: no allocation was ever refused)
        long pairs = 64;                  // capacity of a pass: (row, group) pairs per row on average (exact_reserve)
        // block skipping (exact_skip.hpp): resident rows from their second epoch on, input_len <= 128
        int skip_mode = 1;                // SOM_EXACT_SKIP: 0 off, 1 on for maps of >= 4096 units (default), 2 on for every map of >= 2 groups (tests), 3: plan, keep everything
        bool skip_live = false;           // this launch plans and skips
        int skip_cooldown = 0;            // launches to run without a plan (the last two plans kept > 97 % of the blocks)
        int skip_idle = 0;                // plans in a row that kept > 97 % of the blocks
        int skip_pause = 2;               // launches the next pause lasts (doubles while the plans stay idle)
        bool sub_blocks = true;           // SOM_EXACT_SUBBLOCKS=0: the plan stops at the groups (A/B)
        // the SORTED PASS: rows in the order of their (pseudo) last BMU's patch (position -> row: `order`), with sorted copies of
        // the half image, its second half, the float32 rows and the norms.  srt[0]: the RESIDENT rows (all passes; valid for
        // (res_rows, res_n), re-sorted when the order has gone stale -- res_* below --, not every epoch); srt[1]: ONE pass of
        // a TRANSIENT row set (query rows, streamed chunks: sorted by the scout of exact_skip.hpp, used once).
        struct SortedRows {
            long cap = 0;                 // positions the buffers hold (each pass padded to the tile)
            int* order = nullptr;
            __bf16* Xb_s = nullptr;
            __bf16* Xl_s = nullptr;       // ... the rows' second half image (the refinement pass): allocated when it first engages
            bool xl_filled = false;       //     ... and written by a gather since
            float *Xf_s = nullptr, *xsq_s = nullptr, *xerr_s = nullptr, *seed_s = nullptr, *sU_s = nullptr;
            int* lastpos_s = nullptr;     // position (patch order) of every sorted row's (pseudo) last BMU
        } srt[2];
        bool cen_ready = false;           // both centroid levels are allocated
        long sk_stride = 0;               // rows per pass the per-pass plan buffers hold
        const void* res_rows = nullptr; long res_n = -1;
        bool res_valid = false;
        int res_every = 0;                // SOM_EXACT_RESORT=n: re-sort every n-th planned epoch (0: when the order has gone stale)
        int res_since = 0;                // planned epochs since the last sort
        double res_share_sort = 1.0;      // executed share of the first epoch after the last sort
        bool res_l2_sort = false, res_l2_last = false;   // ... whether level 2 ran in that epoch / in the last one (shares compare like with like)
        int res_forced = 8;               // planned epochs after which the rows are sorted in any case (doubles after a forced sort that did not pay)
        double res_share_last = 1.0;      // ... of the last planned epoch
        int64_t resorts = 0, planned = 0; // som_exact_resident_stats
        int *sk_keys = nullptr, *sk_keys2 = nullptr, *sk_vals = nullptr;
        void* sk_tmp = nullptr; size_t sk_tmp_bytes = 0;
        bool refine_on = true;            // SOM_EXACT_REFINE=0: no refinement pass (A/B)
        bool refine_live = false;
        double pairs_per_row_last = 0.0;  // candidate (row, group) pairs per row of the last planned epoch
        int64_t pairs_refined_in = 0, pairs_refined_out = 0;   // som_exact_refine_stats
        // the SCOUT (exact_skip.hpp): pseudo last BMUs for rows that have none, or whose last BMUs say little
        int grid_mult = 2;                // persistent re-score / refinement kernels: workgroups per resident slot (SOM_EXACT_GRID_MULT: A/B)
        bool scout_on = true;             // SOM_EXACT_SCOUT=0: plans only from last epoch's BMUs (A/B)
        bool scout_live = false;          // this launch runs the scout
        int* scout_g = nullptr;           // [stride] nearest group centroid of every row of the pass
        float* tq = nullptr;              // [stride] wide plan: the float32 score of every sorted row's last BMU under the current codebook
        double scout_est_last = 0.0;      // executed share the sample tiles forecast at the last estimate
        double scout_f_now = 0.0, scout_f_declined = 0.0; int scout_f_age = 0;   // the sampled rows' need now / when the sample tiles last declined a plan
        double scout_win_share = 0.0;     // rows of the last launch whose scout pick beat their last BMU by a tenth of the squared distance
        int64_t scout_declined = 0;       // launches whose estimate said: nothing to skip, no plan
        int64_t scouted = 0, tr_planned = 0;   // som_exact_scout_stats: launches that ran the scout; transient launches under a plan
        double tr_share_last = 1.0;       // executed share of the last transient launch under a plan
        double share_forecast = 1.0;      // ... of the launch about to run (the screen sizes its codebook parts by it)
        int tr_idle = 0, tr_cooldown = 0, tr_pause = 2;   // transient launches: idle plans pause the plan as for the resident rows
        // level 2 of the plan runs where it pays (l2_pays: measured whenever it runs), is probed again after l2_wait epochs or
        // when level 1's share has moved by half since the last probe
        double l1_share_last = 1.0, l1_share_probe = -1.0;
        bool l2_live = false, l2_pays = true;
        int l2_wait = 0;
        // MEASURED COSTS (hipEvents on the handle's stream; per ROW, in ms, so that launches of different sizes compare): what the
        // policy decides from -- is a plan worth its launches, does level 2 pay, did a sort pay, is the plan idle.  The launch as
        // a whole is timed every time (two records); its phases under a plan -- screen, level 2, sort + gather -- in the first
        // planned launches, in every launch that sorts, and every fourth one after that (a record costs a few microseconds of
        // stream bubble: eight of them a launch would be 2 % of a 1.8 ms epoch).
        struct Cost : policy::Costs {     // (the numbers themselves and what is decided from them: exact_policy.hpp)
            hipEvent_t ev[8] = {};            // [0] launch / pass start, [1, 2] screen, [3, 4] level 2, [5, 6] sort + gather, [7] pass end
            bool have = false;
            int since = 99;                   // planned launches since the phases were last timed
        } cost;
        // the centroid sets of the plan: [0] the 64-unit groups, [1] their 16-unit sub-blocks (exact_centroid_kernel's slot order)
        struct Centroids { float *Cc = nullptr, *rg = nullptr, *csq = nullptr, *cmax2 = nullptr; char* Cst = nullptr;
                           char* Cst_plain = nullptr;   // level 1 only: the scout's copy (plain initial accumulators)
                           int n_slots = 0, n_cstages = 0, n_img_stages = 0; } cen[2];
        unsigned long long *need = nullptr, *need2 = nullptr;
        int *glist = nullptr, *gcnt = nullptr;   // per tile: (group << 4 | sub-block mask) items: what the select kernel walks
        int *tlist = nullptr, *tcnt = nullptr;   // per tile: the same blocks as a dense list of 16-unit tiles: what the screen walks
        int2* tile_counts = nullptr;
        int2* items = nullptr;            // the listed screen's work queue: [0] = (items, counter), from [8] on (tile, part | parts << 16)
        int item_slots = 0;               // ... sized for this many resident workgroups
        int screen_slots = 0;             // ... the listed screen's last grid (what the next plan cuts its lists for)
        bool item_queue = true;           // SOM_EXACT_QUEUE=0: one workgroup per tile (and part) instead (A/B)
        int item_len_pct = 125;           // ... =<pct >= 25>: an item's length in per cent of the mean list
    } ex;
    int n_kchunks = 0;       // tiled: 64-feature chunks
    int n_ublocks = 0;       // tiled: unit blocks of tl_bn
    bool tl_big = false;     // tiled: 256 x 256 workgroup tiles (8 waves) instead of 128 x 128
    bool wide = false;       // bf16, 128 < input_len <= 800, big maps: samples resident in registers (bmu_bf16_wide.hpp)
    int tl_bm = 128, tl_bn = 128, tl_xtile = 0, tl_wfrag = 0, tl_wtile = 0;
    int dp = 0;              // feature stride of the bf16 row image
    int stage_bytes = 0;     // bytes of one codebook stage image
    int stage_units = 0;     // units per stage
    int nt = 1;              // neighbourhood terms
    bool swapped = false;    // mexican_hat + compact_support, rectangular: row stage, mask, column stage (update.hpp)
    int norm_p = 2;          // exponent of the norm_p distances
    double norm_pr = 0.0;    // ... when it is not an integer (0: norm_p)
    hipStream_t stream = nullptr;
    bool own_stream = false;

    float *W = nullptr, *wsq = nullptr, *SC = nullptr, *T = nullptr, *ACC = nullptr, *P1 = nullptr, *P2 = nullptr;
    float* Ud = nullptr;     // the count column after stage 1 of the transform, dense [nt][X][Y]
    char* Wst = nullptr;
    char* Wst_lo = nullptr;  // exact mode, input_len <= 128: the units' second half image (the refinement pass, bmu_exact.hpp)
    char* Wfst = nullptr;    // f32 parity mode, input_len <= 128: float32 stage image (bmu_f32_res.hpp)
    int fr_kg = 0, fr_stages = 0;
    char* Wfimg = nullptr;   // f32 parity mode, input_len > 128: float32 tile image (bmu_f32_tiled.hpp)
    char* ftX = nullptr;     //   sample tile image of the rows being scanned (scratch, grown on demand)
    long ftX_cap = 0;
    int ft_kchunks = 0, ft_ublocks = 0;
    int n_stages = 0;
    // operands derived from W, rebuilt lazily: the bf16 stage image (w_dirty), |w|^2 (wsq_dirty) and the
    // float32 stage / tile images (wf_dirty).  Training in bf16 precision never touches the float32 images.
    bool w_dirty = true, wsq_dirty = true, wf_dirty = true, wp_dirty = true;   // (wp: the patch-order copy, exact mode)

    // resident training rows
    const float* Xd = nullptr;
    float* X_owned = nullptr;
    long N = 0, Np = 0;
    int* bmu = nullptr;
    bool bmu_valid = false;  // bmu holds the ids of a completed BMU pass over the resident rows
    // precision 'exact': the update-side work that does not depend on the BMUs (the zeroed segment sums, the neighbourhood
    // tables) is queued BEFORE the pass's counter read-back, so the GPU has it to do while the host wakes up
    struct EarlyUpdate { bool armed = false, done = false; double sigma = 0, eta = 0; int neigh_f64 = 0; } early;
    unsigned long long* best64 = nullptr;   // bf16 path: per-row (value bits | unit) merged across codebook parts
    long best64_cap = 0;
    int n_cus = 0;
    // BMU-ordered view of a row set + the partial lists of the segment sum's upper levels (update.hpp)
    struct SegScratch {
        int *iota = nullptr, *skey = nullptr, *srow = nullptr;
        void* tmp = nullptr;
        size_t tmp_bytes = 0;
        int *kA = nullptr, *kB = nullptr;        // keys of the level-1 / level-2 lists (ping-pong from there on)
        float *vA = nullptr, *vB = nullptr;      // their vectors [entries][D1p]
        long cap = 0;                            // rows this scratch serves
        int* cs_table = nullptr;                 // counting sort (small maps): [K][blocks of 1 024 rows] counts, then [K] totals
        long cs_blocks = 0;
    };
    SegScratch seg;                              // resident rows
    float* xsq = nullptr;
    __bf16* Xb = nullptr;
    float* xmax2 = nullptr;  // [0] resident rows, [1] query scratch: max_n |x~_n|^2
    float* wn = nullptr;     // |w~_k|^2 per unit
    float* wmax2 = nullptr;

    // scratch for som_bmu / som_quantization_error
    float* qX = nullptr; int* qbmu = nullptr; int* qbmu2 = nullptr; float* qxsq = nullptr; __bf16* qXb = nullptr;
    double* qX64 = nullptr; size_t qX64_cap = 0;   // som_bmu_f64: float64 query rows
    long qcap = 0, qX_cap = 0;
    double* dsum = nullptr;
    // streamed epochs (rows that do not stay resident): per-chunk sort scratch, grown on demand
    SegScratch st_seg;
    bool streaming = false;
    // double-buffered device staging for chunks that arrive in pinned host memory
    struct Slot {
        float* dX = nullptr; __bf16* dXb = nullptr; float* dxsq = nullptr; int* dbmu = nullptr;
        long cap = 0;
        hipEvent_t copied = nullptr, consumed = nullptr;
        bool used = false;
    } slot[2];
    hipStream_t copy_stream = nullptr;
    int slot_idx = 0;

    // whole-epoch hipGraph (som_epoch_accumulate on resident rows), opt-in with SOM_GRAPH=1: captured on the
    // second epoch over the same buffers, replayed afterwards; any (re)allocation or new row set makes it
    // stale.  Off by default: measured on ROCm 7.2 the replay is no faster than the eager launches
    // (6x6x4 map, 150 rows: 49.5 us vs 41.2 us per epoch; 64x64x32, 100k rows: 0.366 vs 0.356 ms).
    bool use_graph = false;
    bool capturing = false;
    hipGraphExec_t gexec = nullptr;
    unsigned long alloc_gen = 0, gexec_gen = 0;
    const void* gexec_rows = nullptr;
    long gexec_n = -1;
    int graph_warm = 0;
    void* np_dev = nullptr;  // NeighParams read by the captured neigh_tables_kernel
    int2* bands = nullptr;   // nonzero column ranges of the neighbourhood tables per 128-row block (update.hpp)
    bool use_bands = true;

    // canary (som_set_verify / SOM_VERIFY=n): n strided rows of every BMU launch re-scored by the float32 kernel
    int verify_rows = 0;
    bool verifying = false;
    int *vf_rows = nullptr, *vf_picks = nullptr, *vf_best = nullptr, *vf_bad = nullptr;
    float* vf_X = nullptr;
    int vf_cap = 0;
    int64_t verify_launches = 0, verify_rows_checked = 0;
    bool fuse_merge_prep = true; // SOM_FUSE_MERGE=0: separate merge and operand-preparation launches (A/B)
    bool counting_sort = true;   // SOM_COUNTING_SORT=0: rocPRIM's sort on small maps too (A/B)
    // read once in som_create (experiments / A-B runs): forced part counts, launch-geometry printing
    int env_bf16_parts = 0;
    bool debug = false;
    // per kernel function: the dynamic-LDS attribute is set and the occupancy queried once, not per launch
    struct KernelSlots { const void* fn; size_t lds; int per_cu; };
    std::vector<KernelSlots> kernel_slots;
    int staged_blocks_done = -1; // som_epoch_accumulate_begin / _block: next block expected, -1 = none pending
    bool async_copies = false;   // SOM_ASYNC_COPIES=1: round 1's original copy path (fresh_process_stress.py)
    int prof = 0;            // som_profile_enable: 0 off, 1 every kernel family, 2 the BMU family only
    std::vector<EventPair> pending, pool;
    double ms[SOM_K_COUNT] = {0};
    int64_t launches[SOM_K_COUNT] = {0};

    std::string err;
};

namespace {

int fail(som_handle* h, const std::string& msg) {
    if (h) h->err = msg; else g_create_error = msg;
    return 1;
}
int fail_hip(som_handle* h, const char* what, hipError_t e) {
    return fail(h, std::string(what) + ": " + hipGetErrorString(e));
}

// Every entry point runs on the handle's device and leaves the calling thread's current device as it found
// it (the host process shares this HIP runtime with torch, whose current device must not move under it).
struct DeviceGuard {
    int prev = -1;
    bool moved = false;
    explicit DeviceGuard(const som_handle* h) {
        if (!h) return;
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != h->cfg.device) { (void)hipSetDevice(h->cfg.device); moved = true; }
    }
    ~DeviceGuard() { if (moved && prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

#define HIPCHK(h, call)                                              \
    do {                                                             \
        hipError_t e_ = (call);                                      \
        if (e_ != hipSuccess) return fail_hip((h), #call, e_);       \
    } while (0)

template <typename T>
int dev_alloc(som_handle* h, T** p, size_t count) {
    *p = nullptr;
    if (count == 0) count = 1;
    if (h->capturing) return fail(h, "allocation during graph capture");
    HIPCHK(h, hipMalloc((void**)p, count * sizeof(T)));
    ++h->alloc_gen;                                   // device pointers baked into a captured graph may be stale
    return 0;
}

// Host -> device copy of caller-owned (usually pageable) memory.  Blocking on purpose: the runtime stages
// pageable sources through its own pinned buffers; the copy is complete before anything that reads `dst` is
// launched, and the stream is drained first, so nothing of ours still touches dst.
// SOM_ASYNC_COPIES=1 restores round 1's original form (hipMemcpyAsync on the engine's stream, no host wait)
// for tools/fresh_process_stress.py, which exists to confirm or kill the hypothesis that that form produced the
// one wrong-BMU event on record.
int h2d_blocking(som_handle* h, void* dst, const void* src, size_t bytes) {
    if (h->async_copies) {
        HIPCHK(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
        return 0;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return 0;
}
// ... and results back to caller-owned memory, the same way: drain the stream, then a blocking copy.
int d2h_blocking(som_handle* h, void* dst, const void* src, size_t bytes) {
    if (h->async_copies) {
        HIPCHK(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return 0;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return 0;
}

// Workgroups of `fn` (threads per workgroup, dynamic LDS bytes) one CU holds at once; the first call per
// (function, LDS size) raises the function's dynamic-LDS limit and asks the runtime, later calls read the cache.
int kernel_per_cu(som_handle* h, const void* fn, int threads, size_t lds, int* per_cu) {
    for (const auto& k : h->kernel_slots)
        if (k.fn == fn && k.lds == lds) { *per_cu = k.per_cu; return 0; }
    HIPCHK(h, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int n = 0;
    HIPCHK(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, threads, lds));
    h->kernel_slots.push_back({fn, lds, n > 0 ? n : 1});
    *per_cu = n > 0 ? n : 1;
    return 0;
}

inline long cdiv(long a, long b) { return (a + b - 1) / b; }
inline long round_up(long a, long b) { return cdiv(a, b) * b; }
constexpr float HALF_MAX = 65504.0f;   // largest finite _Float16
constexpr long ROW_PAD = 3072;   // bf16 row images are padded to a multiple of every kernel's workgroup tile

// ---- profiling: event pairs recorded around kernel families, resolved lazily ---------------
struct Timed {
    som_handle* h; int kernel; EventPair ep{}; bool on;
    Timed(som_handle* h_, int k) : h(h_), kernel(k), on(h_->prof == 1 || (h_->prof == 2 && (k == SOM_K_BMU || k == SOM_K_SCREEN))) {
        if (!on) return;
        if (!h->pool.empty()) { ep = h->pool.back(); h->pool.pop_back(); }
        else { (void)hipEventCreate(&ep.a); (void)hipEventCreate(&ep.b); }
        ep.kernel = kernel;
        (void)hipEventRecord(ep.a, h->stream);
    }
    ~Timed() {
        if (!on) return;
        (void)hipEventRecord(ep.b, h->stream);
        h->pending.push_back(ep);
    }
};

int resolve_profile(som_handle* h) {
    if (h->pending.empty()) return 0;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (auto& ep : h->pending) {
        float ms = 0.f;
        HIPCHK(h, hipEventElapsedTime(&ms, ep.a, ep.b));
        h->ms[ep.kernel] += ms;
        h->launches[ep.kernel] += 1;
        h->pool.push_back(ep);
    }
    h->pending.clear();
    return 0;
}

// ---- codebook-derived operands (w_sq cache, xpysom.py:529-537; bf16 / f16 stage image) -----
// The half-precision paths are templates on the operand type's tag E (Bf16 or F16, som_common.hpp); SOM_HALF picks
// the instance from the handle.
#define SOM_HALF(h, fn, ...) ((h)->f16 ? fn<F16>(__VA_ARGS__) : fn<Bf16>(__VA_ARGS__))

void mark_codebook_changed(som_handle* h) { h->w_dirty = h->wsq_dirty = h->wf_dirty = h->wp_dirty = true; }

// the 16-bit operand images of the codebook: stage / tile image, |w~|^2 per unit and its maximum
template <class E>
int prep_codebook_half(som_handle* h) {
    const float* unit = h->cfg.distance == SOM_DIST_COSINE ? h->wsq : nullptr;
    // exact mode in patch order: the screen's image from the permuted codebook (refresh_codebook_operands wrote it)
    const float* Wex = h->ex_patch ? h->Wp : h->W;
    const float* qex = h->ex_patch ? h->wsq_p : h->wsq;
    const dim3 block(256);
    if (h->tiled) {
        if (h->wide && h->exact) {
            if (unit) unit = qex;
            // exact mode beyond 128 features: the float32 kernel's own |w|^2 as the norm term (euclidean) or none (cosine:
            // unit-length units, max |w| = 1), the units scaled by a power of two, their rounding errors measured
            if (unit) {
                HIPCHK(h, hipMemsetAsync(h->wn, 0, (size_t)h->K * sizeof(float), h->stream));
                HIPCHK(h, hipMemsetD32Async((hipDeviceptr_t)h->wmax2, 0x3F800000, 1, h->stream));
            } else {
                HIPCHK(h, hipMemsetAsync(h->wmax2, 0, sizeof(float), h->stream));
                exact_copy_wsq_kernel<<<dim3((unsigned)cdiv(h->K, 1024)), dim3(1024), 0, h->stream>>>(qex, h->K, h->wn, h->wmax2);
            }
            long total = (long)h->n_stages * WD_T * h->n_kchunks * 64;
            prep_w_bf16_wide_kernel<E><<<dim3((unsigned)cdiv(total, 256)), block, 0, h->stream>>>(
                Wex, h->K, h->D, h->n_kchunks, h->Wst, h->n_stages, unit, h->wmax2);
            HIPCHK(h, hipMemsetAsync(h->wmax2 + 1, 0, sizeof(float), h->stream));
            exact_werr_kernel<E><<<dim3((unsigned)cdiv(h->K, 4 * EX_WERR_UNITS)), block, 0, h->stream>>>(Wex, h->K, h->D, h->wmax2, h->wmax2 + 1, unit);
            return 0;
        }
        if (h->wide) {
            long total = (long)h->n_stages * WD_T * h->n_kchunks * 64;
            prep_w_bf16_wide_kernel<E><<<dim3((unsigned)cdiv(total, 256)), block, 0, h->stream>>>(
                h->W, h->K, h->D, h->n_kchunks, h->Wst, h->n_stages, unit);
        } else {
            long total = (long)h->n_ublocks * h->n_kchunks * (h->tl_bn / 16) * TL_KS * 64;
            prep_tiles_bf16_kernel<E><<<dim3((unsigned)cdiv(total, 256)), block, 0, h->stream>>>(
                h->W, h->K, h->D, h->n_kchunks, h->n_ublocks, h->tl_bn, h->tl_wtile, -1.0f, unit, h->Wst);
        }
        HIPCHK(h, hipMemsetAsync(h->wmax2, 0, sizeof(float), h->stream));
        rownorm_bf16_kernel<E><<<dim3((unsigned)cdiv(h->K, 4)), block, 0, h->stream>>>(
            h->W, h->K, h->D, unit, unit != nullptr, h->wn, h->wmax2);
        return 0;
    }
    const long total = (long)h->n_stages * K16_T * h->ks32 * 64;
    const dim3 grid((unsigned)cdiv(total, 256));
    if (h->exact) {
        // the float32 kernel's own |w|^2 (refreshed just before) and its maximum first: the units go in scaled by
        // ex_scale(max |w|^2); then the scaled stage image and the units' rounding errors in one pass
        HIPCHK(h, hipMemsetAsync(h->wmax2, 0, 2 * sizeof(float), h->stream));     // [0]: max |w|^2, [1]: max_k |w^_k - w~_k|^2
        exact_copy_wsq_kernel<<<dim3((unsigned)cdiv(h->K, 1024)), dim3(1024), 0, h->stream>>>(qex, h->K, h->wn, h->wmax2);
        const dim3 tgrid((unsigned)cdiv((long)h->n_stages * K16_T, 4));
        switch (h->ks32) {
        case 1: prep_w_exact_k16_kernel<1, E><<<tgrid, block, 0, h->stream>>>(Wex, h->K, h->D, h->Wst, h->n_stages, h->wmax2, h->wmax2 + 1, h->Wst_lo); break;
        case 2: prep_w_exact_k16_kernel<2, E><<<tgrid, block, 0, h->stream>>>(Wex, h->K, h->D, h->Wst, h->n_stages, h->wmax2, h->wmax2 + 1, h->Wst_lo); break;
        case 3: prep_w_exact_k16_kernel<3, E><<<tgrid, block, 0, h->stream>>>(Wex, h->K, h->D, h->Wst, h->n_stages, h->wmax2, h->wmax2 + 1, h->Wst_lo); break;
        case 4: prep_w_exact_k16_kernel<4, E><<<tgrid, block, 0, h->stream>>>(Wex, h->K, h->D, h->Wst, h->n_stages, h->wmax2, h->wmax2 + 1, h->Wst_lo); break;
        default: return fail(h, "the resident half-precision kernel supports input_len <= 128");
        }
        return 0;
    }
    const float* sc = nullptr;
    switch (h->ks32) {
    case 1: prep_w_bf16_k16_kernel<1, E><<<grid, block, 0, h->stream>>>(h->W, h->K, h->D, h->Wst, h->n_stages, unit, sc); break;
    case 2: prep_w_bf16_k16_kernel<2, E><<<grid, block, 0, h->stream>>>(h->W, h->K, h->D, h->Wst, h->n_stages, unit, sc); break;
    case 3: prep_w_bf16_k16_kernel<3, E><<<grid, block, 0, h->stream>>>(h->W, h->K, h->D, h->Wst, h->n_stages, unit, sc); break;
    case 4: prep_w_bf16_k16_kernel<4, E><<<grid, block, 0, h->stream>>>(h->W, h->K, h->D, h->Wst, h->n_stages, unit, sc); break;
    default: return fail(h, "the resident half-precision kernel supports input_len <= 128");
    }
    HIPCHK(h, hipMemsetAsync(h->wmax2, 0, sizeof(float), h->stream));
    prep_wnorm_kernel<E><<<dim3((unsigned)cdiv(h->K, 256)), block, 0, h->stream>>>(h->W, h->K, h->D, h->wn, h->wmax2, unit);
    return 0;
}

// need_f32: the caller is about to run a float32 kernel (parity-mode BMU, top-2, distance matrix).
// patch: the caller is the exact mode's screen + re-score (float32 image in patch order, where the handle uses one).
int refresh_codebook_operands(som_handle* h, bool need_f32, bool patch = false) {
    if (h->exact) need_f32 = true;                       // the re-score reads the float32 stage image and |w|^2
    patch = patch && h->ex_patch;
    const bool bf = h->cfg.precision != SOM_PREC_F32;
    const bool do_f32 = (need_f32 || !bf) && (h->wf_dirty || h->wf_patch != patch);
    const bool do_bf = bf && h->w_dirty;
    // (cosine scales the 16-bit images by 1/|w|: |w|^2 is wanted whenever they are rebuilt)
    const bool do_wsq = h->wsq_dirty && (need_f32 || !bf || (do_bf && h->cfg.distance == SOM_DIST_COSINE));
    if (!do_f32 && !do_wsq && !do_bf) return 0;
    Timed t(h, SOM_K_PREP);
    if (do_wsq) {
        row_sq_f32_kernel<<<dim3((unsigned)cdiv(h->K, 256)), dim3(256), 0, h->stream>>>(h->W, h->K, h->D, h->wsq, h->wsq_p,
                                                                                        h->ex_inv);
        h->wsq_dirty = false;
    }
    if (h->ex_patch && h->wp_dirty && ((do_f32 && patch) || do_bf)) {
        const long total = (long)h->K * ((h->D & 3) == 0 ? h->D / 4 : h->D);
        exact_permute_kernel<<<dim3((unsigned)cdiv(total, 256)), dim3(256), 0, h->stream>>>(h->W, h->wsq, h->K, h->D, h->ex_perm, h->Wp,
                                                                                           h->wsq_p);
        h->wp_dirty = false;
    }
    if (do_f32) {
        const float* Wsrc = patch ? h->Wp : h->W;
        const float* qsrc = patch ? h->wsq_p : h->wsq;
        if (h->Wfimg) {
            long total = (long)h->ft_ublocks * h->ft_kchunks * (4 * 4 * 64 + 128);
            prep_tiles_f32_kernel<<<dim3((unsigned)cdiv(total, 256)), dim3(256), 0, h->stream>>>(
                Wsrc, h->K, h->D, h->ft_kchunks, h->ft_ublocks, FT_WTILE, qsrc, h->Wfimg);
        }
        if (h->Wfst) {
            long total = (long)h->fr_stages * ((long)FR_UT * h->fr_kg * 64 + 64);
            prep_w_f32_res_kernel<<<dim3((unsigned)cdiv(total, 256)), dim3(256), 0, h->stream>>>(
                Wsrc, qsrc, h->K, h->D, h->fr_kg, h->Wfst, h->fr_stages);
        }
        h->wf_dirty = false;
        h->wf_patch = patch;
    }
    if (do_bf) {
        if (int rc = SOM_HALF(h, prep_codebook_half, h)) return rc;
        h->w_dirty = false;
    }
    HIPCHK(h, hipGetLastError());
    return 0;
}

// ---- BMU launches ----------------------------------------------------------------------------
int choose_parts(som_handle* h, long blocks, long slots, int max_parts_hint);

template <int MODE, int KG, bool TOP2 = false>
int launch_bmu_f32_res_kg(som_handle* h, const float* X, long N, const float* xsq, int* out, int* out2 = nullptr) {
    auto kern = bmu_f32_res_kernel<MODE, KG, TOP2>;
    size_t lds = 2 * (size_t)fr_stage_bytes(KG);
    int per_cu = 1;
    if (int rc = kernel_per_cu(h, (const void*)kern, 256, lds, &per_cu)) return rc;
    long grid = cdiv(N, FR_WG_SAMPLES);
    if (grid <= 0 || grid > 0x7fffffffL) return fail(h, "bmu_f32: row count out of range");
    int parts = 1;
    if (!TOP2) {
        const long slots = (long)per_cu * (h->n_cus > 0 ? h->n_cus : 256);
        parts = choose_parts(h, grid, slots, h->fr_stages);
        if (parts > h->fr_stages) parts = h->fr_stages;
        if (h->debug)
            std::fprintf(stderr, "[somhip] bmu_f32_res: blocks=%ld per_cu=%d slots=%ld parts=%d stages=%d\n", grid, per_cu,
                         slots, parts, h->fr_stages);
    }
    if (parts == 1) {
        kern<<<dim3((unsigned)grid), dim3(256), lds, h->stream>>>(X, N, h->D, xsq, h->Wfst, h->fr_stages, h->K, out, out2,
                                                                 nullptr);
    } else {
        if (N > h->best64_cap) {
            (void)hipFree(h->best64);
            h->best64 = nullptr; h->best64_cap = 0;
            if (int rc = dev_alloc(h, &h->best64, (size_t)round_up(N, 1024))) return rc;
            h->best64_cap = round_up(N, 1024);
        }
        HIPCHK(h, hipMemsetAsync(h->best64, 0xFF, (size_t)N * sizeof(unsigned long long), h->stream));
        kern<<<dim3((unsigned)grid, (unsigned)parts), dim3(256), lds, h->stream>>>(X, N, h->D, xsq, h->Wfst, h->fr_stages,
                                                                                  h->K, out, out2, h->best64);
        bmu_finalize_kernel<<<dim3((unsigned)cdiv(N, 256)), dim3(256), 0, h->stream>>>(h->best64, N, h->K, out);
    }
    HIPCHK(h, hipGetLastError());
    return 0;
}

template <int MODE, bool TOP2>
int launch_bmu_f32_tiled(som_handle* h, const float* X, long N, const float* xsq, int* out, int* out2) {
    const long n_blocks = cdiv(N, FT_BM);
    if (n_blocks <= 0 || n_blocks > 0x7fffffffL) return fail(h, "bmu_f32: row count out of range");
    if (n_blocks > h->ftX_cap) {
        (void)hipFree(h->ftX);
        h->ftX = nullptr; h->ftX_cap = 0;
        if (int rc = dev_alloc(h, &h->ftX, (size_t)n_blocks * h->ft_kchunks * FT_TILE)) return rc;
        h->ftX_cap = n_blocks;
    }
    long total = n_blocks * h->ft_kchunks * (4 * 4 * 64);
    prep_tiles_f32_kernel<<<dim3((unsigned)cdiv(total, 256)), dim3(256), 0, h->stream>>>(
        X, N, h->D, h->ft_kchunks, n_blocks, FT_TILE, nullptr, h->ftX);
    size_t lds = 2 * (size_t)FT_STAGE;
    { int pc; if (int rc = kernel_per_cu(h, (const void*)bmu_f32_tiled_kernel<MODE, TOP2>, 256, lds, &pc)) return rc; }
    bmu_f32_tiled_kernel<MODE, TOP2><<<dim3((unsigned)n_blocks), dim3(256), lds, h->stream>>>(
        h->ftX, N, xsq, h->Wfimg, h->ft_ublocks, h->ft_kchunks, h->K, out, out2);
    HIPCHK(h, hipGetLastError());
    return 0;
}

template <int MODE>
int launch_bmu_f32_any(som_handle* h, const float* X, long N, const float* xsq, int* out) {
    if (h->Wfimg) return launch_bmu_f32_tiled<MODE, false>(h, X, N, xsq, out, nullptr);   // input_len > 128
    switch (h->fr_kg) {
    case 1: return launch_bmu_f32_res_kg<MODE, 1>(h, X, N, xsq, out);
    case 2: return launch_bmu_f32_res_kg<MODE, 2>(h, X, N, xsq, out);
    case 4: return launch_bmu_f32_res_kg<MODE, 4>(h, X, N, xsq, out);
    case 8: return launch_bmu_f32_res_kg<MODE, 8>(h, X, N, xsq, out);
    case 16: return launch_bmu_f32_res_kg<MODE, 16>(h, X, N, xsq, out);
    }
    return fail(h, "bmu_f32: bad k-group count");
}

// best AND second-best unit under the sqrt'd Euclidean distance (topographic error)
int launch_bmu_top2(som_handle* h, const float* X, long N, const float* xsq, int* out, int* out2) {
    constexpr int M = SCORE_EUCLID_SQRT;
    if (h->Wfimg) return launch_bmu_f32_tiled<M, true>(h, X, N, xsq, out, out2);
    switch (h->fr_kg) {
    case 1: return launch_bmu_f32_res_kg<M, 1, true>(h, X, N, xsq, out, out2);
    case 2: return launch_bmu_f32_res_kg<M, 2, true>(h, X, N, xsq, out, out2);
    case 4: return launch_bmu_f32_res_kg<M, 4, true>(h, X, N, xsq, out, out2);
    case 8: return launch_bmu_f32_res_kg<M, 8, true>(h, X, N, xsq, out, out2);
    case 16: return launch_bmu_f32_res_kg<M, 16, true>(h, X, N, xsq, out, out2);
    }
    return fail(h, "bmu_f32: bad k-group count");
}

// Number of codebook parts the scan is split into: fill whole rounds of the resident workgroup
// slots (large inputs) or spread a short scan over the chip (few rows: winner / small data).
int choose_parts(som_handle* h, long blocks, long slots, int max_parts_hint) {
    (void)h;
    int parts = 1;
    if (blocks < slots) {
        // fewer workgroups than resident slots: split the scan so about two slots' worth of workgroups
        // exist -- co-resident workgroups share a CU's MFMA pipe, so finer pieces balance the CUs
        // (measured: batch 65 536 at 256x256x128 bf16 +5 %, configs[1] f32 +15 % over one exact round)
        parts = (int)cdiv(2 * slots, blocks);
        // a handful of query rows: spread the scan itself (one or two workgroups' worth of rows -- winner() of a few
        // samples, the exact mode's fallback rows -- over up to 256 pieces: every CU streams a slice of the codebook)
        const int cap = blocks <= 2 ? 256 : 32;
        if (parts > cap) parts = cap;
    } else {
        double best_eff = 0.0;
        for (int p = 1; p <= 4; ++p) {
            long wgs = blocks * p;
            double eff = (double)wgs / (double)(cdiv(wgs, slots) * slots);
            if (eff > best_eff + 0.02) { best_eff = eff; parts = p; }
        }
    }
    if (parts > max_parts_hint) parts = max_parts_hint;
    if (parts < 1) parts = 1;
    return parts;
}

template <int KS32, class E>
int launch_bmu_bf16_k16(som_handle* h, const __bf16* Xb, long N, int* out) {
    size_t lds = 2 * (size_t)k16_stage_bytes(KS32);
    int per_cu = 1;
    if (int rc = kernel_per_cu(h, (const void*)bmu_bf16_k16_kernel<KS32, E>, 64 * K16_NW, lds, &per_cu)) return rc;
    long blocks = cdiv(N, K16_WG_SAMPLES);
    if (blocks <= 0 || blocks > 0x7fffffffL) return fail(h, "bmu_bf16: row count out of range");
    // split the codebook scan into `parts` so the grid fills whole rounds of resident workgroups
    const long slots = (long)per_cu * (h->n_cus > 0 ? h->n_cus : 256);
    int parts = choose_parts(h, blocks, slots, h->n_stages);
    if (h->env_bf16_parts > 0) parts = h->env_bf16_parts;   // experiments
    if (h->debug)
        std::fprintf(stderr, "[somhip] bmu_bf16_k16: blocks=%ld per_cu=%d cus=%d slots=%ld parts=%d stages=%d\n", blocks,
                     per_cu, h->n_cus, slots, parts, h->n_stages);
    // (best64[0..N) was reset by prep_wsqh_kernel, launch_bmu_bf16)
    bmu_bf16_k16_kernel<KS32, E><<<dim3((unsigned)blocks, (unsigned)parts), dim3(64 * K16_NW), lds, h->stream>>>(
        Xb, N, h->Wst, h->n_stages, h->K, h->best64);
    bmu_finalize_kernel<<<dim3((unsigned)cdiv(N, 256)), dim3(256), 0, h->stream>>>(h->best64, N, h->K, out);
    HIPCHK(h, hipGetLastError());
    return 0;
}

template <int WS, int NWR, int NWC, class E>
int launch_bmu_bf16_tiled_cfg(som_handle* h, const __bf16* Ximg, long N, int* out) {
    using C = TileCfg<WS, NWR, NWC>;
    auto kern = bmu_bf16_tiled_kernel<WS, NWR, NWC, E>;
    size_t lds = (size_t)C::LDS_BYTES;
    int per_cu = 1;
    if (int rc = kernel_per_cu(h, (const void*)kern, 64 * C::WAVES, lds, &per_cu)) return rc;
    long blocks = cdiv(N, C::BM);
    if (blocks <= 0 || blocks > 0x7fffffffL) return fail(h, "bmu_bf16: row count out of range");
    const long slots = (long)per_cu * (h->n_cus > 0 ? h->n_cus : 256);
    int parts = choose_parts(h, blocks, slots, h->n_ublocks);
    // with many unit blocks, 8 parts let one XCD's resident workgroups share sample tiles through its L2
    if (h->n_ublocks >= 64 && blocks * 8 >= slots) parts = 8;
    if (h->env_bf16_parts > 0) parts = h->env_bf16_parts;
    if (parts > h->n_ublocks) parts = h->n_ublocks;
    if (N > h->best64_cap) {
        (void)hipFree(h->best64);
        h->best64 = nullptr; h->best64_cap = 0;
        if (int rc = dev_alloc(h, &h->best64, (size_t)round_up(N, 1024))) return rc;
        h->best64_cap = round_up(N, 1024);
    }
    HIPCHK(h, hipMemsetAsync(h->best64, 0xFF, (size_t)N * sizeof(unsigned long long), h->stream));
    const long grid = round_up(blocks, 8) * parts;
    if (grid > 0x7fffffffL) return fail(h, "bmu_bf16: grid too large");
    kern<<<dim3((unsigned)grid), dim3(64 * C::WAVES), lds, h->stream>>>(
        (const char*)Ximg, N, h->Wst, h->n_ublocks, h->n_kchunks, h->K, h->best64, (int)blocks, parts);
    bmu_finalize_kernel<<<dim3((unsigned)cdiv(N, 256)), dim3(256), 0, h->stream>>>(h->best64, N, h->K, out);
    HIPCHK(h, hipGetLastError());
    return 0;
}

template <int KS32, class E>
int launch_bmu_bf16_wide(som_handle* h, const __bf16* Ximg, const float* xmax2, long N, int* out) {
    auto kern = bmu_bf16_wide_kernel<KS32, E>;
    const size_t lds = (size_t)WD_SLOTS * wd_stage_bytes(KS32);
    int per_cu = 1;
    if (int rc = kernel_per_cu(h, (const void*)kern, 64 * WD_NW, lds, &per_cu)) return rc;
    const long blocks = cdiv(N, WD_WG_SAMPLES);
    if (blocks <= 0 || blocks > 0x7fffffffL) return fail(h, "bmu_bf16: row count out of range");
    const long slots = (long)per_cu * (h->n_cus > 0 ? h->n_cus : 256);
    // parts: whole rounds of the resident slots (every round scans one part in step: one L2 miss per XCD and stage)
    int parts = 1;
    if (blocks < slots) {
        parts = (int)std::min<long>(cdiv(slots, blocks), 64);
    } else {
        double best_eff = 0.0;
        for (int p = 1; p <= 8; ++p) {
            const long wgs = blocks * p;
            const double eff = (double)wgs / (double)(cdiv(wgs, slots) * slots);
            if (eff > best_eff + 0.01) { best_eff = eff; parts = p; }
        }
    }
    if (h->env_bf16_parts > 0) parts = h->env_bf16_parts;
    if (parts > h->n_stages) parts = h->n_stages;
    if (N > h->best64_cap) {
        (void)hipFree(h->best64);
        h->best64 = nullptr; h->best64_cap = 0;
        if (int rc = dev_alloc(h, &h->best64, (size_t)round_up(N, 1024))) return rc;
        h->best64_cap = round_up(N, 1024);
    }
    const long units = (long)h->n_stages * h->stage_units;
    prep_wsqh_kernel<<<dim3((unsigned)cdiv(std::max(units, N), 256)), dim3(256), 0, h->stream>>>(
        h->wn, h->K, h->wmax2, xmax2, h->Wst, h->n_stages, h->stage_bytes, h->stage_units, h->best64, N);
    if (h->debug)
        std::fprintf(stderr, "[somhip] bmu_bf16_wide: blocks=%ld per_cu=%d slots=%ld parts=%d stages=%d\n", blocks, per_cu, slots,
                     parts, h->n_stages);
    bmu_bf16_wide_kernel<KS32, E><<<dim3((unsigned)blocks, (unsigned)parts), dim3(64 * WD_NW), lds, h->stream>>>(
        (const char*)Ximg, N, h->Wst, h->n_stages, h->best64);
    bmu_finalize_kernel<<<dim3((unsigned)cdiv(N, 256)), dim3(256), 0, h->stream>>>(h->best64, N, h->K, out);
    HIPCHK(h, hipGetLastError());
    return 0;
}

template <class E>
int launch_bmu_bf16_tiled(som_handle* h, const __bf16* Ximg, const float* xmax2, long N, int* out) {
    if (h->wide) {
        switch (h->n_kchunks) {
#define SOM_WIDE_CASE(n) case n: return launch_bmu_bf16_wide<n, E>(h, Ximg, xmax2, N, out);
        SOM_WIDE_CASE(5) SOM_WIDE_CASE(6) SOM_WIDE_CASE(7) SOM_WIDE_CASE(8) SOM_WIDE_CASE(9) SOM_WIDE_CASE(10)
        SOM_WIDE_CASE(11) SOM_WIDE_CASE(12) SOM_WIDE_CASE(13) SOM_WIDE_CASE(14) SOM_WIDE_CASE(15) SOM_WIDE_CASE(16)
        SOM_WIDE_CASE(17) SOM_WIDE_CASE(18) SOM_WIDE_CASE(19) SOM_WIDE_CASE(20) SOM_WIDE_CASE(21) SOM_WIDE_CASE(22)
        SOM_WIDE_CASE(23) SOM_WIDE_CASE(24) SOM_WIDE_CASE(25)
#undef SOM_WIDE_CASE
        }
        return fail(h, "bmu_bf16_wide: no instance for this input_len");
    }
    long cin = (long)h->n_ublocks * h->n_kchunks * h->tl_bn;
    prep_tiles_cin_kernel<<<dim3((unsigned)cdiv(cin, 256)), dim3(256), 0, h->stream>>>(
        h->wn, h->K, h->wmax2, xmax2, h->n_kchunks, h->n_ublocks, h->tl_bn, h->tl_wfrag, h->tl_wtile, h->Wst);
    if (h->tl_big) return launch_bmu_bf16_tiled_cfg<8, 2, 4, E>(h, Ximg, N, out);
    return launch_bmu_bf16_tiled_cfg<4, 2, 2, E>(h, Ximg, N, out);
}

template <class E>
int launch_bmu_half(som_handle* h, const __bf16* Xb, const float* xmax2, long N, int* out) {
    if (h->tiled) return launch_bmu_bf16_tiled<E>(h, Xb, xmax2, N, out);
    // the stage image's initial accumulators depend on the row set through B = xmax * wmax; the same launch
    // resets the per-row merge keys of the 16x16x32 kernel
    long units = (long)h->n_stages * h->stage_units;
    if (N > h->best64_cap) {
        (void)hipFree(h->best64);
        h->best64 = nullptr; h->best64_cap = 0;
        if (int rc = dev_alloc(h, &h->best64, (size_t)round_up(N, 1024))) return rc;
        h->best64_cap = round_up(N, 1024);
    }
    prep_wsqh_kernel<<<dim3((unsigned)cdiv(std::max(units, N), 256)), dim3(256), 0, h->stream>>>(
        h->wn, h->K, h->wmax2, xmax2, h->Wst, h->n_stages, h->stage_bytes, h->stage_units, h->best64, N);
    switch (h->ks32) {
    case 1: return launch_bmu_bf16_k16<1, E>(h, Xb, N, out);
    case 2: return launch_bmu_bf16_k16<2, E>(h, Xb, N, out);
    case 3: return launch_bmu_bf16_k16<3, E>(h, Xb, N, out);
    case 4: return launch_bmu_bf16_k16<4, E>(h, Xb, N, out);
    }
    return fail(h, "bf16 precision supports input_len <= 128");
}

int launch_bmu_bf16(som_handle* h, const __bf16* Xb, const float* xmax2, long N, int* out) {
    return SOM_HALF(h, launch_bmu_half, h, Xb, xmax2, N, out);
}

// precision 'exact': every |x|^2 buffer is allocated with twice its row capacity; the second half holds the rows'
// measured operand rounding errors (bmu_exact.hpp).  Which half-way point belongs to this buffer:
float* exact_err_of(som_handle* h, const float* xsq) {
    if (!xsq) return nullptr;
    if (xsq == h->xsq) return h->xsq + h->N;
    if (xsq == h->qxsq) return h->qxsq + h->qcap;
    for (auto& sl : h->slot) if (xsq == sl.dxsq) return sl.dxsq + sl.cap;
    return nullptr;
}

// rows -> bf16 operand image (+ max |x~|^2 for the offset B).  Cosine: the rows go in at unit length -- the
// argmin does not depend on |x|, and B = max|x~| max|w~| then resolves every row alike (a short row next to
// long ones would otherwise be compared at B's absolute precision).  xsq_scratch: N floats, cosine + tiled only.
template <class E>
int prep_rows_half(som_handle* h, const float* X, long N, long Np, __bf16* Xb, float* xmax2, float* xsq_scratch) {
    const int Dp = h->dp;
    const bool unit = h->cfg.distance == SOM_DIST_COSINE;
    HIPCHK(h, hipMemsetAsync(xmax2, 0, sizeof(float), h->stream));
    if (h->tiled && h->exact) {
        // |x|^2 in NumPy's order is in xsq_scratch (the caller's row_sq); its second half takes the rounding errors
        float* xerr = exact_err_of(h, xsq_scratch);
        if ((!xsq_scratch || !xerr) && N > 0) return fail(h, "exact: no row-norm buffer");
        const float* usq = unit ? xsq_scratch : nullptr;
        if (unit) HIPCHK(h, hipMemsetD32Async((hipDeviceptr_t)xmax2, 0x3F800000, 1, h->stream));   // unit-length rows: max |x| = 1
        else if (N > 0) exact_max_kernel<<<dim3((unsigned)std::min<long>(cdiv(N, 256), 512)), dim3(256), 0, h->stream>>>(xsq_scratch, N, xmax2);
        const long n_blocks = Np / h->tl_bm;
        const long total = n_blocks * h->n_kchunks * (h->tl_bm / 16) * TL_KS * 64;
        prep_tiles_bf16_kernel<E><<<dim3((unsigned)cdiv(total, 256)), dim3(256), 0, h->stream>>>(
            X, N, h->D, h->n_kchunks, n_blocks, h->tl_bm, h->tl_xtile, 1.0f, usq, (char*)Xb, xmax2);
        if (N > 0) exact_rowerr_kernel<E><<<dim3((unsigned)cdiv(N, 4)), dim3(256), 0, h->stream>>>(X, N, h->D, usq, xmax2, xerr);
        HIPCHK(h, hipGetLastError());
        return 0;
    }
    if (h->tiled) {
        const float* usq = nullptr;
        if (unit && N > 0) {
            if (!xsq_scratch) return fail(h, "prep_rows_bf16: no row-norm scratch");
            row_sq_f32_kernel<<<dim3((unsigned)cdiv(N, 256)), dim3(256), 0, h->stream>>>(X, N, h->D, xsq_scratch);
            usq = xsq_scratch;
        }
        long n_blocks = Np / h->tl_bm;
        long total = n_blocks * h->n_kchunks * (h->tl_bm / 16) * TL_KS * 64;
        prep_tiles_bf16_kernel<E><<<dim3((unsigned)cdiv(total, 256)), dim3(256), 0, h->stream>>>(
            X, N, h->D, h->n_kchunks, n_blocks, h->tl_bm, h->tl_xtile, 1.0f, usq, (char*)Xb);
        if (N > 0)
            rownorm_bf16_kernel<E><<<dim3((unsigned)cdiv(N, 4)), dim3(256), 0, h->stream>>>(X, N, h->D, usq, 0, nullptr, xmax2);
        HIPCHK(h, hipGetLastError());
        return 0;
    }
    if (h->exact) {
        // max |x|^2 from the rows' float32 norms (xsq_scratch: computed by the caller), then the rows scaled by ex_scale of it
        if (!xsq_scratch && N > 0) return fail(h, "exact: no row norms");
        float* xerr = exact_err_of(h, xsq_scratch);
        if (!xerr && N > 0) return fail(h, "exact: unknown row-norm buffer");
        if (N > 0) exact_max_kernel<<<dim3((unsigned)std::min<long>(cdiv(N, 256), 512)), dim3(256), 0, h->stream>>>(xsq_scratch, N, xmax2);
        prep_x_bf16_kernel<E><<<dim3((unsigned)cdiv(Np, 4)), dim3(256), 0, h->stream>>>(X, N, h->D, Dp, Np, Xb, nullptr, 0, xmax2, xerr);
    } else
        prep_x_bf16_kernel<E><<<dim3((unsigned)cdiv(Np, 4)), dim3(256), 0, h->stream>>>(X, N, h->D, Dp, Np, Xb, xmax2, unit ? 1 : 0);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int prep_rows_bf16(som_handle* h, const float* X, long N, long Np, __bf16* Xb, float* xmax2, float* xsq_scratch) {
    return SOM_HALF(h, prep_rows_half, h, X, N, Np, Xb, xmax2, xsq_scratch);
}

// single-pass half precision ('bf16' or 'f16'), as opposed to the hi/lo split modes
inline bool is_half1(const som_handle* h) { return h->cfg.precision == SOM_PREC_BF16 || h->cfg.precision == SOM_PREC_F16; }

template <class E>
int merge_prep_half(som_handle* h) {
    HIPCHK(h, hipMemsetAsync(h->wmax2, 0, sizeof(float), h->stream));
    if (h->wide) {
        merge_prep_wide_kernel<E><<<dim3((unsigned)(h->n_stages * WD_T)), dim3(256), 0, h->stream>>>(
            h->W, h->ACC, h->K, h->D, h->D1p, h->n_kchunks, h->Wst, h->wn, h->wmax2, h->cfg.distance == SOM_DIST_COSINE);
        return 0;
    }
    const long n_tiles = (long)h->n_stages * K16_T;
    const dim3 grid((unsigned)cdiv(n_tiles, MP_TILES));
    switch (h->ks32) {
    case 1: merge_prep_k16_kernel<1, E><<<grid, dim3(64), 0, h->stream>>>(h->W, h->ACC, h->K, h->D, h->D1p, h->Wst, h->wn, h->wmax2, n_tiles); break;
    case 2: merge_prep_k16_kernel<2, E><<<grid, dim3(128), 0, h->stream>>>(h->W, h->ACC, h->K, h->D, h->D1p, h->Wst, h->wn, h->wmax2, n_tiles); break;
    case 3: merge_prep_k16_kernel<3, E><<<grid, dim3(192), 0, h->stream>>>(h->W, h->ACC, h->K, h->D, h->D1p, h->Wst, h->wn, h->wmax2, n_tiles); break;
    case 4: merge_prep_k16_kernel<4, E><<<grid, dim3(256), 0, h->stream>>>(h->W, h->ACC, h->K, h->D, h->D1p, h->Wst, h->wn, h->wmax2, n_tiles); break;
    default: return fail(h, "the resident half-precision kernel supports input_len <= 128");
    }
    return 0;
}

int launch_bmu_pairwise(som_handle* h, const float* X, long N, int p, bool even, int* out) {
    const double pr = h->norm_pr;                        // (a real exponent: the generic form, whatever `even` says)
    if (pr != 0.0) even = false;
    size_t w_bytes = (size_t)PW_UNITS * h->D * sizeof(float);
    size_t x_bytes = (size_t)PW_SAMPLES * (h->D + 1) * sizeof(float);
    int x_in_lds = w_bytes + x_bytes <= 150 * 1024;
    size_t lds = w_bytes + (x_in_lds ? x_bytes : 0);
    if (lds > 150 * 1024) return fail(h, "pairwise distance: input_len too large for the LDS unit tile");
    long grid = cdiv(N, PW_SAMPLES);
    if (grid <= 0 || grid > 0x7fffffffL) return fail(h, "bmu_pairwise: row count out of range");
    if (even) {
        { int pc; if (int rc = kernel_per_cu(h, (const void*)bmu_pairwise_kernel<PW_EVEN>, PW_SAMPLES, lds, &pc)) return rc; }
        bmu_pairwise_kernel<PW_EVEN><<<dim3((unsigned)grid), dim3(PW_SAMPLES), lds, h->stream>>>(X, N, h->D, h->W, h->K, p, x_in_lds, out);
    } else {
        { int pc; if (int rc = kernel_per_cu(h, (const void*)bmu_pairwise_kernel<PW_GENERIC>, PW_SAMPLES, lds, &pc)) return rc; }
        bmu_pairwise_kernel<PW_GENERIC><<<dim3((unsigned)grid), dim3(PW_SAMPLES), lds, h->stream>>>(X, N, h->D, h->W, h->K, p, x_in_lds, out, pr);
    }
    HIPCHK(h, hipGetLastError());
    return 0;
}

#include "exact_host.hpp"   // launch_bmu_exact and everything it drives (precision 'exact'): same translation unit

// ---- canary: re-score a strided sample of a BMU launch's rows with the float32 kernel (bmu_exact.hpp) -------------
int verify_bmu_launch(som_handle* h, const float* X, long N, const int* ids) {
    const int n = (int)std::min<long>(h->verify_rows, N);
    if (n <= 0 || h->cfg.distance > SOM_DIST_COSINE) return 0;     // (the VALU distances have one precision: nothing to cross-check)
    if (h->capturing) return fail(h, "SOM_VERIFY reads a flag back per launch: not capturable");
    if (n > h->vf_cap) {
        void* old[] = {h->vf_rows, h->vf_picks, h->vf_best, h->vf_X};
        for (void* p : old) if (p) (void)hipFree(p);
        h->vf_rows = h->vf_picks = h->vf_best = nullptr; h->vf_X = nullptr; h->vf_cap = 0;
        if (int rc = dev_alloc(h, &h->vf_rows, (size_t)n)) return rc;
        if (int rc = dev_alloc(h, &h->vf_picks, (size_t)n)) return rc;
        if (int rc = dev_alloc(h, &h->vf_best, (size_t)n)) return rc;
        if (int rc = dev_alloc(h, &h->vf_X, (size_t)n * h->D)) return rc;
        if (!h->vf_bad) if (int rc = dev_alloc(h, &h->vf_bad, 4)) return rc;
        h->vf_cap = n;
    }
    verify_pick_rows_kernel<<<dim3((unsigned)cdiv(n, 256)), dim3(256), 0, h->stream>>>(N, n, ids, h->vf_rows, h->vf_picks);
    exact_gather_rows_kernel<<<dim3((unsigned)cdiv((long)n * h->D, 256)), dim3(256), 0, h->stream>>>(X, h->vf_rows, n, h->D, h->vf_X);
    HIPCHK(h, hipMemsetAsync(h->vf_bad, 0, 4 * sizeof(int), h->stream));
    if (h->wsq_dirty) {                                            // |w|^2 in NumPy's order (the parity kernels' own)
        row_sq_f32_kernel<<<dim3((unsigned)cdiv(h->K, 256)), dim3(256), 0, h->stream>>>(h->W, h->K, h->D, h->wsq, h->wsq_p,
                                                                                        h->ex_inv);
        h->wsq_dirty = false;
    }
    const dim3 grid((unsigned)n), block(256);
    const size_t lds = (size_t)h->D * sizeof(float);
    if (lds > 60 * 1024) return fail(h, "SOM_VERIFY: input_len too large for the canary's row buffer");
    switch (h->cfg.distance) {
    case SOM_DIST_EUCLIDEAN: verify_best_kernel<SCORE_EUCLID_PART><<<grid, block, lds, h->stream>>>(h->vf_X, h->D, h->W, h->wsq, h->K, h->vf_best); break;
    case SOM_DIST_EUCLIDEAN_NO_OPT: verify_best_kernel<SCORE_EUCLID_SQ><<<grid, block, lds, h->stream>>>(h->vf_X, h->D, h->W, h->wsq, h->K, h->vf_best); break;
    default: verify_best_kernel<SCORE_COSINE><<<grid, block, lds, h->stream>>>(h->vf_X, h->D, h->W, h->wsq, h->K, h->vf_best); break;
    }
    // the mode's bound on the score gap: 0 = the float32 pick itself (f32, exact); half operands: 4 ub |x||w| (two
    // units, two operands each) with a margin
    float tol = 0.0f;
    const int prec = h->cfg.precision;
    const bool cosine = h->cfg.distance == SOM_DIST_COSINE;
    if (prec == SOM_PREC_BF16) tol = 8.0f / 256.0f;
    else if (prec == SOM_PREC_F16) tol = 8.0f / 2048.0f;
    if (cosine && tol > 0.0f) tol *= 0.5f;                         // (unit-length operands: |x||w| = 1)
    verify_picks_kernel<<<dim3((unsigned)cdiv(n, 256)), dim3(256), 0, h->stream>>>(
        h->vf_X, n, h->D, h->W, h->wsq, h->wmax2 ? h->wmax2 : h->wsq, h->vf_rows, h->vf_picks, h->vf_best, cosine ? 1 : 0, tol, h->vf_bad);
    HIPCHK(h, hipGetLastError());
    int bad[4] = {0, 0, 0, 0};
    if (int rc2 = d2h_blocking(h, bad, h->vf_bad, sizeof(bad))) return rc2;
    h->verify_launches += 1; h->verify_rows_checked += n;
    if (bad[0] > 0) {
        char msg[256];
        std::snprintf(msg, sizeof(msg), "SOM_VERIFY: %d of %d re-scored rows carry a pick outside the precision mode's bound "
                      "(e.g. row %d: unit %d, the float32 kernel says %d)", bad[0], n, bad[1], bad[2], bad[3]);
        return fail(h, msg);
    }
    return 0;
}

// BMU of `N` device rows with the configured activation distance (xpysom.py:410-417)
int run_activation_bmu_launch(som_handle* h, const float* X, long N, const float* xsq, const __bf16* Xb, const float* xmax2,
                              int* out) {
    if (int rc = refresh_codebook_operands(h, h->cfg.precision == SOM_PREC_F32, h->exact)) return rc;
    Timed t(h, SOM_K_BMU);
    if (h->exact) return launch_bmu_exact(h, X, N, xsq, Xb, xmax2, out);
    if (h->cfg.precision != SOM_PREC_F32) return launch_bmu_bf16(h, Xb, xmax2, N, out);
    switch (h->cfg.distance) {
    case SOM_DIST_EUCLIDEAN: return launch_bmu_f32_any<SCORE_EUCLID_PART>(h, X, N, xsq, out);
    case SOM_DIST_EUCLIDEAN_NO_OPT: return launch_bmu_f32_any<SCORE_EUCLID_SQ>(h, X, N, xsq, out);
    case SOM_DIST_COSINE: return launch_bmu_f32_any<SCORE_COSINE>(h, X, N, xsq, out);
    case SOM_DIST_MANHATTAN: return launch_bmu_pairwise(h, X, N, 1, false, out);
    case SOM_DIST_NORM_P_NO_OPT: return launch_bmu_pairwise(h, X, N, h->norm_p, false, out);
    case SOM_DIST_NORM_P: return launch_bmu_pairwise(h, X, N, h->norm_p, h->norm_p % 2 == 0, out);
    }
    return fail(h, "unknown distance id");
}

int run_activation_bmu(som_handle* h, const float* X, long N, const float* xsq, const __bf16* Xb, const float* xmax2,
                       int* out) {
    if (N == 0) return 0;
    if (int rc = run_activation_bmu_launch(h, X, N, xsq, Xb, xmax2, out)) return rc;
    if (out == h->bmu) h->bmu_valid = true;               // (the next epoch's exact screen seeds its thresholds from these ids)
    if (h->verify_rows > 0 && !h->verifying) return verify_bmu_launch(h, X, N, out);
    return 0;
}

bool needs_xsq(const som_handle* h) {
    if (h->exact) return true;                          // |x_n| scales the row's error bound (bmu_exact.hpp)
    return h->cfg.precision == SOM_PREC_F32 &&
           (h->cfg.distance == SOM_DIST_EUCLIDEAN_NO_OPT || h->cfg.distance == SOM_DIST_COSINE);
}

int row_sq(som_handle* h, const float* X, long N, float* out) {
    if (N == 0) return 0;
    row_sq_f32_kernel<<<dim3((unsigned)cdiv(N, 256)), dim3(256), 0, h->stream>>>(X, N, h->D, out);
    HIPCHK(h, hipGetLastError());
    return 0;
}

// ---- update path: segment sum + separable neighbourhood transform --------------------------
// SC[b] += sum of the rows whose BMU is b (and their count): sort by BMU, chunked register sums
void seg_free(som_handle::SegScratch& sg) {
    void* b[] = {sg.iota, sg.skey, sg.srow, sg.tmp, sg.kA, sg.kB, sg.vA, sg.vB, sg.cs_table};
    for (void* p : b) if (p) (void)hipFree(p);
    sg = som_handle::SegScratch();
}

// scratch for the segment sum of up to `rows` rows (sort buffers + the partial lists of levels 1 and 2;
// level 3 reuses level 1's, and so on)
int seg_reserve(som_handle* h, som_handle::SegScratch& sg, long rows) {
    if (rows <= sg.cap) return 0;
    seg_free(sg);
    if (rows > 0x7fffffffL) return fail(h, "more than 2^31-1 rows per GPU in one row set");
    if (int rc = dev_alloc(h, &sg.skey, (size_t)rows)) return rc;
    if (int rc = dev_alloc(h, &sg.srow, (size_t)rows)) return rc;
    const size_t bytes = radix_scratch_ints(rows) * sizeof(int);
    if (int rc = dev_alloc(h, (char**)&sg.tmp, bytes)) return rc;
    sg.tmp_bytes = bytes;
    const int nw = seg_waves_per_block(h->D1p);
    const long n1 = seg_next_entries(rows, SEG_CHUNK, nw), n2 = seg_next_entries(n1, SEG_CHUNK_UP, nw);
    if (int rc = dev_alloc(h, &sg.kA, (size_t)n1)) return rc;
    if (int rc = dev_alloc(h, &sg.vA, (size_t)n1 * h->D1p)) return rc;
    if (int rc = dev_alloc(h, &sg.kB, (size_t)n2)) return rc;
    if (int rc = dev_alloc(h, &sg.vB, (size_t)n2 * h->D1p)) return rc;
    // small maps: the counting sort's table (update.hpp); SOM_COUNTING_SORT=0 keeps rocPRIM's sort (A/B)
    const long csb = cdiv(rows, CS_BLOCK);
    if (h->counting_sort && h->K <= CS_MAX_K && rows >= 2 * CS_BLOCK && csb * h->K <= (1L << 21)) {
        if (int rc = dev_alloc(h, &sg.cs_table, (size_t)(csb + 1) * h->K)) return rc;
        sg.cs_blocks = csb;
    }
    sg.cap = rows;
    return 0;
}

// one level of the run sum over n entries; returns the number of workgroups it used
template <bool LEVEL0>
long launch_runsum(som_handle* h, const float* X, const int* keys, const int* srow, const float* vin, long n,
                   int accumulate, int* kout, float* vout) {
    const int nw = seg_waves_per_block(h->D1p);
    const int chunk = LEVEL0 ? SEG_CHUNK : seg_chunk_up(n, nw);
    const long blocks = cdiv(n, (long)nw * chunk);
    const dim3 grid((unsigned)blocks), block(64 * nw);
    const size_t lds = nw > 1 ? (size_t)2 * nw * (h->D1p + 1) * sizeof(float) : 0;
    if ((h->D & 1) == 0)
        runsum_kernel<LEVEL0, true><<<grid, block, lds, h->stream>>>(X, keys, srow, vin, n, chunk, h->D, h->D1p, accumulate, h->SC, h->SC + (size_t)h->K * h->D1p, kout, vout);
    else
        runsum_kernel<LEVEL0, false><<<grid, block, lds, h->stream>>>(X, keys, srow, vin, n, chunk, h->D, h->D1p, accumulate, h->SC, h->SC + (size_t)h->K * h->D1p, kout, vout);
    return blocks;
}

// SC[b] += sum of the rows whose BMU is b (and their count): sort by BMU, then the levels of update.hpp's
// run sum (rows -> partial lists) until one workgroup holds the whole list.  zero_first: SC starts from zero and
// every unit is written once (plain stores); otherwise the chunk's sums are added to what SC holds.
int segsum_rows(som_handle* h, const float* X, const int* bmu, long N, som_handle::SegScratch& sg, bool zero_first) {
    Timed t(h, SOM_K_SEGSUM);
    if (zero_first && !(h->early.done && X == h->Xd))   // (exact mode: already zeroed before the pass's read-back)
        HIPCHK(h, hipMemsetAsync(h->SC, 0, (size_t)h->K * (h->D1p + 1) * sizeof(float), h->stream));
    if (N > 0) {
        if (N > sg.cap) return fail(h, "segment sum: scratch smaller than the row set");
        int bits = 1;
        while ((1L << bits) < h->K) ++bits;
        const int B = (int)cdiv(N, CS_BLOCK);
        if (sg.cs_table != nullptr && B <= sg.cs_blocks && N >= 2 * CS_BLOCK) {
            // few units: a stable counting sort in three launches (same order as the radix sort: unit, then row)
            int* tot = sg.cs_table + (long)B * h->K;
            const size_t lds = (size_t)h->K * sizeof(int);
            cs_hist_kernel<<<dim3((unsigned)B), dim3(256), lds, h->stream>>>(bmu, N, h->K, B, sg.cs_table);
            cs_scan_kernel<<<dim3((unsigned)cdiv(h->K, 4)), dim3(256), 0, h->stream>>>(sg.cs_table, h->K, B, tot);
            cs_scatter_kernel<<<dim3((unsigned)B), dim3(CS_BLOCK), lds, h->stream>>>(bmu, N, h->K, B, bits, sg.cs_table, tot, sg.skey,
                                                                                    sg.srow);
            HIPCHK(h, hipGetLastError());
        } else {
            if (int rc = radix_sort_rows(h, bmu, N, bits, sg.skey, sg.srow, (int*)sg.tmp)) return rc;
        }
        const int acc = zero_first ? 0 : 1;
        long blocks = launch_runsum<true>(h, X, sg.skey, sg.srow, nullptr, N, acc, sg.kA, sg.vA);
        int *kin = sg.kA, *kout = sg.kB;
        float *vin = sg.vA, *vout = sg.vB;
        while (blocks > 1) {                             // 2 * blocks entries are waiting in (kin, vin)
            blocks = launch_runsum<false>(h, nullptr, kin, nullptr, vin, 2 * blocks, acc, kout, vout);
            std::swap(kin, kout);
            std::swap(vin, vout);
        }
        HIPCHK(h, hipGetLastError());
    }
    return 0;
}

int run_transform(som_handle* h, double sigma, double eta, int neigh_f64);

NeighParams make_neigh_params(const som_handle* h, double sigma, double eta, int neigh_f64) {
    NeighParams p{};
    p.sigma = sigma; p.eta = eta;
    p.d = 2.0 * (h->cfg.std_coeff * h->cfg.std_coeff) * (sigma * sigma);
    p.kind = h->cfg.neighborhood; p.compact = h->cfg.compact_support; p.wide = neigh_f64 ? 1 : 0;
    // the reference's triangle is float64 whatever sigma's type (int64 - |...| + sigma, neighborhoods.py:121-122)
    if (p.kind == SOM_NEIGH_TRIANGLE) p.wide = 1;
    p.X = h->X; p.Y = h->Y; p.nt = h->nt;
    p.hex = h->cfg.topology == SOM_TOPO_HEXAGONAL && h->cfg.neighborhood != SOM_NEIGH_BUBBLE;
    p.ncls = !p.hex ? 1 : h->cfg.compact_support ? 4 : 3;
    p.base_nt = h->nt / p.ncls;
    p.swapped = h->swapped ? 1 : 0;
    return p;
}

int run_update(som_handle* h, double sigma, double eta, int neigh_f64) {
    if (int rc = segsum_rows(h, h->Xd, h->bmu, h->N, h->seg, true)) return rc;
    return run_transform(h, sigma, eta, neigh_f64);
}

// [num|den] = sum_t (Px_t (x) Py_t) [S|c]
// the factor tables of this epoch's neighbourhood (+ their nonzero bands)
int build_tables(som_handle* h, double sigma, double eta, int neigh_f64, hipStream_t st) {
    const NeighParams p = make_neigh_params(h, sigma, eta, neigh_f64);
    long ntab = (long)h->nt * h->Y * h->Y + (long)h->X * h->nt * h->X;
    neigh_tables_kernel<<<dim3((unsigned)cdiv(ntab, 256)), dim3(256), 0, st>>>(
        p, h->capturing ? (const NeighParams*)h->np_dev : nullptr, h->P1, h->P2);
    HIPCHK(h, hipGetLastError());
    // nonzero bands of the tables (late epochs: most of Px, Py is exact zeros); SOM_NO_BANDS=1 walks everything
    const int nyb = (int)cdiv(h->Y, LM_BM), nxb = (int)cdiv(h->X, LM_BM);
    if (h->use_bands) {
        int2* bands1 = h->bands;                             // [nt][nyb]
        int2* bands2 = h->bands + (long)h->nt * nyb;         // [nxb][nt]
        HIPCHK(h, hipMemsetAsync(h->bands, 0, (size_t)h->nt * (nyb + nxb) * sizeof(int2), st));
        band_ranges_kernel<<<dim3((unsigned)nyb, (unsigned)cdiv(h->Y, 64), (unsigned)h->nt), dim3(256), 0, st>>>(
            h->P1, h->Y, h->Y, 1, h->Y, bands1, (long)h->Y * h->Y, nyb);
        band_ranges_kernel<<<dim3((unsigned)nxb, (unsigned)(h->nt * cdiv(h->X, 64)), 1), dim3(256), 0, st>>>(
            h->P2, h->X, h->nt * h->X, h->nt, h->X, bands2, 0, 0);
        HIPCHK(h, hipGetLastError());
    }
    return 0;
}

// OUT = H * M over C columns (rows of M / OUT ld floats apart): 128-wide MFMA tiles, and a VALU kernel for a last
// tile of <= LM_NARROW columns
void launch_leftmul(som_handle* h, const float* H, int Ro, int Ri, const float* M, long mstride, float* OUT, long ostride,
                    long C, long ld, int row_blocks, int batch, const int2* ranges, int nseg, int segw, long ldo = 0) {
    if (ldo == 0) ldo = ld;                              // rows of OUT as far apart as rows of M unless told otherwise
    long wide = cdiv(C, LM_BN);
    const long rem = C - (C / LM_BN) * LM_BN;
    if (rem > 0 && rem <= LM_NARROW) {
        --wide;
        const dim3 g(1, (unsigned)row_blocks, (unsigned)batch), b(256);
        if (rem <= 4) leftmul_narrow_f32_kernel<2><<<g, b, 0, h->stream>>>(H, Ro, Ri, M, mstride, OUT, ostride, C, ld, ldo, C - rem, ranges, nseg, segw);
        else leftmul_narrow_f32_kernel<4><<<g, b, 0, h->stream>>>(H, Ro, Ri, M, mstride, OUT, ostride, C, ld, ldo, C - rem, ranges, nseg, segw);
    }
    if (wide <= 0) return;
    // 64-row tiles where 128-row ones would be half empty (64 x 64 x 32: transform 34.6 -> 26.6 us); on grids that leave
    // two or three 128-row workgroups per CU (256 x 256 x 128: 528 on 256 CUs) they were measured neutral (109 vs 110 us)
    if (Ro <= 64)
        leftmul_f32_kernel<64><<<dim3((unsigned)wide, (unsigned)cdiv(Ro, 64), (unsigned)batch), dim3(256), 0, h->stream>>>(
            H, Ro, Ri, M, mstride, OUT, ostride, C, ld, ldo, ranges, nseg, segw);
    else
        leftmul_f32_kernel<128><<<dim3((unsigned)wide, (unsigned)row_blocks, (unsigned)batch), dim3(256), 0, h->stream>>>(
            H, Ro, Ri, M, mstride, OUT, ostride, C, ld, ldo, ranges, nseg, segw);
}

// tables + stage 1: T_t[a] = Py_t (Y x Y) * SC[a] (Y x D), batched
// over the X map rows; the count column goes its own way: U_t = C Py_t^T (X x Y, update.hpp)
int run_transform_stage1(som_handle* h, double sigma, double eta, int neigh_f64) {
    // (built on this stream: handing them to a side stream under the BMU kernel was measured -- the two event
    //  waits cost more than the 7 us kernel, epoch 1.058 vs 1.035 ms at 65 536 rows)
    if (!h->early.done)                                  // (exact mode: built before the pass's read-back)
        if (int rc = build_tables(h, sigma, eta, neigh_f64, h->stream)) return rc;
    const int nyb = (int)cdiv(h->Y, LM_BM);
    const int2* bands1 = h->use_bands ? h->bands : nullptr;             // [nt][nyb]
    const long slab = (long)h->Y * h->D1p;
    if (h->swapped) {
        // mexican_hat + compact_support, rectangular (X == Y): the row stage first, V_t = Fx_t (X x X) * SC (X x Y*D1p),
        // laid out T[i][t][b][:] (rows of term t: T + t*slab, nt*slab apart); then the reference's second mask,
        // m2(i, b), on the three masked terms; the column stage follows in run_transform_stage2
        const int nxb = (int)cdiv(h->X, LM_BM);
        for (int t1 = 0; t1 < h->nt; ++t1)
            launch_leftmul(h, h->P1 + (long)t1 * h->X * h->X, h->X, h->X, h->SC, 0, h->T + (long)t1 * slab, 0, slab, slab, nxb,
                           1, nullptr, 1, h->X, (long)h->nt * slab);
        const NeighParams p = make_neigh_params(h, sigma, eta, neigh_f64);
        const long total = (long)h->X * h->nt * slab;
        mask_rows_kernel<<<dim3((unsigned)cdiv(total, 256)), dim3(256), 0, h->stream>>>(
            p, h->capturing ? (const NeighParams*)h->np_dev : nullptr, h->T, h->nt - 1, h->D1p);
        HIPCHK(h, hipGetLastError());
        return 0;
    }
    for (int t1 = 0; t1 < h->nt; ++t1)
        launch_leftmul(h, h->P1 + (long)t1 * h->Y * h->Y, h->Y, h->Y, h->SC, slab, h->T + (long)t1 * h->X * slab, slab,
                       h->D, h->D1p, nyb, h->X, bands1 ? bands1 + (long)t1 * nyb : nullptr, 1, h->Y);
    // U_t[a][j] = sum_b C[a][b] Py_t[j][b] from the dense counts; stage 2 reads it densely (Ud) when it batches over
    // the map columns, or finds it where stage 1 would have put it, T_t[a][j][D], when it runs over whole rows
    const bool per_column = h->D % LM_BN == 0;
    strided_gemm_f32_kernel<<<dim3((unsigned)cdiv(h->Y, SG_T), (unsigned)cdiv(h->X, SG_T), (unsigned)h->nt), dim3(256), 0, h->stream>>>(
        h->SC + (size_t)h->K * h->D1p, h->Y, 1, h->P1, 1, h->Y, per_column ? h->Ud : h->T + h->D, per_column ? h->Y : slab,
        per_column ? 1 : h->D1p, h->X, h->Y, h->Y, 0, (long)h->Y * h->Y, per_column ? (long)h->K : (long)h->X * slab);
    HIPCHK(h, hipGetLastError());
    return 0;
}

// stage 2 for the map-row blocks [b0, b1) of LM_BM rows: ACC = [Px_0 | Px_1 ...] (X x nt*X) * T (nt*X x Y*D1p).
// input_len a multiple of the 128-column tile: batched over the Y map columns, one exact tile set per (j, features)
// -- 256 x 2 workgroups at 256 x 256 x 128, one round of the resident slots, where 132-float rows cut into 128-wide
// tiles made 264 x 2 -- and the count column through the strided kernel.  Otherwise: over the whole Y*D1p-wide rows.
int run_transform_stage2(som_handle* h, int b0, int b1) {
    const int nyb = (int)cdiv(h->Y, LM_BM), nxb = (int)cdiv(h->X, LM_BM);
    if (b0 < 0 || b1 > nxb || b0 >= b1) return fail(h, "transform: map-row block out of range");
    const int2* bands2 = h->use_bands ? h->bands + (long)h->nt * nyb : nullptr;   // [nxb][nt]
    const long slab = (long)h->Y * h->D1p;
    const long i0 = (long)b0 * LM_BM;
    const int rows = (int)std::min<long>((long)b1 * LM_BM, h->X) - (int)i0;
    if (h->swapped) {
        // the column stage, batched over the block's map rows: ACC[i] = [Gy_0 | Gy_1 ...] (Y x nt*Y) * T[i] (nt*Y x D1p)
        launch_leftmul(h, h->P2, h->Y, h->nt * h->Y, h->T + i0 * h->nt * slab, (long)h->nt * slab, h->ACC + i0 * slab, slab,
                       h->D1p, h->D1p, nyb, rows, nullptr, 1, h->nt * h->Y);
        HIPCHK(h, hipGetLastError());
        return 0;
    }
    const float* H = h->P2 + i0 * h->nt * h->X;
    const int2* ranges = bands2 ? bands2 + (long)b0 * h->nt : nullptr;
    if (h->D % LM_BN == 0) {
        launch_leftmul(h, H, rows, h->nt * h->X, h->T, h->D1p, h->ACC + i0 * slab, h->D1p, h->D, slab, b1 - b0, h->Y, ranges,
                       h->nt, h->X);
        // den[i][j] = sum_k [Px_0 | Px_1 ...][i][k] U[k][j], U = [U_0; U_1; ...] dense (stage 1)
        strided_gemm_f32_kernel<<<dim3((unsigned)cdiv(h->Y, SG_T), (unsigned)cdiv(rows, SG_T), 1), dim3(256), 0, h->stream>>>(
            H, (long)h->nt * h->X, 1, h->Ud, h->Y, 1, h->ACC + i0 * slab + h->D, slab, h->D1p, rows, h->Y, h->nt * h->X, 0, 0, 0);
    } else {
        launch_leftmul(h, H, rows, h->nt * h->X, h->T, 0, h->ACC + i0 * slab, 0, slab, slab, b1 - b0, 1, ranges, h->nt, h->X);
    }
    HIPCHK(h, hipGetLastError());
    return 0;
}

int run_transform(som_handle* h, double sigma, double eta, int neigh_f64) {
    Timed t(h, SOM_K_KRON);
    if (int rc = run_transform_stage1(h, sigma, eta, neigh_f64)) return rc;
    return run_transform_stage2(h, 0, (int)cdiv(h->X, LM_BM));
}

// with_rows: the caller stages host rows through qX (device rows of the caller's are read where they are)
int ensure_query_scratch(som_handle* h, long n, bool with_rows = true) {
    if (with_rows && n > h->qX_cap) {
        (void)hipFree(h->qX);
        h->qX = nullptr; h->qX_cap = 0;
        if (int rc = dev_alloc(h, &h->qX, (size_t)round_up(n, 1024) * h->D)) return rc;
        h->qX_cap = round_up(n, 1024);
    }
    if (n <= h->qcap) return 0;
    long cap = round_up(n, 1024);
    (void)hipFree(h->qbmu); (void)hipFree(h->qbmu2); (void)hipFree(h->qxsq); (void)hipFree(h->qXb);
    h->qbmu = nullptr; h->qbmu2 = nullptr; h->qxsq = nullptr; h->qXb = nullptr; h->qcap = 0;
    if (int rc = dev_alloc(h, &h->qbmu, (size_t)cap)) return rc;
    if (int rc = dev_alloc(h, &h->qbmu2, (size_t)cap)) return rc;
    if (int rc = dev_alloc(h, &h->qxsq, (size_t)cap * (h->exact ? 2 : 1))) return rc;
    if (h->cfg.precision != SOM_PREC_F32) {
        long capp = round_up(cap, ROW_PAD);
        if (int rc = dev_alloc(h, &h->qXb, (size_t)capp * h->dp)) return rc;
    }
    h->qcap = cap;
    return 0;
}

template <int MODE>
int launch_dist_matrix(som_handle* h, long N, float* out) {
    const int Dp = (int)round_up(h->D, F32_KC);
    size_t lds = (size_t)(F32_UB * (F32_KC + 1) + F32_UB + F32_SB * (F32_KC + 1)) * sizeof(float);
    dim3 grid((unsigned)cdiv(N, F32_SB), (unsigned)cdiv(h->K, F32_UB));
    dist_matrix_f32_kernel<MODE><<<grid, dim3(256), lds, h->stream>>>(h->qX, N, h->D, Dp, h->W, h->wsq, h->K, h->qxsq, out);
    HIPCHK(h, hipGetLastError());
    return 0;
}

}  // namespace

// ==============================================================================================
// the exact mode's patch order (som_common.hpp): position -> unit.  Host arithmetic only (som_patch_order exports it).
// Where both sides are multiples of 8 every group is a whole 8 x 8 patch, held in FOUR-BY-FOUR blocks (ex_rank44, bmu_f32.hpp):
// the 16-unit sub-blocks the plan tests (exact_skip.hpp) are then 4 x 4 squares of the map -- at the end of the benchmark's
// schedule a third fewer of them must run than of 2 x 8 strips (tools/bound_probe.py) -- and the re-score kernels decide equal
// scores by rank.  Elsewhere the units of a group ascend.
static bool patch_order_blocks(int X, int Y) { return X % 8 == 0 && Y % 8 == 0; }
static void patch_order(int X, int Y, std::vector<int>& perm, bool blocks = true) {
    const long K = (long)X * Y;
    perm.clear(); perm.reserve((size_t)K);
    for (int x0 = 0; x0 < X; x0 += 8)
        for (int y = 0; y < Y; ++y)
            for (int x = x0; x < std::min(x0 + 8, X); ++x) perm.push_back(x * Y + y);
    for (long g = 0; g < K; g += EX_GROUP) std::sort(perm.begin() + g, perm.begin() + std::min<long>(g + EX_GROUP, K));
    if (!blocks || !patch_order_blocks(X, Y)) return;
    int sorted[EX_GROUP];
    for (long g = 0; g < K; g += EX_GROUP) {
        for (int i = 0; i < EX_GROUP; ++i) sorted[i] = perm[g + i];
        for (int w = 0; w < EX_GROUP; ++w)
            perm[g + w] = sorted[8 * (4 * ((w >> 5) & 1) + ((w >> 2) & 3)) + 4 * ((w >> 4) & 1) + (w & 3)];
    }
}

extern "C" {

#ifndef SOM_SRC_HASH
#define SOM_SRC_HASH "unversioned-build"
#endif
// ends in the hash of the sources this binary was built from (xpysom_dask_amd/build.py checks it against the tree)
const char* som_version(void) { return "somhip 0.2 (gfx950) somhip-src:" SOM_SRC_HASH; }

int som_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* som_last_error(const som_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int som_create(const som_config* cfg, som_handle** out) {
    if (!out) return fail(nullptr, "som_create: out is NULL");
    *out = nullptr;
    if (!cfg) return fail(nullptr, "som_create: cfg is NULL");
    if (cfg->x < 1 || cfg->y < 1 || cfg->input_len < 1) return fail(nullptr, "som_create: x, y, input_len must be >= 1");
    if ((long)cfg->x * cfg->y > (1L << 30)) return fail(nullptr, "som_create: map too large");
    if (cfg->distance < 0 || cfg->distance > SOM_DIST_NORM_P_NO_OPT) return fail(nullptr, "som_create: unknown distance id");
    if (cfg->norm_p < 0 || cfg->norm_p > PW_MAX_P) return fail(nullptr, "som_create: norm_p out of range (1..16)");
    if (cfg->norm_p_real != 0.0 && !(cfg->norm_p_real > 0.0 && cfg->norm_p_real <= 64.0))
        return fail(nullptr, "som_create: norm_p_real out of range (0 < p <= 64)");
    if (cfg->distance >= SOM_DIST_MANHATTAN && cfg->precision != SOM_PREC_F32 && cfg->precision != SOM_PREC_EXACT)
        return fail(nullptr, "som_create: manhattan / norm_p distances are float32 VALU kernels (precision f32)");
    if (cfg->neighborhood < 0 || cfg->neighborhood > SOM_NEIGH_TRIANGLE)
        return fail(nullptr, "som_create: unknown neighbourhood id");
    if (cfg->neighborhood == SOM_NEIGH_MEXICAN_HAT && cfg->compact_support && cfg->topology == SOM_TOPO_RECTANGULAR &&
        cfg->x != cfg->y)
        return fail(nullptr, "som_create: mexican_hat with compact_support needs a square map on the rectangular topology "
                             "(the reference's second mask on px does not broadcast otherwise, neighborhoods.py:70)");
    if (cfg->topology != SOM_TOPO_RECTANGULAR && cfg->topology != SOM_TOPO_HEXAGONAL)
        return fail(nullptr, "som_create: unknown topology id");
    if (cfg->topology == SOM_TOPO_HEXAGONAL && cfg->neighborhood == SOM_NEIGH_TRIANGLE)
        return fail(nullptr, "som_create: the hexagonal topology has no triangle neighbourhood (xpysom.py:271-279)");
    if (cfg->precision < SOM_PREC_F32 || cfg->precision > SOM_PREC_EXACT)
        return fail(nullptr, "som_create: unknown precision id");
    if (cfg->precision == SOM_PREC_RETIRED_2 || cfg->precision == SOM_PREC_RETIRED_4)
        return fail(nullptr, "som_create: the split-operand precisions (ids 2 and 4: 'bf16x3', 'f16x3') are retired -- "
                             "precision 'exact' returns float32's own BMUs, faster");
    if (cfg->precision != SOM_PREC_F32 && cfg->precision != SOM_PREC_EXACT) {
        if (cfg->distance == SOM_DIST_EUCLIDEAN_NO_OPT)
            return fail(nullptr, "som_create: bf16 precision implements 'euclidean' and 'cosine' "
                                 "('euclidean_no_opt' has the same argmin as 'euclidean')");
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1) return fail(nullptr, "som_create: no HIP device available");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, "som_create: device ordinal out of range");

    som_handle* h = new som_handle();
    h->cfg = *cfg;
    if (cfg->precision == SOM_PREC_EXACT) {
        // the screen + re-score scheme covers the euclidean distance up to 128 features; everywhere else the
        // float32 kernels ARE the exact mode
        // (<= 128 features: the resident screen, euclidean; 129 .. 800 features on maps of >= 4096 units: the wide
        //  screen, euclidean and cosine -- decided below, once the tiling is known)
        if ((cfg->distance == SOM_DIST_EUCLIDEAN && cfg->input_len <= 128) ||
            ((cfg->distance == SOM_DIST_EUCLIDEAN || cfg->distance == SOM_DIST_COSINE) && cfg->input_len > 128)) h->exact = true;
        else h->cfg.precision = SOM_PREC_F32;
    }
    cfg = &h->cfg;
    h->X = cfg->x; h->Y = cfg->y; h->K = cfg->x * cfg->y; h->D = cfg->input_len;
    h->D1p = (int)round_up(h->D + 1, 4);
    h->norm_p = cfg->norm_p > 0 ? cfg->norm_p : 2;
    if (cfg->norm_p_real != 0.0 && (cfg->distance == SOM_DIST_NORM_P || cfg->distance == SOM_DIST_NORM_P_NO_OPT)) h->norm_pr = cfg->norm_p_real;
    h->ks32 = (int)cdiv(h->D, 32);
    h->f16 = cfg->precision == SOM_PREC_F16 || h->exact;   // (the exact mode's screen: IEEE half)
    h->tiled = (cfg->precision == SOM_PREC_BF16 || cfg->precision == SOM_PREC_F16 || h->exact) && h->D > 128;
    if (h->tiled) {
        // 256 x 256 tiles need enough units to amortise them; SOM_BF16_TILE=128|256 overrides
        h->tl_big = h->K >= 4096;
        if (h->tl_big) {
            using C = TileCfg<8, 2, 4>;
            h->tl_bm = C::BM; h->tl_bn = C::BN; h->tl_xtile = C::XTILE; h->tl_wfrag = C::WFRAG; h->tl_wtile = C::WTILE;
        } else {
            using C = TileCfg<4, 2, 2>;
            h->tl_bm = C::BM; h->tl_bn = C::BN; h->tl_xtile = C::XTILE; h->tl_wfrag = C::WFRAG; h->tl_wtile = C::WTILE;
        }
        h->n_kchunks = (int)cdiv((long)h->D, TL_BK);
        h->n_ublocks = (int)cdiv(h->K, h->tl_bn);
    }
    h->wide = h->tiled && h->tl_big && h->n_kchunks <= 25;
    if (const char* e = dev_env("SOM_BF16_WIDE")) if (std::atoi(e) == 0) h->wide = false;   // A/B: the two-sided tiling
    if (h->exact && h->tiled && !h->wide) {              // no exact screen on the two-sided tiling: the float32 kernels serve
        h->exact = false; h->f16 = false; h->tiled = false;
        h->cfg.precision = SOM_PREC_F32;
    }
    h->ex_patch = h->exact && h->K >= 2 * EX_GROUP;      // (a map of one group has nothing to order)
    if (const char* e = dev_env("SOM_EXACT_PATCH")) if (std::atoi(e) == 0) h->ex_patch = false;   // A/B: groups = strips of a map row
    h->dp = h->tiled ? TL_BK * h->n_kchunks : 32 * h->ks32;
    h->stage_bytes = h->wide ? wd_stage_bytes(h->n_kchunks) : k16_stage_bytes(h->ks32);
    h->stage_units = h->wide ? WD_STAGE_UNITS : K16_STAGE_UNITS;
    h->nt = cfg->neighborhood == SOM_NEIGH_MEXICAN_HAT ? (cfg->compact_support ? 4 : 2) : 1;
    h->swapped = cfg->neighborhood == SOM_NEIGH_MEXICAN_HAT && cfg->compact_support && cfg->topology == SOM_TOPO_RECTANGULAR;
    // hexagonal: one copy of the terms per parity class of (unit row, BMU row) -- three classes by the x offset
    // difference (0, +0.5, -0.5); four with compact_support, whose mask compares ABSOLUTE x coordinates (update.hpp)
    if (cfg->topology == SOM_TOPO_HEXAGONAL && cfg->neighborhood != SOM_NEIGH_BUBBLE) h->nt *= cfg->compact_support ? 4 : 3;
    int rc = 0;
    auto bail = [&](int code) { g_create_error = h->err; som_destroy(h); return code; };
    DeviceGuard dev_guard(h);
    {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess || cur != cfg->device) return bail(fail(h, "hipSetDevice failed"));
    }
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess) h->n_cus = prop.multiProcessorCount;
    }
    if (cfg->stream) { h->stream = (hipStream_t)cfg->stream; }
    else {
        if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess)
            return bail(fail(h, "hipStreamCreate failed"));
        h->own_stream = true;
    }
    const size_t KD1 = (size_t)h->K * h->D1p;
    if ((rc = dev_alloc(h, &h->W, (size_t)h->K * h->D))) return bail(rc);
    if ((rc = dev_alloc(h, &h->wsq, (size_t)h->K))) return bail(rc);
    if (h->ex_patch) {
        if ((rc = dev_alloc(h, &h->Wp, (size_t)h->K * h->D))) return bail(rc);
        if ((rc = dev_alloc(h, &h->wsq_p, (size_t)h->K))) return bail(rc);
        if ((rc = dev_alloc(h, &h->ex_perm, (size_t)h->K))) return bail(rc);
        if ((rc = dev_alloc(h, &h->ex_inv, (size_t)h->K))) return bail(rc);
        // bands of 8 map rows, column by column: 64 consecutive positions = 8 columns of a band = an 8 x 8 patch (where
        // the sides are no multiples of 8 a group may straddle two bands or hold a narrower band's 64 / h columns: still
        // compact); then every group's units in ascending order (the first-minimum rule inside a re-score tile)
        std::vector<int> perm;
        bool blocks = true;
        if (const char* e = dev_env("SOM_EXACT_SUB44")) blocks = std::atoi(e) != 0;   // A/B: every group's units ascending (2 x 8 sub-blocks)
        h->ex_sub44 = blocks && patch_order_blocks(h->X, h->Y);
        patch_order(h->X, h->Y, perm, blocks);
        std::vector<int> inv((size_t)h->K);
        for (int p = 0; p < h->K; ++p) inv[(size_t)perm[(size_t)p]] = p;
        if (hipMemcpy(h->ex_perm, perm.data(), (size_t)h->K * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(h->ex_inv, inv.data(), (size_t)h->K * sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
            return bail(fail(h, "hipMemcpy of the patch-order tables failed"));
    }
    if ((rc = dev_alloc(h, &h->SC, KD1 + (size_t)h->K))) return bail(rc);        // [K][D1p] sums|counts, then the counts densely [K]
    if ((rc = dev_alloc(h, &h->Ud, (size_t)h->nt * h->K))) return bail(rc);
    if ((rc = dev_alloc(h, &h->T, KD1 * h->nt))) return bail(rc);
    if ((rc = dev_alloc(h, &h->ACC, KD1))) return bail(rc);
    if ((rc = dev_alloc(h, &h->P1, (size_t)h->nt * h->Y * h->Y))) return bail(rc);
    if ((rc = dev_alloc(h, &h->P2, (size_t)h->X * h->nt * h->X))) return bail(rc);
    if ((rc = dev_alloc(h, &h->dsum, 1))) return bail(rc);
    {
        NeighParams* npd = nullptr;
        if ((rc = dev_alloc(h, &npd, 1))) return bail(rc);
        h->np_dev = npd;
        if (const char* e = std::getenv("SOM_GRAPH")) h->use_graph = std::atoi(e) != 0;
        if (const char* e = dev_env("SOM_BF16_PARTS")) h->env_bf16_parts = std::atoi(e);
        h->debug = std::getenv("SOM_DEBUG") != nullptr;
        if (const char* e = std::getenv("SOM_VERIFY")) h->verify_rows = std::max(0, std::atoi(e));
        if (const char* e = dev_env("SOM_EXACT_PASS_ROWS")) h->ex.pass_rows_override = std::atol(e);
        if (const char* e = dev_env("SOM_EXACT_DEBUG_REFUSE_ABOVE")) h->ex.hook_refuse_above = std::atol(e);
        if (const char* e = dev_env("SOM_EXACT_PAIRS")) h->ex.hook_pairs = std::max(1L, std::atol(e));
        h->ex.hook_refuse_skip = dev_env("SOM_EXACT_DEBUG_REFUSE_SKIP") != nullptr;
        if (const char* e = dev_env("SOM_EXACT_TWO_ROUND")) h->ex.two_round = std::atoi(e) != 0 ? 1 : 0;
        if (const char* e = dev_env("SOM_EXACT_SEED")) h->ex.seed_on = std::atoi(e) != 0;
        if (const char* e = std::getenv("SOM_EXACT_SKIP")) h->ex.skip_mode = std::atoi(e);
        if (const char* e = dev_env("SOM_EXACT_SUBBLOCKS")) h->ex.sub_blocks = std::atoi(e) != 0;
        if (const char* e = dev_env("SOM_EXACT_REFINE")) h->ex.refine_on = std::atoi(e) != 0;
        if (const char* e = dev_env("SOM_EXACT_QUEUE")) { h->ex.item_queue = std::atoi(e) != 0; if (std::atoi(e) >= 25) h->ex.item_len_pct = std::atoi(e); }
        if (const char* e = dev_env("SOM_EXACT_SCOUT")) h->ex.scout_on = std::atoi(e) != 0;
        if (const char* e = dev_env("SOM_EXACT_GRID_MULT")) h->ex.grid_mult = std::max(1, std::atoi(e));
        if (const char* e = dev_env("SOM_EXACT_RESORT")) h->ex.res_every = std::max(0, std::atoi(e));
        if (const char* e = dev_env("SOM_ASYNC_COPIES")) h->async_copies = std::atoi(e) != 0;
        if (const char* e = dev_env("SOM_FUSE_MERGE")) h->fuse_merge_prep = std::atoi(e) != 0;
        if (const char* e = dev_env("SOM_COUNTING_SORT")) h->counting_sort = std::atoi(e) != 0;
        // a 128-row block of a table already spans most of a map side up to 256: nothing to skip there
        h->use_bands = h->X > 256 || h->Y > 256;
        if (const char* e = dev_env("SOM_NO_BANDS")) h->use_bands = std::atoi(e) == 0;
        const size_t nb = (size_t)h->nt * (cdiv(h->Y, LM_BM) + cdiv(h->X, LM_BM));
        if ((rc = dev_alloc(h, &h->bands, nb))) return bail(rc);
    }
    // (T: stage 1 of the transform writes the feature columns and the count column only; the padding columns behind
    //  them are read by a whole-row stage 2 and end up in ACC's padding, which is all-reduced: keep them defined)
    if (hipMemsetAsync(h->W, 0, (size_t)h->K * h->D * sizeof(float), h->stream) != hipSuccess ||
        hipMemsetAsync(h->T, 0, KD1 * h->nt * sizeof(float), h->stream) != hipSuccess ||
        hipMemsetAsync(h->ACC, 0, KD1 * sizeof(float), h->stream) != hipSuccess)
        return bail(fail(h, "hipMemsetAsync failed"));
    if (h->D > 128) {
        h->ft_kchunks = (int)cdiv(h->D, FT_BK);
        h->ft_ublocks = (int)cdiv(h->K, FT_BN);
        if ((rc = dev_alloc(h, &h->Wfimg, (size_t)h->ft_ublocks * h->ft_kchunks * FT_WTILE))) return bail(rc);
    }
    if (h->D <= 128) {
        int kg = 1;
        while (kg * 8 < h->D) kg *= 2;                      // 8, 16, 32, 64 or 128 features per row image
        h->fr_kg = kg;
        h->fr_stages = (int)cdiv(h->K, FR_STAGE_UNITS);
        if ((rc = dev_alloc(h, &h->Wfst, (size_t)h->fr_stages * fr_stage_bytes(kg)))) return bail(rc);
    }
    if (h->cfg.precision != SOM_PREC_F32) {
        h->n_stages = (int)cdiv(h->K, h->stage_units);
        size_t bytes = (size_t)h->n_stages * h->stage_bytes;
        if (h->tiled && !h->wide) bytes = (size_t)h->n_ublocks * h->n_kchunks * h->tl_wtile;
        if ((rc = dev_alloc(h, &h->Wst, bytes))) return bail(rc);
        if (h->exact && !h->tiled) {
            if ((rc = dev_alloc(h, &h->Wst_lo, bytes))) return bail(rc);
            if (hipMemsetAsync(h->Wst_lo, 0, bytes, h->stream) != hipSuccess) return bail(fail(h, "hipMemsetAsync failed"));
        }
        if ((rc = dev_alloc(h, &h->xmax2, 2))) return bail(rc);
        if ((rc = dev_alloc(h, &h->wn, (size_t)h->K))) return bail(rc);
        if ((rc = dev_alloc(h, &h->wmax2, 2))) return bail(rc);
        if (hipMemsetAsync(h->Wst, 0, bytes, h->stream) != hipSuccess) return bail(fail(h, "hipMemsetAsync failed"));
    }
    if (hipStreamSynchronize(h->stream) != hipSuccess) return bail(fail(h, "hipStreamSynchronize failed"));
    *out = h;
    return 0;
}

void som_destroy(som_handle* h) {
    if (!h) return;
    DeviceGuard dev_guard(h);
    (void)som_comm_destroy(h);
    if (h->comm_stream) (void)hipStreamDestroy(h->comm_stream);
    if (h->ev_block) (void)hipEventDestroy(h->ev_block);
    if (h->ev_comm) (void)hipEventDestroy(h->ev_comm);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->gexec) (void)hipGraphExecDestroy(h->gexec);
    if (h->np_dev) (void)hipFree(h->np_dev);
    if (h->bands) (void)hipFree(h->bands);
    for (auto& ep : h->pending) { (void)hipEventDestroy(ep.a); (void)hipEventDestroy(ep.b); }
    for (auto& ep : h->pool) { (void)hipEventDestroy(ep.a); (void)hipEventDestroy(ep.b); }
    void* bufs[] = {h->Ud, h->W, h->wsq, h->SC, h->T, h->ACC, h->P1, h->P2, h->Wst, h->X_owned, h->bmu, h->xsq, h->Xb,
                    h->xmax2, h->wn, h->wmax2, h->qX, h->qbmu, h->qbmu2, h->qxsq, h->qXb, h->dsum,
                    h->best64, h->Wfst, h->Wfimg, h->ftX, h->qX64, h->Wp, h->wsq_p, h->ex_perm, h->ex_inv, h->Wst_lo};
    for (void* b : bufs) if (b) (void)hipFree(b);
    seg_free(h->seg);
    seg_free(h->st_seg);
    {
        void* vb[] = {h->vf_rows, h->vf_picks, h->vf_best, h->vf_bad, h->vf_X};
        for (void* b : vb) if (b) (void)hipFree(b);
    }
    {
        void* eb[] = {h->ex.gmin, h->ex.gflags, h->ex.rowcnt, h->ex.rowarg, h->ex.seed, h->ex.fb_list, h->ex.ctr, h->ex.fb_ids, h->ex.fbX, h->ex.plist, h->ex.tile_tab,
                      h->ex.sk_keys, h->ex.sk_keys2, h->ex.sk_vals, h->ex.sk_tmp, h->ex.scout_g, h->ex.tq,
                      h->ex.need, h->ex.need2, h->ex.glist, h->ex.gcnt, h->ex.tile_counts, h->ex.tlist, h->ex.tcnt};
        for (void* b : eb) if (b) (void)hipFree(b);
        for (auto& sr : h->ex.srt) {
            void* sb[] = {sr.order, sr.Xb_s, sr.Xl_s, sr.Xf_s, sr.xsq_s, sr.xerr_s, sr.seed_s, sr.sU_s, sr.lastpos_s};
            for (void* b : sb) if (b) (void)hipFree(b);
        }
        for (auto& c : h->ex.cen) {
            void* cb[] = {c.Cc, c.rg, c.csq, c.cmax2, c.Cst, c.Cst_plain};
            for (void* b : cb) if (b) (void)hipFree(b);
        }
        if (h->ex.cost.have) for (auto& e : h->ex.cost.ev) (void)hipEventDestroy(e);
        if (h->ex.fb_count_host) (void)hipHostFree(h->ex.fb_count_host);
        if (h->ex.fb_ready) (void)hipEventDestroy(h->ex.fb_ready);
    }
    for (auto& sl : h->slot) {
        void* sb[] = {sl.dX, sl.dXb, sl.dxsq, sl.dbmu};
        for (void* b : sb) if (b) (void)hipFree(b);
        if (sl.copied) (void)hipEventDestroy(sl.copied);
        if (sl.consumed) (void)hipEventDestroy(sl.consumed);
    }
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int som_set_weights(som_handle* h, const float* w_host) {
    DeviceGuard dev_guard(h);
    if (!h || !w_host) return fail(h, "som_set_weights: NULL argument");
    if (h->f16 && !h->exact && h->cfg.distance != SOM_DIST_COSINE) {   // (cosine rounds unit-length rows; exact scales)
        // every unit's own norm must fit IEEE half (NaN / infinite units are left to the kernels' NaN rules)
        for (long k = 0; k < h->K; ++k) {
            double q = 0.0;
            for (int d = 0; d < h->D; ++d) { const double v = w_host[k * h->D + d]; q += v * v; }
            if (q > (double)HALF_MAX * HALF_MAX && q <= 1.0e300)
                return fail(h, "som_set_weights: precision 'f16' needs units of norm <= 65504 (float16 range): scale the data, or use 'bf16' / 'f32'");
        }
    }
    if (int rc = h2d_blocking(h, h->W, w_host, (size_t)h->K * h->D * sizeof(float))) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    mark_codebook_changed(h);
    return 0;
}

int som_get_weights(som_handle* h, float* w_host) {
    DeviceGuard dev_guard(h);
    if (!h || !w_host) return fail(h, "som_get_weights: NULL argument");
    return d2h_blocking(h, w_host, h->W, (size_t)h->K * h->D * sizeof(float));
}

static int adopt_rows(som_handle* h, int64_t n_rows) {
    (void)hipFree(h->bmu); (void)hipFree(h->xsq); (void)hipFree(h->Xb);
    h->bmu = nullptr; h->xsq = nullptr; h->Xb = nullptr;
    h->bmu_valid = false;
    h->ex.res_valid = false;                             // (the resident sorted pass belongs to the rows it was sorted from)
    seg_free(h->seg);
    h->N = n_rows;
    h->Np = round_up(n_rows, ROW_PAD);
    if (n_rows > 0x7fffffffL) return fail(h, "som_set_data: more than 2^31-1 rows per GPU");
    if (int rc = dev_alloc(h, &h->bmu, (size_t)n_rows)) return rc;
    if (n_rows > 0)
        if (int rc = seg_reserve(h, h->seg, n_rows)) return rc;
    const bool bf_cos_tiled = h->cfg.precision != SOM_PREC_F32 && h->cfg.distance == SOM_DIST_COSINE && h->tiled;
    if (needs_xsq(h) || bf_cos_tiled) {
        if (int rc = dev_alloc(h, &h->xsq, (size_t)n_rows * (h->exact ? 2 : 1))) return rc;
        if (needs_xsq(h)) if (int rc = row_sq(h, h->Xd, n_rows, h->xsq)) return rc;
    }
    if (h->cfg.precision != SOM_PREC_F32 && n_rows > 0) {
        if (int rc = dev_alloc(h, &h->Xb, (size_t)h->Np * h->dp)) return rc;
        if (int rc = prep_rows_bf16(h, h->Xd, n_rows, h->Np, h->Xb, h->xmax2, h->xsq)) return rc;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->f16 && !h->exact && n_rows > 0) {             // IEEE half tops out at 65504: refuse rows that do not fit
        float m2 = 0.0f;
        if (int rc = d2h_blocking(h, &m2, h->xmax2, sizeof(float))) return rc;
        if (!(m2 <= HALF_MAX * HALF_MAX))
            return fail(h, "som_set_data: precision 'f16' needs rows of norm <= 65504 (float16 range): scale the data, or use 'bf16' / 'f32'");
    }
    if (h->exact && n_rows > 0) {
        // the exact mode's pass scratch and, where block skipping can engage, the resident sorted pass's buffers: allocated with
        // the rows, not inside the first epochs (gigabytes of fresh device memory can take a driver tens to hundreds of
        // milliseconds).  A refusal here is not an error: launch_bmu_exact asks again and settles it (smaller passes, no plan).
        bool ok = exact_reserve(h, n_rows) == 0;
        const long n_groups = cdiv(h->K, EX_GROUP);
        if (ok && h->ex.skip_mode > 0 && h->ex.seed_on && !h->wide && (h->ex.skip_mode > 1 ? n_groups >= 2 : h->K >= 4096))
            ok = exact_skip_reserve(h, h->ex.srt[0], n_rows, h->ex.stride) == 0;
        if (!ok) { (void)hipGetLastError(); h->err.clear(); }
    }
    return 0;
}

int som_set_data(som_handle* h, const float* x_host, int64_t n_rows) {
    DeviceGuard dev_guard(h);
    if (!h || n_rows < 0 || (!x_host && n_rows > 0)) return fail(h, "som_set_data: bad argument");
    (void)hipFree(h->X_owned);
    h->X_owned = nullptr; h->Xd = nullptr;
    if (int rc = dev_alloc(h, &h->X_owned, (size_t)n_rows * h->D)) return rc;
    if (n_rows > 0)
        if (int rc = h2d_blocking(h, h->X_owned, x_host, (size_t)n_rows * h->D * sizeof(float))) return rc;
    h->Xd = h->X_owned;
    return adopt_rows(h, n_rows);
}

int som_set_data_device(som_handle* h, const void* x_dev, int64_t n_rows) {
    DeviceGuard dev_guard(h);
    if (!h || n_rows < 0 || (!x_dev && n_rows > 0)) return fail(h, "som_set_data_device: bad argument");
    (void)hipFree(h->X_owned);
    h->X_owned = nullptr;
    h->Xd = (const float*)x_dev;
    return adopt_rows(h, n_rows);
}

// Order the engine behind whoever produced device rows handed to som_set_data_device: a
// __cuda_array_interface__ `stream` value (1 = legacy default stream, 2 = per-thread default stream, else a
// hipStream_t), or -- has_stream == 0, the producer named none -- the whole device.
int som_sync_producer(som_handle* h, uint64_t stream, int32_t has_stream) {
    DeviceGuard dev_guard(h);
    if (!h) return 1;
    if (!has_stream) { HIPCHK(h, hipDeviceSynchronize()); return 0; }
    hipStream_t st = stream == 1 ? (hipStream_t)0 : stream == 2 ? hipStreamPerThread : (hipStream_t)(uintptr_t)stream;
    HIPCHK(h, hipStreamSynchronize(st));
    return 0;
}

// Device rows the caller owns, copied back to host memory (analysis calls on data that lives in HBM).
int som_copy_to_host(som_handle* h, const void* x_dev, uint64_t bytes, void* dst_host) {
    DeviceGuard dev_guard(h);
    if (!h || (bytes > 0 && (!x_dev || !dst_host))) return fail(h, "som_copy_to_host: bad argument");
    if (bytes == 0) return 0;
    return d2h_blocking(h, dst_host, x_dev, (size_t)bytes);
}

namespace {

int epoch_accumulate_eager(som_handle* h, double sigma, double eta, int neigh_f64) {
    h->early = som_handle::EarlyUpdate();
    h->early.armed = h->exact && !h->capturing && h->N > 0;
    h->early.sigma = sigma; h->early.eta = eta; h->early.neigh_f64 = neigh_f64;
    int rc = run_activation_bmu(h, h->Xd, h->N, h->xsq, h->Xb, h->xmax2, h->bmu);
    if (rc == 0) rc = run_update(h, sigma, eta, neigh_f64);
    h->early = som_handle::EarlyUpdate();
    return rc;
}

void drop_graph(som_handle* h) {
    if (h->gexec) (void)hipGraphExecDestroy(h->gexec);
    h->gexec = nullptr;
}

// Capture one epoch (codebook prep, BMU, sort, segment sum, tables, transform) into a hipGraph.
// Every launch parameter except sigma / eta / the neighbourhood dtype is a function of the handle's
// buffers and row count, which the caller has checked are unchanged; those three travel through np_dev.
// Returns 0 with h->gexec set, or 0 with h->use_graph cleared (the eager path then runs as before).
int capture_epoch_graph(som_handle* h) {
    drop_graph(h);
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeRelaxed) != hipSuccess) {
        (void)hipGetLastError();
        h->use_graph = false;
        return 0;
    }
    h->capturing = true;
    mark_codebook_changed(h);                           // the replayed epoch always rebuilds the operands its BMU kernel needs
    const std::string saved_err = h->err;
    int rc = epoch_accumulate_eager(h, 1.0, 1.0, 1);
    h->capturing = false;
    hipError_t e = hipStreamEndCapture(h->stream, &graph);
    if (rc == 0 && e == hipSuccess && graph != nullptr) e = hipGraphInstantiate(&h->gexec, graph, nullptr, nullptr, 0);
    if (graph) (void)hipGraphDestroy(graph);
    if (rc != 0 || e != hipSuccess || h->gexec == nullptr) {
        (void)hipGetLastError();
        h->gexec = nullptr;
        h->use_graph = false;                           // this handle stays on the eager path
        h->err = saved_err;
        mark_codebook_changed(h);
        if (h->debug) std::fprintf(stderr, "[somhip] epoch graph capture failed; eager launches\n");
        return 0;
    }
    h->gexec_gen = h->alloc_gen;
    h->gexec_rows = h->Xd;
    h->gexec_n = h->N;
    return 0;
}

}  // namespace

int som_epoch_accumulate(som_handle* h, double sigma, double eta, int neigh_f64) {
    DeviceGuard dev_guard(h);
    if (!h) return 1;
    if (!h->Xd && h->N > 0) return fail(h, "som_epoch_accumulate: no resident data (call som_set_data)");
    // launch-bound maps (many short kernels per epoch) replay a captured graph; profiling needs the
    // per-kernel events of the eager path
    if (h->use_graph && !h->prof && h->N > 0 && !h->exact) {
        const bool same = h->gexec && h->gexec_gen == h->alloc_gen && h->gexec_rows == h->Xd && h->gexec_n == h->N;
        if (!same) {
            if (h->graph_warm == 0 || h->gexec_rows != h->Xd || h->gexec_n != h->N) {
                // first epoch over these rows runs eagerly: it sizes every scratch buffer
                drop_graph(h);
                h->gexec_rows = h->Xd; h->gexec_n = h->N; h->graph_warm = 1;
                return epoch_accumulate_eager(h, sigma, eta, neigh_f64);
            }
            if (int rc = capture_epoch_graph(h)) return rc;
        }
        if (h->gexec) {
            const NeighParams p = make_neigh_params(h, sigma, eta, neigh_f64);
            store_params_kernel<<<dim3(1), dim3(64), 0, h->stream>>>(p, (NeighParams*)h->np_dev);
            HIPCHK(h, hipGraphLaunch(h->gexec, h->stream));
            // what the captured refresh_codebook_operands(h, precision == F32) rebuilt
            const bool bf = h->cfg.precision != SOM_PREC_F32;
            h->w_dirty = false;
            if (!bf) h->wf_dirty = false;
            if (!bf || h->cfg.distance == SOM_DIST_COSINE) h->wsq_dirty = false;
            return 0;
        }
    }
    return epoch_accumulate_eager(h, sigma, eta, neigh_f64);
}

// the same accumulator through the update as the reference states it (update.hpp, faithful_update_f32_kernel)
int som_epoch_accumulate_faithful(som_handle* h, double sigma, double eta, int neigh_f64) {
    DeviceGuard dev_guard(h);
    if (!h) return 1;
    if (!h->Xd && h->N > 0) return fail(h, "som_epoch_accumulate_faithful: no resident data (call som_set_data)");
    if (h->swapped) return fail(h, "som_epoch_accumulate_faithful: mexican_hat with compact_support on the rectangular "
                                   "topology is not a sum of row factor x column factor terms");
    if (int rc = run_activation_bmu(h, h->Xd, h->N, h->xsq, h->Xb, h->xmax2, h->bmu)) return rc;
    Timed t(h, SOM_K_KRON);
    if (int rc = build_tables(h, sigma, eta, neigh_f64, h->stream)) return rc;
    HIPCHK(h, hipMemsetAsync(h->ACC, 0, (size_t)h->K * h->D1p * sizeof(float), h->stream));
    const dim3 grid((unsigned)cdiv(h->D, LM_BN), (unsigned)cdiv(h->K, LM_BM));
    faithful_update_f32_kernel<<<grid, dim3(256), 0, h->stream>>>(h->Xd, h->bmu, h->N, h->D, h->D1p, h->X, h->Y, h->nt, h->P1,
                                                              h->P2, h->ACC);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int som_epoch_accumulate_forced(som_handle* h, const int32_t* bmu_host, double sigma, double eta, int neigh_f64) {
    DeviceGuard dev_guard(h);
    if (!h || (!bmu_host && h->N > 0)) return fail(h, "som_epoch_accumulate_forced: bad argument");
    for (long i = 0; i < h->N; ++i)
        if (bmu_host[i] < 0 || bmu_host[i] >= h->K) return fail(h, "som_epoch_accumulate_forced: id out of range");
    if (h->N > 0)
        if (int rc = h2d_blocking(h, h->bmu, bmu_host, (size_t)h->N * sizeof(int))) return rc;
    return run_update(h, sigma, eta, neigh_f64);
}

// ---- streamed epoch: rows pass through HBM chunk by chunk (out-of-core data) ------------------
int som_stream_begin(som_handle* h) {
    DeviceGuard dev_guard(h);
    if (!h) return 1;
    if (int rc = refresh_codebook_operands(h, h->cfg.precision == SOM_PREC_F32, h->exact)) return rc;
    HIPCHK(h, hipMemsetAsync(h->SC, 0, (size_t)h->K * (h->D1p + 1) * sizeof(float), h->stream));
    h->streaming = true;
    return 0;
}

static int ensure_slot(som_handle* h, som_handle::Slot& sl, long n) {
    if (!sl.copied) {
        HIPCHK(h, hipEventCreateWithFlags(&sl.copied, hipEventDisableTiming));
        HIPCHK(h, hipEventCreateWithFlags(&sl.consumed, hipEventDisableTiming));
    }
    if (n <= sl.cap) return 0;
    if (sl.used) HIPCHK(h, hipEventSynchronize(sl.consumed));
    (void)hipFree(sl.dX); (void)hipFree(sl.dXb); (void)hipFree(sl.dxsq); (void)hipFree(sl.dbmu);
    sl.dX = nullptr; sl.dXb = nullptr; sl.dxsq = nullptr; sl.dbmu = nullptr; sl.cap = 0;
    long cap = round_up(n, ROW_PAD);
    if (int rc = dev_alloc(h, &sl.dX, (size_t)cap * h->D)) return rc;
    if (int rc = dev_alloc(h, &sl.dbmu, (size_t)cap)) return rc;
    if (int rc = dev_alloc(h, &sl.dxsq, (size_t)cap * (h->exact ? 2 : 1))) return rc;
    if (h->cfg.precision != SOM_PREC_F32)
        if (int rc = dev_alloc(h, &sl.dXb, (size_t)cap * h->dp)) return rc;
    sl.cap = cap;
    return 0;
}

static bool is_pinned_host(const void* p) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return attr.type == hipMemoryTypeHost;
}

int som_stream_rows(som_handle* h, const float* x_host, int64_t n_rows) {
    DeviceGuard dev_guard(h);
    if (!h || n_rows < 0 || (n_rows > 0 && !x_host)) return fail(h, "som_stream_rows: bad argument");
    if (!h->streaming) return fail(h, "som_stream_rows: call som_stream_begin first");
    if (n_rows == 0) return 0;
    if (n_rows > 0x7fffffffL) return fail(h, "som_stream_rows: more than 2^31-1 rows in one chunk");
    const bool pinned = is_pinned_host(x_host);
    if (!pinned) if (int rc = ensure_query_scratch(h, n_rows)) return rc;
    if (n_rows > h->st_seg.cap) {
        HIPCHK(h, hipStreamSynchronize(h->stream));       // earlier chunks' kernels still read the old scratch
        if (int rc = seg_reserve(h, h->st_seg, round_up(n_rows, 1024))) return rc;
    }
    const size_t bytes = (size_t)n_rows * h->D * sizeof(float);
    if (pinned) {
        // Pinned chunk: copy on a second stream into one of two device slots, so the transfer of
        // chunk i+1 runs under the kernels of chunk i.  The caller may reuse the buffer of call i
        // once call i+1 has returned (its copy is waited for here).
        if (!h->copy_stream) HIPCHK(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
        som_handle::Slot& prev = h->slot[h->slot_idx ^ 1];
        if (prev.used) HIPCHK(h, hipEventSynchronize(prev.copied));
        som_handle::Slot& sl = h->slot[h->slot_idx];
        h->slot_idx ^= 1;
        if (int rc = ensure_slot(h, sl, n_rows)) return rc;
        if (sl.used) HIPCHK(h, hipEventSynchronize(sl.consumed));     // the kernels that read this slot are done
        HIPCHK(h, hipMemcpyAsync(sl.dX, x_host, bytes, hipMemcpyHostToDevice, h->copy_stream));
        HIPCHK(h, hipEventRecord(sl.copied, h->copy_stream));
        HIPCHK(h, hipStreamWaitEvent(h->stream, sl.copied, 0));
        if (needs_xsq(h)) if (int rc = row_sq(h, sl.dX, n_rows, sl.dxsq)) return rc;
        if (h->cfg.precision != SOM_PREC_F32)
            if (int rc = prep_rows_bf16(h, sl.dX, n_rows, round_up(n_rows, ROW_PAD), sl.dXb, h->xmax2 + 1, sl.dxsq)) return rc;
        if (int rc = run_activation_bmu(h, sl.dX, n_rows, sl.dxsq, sl.dXb, h->xmax2 + 1, sl.dbmu)) return rc;
        if (int rc = segsum_rows(h, sl.dX, sl.dbmu, n_rows, h->st_seg, false)) return rc;
        HIPCHK(h, hipEventRecord(sl.consumed, h->stream));
        sl.used = true;
        return 0;
    }
    // pageable chunk: staged by the runtime, synchronous; the caller's buffer is free on return
    if (int rc = h2d_blocking(h, h->qX, x_host, bytes)) return rc;
    if (needs_xsq(h)) if (int rc = row_sq(h, h->qX, n_rows, h->qxsq)) return rc;
    if (h->cfg.precision != SOM_PREC_F32)
        if (int rc = prep_rows_bf16(h, h->qX, n_rows, round_up(n_rows, ROW_PAD), h->qXb, h->xmax2 + 1, h->qxsq)) return rc;
    if (int rc = run_activation_bmu(h, h->qX, n_rows, h->qxsq, h->qXb, h->xmax2 + 1, h->qbmu)) return rc;
    if (int rc = segsum_rows(h, h->qX, h->qbmu, n_rows, h->st_seg, false)) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

int som_pinned_alloc(uint64_t bytes, void** out) {
    if (!out) return 1;
    *out = nullptr;
    if (hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return 1; }
    return 0;
}

int som_pinned_free(void* p) {
    if (p && hipHostFree(p) != hipSuccess) { (void)hipGetLastError(); return 1; }
    return 0;
}

int som_stream_end(som_handle* h, double sigma, double eta, int neigh_f64) {
    DeviceGuard dev_guard(h);
    if (!h) return 1;
    if (!h->streaming) return fail(h, "som_stream_end: call som_stream_begin first");
    h->streaming = false;
    return run_transform(h, sigma, eta, neigh_f64);
}

int som_epoch_merge(som_handle* h) {
    DeviceGuard dev_guard(h);
    if (!h) return 1;
    Timed t(h, SOM_K_MERGE);
    // the half-precision paths whose operand image is a stage image (resident kernel, euclidean; wide kernel,
    // euclidean and cosine): the merge also writes the next epoch's 16-bit operands
    if (h->fuse_merge_prep && ((is_half1(h) && !h->tiled && h->cfg.distance == SOM_DIST_EUCLIDEAN) ||
                               (h->wide && !h->exact && h->n_kchunks <= 4 * WD_MP_ITERS))) {
        if (int rc = SOM_HALF(h, merge_prep_half, h)) return rc;
        HIPCHK(h, hipGetLastError());
        mark_codebook_changed(h);
        h->w_dirty = false;                              // the stage image and |w~|^2 are already the new codebook's
        return 0;
    }
    long total = (long)h->K * h->D;
    // (exact mode in patch order: a copy that was in step with the codebook stays in step -- the merge writes both)
    const bool keep_wp = h->ex_patch && !h->wp_dirty;
    merge_kernel<<<dim3((unsigned)cdiv(total, 256)), dim3(256), 0, h->stream>>>(h->W, h->ACC, h->K, h->D, h->D1p,
                                                                                keep_wp ? h->Wp : nullptr, h->ex_inv);
    HIPCHK(h, hipGetLastError());
    mark_codebook_changed(h);
    if (keep_wp) h->wp_dirty = false;
    return 0;
}

int som_epoch_accumulate_begin(som_handle* h, double sigma, double eta, int neigh_f64);
int som_epoch_accumulate_block(som_handle* h, int32_t block, int64_t* offset, int64_t* n_floats);

// ---- the exchange step inside the library: RCCL all-reduce of the fused accumulator over xGMI ----------------
namespace {
int rccl_load(const char* path) {
    if (g_rccl.lib) return 0;
    // an explicit path means that library and no other; NULL: a copy some other component of the process already
    // loaded (torch's), else the system's
    const char* search[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    const char* only[] = {path};
    const char* const* cands = path ? only : search;
    const int n_cands = path ? 1 : 3;
    void* lib = nullptr;
    for (int i = 0; i < n_cands && !lib; ++i) lib = dlopen(cands[i], RTLD_NOW | RTLD_NOLOAD);
    std::string why;
    for (int i = 0; i < n_cands && !lib; ++i) {
        lib = dlopen(cands[i], RTLD_NOW | RTLD_LOCAL);
        if (!lib) { const char* e = dlerror(); why = e ? e : ""; }   // dlerror() clears the message: read it once
    }
    if (!lib) { g_rccl_error = std::string("librccl not found: ") + why; return 1; }
    RcclApi a;
    a.lib = lib;
    a.GetUniqueId = (int (*)(som_nccl_id*))dlsym(lib, "ncclGetUniqueId");
    a.CommInitRank = (int (*)(void**, int, som_nccl_id, int))dlsym(lib, "ncclCommInitRank");
    a.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(lib, "ncclAllReduce");
    a.CommDestroy = (int (*)(void*))dlsym(lib, "ncclCommDestroy");
    a.GetErrorString = (const char* (*)(int))dlsym(lib, "ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.AllReduce || !a.CommDestroy) { g_rccl_error = "librccl: symbols missing"; return 1; }
    g_rccl = a;
    return 0;
}
int fail_rccl(som_handle* h, const char* what, int code) {
    return fail(h, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(code) : "RCCL error"));
}
constexpr int NCCL_FLOAT32 = 7, NCCL_SUM = 0;           // rccl.h: ncclFloat32, ncclSum

int comm_allreduce(som_handle* h, long offset, long n_floats, hipStream_t st) {
    int rc = g_rccl.AllReduce(h->ACC + offset, h->ACC + offset, (size_t)n_floats, NCCL_FLOAT32, NCCL_SUM, h->comm, st);
    if (rc != 0) return fail_rccl(h, "ncclAllReduce", rc);
    return 0;
}
}  // namespace

int som_comm_load(const char* librccl_path) {
    if (rccl_load(librccl_path)) { g_create_error = g_rccl_error; return 1; }
    return 0;
}

int som_comm_unique_id(void* id_out) {
    if (!id_out) return 1;
    if (rccl_load(nullptr)) { g_create_error = g_rccl_error; return 1; }
    som_nccl_id id;
    int rc = g_rccl.GetUniqueId(&id);
    if (rc != 0) { g_create_error = "ncclGetUniqueId failed"; return 1; }
    std::memcpy(id_out, id.internal, sizeof(id.internal));
    return 0;
}

int som_comm_init(som_handle* h, int32_t world, int32_t rank, const void* id_bytes) {
    DeviceGuard dev_guard(h);
    if (!h || !id_bytes || world < 1 || rank < 0 || rank >= world) return fail(h, "som_comm_init: bad argument");
    if (h->comm) return fail(h, "som_comm_init: this handle already has a communicator");
    if (rccl_load(nullptr)) return fail(h, g_rccl_error);
    som_nccl_id id;
    std::memcpy(id.internal, id_bytes, sizeof(id.internal));
    int rc = g_rccl.CommInitRank(&h->comm, world, id, rank);
    if (rc != 0) { h->comm = nullptr; return fail_rccl(h, "ncclCommInitRank", rc); }
    h->comm_world = world;
    if (!h->comm_stream) {
        HIPCHK(h, hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
        HIPCHK(h, hipEventCreateWithFlags(&h->ev_block, hipEventDisableTiming));
        HIPCHK(h, hipEventCreateWithFlags(&h->ev_comm, hipEventDisableTiming));
    }
    return 0;
}

int som_comm_destroy(som_handle* h) {
    DeviceGuard dev_guard(h);
    if (!h) return 1;
    if (h->comm) {
        (void)hipStreamSynchronize(h->stream);
        if (h->comm_stream) (void)hipStreamSynchronize(h->comm_stream);
        (void)g_rccl.CommDestroy(h->comm);
        h->comm = nullptr;
        h->comm_world = 1;
    }
    return 0;
}

int som_epoch_allreduce(som_handle* h) {
    DeviceGuard dev_guard(h);
    if (!h) return 1;
    if (!h->comm) return 0;                             // one GPU: the local sums are the global ones
    return comm_allreduce(h, 0, (long)h->K * h->D1p, h->stream);
}

// accumulate (+ all-reduce, when som_comm_init gave this handle a communicator) + merge.  With more than one
// 128-row block of the map the collective runs block by block on a second stream, each block's all-reduce under
// the next block's transform; one block: one all-reduce on the handle's own stream.
int som_epoch(som_handle* h, double sigma, double eta, int neigh_f64) {
    if (!h) return 1;
    const int nb = (int)cdiv(h->X, LM_BM);
    if (!h->comm || nb < 2) {
        if (int rc = som_epoch_accumulate(h, sigma, eta, neigh_f64)) return rc;
        if (int rc = som_epoch_allreduce(h)) return rc;
        return som_epoch_merge(h);
    }
    DeviceGuard dev_guard(h);
    if (int rc = som_epoch_accumulate_begin(h, sigma, eta, neigh_f64)) return rc;
    for (int b = 0; b < nb; ++b) {
        int64_t off = 0, n = 0;
        if (int rc = som_epoch_accumulate_block(h, b, &off, &n)) return rc;
        HIPCHK(h, hipEventRecord(h->ev_block, h->stream));
        HIPCHK(h, hipStreamWaitEvent(h->comm_stream, h->ev_block, 0));
        if (int rc = comm_allreduce(h, off, n, h->comm_stream)) return rc;
    }
    HIPCHK(h, hipEventRecord(h->ev_comm, h->comm_stream));
    HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_comm, 0));
    return som_epoch_merge(h);
}

// ---- the epoch in stages: finished map-row blocks of the accumulator can be all-reduced while the next are computed
int som_epoch_accumulate_begin(som_handle* h, double sigma, double eta, int neigh_f64) {
    DeviceGuard dev_guard(h);
    if (!h) return 1;
    if (!h->Xd && h->N > 0) return fail(h, "som_epoch_accumulate_begin: no resident data (call som_set_data)");
    if (int rc = run_activation_bmu(h, h->Xd, h->N, h->xsq, h->Xb, h->xmax2, h->bmu)) return rc;
    if (int rc = segsum_rows(h, h->Xd, h->bmu, h->N, h->seg, true)) return rc;
    Timed t(h, SOM_K_KRON);
    if (int rc = run_transform_stage1(h, sigma, eta, neigh_f64)) return rc;
    h->staged_blocks_done = 0;
    return 0;
}

int som_epoch_block_count(som_handle* h, int32_t* n_blocks) {
    if (!h || !n_blocks) return fail(h, "som_epoch_block_count: NULL argument");
    *n_blocks = (int32_t)cdiv(h->X, LM_BM);
    return 0;
}

int som_epoch_accumulate_block(som_handle* h, int32_t block, int64_t* offset, int64_t* n_floats) {
    DeviceGuard dev_guard(h);
    if (!h) return 1;
    if (h->staged_blocks_done < 0 || block != h->staged_blocks_done)
        return fail(h, "som_epoch_accumulate_block: blocks go in order, after som_epoch_accumulate_begin");
    {
        Timed t(h, SOM_K_KRON);
        if (int rc = run_transform_stage2(h, block, block + 1)) return rc;
    }
    const long slab = (long)h->Y * h->D1p, i0 = (long)block * LM_BM;
    const long rows = std::min<long>(i0 + LM_BM, h->X) - i0;
    if (offset) *offset = i0 * slab;
    if (n_floats) *n_floats = rows * slab;
    h->staged_blocks_done = block + 1 == (int)cdiv(h->X, LM_BM) ? -1 : block + 1;
    return 0;
}

int som_accum_device_ptr(som_handle* h, void** dev_ptr, int64_t* n_floats) {
    if (!h || !dev_ptr || !n_floats) return fail(h, "som_accum_device_ptr: NULL argument");
    *dev_ptr = h->ACC;
    *n_floats = (int64_t)h->K * h->D1p;
    return 0;
}

int som_get_stream(som_handle* h, void** stream_out) {
    if (!h || !stream_out) return fail(h, "som_get_stream: NULL argument");
    *stream_out = (void*)h->stream;
    return 0;
}

int som_epoch_fetch(som_handle* h, float* num, float* den, int32_t* bmu) {
    DeviceGuard dev_guard(h);
    if (!h) return 1;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (num || den) {
        std::vector<float> acc((size_t)h->K * h->D1p);
        HIPCHK(h, hipMemcpy(acc.data(), h->ACC, acc.size() * sizeof(float), hipMemcpyDeviceToHost));
        for (long k = 0; k < h->K; ++k) {
            if (num) std::memcpy(num + k * h->D, acc.data() + k * h->D1p, (size_t)h->D * sizeof(float));
            if (den) den[k] = acc[k * h->D1p + h->D];
        }
    }
    if (bmu && h->N > 0) HIPCHK(h, hipMemcpy(bmu, h->bmu, (size_t)h->N * sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}

namespace {
// BMU under the full Euclidean distance for quantization / quantization_error (xpysom.py:632-645, 699-707)
// of the n_rows rows in the query scratch.  f32 precision: the reference's sqrt'd distance, bit for bit.
// bf16 / f16 precision with the 'euclidean' activation distance: the same argmin through the configured
// MFMA path (the squared distance is monotone in it); the caller evaluates the distance itself exactly.
// value_only (quantization_error): the caller evaluates the distance to the chosen unit itself and wants no id -- in EXACT
// precision with the 'euclidean' activation distance the screen + re-score then serves (its pick is float32's argmin of the
// squared distance's unit part; the sqrt'd distance can only tie where that one is within an ulp: the same distance).
int run_quantization_bmu(som_handle* h, const float* X, long n_rows, bool value_only = false) {
    if (h->cfg.precision != SOM_PREC_F32 && h->cfg.distance == SOM_DIST_EUCLIDEAN && (!h->exact || value_only)) {
        if (h->exact) if (int rc = row_sq(h, X, n_rows, h->qxsq)) return rc;
        if (int rc = prep_rows_bf16(h, X, n_rows, round_up(n_rows, ROW_PAD), h->qXb, h->xmax2 + 1, h->qxsq)) return rc;
        return run_activation_bmu(h, X, n_rows, h->qxsq, h->qXb, h->xmax2 + 1, h->qbmu);
    }
    if (int rc = refresh_codebook_operands(h, true)) return rc;
    if (int rc = row_sq(h, X, n_rows, h->qxsq)) return rc;
    Timed t(h, SOM_K_BMU);
    return launch_bmu_f32_any<SCORE_EUCLID_SQRT>(h, X, n_rows, h->qxsq, h->qbmu);
}

// BMUs (activation or quantization rule) of n device rows into qbmu
int run_query_bmu(som_handle* h, const float* X, long n_rows, int mode) {
    if (mode == SOM_BMU_QUANTIZATION) return run_quantization_bmu(h, X, n_rows);
    if (needs_xsq(h)) if (int rc = row_sq(h, X, n_rows, h->qxsq)) return rc;
    if (h->cfg.precision != SOM_PREC_F32)
        if (int rc = prep_rows_bf16(h, X, n_rows, round_up(n_rows, ROW_PAD), h->qXb, h->xmax2 + 1, h->qxsq)) return rc;
    return run_activation_bmu(h, X, n_rows, h->qxsq, h->qXb, h->xmax2 + 1, h->qbmu);
}

int run_quantization_error(som_handle* h, const float* X, long n_rows, double* qe_out) {
    if (int rc = run_quantization_bmu(h, X, n_rows, true)) return rc;
    HIPCHK(h, hipMemsetAsync(h->dsum, 0, sizeof(double), h->stream));
    qe_kernel<<<dim3((unsigned)std::min<long>(cdiv(n_rows, 4), 4096)), dim3(256), 0, h->stream>>>(X, h->qbmu, h->W, n_rows, h->D, h->dsum);
    HIPCHK(h, hipGetLastError());
    double s = 0.0;
    if (int rc = d2h_blocking(h, &s, h->dsum, sizeof(double))) return rc;
    *qe_out = s / (double)n_rows;
    return 0;
}
}  // namespace

int som_bmu(som_handle* h, const float* x_host, int64_t n_rows, int32_t mode, int32_t* ids_out) {
    DeviceGuard dev_guard(h);
    if (!h || n_rows < 0 || (n_rows > 0 && (!x_host || !ids_out))) return fail(h, "som_bmu: bad argument");
    if (mode != SOM_BMU_ACTIVATION && mode != SOM_BMU_QUANTIZATION) return fail(h, "som_bmu: unknown mode");
    if (n_rows == 0) return 0;
    if (int rc = ensure_query_scratch(h, n_rows)) return rc;
    if (int rc = h2d_blocking(h, h->qX, x_host, (size_t)n_rows * h->D * sizeof(float))) return rc;
    if (int rc = run_query_bmu(h, h->qX, n_rows, mode)) return rc;
    return d2h_blocking(h, ids_out, h->qbmu, (size_t)n_rows * sizeof(int));
}

int som_bmu_device(som_handle* h, const void* x_dev, int64_t n_rows, int32_t mode, int32_t* ids_out) {
    DeviceGuard dev_guard(h);
    if (!h || n_rows < 0 || (n_rows > 0 && (!x_dev || !ids_out))) return fail(h, "som_bmu_device: bad argument");
    if (mode != SOM_BMU_ACTIVATION && mode != SOM_BMU_QUANTIZATION) return fail(h, "som_bmu_device: unknown mode");
    if (n_rows == 0) return 0;
    if (n_rows > 0x7fffffffL) return fail(h, "som_bmu_device: more than 2^31-1 rows in one call");
    if (int rc = ensure_query_scratch(h, n_rows, false)) return rc;
    if (int rc = run_query_bmu(h, (const float*)x_dev, n_rows, mode)) return rc;
    return d2h_blocking(h, ids_out, h->qbmu, (size_t)n_rows * sizeof(int));
}

int som_quantization_error_device(som_handle* h, const void* x_dev, int64_t n_rows, double* qe_out) {
    DeviceGuard dev_guard(h);
    if (!h || !qe_out || n_rows < 0 || (n_rows > 0 && !x_dev)) return fail(h, "som_quantization_error_device: bad argument");
    if (n_rows == 0) { *qe_out = NAN; return 0; }
    if (n_rows > 0x7fffffffL) return fail(h, "som_quantization_error_device: more than 2^31-1 rows in one call");
    if (int rc = ensure_query_scratch(h, n_rows, false)) return rc;
    return run_quantization_error(h, (const float*)x_dev, n_rows, qe_out);
}

int som_bmu_f64(som_handle* h, const double* x_host, int64_t n_rows, int32_t* ids_out) {
    DeviceGuard dev_guard(h);
    if (!h || n_rows < 0 || (n_rows > 0 && (!x_host || !ids_out))) return fail(h, "som_bmu_f64: bad argument");
    if (h->cfg.distance != SOM_DIST_EUCLIDEAN) return fail(h, "som_bmu_f64: the float64 query path serves the 'euclidean' activation distance");
    if (n_rows == 0) return 0;
    const size_t w_bytes = (size_t)PW_UNITS * h->D * sizeof(float);
    if (w_bytes > 150 * 1024) return fail(h, "som_bmu_f64: input_len too large for the LDS unit tile");
    if (int rc = ensure_query_scratch(h, n_rows)) return rc;
    if ((size_t)n_rows * h->D > h->qX64_cap) {
        (void)hipFree(h->qX64);
        h->qX64 = nullptr; h->qX64_cap = 0;
        const size_t cap = (size_t)round_up(n_rows, 1024) * h->D;
        if (int rc = dev_alloc(h, &h->qX64, cap)) return rc;
        h->qX64_cap = cap;
    }
    if (int rc = h2d_blocking(h, h->qX64, x_host, (size_t)n_rows * h->D * sizeof(double))) return rc;
    if (int rc = refresh_codebook_operands(h, true)) return rc;       // |w|^2 in NumPy's float32 order
    {
        Timed t(h, SOM_K_BMU);
        bmu_f64_kernel<<<dim3((unsigned)cdiv(n_rows, PW_SAMPLES)), dim3(PW_SAMPLES), w_bytes, h->stream>>>(
            h->qX64, n_rows, h->D, h->W, h->wsq, h->K, h->qbmu);
        HIPCHK(h, hipGetLastError());
    }
    return d2h_blocking(h, ids_out, h->qbmu, (size_t)n_rows * sizeof(int));
}

int som_bmu_top2(som_handle* h, const float* x_host, int64_t n_rows, int32_t* ids1_out, int32_t* ids2_out) {
    DeviceGuard dev_guard(h);
    if (!h || n_rows < 0 || (n_rows > 0 && (!x_host || !ids1_out || !ids2_out))) return fail(h, "som_bmu_top2: bad argument");
    if (n_rows == 0) return 0;
    if (int rc = ensure_query_scratch(h, n_rows)) return rc;
    if (int rc = h2d_blocking(h, h->qX, x_host, (size_t)n_rows * h->D * sizeof(float))) return rc;
    if (int rc = refresh_codebook_operands(h, true)) return rc;
    if (int rc = row_sq(h, h->qX, n_rows, h->qxsq)) return rc;
    {
        Timed t(h, SOM_K_BMU);
        if (int rc = launch_bmu_top2(h, h->qX, n_rows, h->qxsq, h->qbmu, h->qbmu2)) return rc;
    }
    if (int rc = d2h_blocking(h, ids1_out, h->qbmu, (size_t)n_rows * sizeof(int))) return rc;
    return d2h_blocking(h, ids2_out, h->qbmu2, (size_t)n_rows * sizeof(int));
}

int som_distance_matrix(som_handle* h, const float* x_host, int64_t n_rows, int32_t mode, float* dist_out) {
    DeviceGuard dev_guard(h);
    if (!h || n_rows < 0 || (n_rows > 0 && (!x_host || !dist_out))) return fail(h, "som_distance_matrix: bad argument");
    if (mode != SOM_BMU_ACTIVATION && mode != SOM_BMU_QUANTIZATION) return fail(h, "som_distance_matrix: unknown mode");
    if (mode == SOM_BMU_ACTIVATION && h->cfg.distance > SOM_DIST_COSINE)
        return fail(h, "som_distance_matrix: only the GEMM-form distances (euclidean, euclidean_no_opt, cosine)");
    if (n_rows == 0) return 0;
    if ((double)n_rows * h->K > 2.0e9) return fail(h, "som_distance_matrix: n_rows * K too large (analysis call, chunk it)");
    if (int rc = ensure_query_scratch(h, n_rows)) return rc;
    if (int rc = h2d_blocking(h, h->qX, x_host, (size_t)n_rows * h->D * sizeof(float))) return rc;
    if (int rc = refresh_codebook_operands(h, true)) return rc;
    if (int rc = row_sq(h, h->qX, n_rows, h->qxsq)) return rc;
    float* dm = nullptr;
    if (int rc = dev_alloc(h, &dm, (size_t)n_rows * h->K)) return rc;
    int rc = 0;
    if (mode == SOM_BMU_QUANTIZATION) rc = launch_dist_matrix<SCORE_EUCLID_SQRT>(h, n_rows, dm);
    else if (h->cfg.distance == SOM_DIST_EUCLIDEAN) rc = launch_dist_matrix<SCORE_EUCLID_PART>(h, n_rows, dm);
    else if (h->cfg.distance == SOM_DIST_EUCLIDEAN_NO_OPT) rc = launch_dist_matrix<SCORE_EUCLID_SQ>(h, n_rows, dm);
    else rc = launch_dist_matrix<SCORE_COSINE>(h, n_rows, dm);
    if (!rc) {
        rc = d2h_blocking(h, dist_out, dm, (size_t)n_rows * h->K * sizeof(float));
    }
    (void)hipStreamSynchronize(h->stream);
    (void)hipFree(dm);
    return rc;
}

int som_quantization_error(som_handle* h, const float* x_host, int64_t n_rows, double* qe_out) {
    DeviceGuard dev_guard(h);
    if (!h || !qe_out || n_rows < 0 || (n_rows > 0 && !x_host)) return fail(h, "som_quantization_error: bad argument");
    if (n_rows == 0) { *qe_out = NAN; return 0; }     // numpy: mean of an empty array
    if (int rc = ensure_query_scratch(h, n_rows)) return rc;
    if (int rc = h2d_blocking(h, h->qX, x_host, (size_t)n_rows * h->D * sizeof(float))) return rc;
    return run_quantization_error(h, h->qX, n_rows, qe_out);
}

int som_set_verify(som_handle* h, int32_t n_rows) {
    if (!h || n_rows < 0) return fail(h, "som_set_verify: bad argument");
    h->verify_rows = n_rows;
    return 0;
}

int som_verify_stats(som_handle* h, int64_t* launches, int64_t* rows_checked) {
    if (!h) return 1;
    if (launches) *launches = h->verify_launches;
    if (rows_checked) *rows_checked = h->verify_rows_checked;
    return 0;
}

// TEST HOOK of the canary: overwrite the operand images the BMU kernels read (the 16-bit stage / tile image and the
// float32 stage image) with zeros WITHOUT marking them stale -- what a lost or torn staging copy would look like.
int som_debug_corrupt_operands(som_handle* h, int32_t which) {
    DeviceGuard dev_guard(h);
    if (!h) return 1;
    if (int rc = refresh_codebook_operands(h, true, h->exact)) return rc;   // (in the order the next BMU launch wants: no rebuild there)
    if ((which & 1) && h->Wst) {
        size_t bytes = (size_t)h->n_stages * h->stage_bytes;
        if (h->tiled && !h->wide) bytes = (size_t)h->n_ublocks * h->n_kchunks * h->tl_wtile;
        HIPCHK(h, hipMemsetAsync(h->Wst, 0, bytes, h->stream));
    }
    if ((which & 2) && h->Wfst) HIPCHK(h, hipMemsetAsync(h->Wfst, 0, (size_t)h->fr_stages * fr_stage_bytes(h->fr_kg), h->stream));
    if ((which & 2) && h->Wfimg) HIPCHK(h, hipMemsetAsync(h->Wfimg, 0, (size_t)h->ft_ublocks * h->ft_kchunks * FT_WTILE, h->stream));
    return 0;
}

// TEST HOOK (no device needed): the exact mode's policy functions (csrc/exact_policy.hpp) on caller-supplied numbers.
// costs[9] = {full_total, full_screen, plan_total, plan_over, plan_over_scout, blk_ms, l2_ms_group, l2_ratio, sort_ms};
// which / args: 0 commit_scouted_plan(share, blocks_per_row) | 1 level2_from_sample(share, share1, blocks_per_row) |
// 2 level2_pays(share, share1) | 3 sort_paid(share_stale, share_fresh, blocks_per_row, epochs_served) | 4 plan_idle(share) |
// 5 scout_continues(win_share, share_last, blocks_per_row) | 6 rows_worth_a_scout(rows, units, features).  *out = 0 / 1.
int som_policy_eval(int32_t which, const double* costs, const double* args, int32_t* out) {
    if (!costs || !args || !out) return 1;
    policy::Costs c;
    c.full_total = costs[0]; c.full_screen = costs[1]; c.plan_total = costs[2]; c.plan_over = costs[3]; c.plan_over_scout = costs[4];
    c.blk_ms = costs[5]; c.l2_ms_group = costs[6]; c.l2_ratio = costs[7]; c.sort_ms = costs[8];
    switch (which) {
    case 0: *out = policy::commit_scouted_plan(c, args[0], args[1]); return 0;
    case 1: *out = policy::level2_from_sample(c, args[0], args[1], args[2]); return 0;
    case 2: *out = policy::level2_pays(c, args[0], args[1]); return 0;
    case 3: *out = policy::sort_paid(c, args[0], args[1], args[2], (int)args[3]); return 0;
    case 4: *out = policy::plan_idle(c, args[0]); return 0;
    case 5: *out = policy::scout_continues(c, args[0], args[1], args[2]); return 0;
    case 6: *out = policy::rows_worth_a_scout(args[0], args[1], args[2]); return 0;
    }
    return 1;
}

int som_debug_mfma16(som_handle* h, const uint16_t* a_host, const uint16_t* b_host, const float* c_host, float* d_host,
                     int32_t is_f16) {
    DeviceGuard dev_guard(h);
    if (!h || !a_host || !b_host || !c_host || !d_host) return fail(h, "som_debug_mfma16: NULL argument");
    uint16_t *a = nullptr, *b = nullptr;
    float *c = nullptr, *d = nullptr;
    int rc = 0;
    if ((rc = dev_alloc(h, &a, 512)) || (rc = dev_alloc(h, &b, 512)) || (rc = dev_alloc(h, &c, 256)) || (rc = dev_alloc(h, &d, 256))) {
        void* p[] = {a, b, c, d};
        for (void* q : p) if (q) (void)hipFree(q);
        return rc;
    }
    rc = h2d_blocking(h, a, a_host, 1024) || h2d_blocking(h, b, b_host, 1024) || h2d_blocking(h, c, c_host, 1024);
    if (!rc) {
        if (is_f16) debug_mfma16_kernel<F16><<<dim3(1), dim3(64), 0, h->stream>>>(a, b, c, d);
        else debug_mfma16_kernel<Bf16><<<dim3(1), dim3(64), 0, h->stream>>>(a, b, c, d);
        rc = d2h_blocking(h, d_host, d, 1024);
    }
    void* p[] = {a, b, c, d};
    for (void* q : p) (void)hipFree(q);
    return rc;
}

// Diagnostic builds only (-DSOM_STAMPS): hand the BMU kernels a buffer for their in-kernel clock stamps (n_pairs
// workgroups' worth; NULL detaches it), or read it back.  In the product build both calls fail with a message.
int som_debug_stamps(som_handle* h, int64_t n_pairs, uint64_t* out_host) {
    DeviceGuard dev_guard(h);
    if (!h) return 1;
#ifdef SOM_STAMPS
    static unsigned long long* buf = nullptr;
    static int64_t cap = 0;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (out_host) {
        if (n_pairs > cap) return fail(h, "som_debug_stamps: more pairs than the buffer holds");
        HIPCHK(h, hipMemcpy(out_host, buf, (size_t)n_pairs * 16, hipMemcpyDeviceToHost));
        return 0;
    }
    if (buf) { (void)hipFree(buf); buf = nullptr; cap = 0; }
    if (n_pairs > 0) {
        HIPCHK(h, hipMalloc((void**)&buf, (size_t)n_pairs * 16));
        HIPCHK(h, hipMemset(buf, 0, (size_t)n_pairs * 16));
        cap = n_pairs;
    }
    HIPCHK(h, hipMemcpyToSymbol(HIP_SYMBOL(g_som_stamps), &buf, sizeof(buf)));
    return 0;
#else
    (void)n_pairs; (void)out_host;
    return fail(h, "som_debug_stamps: this library was built without -DSOM_STAMPS (tools/stamps.py builds the diagnostic one)");
#endif
}

int som_patch_order(int32_t x, int32_t y, int32_t* perm_out) {
    if (x <= 0 || y <= 0 || !perm_out || (int64_t)x * y > 0x7fffffffLL) return 1;
    std::vector<int> perm;
    patch_order(x, y, perm);
    std::copy(perm.begin(), perm.end(), perm_out);
    return 0;
}

int som_exact_stats(som_handle* h, int64_t* rows, int64_t* rows_fallback, int64_t* passes) {
    if (!h) return 1;
    if (rows) *rows = h->ex.rows_total;
    if (rows_fallback) *rows_fallback = h->ex.rows_fallback;
    if (passes) *passes = h->ex.chunks;
    return 0;
}

int som_exact_skip_stats(som_handle* h, int64_t* blocks_run, int64_t* blocks_total) {
    if (!h || !blocks_run || !blocks_total) return 1;
    *blocks_run = h->ex.blocks_run; *blocks_total = h->ex.blocks_total;
    return 0;
}

int som_exact_resident_stats(som_handle* h, int64_t* planned_epochs, int64_t* sorts) {
    if (!h || !planned_epochs || !sorts) return 1;
    *planned_epochs = h->ex.planned; *sorts = h->ex.resorts;
    return 0;
}

int som_exact_scout_stats(som_handle* h, int64_t* scouted_launches, int64_t* transient_planned) {
    if (!h || !scouted_launches || !transient_planned) return 1;
    *scouted_launches = h->ex.scouted; *transient_planned = h->ex.tr_planned;
    return 0;
}

int som_exact_refine_stats(som_handle* h, int64_t* pairs_in, int64_t* pairs_out) {
    if (!h || !pairs_in || !pairs_out) return 1;
    *pairs_in = h->ex.pairs_refined_in; *pairs_out = h->ex.pairs_refined_out;
    return 0;
}

int som_exact_last_counts(som_handle* h, int32_t* counts_out, int64_t n) {
    DeviceGuard dev_guard(h);
    if (!h || !counts_out || n < 0) return fail(h, "som_exact_last_counts: bad argument");
    if (!h->exact || n > h->ex.stride) return fail(h, "som_exact_last_counts: no screen pass of that many rows");
    if (int rc = d2h_blocking(h, counts_out, h->ex.rowcnt, (size_t)n * sizeof(int32_t))) return rc;
    return 0;
}

int som_sync(som_handle* h) {
    DeviceGuard dev_guard(h);
    if (!h) return 1;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

int som_profile_enable(som_handle* h, int32_t on) {
    DeviceGuard dev_guard(h);
    if (!h) return 1;
    if (!on) if (int rc = resolve_profile(h)) return rc;
    h->prof = on == 2 ? 2 : on != 0;
    return 0;
}

int som_profile_get(som_handle* h, int32_t kernel, double* total_ms, int64_t* launches) {
    DeviceGuard dev_guard(h);
    if (!h || kernel < 0 || kernel >= SOM_K_COUNT) return fail(h, "som_profile_get: bad argument");
    if (int rc = resolve_profile(h)) return rc;
    if (total_ms) *total_ms = h->ms[kernel];
    if (launches) *launches = h->launches[kernel];
    return 0;
}

int som_profile_reset(som_handle* h) {
    DeviceGuard dev_guard(h);
    if (!h) return 1;
    if (int rc = resolve_profile(h)) return rc;
    for (int i = 0; i < SOM_K_COUNT; ++i) { h->ms[i] = 0; h->launches[i] = 0; }
    return 0;
}

}  // extern "C"
