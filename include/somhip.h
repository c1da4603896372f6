/*
 * somhip.h -- C ABI of the MI355X-native batch-SOM engine (libsomhip.so).
 *
 * The reference (jcfaracco/xpysom-dask) has no FFI layer of its own: its hot
 * path is reached through `xp.*` array calls inside the Python class XPySom.
 * This header is the boundary a drop-in replaces those calls with.  Every entry
 * point below names the reference code it stands in for (file:line under
 * /root/reference/xpysom_dask/).  The Python host (xpysom_dask_amd/xpysom.py)
 * binds these with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.
 *   - Every call returns 0 on success, non-zero on failure; the message is
 *     available from som_last_error(handle) (or som_last_error(NULL) when
 *     som_create itself failed).  Nothing throws across the ABI.
 *   - The caller owns all host buffers; the library owns all device buffers.
 *   - One handle = one GPU = one host thread at a time (not locked).
 *   - Codebook layout: float32 [K][D] row-major, K = X*Y, unit k = i*Y + j
 *     (xpysom.py:240 unravel table; distances.py:185 reshape).
 *   - Data layout: float32 [N][D] row-major (xpysom.py:510).
 */
#ifndef SOMHIP_H
#define SOMHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct som_handle som_handle;

/* activation distance: distances.py:162-170 registry (names kept) */
enum { SOM_DIST_EUCLIDEAN = 0,        /* 'euclidean'  -> euclidean_squared_distance_part, distances.py:11-23 */
       SOM_DIST_EUCLIDEAN_NO_OPT = 1, /* 'euclidean_no_opt' -> euclidean_squared_distance, distances.py:25-31 */
       SOM_DIST_COSINE = 2,           /* 'cosine'     -> cosine_distance, distances.py:45-59 */
       SOM_DIST_MANHATTAN = 3,        /* 'manhattan', 'manhattan_no_opt' -> distances.py:109-158 (incl. the `l1norm` CUDA kernel) */
       SOM_DIST_NORM_P = 4,           /* 'norm_p' -> norm_p_power_distance, distances.py:98-107 (even p: binomial form :77-96) */
       SOM_DIST_NORM_P_NO_OPT = 5 };  /* 'norm_p_no_opt' -> norm_p_power_distance_generic, distances.py:61-75 */

/* neighbourhood function on the rectangular topology: neighborhoods.py:14-33, :57-74, :99-130 */
enum { SOM_NEIGH_GAUSSIAN = 0, SOM_NEIGH_MEXICAN_HAT = 1, SOM_NEIGH_BUBBLE = 2, SOM_NEIGH_TRIANGLE = 3 };

/* map topology: xpysom.py:196-206.  HEXAGONAL selects the *_generic neighbourhoods
 * (neighborhoods.py:35-55, :76-97) on the shifted-row coordinates; bubble ignores the shift. */
enum { SOM_TOPO_RECTANGULAR = 0, SOM_TOPO_HEXAGONAL = 1 };

/* arithmetic of the distance GEMM (the -2 x.w^T term, distances.py:22):
 *   F32  : v_mfma_f32_32x32x2_f32, exact float32 fma chain -- the parity mode
 *   BF16 : v_mfma_f32_16x16x32_bf16 on bf16-rounded x and w, f32 accumulate -- the throughput mode
 *   F16  : the BF16 kernels instantiated on IEEE half (v_mfma_f32_16x16x32_f16): 11 significant
 *           bits per operand instead of 8 at the same MFMA rate (97 % of the bf16 throughput under the chip's power
 *           limit); rows and units must fit float16 -- som_set_data / som_set_weights refuse rows and units whose
 *           norm exceeds 65504; streamed chunks and query rows (som_stream_rows, som_bmu) are not checked: values
 *           beyond the range saturate at +-65504
 *   EXACT: the BMUs of F32, row for row and bit for bit (near-ties and exact ties included), at half-precision MFMA
 *           speed: one pass of the MFMA kernel on power-of-two-scaled IEEE-half operands screens every unit and records
 *           the minimum of every 64-unit group per row (a group = a compact patch of the map: 8 x 8 units where both
 *           sides are multiples of 8), and the float32 fma chain itself re-scores the groups the
 *           screen's rigorous (partly measured) error bound cannot rule out; rows it cannot vouch for (NaN / infinite
 *           values, a pass with more candidate pairs than re-scoring is worth) go to the F32 kernel.  Euclidean distance
 *           with input_len <= 128; euclidean and cosine with 128 < input_len <= 800 on maps of >= 4096 units; other
 *           configurations run the F32 kernels under this id (which are the exact mode by definition).  On maps of >= 4096
 *           units the screen SKIPS the (256-row tile, block of units) pairs a centroid-and-radius bound proves empty
 *           (csrc/exact_skip.hpp, exact_skip_wide.hpp; SOM_EXACT_SKIP=0 runs every block): around last epoch's BMU for
 *           resident rows from their second epoch on (input_len <= 128; euclidean beyond), around a pseudo last BMU found
 *           from the current codebook's own group centroids for every other large row set -- query rows (som_bmu,
 *           som_bmu_device, som_quantization_error*), streamed chunks, a row set's first epoch.  A forecast on sample tiles
 *           declines plans with nothing to skip; the plan's own decisions come from timed costs of the handle's launches.
 *           The ids do not change, the time does, by the data.  Everything but the BMU search (update, merge,
 *           quantization) is as in F32. */
enum { SOM_PREC_F32 = 0, SOM_PREC_BF16 = 1,
       SOM_PREC_RETIRED_2 = 2,  /* was BF16X3 (hi/lo-split operands, three MFMAs per product): som_create refuses it --
                                   EXACT returns float32's own BMUs at three to ten times its speed; the id stays reserved */
       SOM_PREC_F16 = 3,      /* the bf16 path on IEEE half operands: 11 significant bits instead of 8, |value| <= 65504 */
       SOM_PREC_RETIRED_4 = 4,  /* was F16X3: refused likewise */
       SOM_PREC_EXACT = 5 };  /* F32's BMUs through an IEEE-half MFMA screen + float32 re-score (bmu_exact.hpp) */

/* which BMU rule som_bmu applies */
enum { SOM_BMU_ACTIVATION = 0,   /* configured activation distance: XPySom._winner, xpysom.py:410-417 */
       SOM_BMU_QUANTIZATION = 1  /* full sqrt'd Euclidean + nan_to_num: _quantization, xpysom.py:640,670 */ };

/* timed kernels for som_profile_get */
enum { SOM_K_BMU = 0, SOM_K_SEGSUM = 1, SOM_K_KRON = 2, SOM_K_MERGE = 3, SOM_K_PREP = 4,
       SOM_K_SCREEN = 5,   /* precision EXACT: the MFMA screen kernel alone (a part of SOM_K_BMU's time) */
       SOM_K_COUNT = 6 };

typedef struct som_config {
    int32_t x, y, input_len;     /* map rows, cols, features: XPySom.__init__, xpysom.py:73 */
    int32_t distance;            /* SOM_DIST_*  */
    int32_t neighborhood;        /* SOM_NEIGH_* */
    int32_t compact_support;     /* neighborhoods.py:29-31; with SOM_NEIGH_MEXICAN_HAT: the reference's double mask on px, :69-71 / :91-93 */
    int32_t precision;           /* SOM_PREC_*  */
    int32_t device;              /* HIP device ordinal */
    double  std_coeff;           /* d = 2*std_coeff^2*sigma^2, neighborhoods.py:19 */
    void*   stream;              /* hipStream_t to launch on; NULL = the library creates its own */
    int32_t topology;            /* SOM_TOPO_* */
    int32_t norm_p;              /* exponent p of SOM_DIST_NORM_P* (activation_distance_kwargs={'p': ...}); 0 = default 2 */
    double  norm_p_real;         /* a non-integer exponent p > 0 (distances.py:61-75 takes any real p); 0 = norm_p holds it */
} som_config;

const char* som_version(void);
int         som_device_count(void);                      /* replaces utils.find_max_cuda_threads' device probe, utils.py:4-13 */
const char* som_last_error(const som_handle* h);

int  som_create(const som_config* cfg, som_handle** out); /* XPySom.__init__ device-side state, xpysom.py:193-240 */
void som_destroy(som_handle* h);

/* host <-> device codebook: train() entry/exit copies, xpysom.py:485, :580-583 */
int som_set_weights(som_handle* h, const float* w_host);
int som_get_weights(som_handle* h, float* w_host);

/* resident training data: xp.asarray(data, float32), xpysom.py:510.  The rows
 * stay on the device for all epochs.  _device: rows already in HBM (borrowed,
 * must outlive the handle's use of them). */
int som_set_data(som_handle* h, const float* x_host, int64_t n_rows);
int som_set_data_device(som_handle* h, const void* x_dev, int64_t n_rows);
/* ... whose producer may still be running: wait for it first.  `stream` is the producer's stream as the
 * __cuda_array_interface__ protocol names it (1 = legacy default, 2 = per-thread default, otherwise a
 * hipStream_t); has_stream == 0 = unknown producer, wait for the whole device.  (The reference's CuPy arrays
 * share CuPy's current stream with the kernels that read them, xpysom.py:487-510; here the engine owns a stream.) */
int som_sync_producer(som_handle* h, uint64_t stream, int32_t has_stream);
/* device rows of the caller copied to host memory (quantization_error(data) after train(device_rows, verbose=True),
 * xpysom.py:589-592 on a CuPy array) */
int som_copy_to_host(som_handle* h, const void* x_dev, uint64_t bytes, void* dst_host);

/* One epoch over the resident rows = the body of the epoch loop, xpysom.py:515-577:
 *   som_epoch_accumulate: w_sq cache (:529-537), every _update (:560-569 -> :420-443:
 *       BMU, neighbourhood*eta, sum_g, g^T x) summed into the fused float32
 *       accumulator [K][D+1] (numerator | denominator), left on the device;
 *   (multi-GPU: the host all-reduces the accumulator here -- replaces the Dask
 *       gather/sum, xpysom.py:545-558)
 *   som_epoch_merge: _merge_updates, xpysom.py:446-455.
 *   som_epoch = accumulate + merge.
 * sigma, eta: this epoch's schedule values (decays.py).  neigh_f64 != 0 evaluates
 * the neighbourhood in float64 (what NumPy >= 2 does when the schedule returns
 * numpy.float64, i.e. 'exponential'); 0 mimics the float32 evaluation. */
int som_epoch_accumulate(som_handle* h, double sigma, double eta, int neigh_f64);
/* the same accumulator computed the way the reference states it, xpysom.py:434-441: g = h * eta per (sample, unit)
 * generated from the neighbourhood tables inside a K x N x D float32 MFMA GEMM (num = g^T x, den = sum_n g), 2*N*K*D
 * flop where som_epoch_accumulate's bucketed algebra needs 2*K*(X+Y)*(D+1).  For cross-checks and the record. */
int som_epoch_accumulate_faithful(som_handle* h, double sigma, double eta, int neigh_f64);
int som_epoch_merge(som_handle* h);
int som_epoch(som_handle* h, double sigma, double eta, int neigh_f64);

/* som_epoch_accumulate in stages, for a host that overlaps the all-reduce with the tail of the epoch (the fused
 * accumulator is K*(D+1) floats -- 823 MB at 512x512x784 -- and the reference's gather/sum of it, xpysom.py:555-558,
 * is the one exchange step of the path):
 *   som_epoch_accumulate_begin    BMUs, segment sums, neighbourhood tables, stage 1 of the separable transform
 *   som_epoch_block_count         number of map-row blocks of stage 2 (128 map rows each, in order)
 *   som_epoch_accumulate_block    stage 2 of block b; afterwards floats [offset, offset + n_floats) of the
 *                                 accumulator (som_accum_device_ptr) are final and may be all-reduced on another
 *                                 stream (ordered behind som_get_stream's) while the next block is computed.
 * All blocks, then som_epoch_merge, equal som_epoch_accumulate + som_epoch_merge bit for bit. */
int som_epoch_accumulate_begin(som_handle* h, double sigma, double eta, int neigh_f64);
int som_epoch_block_count(som_handle* h, int32_t* n_blocks);
int som_epoch_accumulate_block(som_handle* h, int32_t block, int64_t* offset, int64_t* n_floats);

/* The same epoch for rows that do NOT stay resident (more rows than HBM holds, or a producer that
 * hands them over chunk by chunk -- what the reference gets from Dask blocks, xpysom.py:545-556):
 *   som_stream_begin                  w_sq cache, zero the segment sums
 *   som_stream_rows(x_host, n) ...    one _update per chunk: BMU + segment sums, added up on the device
 *   som_stream_end(sigma, eta, f64)   separable transform -> fused accumulator (then all-reduce / merge as above)
 * The sums of all chunks equal som_epoch_accumulate over their concatenation (float32 add order aside). */
/* A chunk in PINNED host memory (som_pinned_alloc, hipHostMalloc, hipHostRegister) is copied on a second
 * stream into one of two device slots, so its transfer overlaps the previous chunk's kernels; the buffer of
 * call i may be reused once call i+1 has returned (or after som_stream_end + som_sync).  A pageable chunk is
 * copied synchronously and is free on return. */
int som_pinned_alloc(uint64_t bytes, void** out);
int som_pinned_free(void* p);
int som_stream_begin(som_handle* h);
int som_stream_rows(som_handle* h, const float* x_host, int64_t n_rows);
int som_stream_end(som_handle* h, double sigma, double eta, int neigh_f64);

/* The exchange step inside the library: one process per GPU, RCCL (bound at run time with dlopen -- the copy a host
 * such as torch already loaded, else librccl.so.1) all-reduces the fused accumulator over xGMI.  Replaces the Dask
 * gather -> sum -> broadcast of xpysom.py:545-558 for a caller that has nothing but this header:
 *   rank 0: som_comm_unique_id(id)  -> the host hands the 128 bytes to every rank (MPI, a file, a socket, ...)
 *   every rank: som_comm_init(h, world, rank, id)   collective; afterwards som_epoch() all-reduces between
 *                                   accumulate and merge (block by block under the transform when the map has
 *                                   more than one 128-row block), and som_epoch_allreduce() does the same for a host
 *                                   that drives accumulate / merge itself
 *   som_comm_destroy(h)             before som_destroy, on every rank
 * som_comm_load(path) picks the RCCL library explicitly (optional; NULL = the search described above). */
#define SOM_COMM_ID_BYTES 128
int som_comm_load(const char* librccl_path);
int som_comm_unique_id(void* id_out);
int som_comm_init(som_handle* h, int32_t world, int32_t rank, const void* id_bytes);
int som_epoch_allreduce(som_handle* h);
int som_comm_destroy(som_handle* h);

/* device address and length (floats) of the fused accumulator, for an in-place
 * all-reduce by the host (RCCL via torch.distributed). */
int som_accum_device_ptr(som_handle* h, void** dev_ptr, int64_t* n_floats);
/* the HIP stream every launch of this handle goes to (som_config.stream, or the handle's own): lets the caller
 * order foreign work -- e.g. the RCCL all-reduce of the accumulator -- on it instead of synchronising the host */
int som_get_stream(som_handle* h, void** stream_out);
/* teacher-forced parity: copy the last accumulate's results to the host.
 * Any pointer may be NULL.  num [K][D], den [K], bmu [n_rows] (raveled ids). */
int som_epoch_fetch(som_handle* h, float* num, float* den, int32_t* bmu);
/* teacher-forcing the other way: run the accumulate with these BMU ids instead
 * of computing them (isolates the update path in tests). */
int som_epoch_accumulate_forced(som_handle* h, const int32_t* bmu_host, double sigma, double eta, int neigh_f64);

/* BMU ids for arbitrary rows: XPySom.winner, xpysom.py:370-408 (mode ACTIVATION)
 * and XPySom._quantization, xpysom.py:632-645 (mode QUANTIZATION). */
int som_bmu(som_handle* h, const float* x_host, int64_t n_rows, int32_t mode, int32_t* ids_out);
/* ... for rows that already live in HBM (float32 [n_rows][input_len], borrowed for the call; order the engine behind their
 * producer with som_sync_producer first): no host round trip -- what the reference's winner() is on a CuPy array
 * (xpysom.py:379-396).  ids_out: host memory.  In EXACT precision large row sets run under a plan (the scout of
 * csrc/exact_skip.hpp gives every row a pseudo last BMU from the current codebook's own group centroids): the same ids,
 * a fraction of the distance GEMM. */
int som_bmu_device(som_handle* h, const void* x_dev, int64_t n_rows, int32_t mode, int32_t* ids_out);
/* best and second-best unit per row under the full Euclidean distance (sqrt + nan_to_num):
 * what XPySom.topographic_error takes from argsort(distances)[:, :2], xpysom.py:727-734 */
/* float64 query rows (XPySom.winner does not coerce its input, xpysom.py:379-396: float64 x against float32 weights is
 * computed in float64 by NumPy): the BMUs of fl64(-2 x.w + |w|^2_f32), euclidean activation distance only.  An analysis
 * call on the vector ALU; rows are staged through device memory as doubles. */
int som_bmu_f64(som_handle* h, const double* x_host, int64_t n_rows, int32_t* ids_out);
int som_bmu_top2(som_handle* h, const float* x_host, int64_t n_rows, int32_t* ids1_out, int32_t* ids2_out);
/* the (n_rows, K) distance matrix itself, row-major, for the analysis calls that return it:
 * mode ACTIVATION = XPySom.activate (xpysom.py:323-354, configured GEMM-form distance),
 * mode QUANTIZATION = XPySom.distance_from_weights (xpysom.py:647-671).  Never used while training. */
int som_distance_matrix(som_handle* h, const float* x_host, int64_t n_rows, int32_t mode, float* dist_out);
/* mean_n |x_n - W[bmu_n]|: XPySom.quantization_error, xpysom.py:673-707.  The distance to the chosen unit is
 * always evaluated exactly (float32 differences, float64 sum).  The BMU search is the reference's sqrt'd
 * Euclidean argmin in F32 precision; in BF16 / F16 precision with the 'euclidean' activation distance it
 * runs through the configured MFMA path (same argmin up to the operand rounding), as does som_bmu's
 * QUANTIZATION mode. */
int som_quantization_error(som_handle* h, const float* x_host, int64_t n_rows, double* qe_out);
/* ... of rows that already live in HBM (see som_bmu_device).  In EXACT precision with the 'euclidean' activation distance
 * the BMU search of both calls is the screen + float32 re-score (the float32 argmin of |w|^2 - 2 x.w: where the sqrt'd
 * distance ties two units this may name the other one -- at the same distance, which is all this call returns). */
int som_quantization_error_device(som_handle* h, const void* x_dev, int64_t n_rows, double* qe_out);

/* The canary.  n_rows > 0: after every BMU launch (epochs, streamed chunks, som_bmu) n_rows strided rows are scored
 * again by the float32 parity kernel and the launch's own picks must be the float32 picks (F32, EXACT) or within the
 * precision mode's operand-rounding bound of them; otherwise the call fails with a message naming a row.  Costs one
 * small float32 launch and one host synchronisation per BMU launch: for smoke tests and stress runs (also: the
 * environment variable SOM_VERIFY=n at som_create).  som_verify_stats: launches / rows checked so far.
 * (The test hooks that go with it -- som_debug_* -- are declared in include/somhip_test.h, not here.) */
int som_set_verify(som_handle* h, int32_t n_rows);
int som_verify_stats(som_handle* h, int64_t* launches, int64_t* rows_checked);

/* precision EXACT, introspection (host arithmetic only, no device needed): the order in which the mode's operand images
 * hold the units of an x * y map -- perm_out[position] = unit id, x * y entries; every 64 consecutive positions are one
 * group of the screen / re-score: where both sides are multiples of 8 an 8 x 8 patch of the map held as four 4 x 4 blocks
 * (positions 0..15: rows 0..3 x columns 0..3 of the patch, row by row; 16..31: columns 4..7; 32..63: rows 4..7 likewise --
 * the 16-unit blocks the exact mode's plan tests); elsewhere a run of whole 8-row bands, ascending inside. */
int som_patch_order(int32_t x, int32_t y, int32_t* perm_out);

/* precision EXACT bookkeeping: rows screened so far, rows that went to the float32 fallback kernel, screen passes */
int som_exact_stats(som_handle* h, int64_t* rows, int64_t* rows_fallback, int64_t* passes);
/* precision EXACT, block skipping (csrc/exact_skip.hpp: resident rows from their second epoch on, input_len <= 128): how
 * many (256-row tile, 16-unit block) blocks the screens ran, of how many a full scan has -- the EXECUTED share of the
 * distance GEMM (launches without skipping count every block) */
int som_exact_skip_stats(som_handle* h, int64_t* blocks_run, int64_t* blocks_total);
/* ... and the resident sorted pass behind it: epochs that ran under a plan, and how many of them (re-)sorted the rows by
 * their last BMU's patch first (the others reused the order of an earlier epoch) */
int som_exact_resident_stats(som_handle* h, int64_t* planned_epochs, int64_t* sorts);
/* ... and the scout (csrc/exact_skip.hpp: pseudo last BMUs from the current codebook's own group centroids, for rows that
 * have no last BMU -- query rows, streamed chunks, a row set's first epoch -- and for a schedule's first epochs): BMU
 * launches that ran it, and launches over transient row sets (queries, streamed chunks) that ran under a plan */
int som_exact_scout_stats(som_handle* h, int64_t* scouted_launches, int64_t* transient_planned);
/* ... and the refinement pass between the screen and the float32 re-score (csrc/bmu_exact.hpp: both operands' second
 * half-precision halves, a window some twenty times narrower than the screen's): candidate (row, group) pairs it was
 * given, and how many of them it left for the re-score */
int som_exact_refine_stats(som_handle* h, int64_t* pairs_in, int64_t* pairs_out);
/* candidate groups per row of the LAST screen pass (its first n rows): how many 64-unit groups the select kernel found for the
 * row.  Under a plan (a sorted pass) entry i belongs to SORTED POSITION i of the pass, not to row i, and the counts are taken
 * BEFORE the refinement pass compacts the lists (som_exact_refine_stats has the totals on both sides of it). */
int som_exact_last_counts(som_handle* h, int32_t* counts_out, int64_t n);

/* stream / timing plumbing */
int som_sync(som_handle* h);
int som_profile_enable(som_handle* h, int32_t on);          /* hipEvent pairs around each kernel family (1), or around the
                                                               BMU kernels only (2: two event records per epoch); 0 = off */
int som_profile_get(som_handle* h, int32_t kernel, double* total_ms, int64_t* launches);
int som_profile_reset(som_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* SOMHIP_H */
