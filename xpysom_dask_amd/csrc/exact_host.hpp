// precision 'exact': the HOST side of the screen + re-score pipeline (bmu_exact.hpp), of block skipping (exact_skip.hpp,
// exact_skip_wide.hpp) and of the scout -- buffers, launch geometry, the per-pass kernel sequence, the policy's bookkeeping.
// Included by somhip.hip INSIDE its anonymous namespace, behind the handle (som_handle, ExactScratch) and the helpers it uses
// (dev_alloc, kernel_per_cu, choose_parts, refresh_codebook_operands, launch_bmu_f32_any, Timed, HIPCHK ...): one translation
// unit, two files.  The decisions themselves -- commit a scouted plan, level 2, did a sort pay, is a plan idle -- are pure
// functions of measured numbers in exact_policy.hpp (unit-tested without a GPU through som_policy_eval, include/somhip_test.h).
// (no #pragma once / include guard on purpose: not a header of its own)

// ---- precision 'exact' (bmu_exact.hpp): screen -> candidate groups -> float32 re-score -> float32 fallback ----------
// E(n) = cA |x_n| wmax + cW wmax^2 + cB Bm in d' units; derivation in bmu_exact.hpp.  KAPPA ulps are charged per MFMA.
constexpr double EX_KAPPA = 6.0;     // measured through som_debug_mfma16: <= 2.4 (tests/test_gpu_exact.py holds it below 3)
ExactBound exact_bound(const som_handle* h) {
    const double u = std::ldexp(1.0, -24);
    // chain length of the float32 kernel (zero padded), MFMAs of the screen's one accumulator chain
    const double Dl = h->wide ? 32.0 * h->ft_kchunks : 8.0 * h->fr_kg;
    const double n_mfma = h->wide ? h->n_kchunks : h->ks32;
    const double gamma = Dl * u / (1.0 - Dl * u);
    const double slop = 1.01;                             // the kernel evaluates E in float32
    ExactBound eb{};
    // one pass on scaled half operands, measured operand errors (bmu_exact.hpp); +1: the initial accumulator's rounding
    eb.cB = (float)(slop * 2.0 * (EX_KAPPA * n_mfma + 1.0) * std::ldexp(1.0, -23));
    eb.cM = (float)(slop * 2.0);
    if (h->cfg.distance == SOM_DIST_COSINE) {
        // scores 1 - cos: the float32 kernel's chain (gamma_D), its two pairwise |.|^2 sums, product, sqrt, division and
        // subtraction (< 27 u together); the screen's two normalisations 1/sqrt(|.|^2) and their products (< 30 u)
        eb.cA = (float)(slop * 2.0 * (gamma + 57.0 * u));
        eb.cW = 0.0f;
        eb.unit = 1;
    } else {
        eb.cA = (float)(slop * (2.0 * gamma + 2.0 * u) * (1.0 + u));   // float32 kernel, relative to A (tau units)
        eb.cW = (float)(slop * u);
    }
    return eb;
}

// rows of one screen pass: the group-minimum matrix of a pass (and, as large again, the groups' row lists) stays within
// 4 GiB -- address space rather than traffic: both are written and read only where a row is near its minimum.  1 Mi rows
// of a 256 x 256 map, or configs[4]'s 250 000-row shard of a 512 x 512 one, are ONE pass (measured against passes of a
// quarter of that: -2.3 % / -3.5 % per epoch: fewer, larger launches and one counter read-back instead of four).
long exact_chunk_rows(const som_handle* h) {
    const long n_groups = cdiv(h->K, EX_GROUP);
    long rows = (4L << 30) / (4 * n_groups);
    if (h->ex.pass_rows_override > 0) rows = h->ex.pass_rows_override;   // SOM_EXACT_PASS_ROWS: tests walk several passes on small data
    rows = rows / 1024 * 1024;                           // (a multiple of every screen kernel's workgroup tile)
    return rows < 1024 ? 1024 : rows;
}

// (the screens write group minima and row masks for whole workgroup tiles: a pass's row stride must hold them)
static_assert(256 % K16_WG_SAMPLES == 0 && 256 % WD_WG_SAMPLES == 0 && K16_WG_SAMPLES % 64 == 0 && WD_WG_SAMPLES % 64 == 0,
              "exact: the pass stride (a multiple of 256 rows) must be a whole number of screen workgroup tiles");
int exact_reserve_stride(som_handle* h, long stride);

// the pass scratch for `rows` rows: as large a pass as the 4 GiB rule allows -- and, where the device cannot give that much
// (other handles, other processes on the card), passes of half the rows, and half again: smaller passes cost a few
// percent, a refused allocation costs the run
int exact_reserve(som_handle* h, long rows) {
    auto& ex = h->ex;
    long stride = round_up(std::min(rows, exact_chunk_rows(h)), 256);
    if (ex.stride_cap > 0) stride = std::min(stride, ex.stride_cap);
    if (stride <= ex.stride) return 0;
    for (;;) {
        const int rc = exact_reserve_stride(h, stride);
        if (rc == 0) return 0;
        if (stride <= 1024) return rc;                   // (the message of the last failed allocation stands)
        (void)hipGetLastError();
        stride = round_up(stride / 2, 256);
        ex.stride_cap = stride;                          // launch_bmu_exact walks passes of this many rows from now on
        if (h->debug) std::fprintf(stderr, "[somhip] exact: pass scratch refused, retrying with passes of %ld rows\n", stride);
    }
}

int exact_reserve_stride(som_handle* h, long stride) {
    auto& ex = h->ex;
    // TEST HOOK (tests/test_gpu_exact.py; SOM_TEST_HOOKS=1): behave as a device that refuses the scratch of passes above n rows
    if (ex.hook_refuse_above > 0 && stride > ex.hook_refuse_above) return fail(h, "exact: pass scratch refused (test hook)");
    void* old[] = {ex.gmin, ex.gflags, ex.rowcnt, ex.rowarg, ex.seed, ex.fb_list, ex.plist, ex.tile_tab};
    for (void* p : old) if (p) (void)hipFree(p);
    ex.gmin = nullptr; ex.gflags = nullptr; ex.rowcnt = nullptr; ex.rowarg = nullptr; ex.seed = nullptr; ex.fb_list = nullptr; ex.plist = nullptr; ex.tile_tab = nullptr;
    ex.stride = 0;
    ex.res_valid = false;                               // (the resident order was built pass by pass: new passes, new order)
    const long n_groups = cdiv(h->K, EX_GROUP);
    // capacity of a pass in (row, group) pairs per row on average: a quarter of the groups -- past that the float32
    // kernel over all of them costs about what the re-score would
    ex.pairs = std::max<long>(EX_PAIRS, std::min<long>(n_groups / 4, 512));
    if (ex.hook_pairs > 0) ex.pairs = ex.hook_pairs;
    if (stride * ex.pairs > 0x7fffffffL) return fail(h, "exact: pass too large");
    if (int rc = dev_alloc(h, &ex.gmin, (size_t)n_groups * stride)) return rc;
    if (int rc = dev_alloc(h, &ex.gflags, (size_t)n_groups * (stride / 64))) return rc;
    if (int rc = dev_alloc(h, &ex.rowcnt, (size_t)stride)) return rc;
    if (int rc = dev_alloc(h, &ex.rowarg, (size_t)stride)) return rc;
    if (int rc = dev_alloc(h, &ex.seed, (size_t)stride)) return rc;
    if (int rc = dev_alloc(h, &ex.plist, (size_t)n_groups * stride)) return rc;   // every group: room for the whole pass
    if (int rc = dev_alloc(h, &ex.fb_list, (size_t)stride)) return rc;
    ex.max_tiles = cdiv(stride * ex.pairs, EX_TR) + n_groups;
    if (int rc = dev_alloc(h, &ex.tile_tab, (size_t)ex.max_tiles)) return rc;
    if (!ex.ctr) {
        if (int rc = dev_alloc(h, &ex.ctr, (size_t)3 * n_groups + 16)) return rc;
        HIPCHK(h, hipHostMalloc((void**)&ex.fb_count_host, 8 * sizeof(int), hipHostMallocDefault));   // fb_count | n_tiles | overflow | 16-unit blocks run | groups run | pairs selected | pairs kept by the refinement
    }
    ex.stride = stride;
    return 0;
}

template <int KS32, class E>
int exact_screen(som_handle* h, const __bf16* Xb, long n, unsigned long long* best64, const float* xsq, const float* xerr,
                 const float* xmax2, const ExactBound& eb, const float* seed, const int* glist, const int* gcnt) {
    const int n_groups = (int)cdiv(h->K, EX_GROUP);
    // one pass on scaled half operands: a stage IS a group
    const bool tl = glist != nullptr;
    const void* kern = tl ? (const void*)bmu_bf16_k16_kernel<KS32, E, true, true> : (const void*)bmu_bf16_k16_kernel<KS32, E, true, false>;
    size_t lds = 2 * (size_t)k16_stage_bytes(KS32);
    int per_cu = 1;
    if (int rc = kernel_per_cu(h, kern, 64 * K16_NW, lds, &per_cu)) return rc;
    const long blocks = cdiv(n, K16_WG_SAMPLES);
    const long slots = (long)per_cu * (h->n_cus > 0 ? h->n_cus : 256);
    int parts = choose_parts(h, blocks, slots, h->n_stages);
    if (tl && blocks >= slots) {
        // a tile's list is short where the plan works (tens of items of 1 024): every part of a tile loads the tile's 64 KB
        // of rows again, so the scan is split only where the lists are long enough to carry that (the last plan's share
        // is the forecast; 1 Mi rows, mid-schedule: three parts re-read 0.8 GB for 0.3 ms of screen)
        const double tiles16 = h->ex.share_forecast * (double)n_groups * K16_T;
        parts = tiles16 < 128.0 ? 1 : tiles16 < 320.0 ? std::min(parts, 2) : parts;
    }
    if (h->env_bf16_parts > 0) parts = std::min(h->env_bf16_parts, h->n_stages);
    if (h->debug)
        std::fprintf(stderr, "[somhip] exact screen: blocks=%ld per_cu=%d slots=%ld parts=%d groups=%d lists=%d\n", blocks, per_cu,
                     slots, parts, n_groups, tl ? 1 : 0);
    const dim3 grid((unsigned)blocks, (unsigned)parts), block(64 * K16_NW);
    if (tl && h->ex.item_queue && glist == h->ex.tlist) {
        // (the plan's lists as a work queue: a workgroup per slot of the chip, items of about equal length -- bmu_bf16_k16.hpp;
        //  the next plan cuts its lists for this many workgroups)
        h->ex.screen_slots = (int)std::min<long>(slots, h->ex.item_slots);
        bmu_bf16_k16_kernel<KS32, E, true, true><<<dim3((unsigned)h->ex.screen_slots), block, lds, h->stream>>>(
            Xb, n, h->Wst, h->n_stages, h->K, best64, h->ex.gmin, h->ex.stride, h->ex.gflags, xsq, xerr, xmax2, h->wmax2, h->wmax2 + 1, eb,
            seed, glist, gcnt, h->ex.items + 8, (const int*)h->ex.items, (int*)h->ex.items + 1);
    } else if (tl)
        bmu_bf16_k16_kernel<KS32, E, true, true><<<grid, block, lds, h->stream>>>(
            Xb, n, h->Wst, h->n_stages, h->K, best64, h->ex.gmin, h->ex.stride, h->ex.gflags, xsq, xerr, xmax2, h->wmax2, h->wmax2 + 1, eb,
            seed, glist, gcnt);
    else
        bmu_bf16_k16_kernel<KS32, E, true, false><<<grid, block, lds, h->stream>>>(
            Xb, n, h->Wst, h->n_stages, h->K, best64, h->ex.gmin, h->ex.stride, h->ex.gflags, xsq, xerr, xmax2, h->wmax2, h->wmax2 + 1, eb,
            seed, nullptr, nullptr);
    return 0;
}

// beyond 128 features: the wide kernel's GM instance (groups = pairs of its 32-unit stages)
template <int KS32, class E>
int exact_screen_wide(som_handle* h, const __bf16* Ximg, long n, unsigned long long* best64, const float* xsq, const float* xerr,
                      const float* xmax2, const ExactBound& eb, const int* glist = nullptr, const int* gcnt = nullptr) {
    if (glist != nullptr) {
        // under a plan (exact_skip_wide.hpp): every workgroup walks its tile's list of groups; parts where the lists are long
        auto kern = bmu_bf16_wide_kernel<KS32, E, true, true>;
        const size_t lds = (size_t)WD_SLOTS * wd_stage_bytes(KS32);
        int per_cu = 1;
        if (int rc = kernel_per_cu(h, (const void*)kern, 64 * WD_NW, lds, &per_cu)) return rc;
        const long blocks = cdiv(n, WD_WG_SAMPLES);
        const long slots = (long)per_cu * (h->n_cus > 0 ? h->n_cus : 256);
        const int n_groups = (int)cdiv(h->n_stages, 2);
        const double groups_forecast = h->ex.share_forecast * (double)n_groups;
        int parts = blocks >= slots ? (groups_forecast < 64.0 ? 1 : groups_forecast < 256.0 ? 2 : 4)
                                    : (int)std::min<long>(cdiv(slots, blocks), 16);
        if (h->env_bf16_parts > 0) parts = h->env_bf16_parts;
        parts = std::max(1, std::min(parts, n_groups));
        if (h->debug)
            std::fprintf(stderr, "[somhip] exact screen (wide, lists): blocks=%ld per_cu=%d slots=%ld parts=%d groups=%d\n", blocks, per_cu, slots, parts, n_groups);
        if (h->ex.item_queue && glist == h->ex.glist) {
            // (the plan's lists as a work queue: a workgroup per slot of the chip -- bmu_bf16_wide.hpp; the next plan cuts for this many)
            h->ex.screen_slots = (int)std::min<long>(slots, h->ex.item_slots);
            bmu_bf16_wide_kernel<KS32, E, true, true><<<dim3((unsigned)h->ex.screen_slots), dim3(64 * WD_NW), lds, h->stream>>>(
                (const char*)Ximg, n, h->Wst, h->n_stages, best64, h->ex.gmin, h->ex.stride, (uint32_t*)h->ex.gflags, xsq, xerr, xmax2,
                h->wmax2, h->wmax2 + 1, eb, glist, gcnt, n_groups, nullptr, nullptr, nullptr, 0, h->ex.items + 8, (const int*)h->ex.items,
                (int*)h->ex.items + 1);
            return 0;
        }
        bmu_bf16_wide_kernel<KS32, E, true, true><<<dim3((unsigned)blocks, (unsigned)parts), dim3(64 * WD_NW), lds, h->stream>>>(
            (const char*)Ximg, n, h->Wst, h->n_stages, best64, h->ex.gmin, h->ex.stride, (uint32_t*)h->ex.gflags, xsq, xerr, xmax2,
            h->wmax2, h->wmax2 + 1, eb, glist, gcnt, n_groups);
        return 0;
    }
    auto kern = bmu_bf16_wide_kernel<KS32, E, true>;
    const size_t lds = (size_t)WD_SLOTS * wd_stage_bytes(KS32);
    int per_cu = 1;
    if (int rc = kernel_per_cu(h, (const void*)kern, 64 * WD_NW, lds, &per_cu)) return rc;
    const long blocks = cdiv(n, WD_WG_SAMPLES);
    const long slots = (long)per_cu * (h->n_cus > 0 ? h->n_cus : 256);
    const int n_groups = (int)cdiv(h->n_stages, 2);
    int parts = 1;
    if (blocks < slots) parts = (int)std::min<long>(cdiv(slots, blocks), 64);
    else {
        double best_eff = 0.0;
        for (int p = 1; p <= 8; ++p) {
            const long wgs = blocks * p;
            const double eff = (double)wgs / (double)(cdiv(wgs, slots) * slots);
            if (eff > best_eff + 0.01) { best_eff = eff; parts = p; }
        }
    }
    if (h->env_bf16_parts > 0) parts = h->env_bf16_parts;
    parts = std::max(1, std::min(parts, n_groups));
    if (h->debug)
        std::fprintf(stderr, "[somhip] exact screen (wide): blocks=%ld per_cu=%d slots=%ld parts=%d groups=%d\n", blocks, per_cu, slots,
                     parts, n_groups);
    bmu_bf16_wide_kernel<KS32, E, true><<<dim3((unsigned)blocks, (unsigned)parts), dim3(64 * WD_NW), lds, h->stream>>>(
        (const char*)Ximg, n, h->Wst, h->n_stages, best64, h->ex.gmin, h->ex.stride, (uint32_t*)h->ex.gflags, xsq, xerr, xmax2,
        h->wmax2, h->wmax2 + 1, eb);
    return 0;
}

template <class E>
int exact_screen_ks(som_handle* h, const __bf16* Xb, long n, unsigned long long* best64, const float* xsq, const float* xerr,
                    const float* xmax2, const ExactBound& eb, const float* seed, const int* glist = nullptr,
                    const int* gcnt = nullptr) {
    if (h->wide) {
        switch (h->n_kchunks) {
#define SOM_WIDE_CASE(k) case k: return exact_screen_wide<k, E>(h, Xb, n, best64, xsq, xerr, xmax2, eb, glist, gcnt);
        SOM_WIDE_CASE(5) SOM_WIDE_CASE(6) SOM_WIDE_CASE(7) SOM_WIDE_CASE(8) SOM_WIDE_CASE(9) SOM_WIDE_CASE(10)
        SOM_WIDE_CASE(11) SOM_WIDE_CASE(12) SOM_WIDE_CASE(13) SOM_WIDE_CASE(14) SOM_WIDE_CASE(15) SOM_WIDE_CASE(16)
        SOM_WIDE_CASE(17) SOM_WIDE_CASE(18) SOM_WIDE_CASE(19) SOM_WIDE_CASE(20) SOM_WIDE_CASE(21) SOM_WIDE_CASE(22)
        SOM_WIDE_CASE(23) SOM_WIDE_CASE(24) SOM_WIDE_CASE(25)
#undef SOM_WIDE_CASE
        }
        return fail(h, "exact: no wide screen instance for this input_len");
    }
    switch (h->ks32) {
    case 1: return exact_screen<1, E>(h, Xb, n, best64, xsq, xerr, xmax2, eb, seed, glist, gcnt);
    case 2: return exact_screen<2, E>(h, Xb, n, best64, xsq, xerr, xmax2, eb, seed, glist, gcnt);
    case 3: return exact_screen<3, E>(h, Xb, n, best64, xsq, xerr, xmax2, eb, seed, glist, gcnt);
    case 4: return exact_screen<4, E>(h, Xb, n, best64, xsq, xerr, xmax2, eb, seed, glist, gcnt);
    }
    return fail(h, "exact: the screen kernel supports input_len <= 128");
}


// The stable radix sort of update.hpp: (keys_in, row index) -> (keys_out, vals_out), `bits` key bits, 8 a pass.  scratch: two
// pairs of n ints + the 256 x blocks table (radix_scratch_ints).  keys_in is left intact.
inline size_t radix_scratch_ints(long n) { return 4 * (size_t)n + 256 * (size_t)cdiv(std::max<long>(n, 1), RS_BLOCK) + 256; }
int radix_sort_rows(som_handle* h, const int* keys_in, long n, int bits, int* keys_out, int* vals_out, int* scratch) {
    if (n <= 0) return 0;
    const int B = (int)cdiv(n, RS_BLOCK);
    int* ka = scratch; int* va = scratch + n; int* kb = scratch + 2 * n; int* vb = scratch + 3 * n;
    int* table = scratch + 4 * n;
    int* tot = table + 256L * B;
    const int passes = std::max(1, (int)cdiv(bits, 8));
    const int* kin = keys_in; const int* vin = nullptr;
    for (int p = 0; p < passes; ++p) {
        const bool last = p == passes - 1;
        int* ko = last ? keys_out : (p & 1) ? kb : ka;
        int* vo = last ? vals_out : (p & 1) ? vb : va;
        rs_hist_kernel<<<dim3((unsigned)B), dim3(256), 0, h->stream>>>(kin, n, 8 * p, B, table);
        rs_scan_kernel<<<dim3(64), dim3(256), 0, h->stream>>>(table, B, tot);
        rs_scatter_kernel<<<dim3((unsigned)B), dim3(256), 0, h->stream>>>(kin, vin, n, 8 * p, B, table, tot, ko, vo);
        kin = ko; vin = vo;
    }
    HIPCHK(h, hipGetLastError());
    return 0;
}

// ---- block skipping (exact_skip.hpp): buffers, the centroid images, the sorted pass, the scout, a pass's plan -----------------
// sr: the sorted copies to (re)size for `rows` positions; stride: rows of one pass (the plan's own buffers)
int exact_skip_reserve(som_handle* h, som_handle::ExactScratch::SortedRows& sr, long rows, long stride) {
    auto& ex = h->ex;
    // TEST HOOK (tests/test_gpu_exact.py): behave as a device without memory for the sorted pass
    if (ex.hook_refuse_skip) return fail(h, "exact: block-skipping scratch refused (test hook)");
    const long n_groups = cdiv(h->K, EX_GROUP);
    if (!ex.cen_ready) {
        // (a refusal part of the way leaves cen_ready unset: the whole block is tried again, nothing half allocated is used)
        for (auto& c : ex.cen) {
            void* cb[] = {c.Cc, c.rg, c.csq, c.cmax2, c.Cst, c.Cst_plain};
            for (void* q : cb) if (q) (void)hipFree(q);
            c = som_handle::ExactScratch::Centroids();
        }
        const int ncs = (int)cdiv(n_groups, K16_STAGE_UNITS);
        for (int lv = 0; lv < (h->wide ? 1 : 2); ++lv) {
            auto& c = ex.cen[lv];
            // level 2: sixteen slots per four groups, 4 * ncs stages (need2 is addressed [tile][4 * ncs]: exact_lists_kernel)
            c.n_slots = lv == 0 ? (int)n_groups : (int)cdiv(n_groups, 4) * 16;
            c.n_cstages = lv == 0 ? ncs : 4 * ncs;
            // (the centroid image's own stages: 64 centroids each up to 128 features, 32 on the wide kernel's tiling -- where
            //  n_cstages stays the number of 64-group WORDS of the need bitmaps)
            c.n_img_stages = h->wide ? (int)cdiv(n_groups, WD_STAGE_UNITS) : c.n_cstages;
            if (int rc = dev_alloc(h, &c.Cc, (size_t)c.n_slots * h->D)) return rc;
            if (int rc = dev_alloc(h, &c.rg, (size_t)c.n_slots)) return rc;
            if (int rc = dev_alloc(h, &c.csq, (size_t)c.n_slots)) return rc;
            if (int rc = dev_alloc(h, &c.cmax2, 2)) return rc;
            if (int rc = dev_alloc(h, &c.Cst, (size_t)c.n_img_stages * h->stage_bytes)) return rc;
            HIPCHK(h, hipMemsetAsync(c.Cst, 0, (size_t)c.n_img_stages * h->stage_bytes, h->stream));
            if (lv == 0) {
                if (int rc = dev_alloc(h, &c.Cst_plain, (size_t)c.n_img_stages * h->stage_bytes)) return rc;
                HIPCHK(h, hipMemsetAsync(c.Cst_plain, 0, (size_t)c.n_img_stages * h->stage_bytes, h->stream));
            }
        }
        ex.cen_ready = true;
    }
    const long need_rows = round_up(rows, SK_TILE);
    if (need_rows > sr.cap) {
        // (kernels of an earlier launch may still read the old copies: a transient set's buffers are reused launch after launch)
        HIPCHK(h, hipStreamSynchronize(h->stream));
        void* old[] = {sr.order, sr.Xb_s, sr.Xl_s, sr.Xf_s, sr.xsq_s, sr.xerr_s, sr.seed_s, sr.sU_s, sr.lastpos_s};
        for (void* p : old) if (p) (void)hipFree(p);
        sr = som_handle::ExactScratch::SortedRows();
        if (&sr == &ex.srt[0]) ex.res_valid = false;
        if (int rc = dev_alloc(h, &sr.order, (size_t)need_rows)) return rc;
        if (int rc = dev_alloc(h, &sr.Xb_s, (size_t)need_rows * h->dp)) return rc;
        // (sr.Xl_s, the rows' second half image: allocated by the first launch whose refinement pass engages -- launch_bmu_exact)
        if (int rc = dev_alloc(h, &sr.Xf_s, (size_t)need_rows * h->D)) return rc;
        if (int rc = dev_alloc(h, &sr.xsq_s, (size_t)need_rows)) return rc;
        if (int rc = dev_alloc(h, &sr.xerr_s, (size_t)need_rows)) return rc;
        if (int rc = dev_alloc(h, &sr.seed_s, (size_t)need_rows)) return rc;
        if (int rc = dev_alloc(h, &sr.sU_s, (size_t)need_rows)) return rc;
        if (int rc = dev_alloc(h, &sr.lastpos_s, (size_t)need_rows)) return rc;
        sr.cap = need_rows;
    }
    if (stride <= ex.sk_stride) return 0;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    void* old[] = {ex.sk_keys, ex.sk_keys2, ex.sk_vals, ex.sk_tmp, ex.need, ex.need2, ex.glist, ex.gcnt, ex.tile_counts, ex.tlist, ex.tcnt, ex.scout_g, ex.items};
    for (void* p : old) if (p) (void)hipFree(p);
    ex.sk_keys = ex.sk_keys2 = ex.sk_vals = nullptr; ex.sk_tmp = nullptr; ex.need = ex.need2 = nullptr; ex.glist = ex.gcnt = nullptr; ex.tile_counts = nullptr; ex.tlist = ex.tcnt = nullptr;
    ex.scout_g = nullptr; ex.items = nullptr;
    ex.sk_stride = 0;
    const long tiles = stride / SK_TILE;
    if (int rc = dev_alloc(h, &ex.sk_keys, (size_t)stride)) return rc;
    if (int rc = dev_alloc(h, &ex.sk_keys2, (size_t)stride)) return rc;
    if (int rc = dev_alloc(h, &ex.sk_vals, (size_t)stride)) return rc;
    if (int rc = dev_alloc(h, &ex.scout_g, (size_t)stride)) return rc;
    if (h->wide) {
        (void)hipFree(ex.tq); ex.tq = nullptr;
        if (int rc = dev_alloc(h, &ex.tq, (size_t)stride)) return rc;
    }
    if (int rc = dev_alloc(h, &ex.need, (size_t)tiles * ex.cen[0].n_cstages)) return rc;
    if (int rc = dev_alloc(h, &ex.need2, (size_t)tiles * ex.cen[1].n_cstages)) return rc;
    if (int rc = dev_alloc(h, &ex.glist, (size_t)tiles * n_groups)) return rc;
    if (int rc = dev_alloc(h, &ex.gcnt, (size_t)tiles)) return rc;
    if (int rc = dev_alloc(h, &ex.tile_counts, (size_t)tiles)) return rc;
    if (int rc = dev_alloc(h, &ex.tlist, (size_t)tiles * n_groups * K16_T)) return rc;
    if (int rc = dev_alloc(h, &ex.tcnt, (size_t)tiles)) return rc;
    // (the listed screen's work items: exact_list_totals_kernel -- at most 2 tiles + 3 slots of them; + the queue's two words)
    ex.item_slots = 8 * (h->n_cus > 0 ? h->n_cus : 256);    // (more workgroups than this never fit a chip: few features, small stages)
    if (int rc = dev_alloc(h, &ex.items, (size_t)(5 * tiles + 4 * ex.item_slots + 16))) return rc;
    int* tmp = nullptr;
    if (int rc = dev_alloc(h, &tmp, radix_scratch_ints(stride))) return rc;
    ex.sk_tmp = tmp; ex.sk_tmp_bytes = radix_scratch_ints(stride) * sizeof(int);
    ex.sk_stride = stride;
    return 0;
}

// centroids and radii of the groups and of their sub-blocks under the current codebook, the centroids' scaled half images
// and initial accumulators
template <class E>
int exact_skip_centroids(som_handle* h, const float* xmax2) {
    auto& ex = h->ex;
    const float* Wsrc = h->ex_patch ? h->Wp : h->W;
    const int n_groups = (int)cdiv(h->K, EX_GROUP);
    auto& c0 = ex.cen[0];
    auto& c1 = ex.cen[1];
    const CentroidLevel l1{c0.Cc, c0.rg, c0.csq, c0.cmax2, c0.n_slots}, l2{c1.Cc, c1.rg, c1.csq, c1.cmax2, c1.n_slots};
    // two launches for both levels: centroids + radii + |c|^2, then the stage images with their tails (the images take the
    // codebook's own power of two: a centroid is no longer than the longest unit)
    exact_centroids_kernel<<<dim3((unsigned)(cdiv(n_groups, 4) * 4)), dim3(512), 0, h->stream>>>(Wsrc, h->K, h->D, n_groups, l1, l2, h->wmax2);
    const int nst2 = ex.l2_live ? c1.n_cstages : 0;
    char* plain = ex.scout_live ? c0.Cst_plain : nullptr;
    const dim3 tgrid((unsigned)cdiv((long)(c0.n_cstages + nst2) * K16_T, 4)), block(256);
    switch (h->ks32) {
#define SOM_CIMG_CASE(k) case k: exact_centroid_image_kernel<k, E><<<tgrid, block, 0, h->stream>>>(l1, c0.Cst, c0.n_cstages, l2, c1.Cst, nst2, h->D, xmax2, h->wmax2, plain); break;
    SOM_CIMG_CASE(1) SOM_CIMG_CASE(2) SOM_CIMG_CASE(3) SOM_CIMG_CASE(4)
#undef SOM_CIMG_CASE
    default: return fail(h, "exact: block skipping supports input_len <= 128");
    }
    HIPCHK(h, hipGetLastError());
    return 0;
}

// (re-)sort one pass: the rows [r0, r0 + n) of the row set in the order of their last BMU's group (prev: last epoch's ids) or
// of their nearest group centroid (scout_g: the scout's): the order (position -> row) into sr at s0, the sorted keys into sk_keys2
int exact_skip_sortkeys(som_handle* h, som_handle::ExactScratch::SortedRows& sr, long s0, long n, const int* prev, const int* scout_g) {
    auto& ex = h->ex;
    const int n_groups = (int)cdiv(h->K, EX_GROUP);
    if (scout_g != nullptr)
        exact_groupkey_kernel<<<dim3((unsigned)cdiv(n, 256)), dim3(256), 0, h->stream>>>(scout_g, n, n_groups, ex.sk_keys, ex.sk_vals);
    else
        exact_sortkey_kernel<<<dim3((unsigned)cdiv(n, 256)), dim3(256), 0, h->stream>>>(prev, h->ex_inv, n, h->K, ex.sk_keys, ex.sk_vals);
    int bits = 1;
    while ((1L << bits) < n_groups) ++bits;
    return radix_sort_rows(h, ex.sk_keys, n, bits, ex.sk_keys2, sr.order + s0, (int*)ex.sk_tmp);
}
// ... and the operands gathered in that order (`order`: n positions -> rows of the pass) into sr at positions s0 ...
template <class E>
int exact_skip_gather(som_handle* h, som_handle::ExactScratch::SortedRows& sr, long s0, const int* order, const float* X, const __bf16* Xb, long n,
                      const float* xsq, const float* xerr, const float* xmax2) {
    const long np = round_up(n, SK_TILE);
    exact_gather_sorted_kernel<E><<<dim3((unsigned)cdiv(np, 16)), dim3(256), 0, h->stream>>>(
        order, n, np, h->dp, h->D, Xb, X, xsq, xerr, xmax2, sr.Xb_s + s0 * h->dp, sr.Xl_s != nullptr ? sr.Xl_s + s0 * h->dp : nullptr, sr.Xf_s + s0 * h->D,
        sr.xsq_s + s0, sr.xerr_s + s0);
    HIPCHK(h, hipGetLastError());
    return 0;
}

// the scout, step 1: every row's nearest group centroid (the plain resident kernel on the plain level-1 centroid image)
template <int KS32, class E>
int exact_scout_nearest_ks(som_handle* h, const __bf16* Xb, long n, unsigned long long* best64, int* g_out) {
    auto& ex = h->ex;
    const auto& c0 = ex.cen[0];
    const void* kern = (const void*)bmu_bf16_k16_kernel<KS32, E, false, false>;
    const size_t lds = 2 * (size_t)k16_stage_bytes(KS32);
    int per_cu = 1;
    if (int rc = kernel_per_cu(h, kern, 64 * K16_NW, lds, &per_cu)) return rc;
    const long blocks = cdiv(n, K16_WG_SAMPLES);
    const long slots = (long)per_cu * (h->n_cus > 0 ? h->n_cus : 256);
    const int parts = std::max(1, std::min(choose_parts(h, blocks, slots, c0.n_cstages), c0.n_cstages));
    bmu_bf16_k16_kernel<KS32, E, false, false><<<dim3((unsigned)blocks, (unsigned)parts), dim3(64 * K16_NW), lds, h->stream>>>(
        Xb, n, c0.Cst_plain, c0.n_cstages, c0.n_slots, best64);
    bmu_finalize_kernel<<<dim3((unsigned)cdiv(n, 256)), dim3(256), 0, h->stream>>>(best64, n, c0.n_slots, g_out);
    HIPCHK(h, hipMemsetAsync(best64, 0xFF, (size_t)n * sizeof(unsigned long long), h->stream));
    HIPCHK(h, hipGetLastError());
    return 0;
}
template <class E>
int exact_scout_nearest(som_handle* h, const __bf16* Xb, long n, unsigned long long* best64, int* g_out) {
    switch (h->ks32) {
    case 1: return exact_scout_nearest_ks<1, E>(h, Xb, n, best64, g_out);
    case 2: return exact_scout_nearest_ks<2, E>(h, Xb, n, best64, g_out);
    case 3: return exact_scout_nearest_ks<3, E>(h, Xb, n, best64, g_out);
    case 4: return exact_scout_nearest_ks<4, E>(h, Xb, n, best64, g_out);
    }
    return fail(h, "exact: the scout supports input_len <= 128");
}

// the scout, step 3: per tile of the SORTED pass the groups of its rows' keys (`keys`: the pass's sorted keys), the plain
// kernel over those groups' units with indices kept: the best of them -> lastpos (the pseudo last BMU)
template <int KS32, class E>
int exact_scout_pick_ks(som_handle* h, som_handle::ExactScratch::SortedRows& sr, long s0, long n, const int* keys, unsigned long long* best64) {
    auto& ex = h->ex;
    const int n_groups = (int)cdiv(h->K, EX_GROUP);
    const long tiles = round_up(n, SK_TILE) / SK_TILE;
    const size_t lds_l = (size_t)cdiv(n_groups, 64) * sizeof(unsigned long long);
    exact_scout_lists_kernel<<<dim3((unsigned)tiles), dim3(64), lds_l, h->stream>>>(keys, nullptr, n, n_groups, ex.tlist, ex.tcnt);
    const void* kern = (const void*)bmu_bf16_k16_kernel<KS32, E, false, true>;
    const size_t lds = 2 * (size_t)k16_stage_bytes(KS32);
    int per_cu = 1;
    if (int rc = kernel_per_cu(h, kern, 64 * K16_NW, lds, &per_cu)) return rc;
    bmu_bf16_k16_kernel<KS32, E, false, true><<<dim3((unsigned)tiles, 1), dim3(64 * K16_NW), lds, h->stream>>>(
        sr.Xb_s + s0 * h->dp, n, h->Wst, h->n_stages, h->K, best64, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, ExactBound(),
        nullptr, ex.tlist, ex.tcnt);
    exact_scout_pos_kernel<<<dim3((unsigned)cdiv(n, 256)), dim3(256), 0, h->stream>>>(best64, n, h->K, sr.lastpos_s + s0);
    HIPCHK(h, hipGetLastError());
    return 0;
}
template <class E>
int exact_scout_pick(som_handle* h, som_handle::ExactScratch::SortedRows& sr, long s0, long n, const int* keys, unsigned long long* best64) {
    switch (h->ks32) {
    case 1: return exact_scout_pick_ks<1, E>(h, sr, s0, n, keys, best64);
    case 2: return exact_scout_pick_ks<2, E>(h, sr, s0, n, keys, best64);
    case 3: return exact_scout_pick_ks<3, E>(h, sr, s0, n, keys, best64);
    case 4: return exact_scout_pick_ks<4, E>(h, sr, s0, n, keys, best64);
    }
    return fail(h, "exact: the scout supports input_len <= 128");
}

// one pass's plan on the sorted rows sr[s0, s0 + n): level 1 (+ the seeds, from lastpos_s), level 2, the tiles' item lists
template <class E>
int exact_skip_plan(som_handle* h, som_handle::ExactScratch::SortedRows& sr, long s0, long n, const float* xmax2, const ExactBound& eb,
                    const int* lastpos2, bool time_l2 = false) {
    auto& ex = h->ex;
    const int n_groups = (int)cdiv(h->K, EX_GROUP);
    const long np = round_up(n, SK_TILE);
    const long tiles = np / SK_TILE;
    // (the select kernel walks the tiles' lists too: the masks of the blocks the screen does not run are never read)
    const dim3 block(64 * K16_NW);
    const auto& c0 = ex.cen[0];
    const auto& c1 = ex.cen[1];
    // (few tiles: their centroid stages split over up to four workgroups each, so that the plan fills the chip)
    const long want = (1024 + tiles - 1) / tiles;
    const dim3 pgrid((unsigned)tiles, (unsigned)std::max<long>(1, std::min<long>({want, 4L, (long)c0.n_cstages})));
    // (two stage slots + the words the workgroup produces; level 2: + its list of active stages)
    const size_t lds1 = 2 * (size_t)h->stage_bytes + (size_t)c0.n_cstages * 8;
    const size_t lds2 = 2 * (size_t)h->stage_bytes + (size_t)c0.n_cstages * 64 * sizeof(int) + (size_t)c1.n_cstages * 8;   // (+ its list of kept groups, its words)
    const bool l2 = ex.l2_live;
    const int force = ex.skip_mode == 3 ? 1 : 0;
    const __bf16* Xs = sr.Xb_s + s0 * h->dp;
    // (level 1 stores every word; level 2 only those of the stages it walks)
    if (l2) HIPCHK(h, hipMemsetAsync(ex.need2, 0, (size_t)tiles * c1.n_cstages * sizeof(unsigned long long), h->stream));
#define SOM_PLAN_CASE(k) case k: { \
        { int pc; if (int rc = kernel_per_cu(h, (const void*)exact_plan_kernel<k, E, false>, 64 * K16_NW, lds1, &pc)) return rc; } \
        exact_plan_kernel<k, E, false><<<pgrid, block, lds1, h->stream>>>(Xs, n, c0.Cst, c0.n_cstages, c0.rg, c0.n_slots, \
            sr.xsq_s + s0, sr.xerr_s + s0, sr.sU_s + s0, xmax2, c0.cmax2, h->wmax2, h->wmax2 + 1, eb, ex.need, sr.lastpos_s + s0, \
            h->Wst, sr.seed_s + s0, nullptr, 0, force, lastpos2, lastpos2 != nullptr ? ex.ctr + 2 * n_groups + 7 : nullptr); \
        if (l2) { \
            if (time_l2) (void)hipEventRecord(ex.cost.ev[3], h->stream); \
            { int pc; if (int rc = kernel_per_cu(h, (const void*)exact_plan_kernel<k, E, true>, 64 * K16_NW, lds2, &pc)) return rc; } \
            exact_plan_kernel<k, E, true><<<dim3((unsigned)tiles, pgrid.y), block, lds2, h->stream>>>(Xs, n, c1.Cst, c1.n_cstages, c1.rg, c1.n_slots, \
                sr.xsq_s + s0, sr.xerr_s + s0, sr.sU_s + s0, xmax2, c1.cmax2, h->wmax2, h->wmax2 + 1, eb, ex.need2, nullptr, \
                nullptr, nullptr, ex.need, c0.n_cstages, force); \
            if (time_l2) (void)hipEventRecord(ex.cost.ev[4], h->stream); \
        } } break;
    switch (h->ks32) {
    SOM_PLAN_CASE(1) SOM_PLAN_CASE(2) SOM_PLAN_CASE(3) SOM_PLAN_CASE(4)
    default: return fail(h, "exact: block skipping supports input_len <= 128");
    }
#undef SOM_PLAN_CASE
    exact_lists_kernel<<<dim3((unsigned)tiles), dim3(64), 0, h->stream>>>(ex.need, c0.n_cstages, l2 ? ex.need2 : nullptr, n_groups,
                                                                         ex.glist, ex.gcnt, ex.tile_counts, ex.tlist, ex.tcnt);
    // (... and the screen's work queue: the lists cut into items of about equal length)
    int2* queue = ex.items + 8;                              // (items[0] = (n_items, counter): the queue's two words)
    exact_list_totals_kernel<<<dim3(1), dim3(1024), 0, h->stream>>>(ex.tile_counts, tiles, ex.ctr + 2 * n_groups + 3, ex.ctr + 2 * n_groups + 4,
                                                                  ex.item_queue ? (ex.screen_slots > 0 ? ex.screen_slots : ex.item_slots) : 0,
                                                                  ex.item_queue ? queue : nullptr,
                                                                  (int*)ex.items, (int*)ex.items + 1, ex.item_len_pct);
    HIPCHK(h, hipGetLastError());
    return 0;
}

// ---- block skipping beyond 128 features (exact_skip_wide.hpp) ---------------------------------------------------------------
// centroids, radii, |c|^2 of the groups under the current codebook; their 32-to-a-stage image with its tails and its measured
// rounding error
template <class E>
int exact_wide_centroids(som_handle* h, const float* xmax2) {
    auto& ex = h->ex;
    auto& c0 = ex.cen[0];
    const float* Wsrc = h->ex_patch ? h->Wp : h->W;
    const int n_groups = (int)cdiv(h->K, EX_GROUP);
    wide_centroids_kernel<<<dim3((unsigned)n_groups), dim3(256), 0, h->stream>>>(Wsrc, h->K, h->D, n_groups, c0.Cc, c0.rg, c0.csq, c0.cmax2, h->wmax2);
    const long total = (long)c0.n_img_stages * WD_T * h->n_kchunks * 64;
    prep_w_bf16_wide_kernel<E><<<dim3((unsigned)cdiv(total, 256)), dim3(256), 0, h->stream>>>(c0.Cc, n_groups, h->D, h->n_kchunks, c0.Cst, c0.n_img_stages,
                                                                                             nullptr, h->wmax2);
    exact_werr_kernel<E><<<dim3((unsigned)cdiv(n_groups, 4 * EX_WERR_UNITS)), dim3(256), 0, h->stream>>>(c0.Cc, n_groups, h->D, h->wmax2, c0.cmax2 + 1, nullptr);
    char* plain = ex.scout_live ? c0.Cst_plain : nullptr;
    if (plain) HIPCHK(h, hipMemcpyAsync(plain, c0.Cst, (size_t)c0.n_img_stages * h->stage_bytes, hipMemcpyDeviceToDevice, h->stream));
    wide_centroid_tail_kernel<<<dim3((unsigned)cdiv((long)c0.n_img_stages * WD_STAGE_UNITS, 256)), dim3(256), 0, h->stream>>>(
        c0.rg, c0.csq, n_groups, c0.Cst, c0.n_img_stages, h->stage_bytes, xmax2, h->wmax2, plain);
    HIPCHK(h, hipGetLastError());
    return 0;
}

// the sorted pass: float32 rows, norms, last BMUs gathered in the order, the tile image built from the sorted rows
template <class E>
int exact_wide_gather(som_handle* h, som_handle::ExactScratch::SortedRows& sr, long s0, const float* X, long n, const float* xsq, const float* xerr,
                      const int* prev, const float* xmax2) {
    const long np = round_up(n, SK_TILE);
    wide_gather_sorted_kernel<<<dim3((unsigned)cdiv(np, 4)), dim3(256), 0, h->stream>>>(sr.order + s0, n, np, h->D, X, xsq, xerr, prev,
                                                                                      sr.Xf_s + s0 * h->D, sr.xsq_s + s0, sr.xerr_s + s0, sr.lastpos_s + s0);
    const long n_blocks = np / h->tl_bm;
    const long total = n_blocks * h->n_kchunks * (h->tl_bm / 16) * TL_KS * 64;
    prep_tiles_bf16_kernel<E><<<dim3((unsigned)cdiv(total, 256)), dim3(256), 0, h->stream>>>(
        sr.Xf_s + s0 * h->D, n, h->D, h->n_kchunks, n_blocks, h->tl_bm, h->tl_xtile, 1.0f, nullptr, (char*)(sr.Xb_s + s0 * h->dp), xmax2);
    HIPCHK(h, hipGetLastError());
    return 0;
}

// the scout beyond 128 features, step 1: every row's nearest group centroid (the plain wide kernel on the plain centroid image)
template <int KS32, class E>
int exact_wide_scout_nearest_ks(som_handle* h, const __bf16* Ximg, long n, unsigned long long* best64, int* g_out) {
    auto& ex = h->ex;
    const auto& c0 = ex.cen[0];
    const size_t lds = (size_t)WD_SLOTS * wd_stage_bytes(KS32);
    int per_cu = 1;
    if (int rc = kernel_per_cu(h, (const void*)bmu_bf16_wide_kernel<KS32, E>, 64 * WD_NW, lds, &per_cu)) return rc;
    const long blocks = cdiv(n, WD_WG_SAMPLES);
    const long slots = (long)per_cu * (h->n_cus > 0 ? h->n_cus : 256);
    const int parts = (int)std::max<long>(1, std::min<long>({cdiv(2 * slots, blocks), 8L, (long)c0.n_img_stages}));
    bmu_bf16_wide_kernel<KS32, E><<<dim3((unsigned)blocks, (unsigned)parts), dim3(64 * WD_NW), lds, h->stream>>>(
        (const char*)Ximg, n, c0.Cst_plain, c0.n_img_stages, best64);
    bmu_finalize_kernel<<<dim3((unsigned)cdiv(n, 256)), dim3(256), 0, h->stream>>>(best64, n, c0.n_slots, g_out);
    HIPCHK(h, hipMemsetAsync(best64, 0xFF, (size_t)n * sizeof(unsigned long long), h->stream));
    HIPCHK(h, hipGetLastError());
    return 0;
}
// ... step 3: per tile of the sorted pass the groups of its rows' keys, the plain wide kernel over those groups' units with
// indices kept: the best of them, as a UNIT id, -> lastpos_s (what the wide plan's float32 seed evaluates)
template <int KS32, class E>
int exact_wide_scout_pick_ks(som_handle* h, som_handle::ExactScratch::SortedRows& sr, long s0, long n, const int* keys, unsigned long long* best64) {
    auto& ex = h->ex;
    const int n_groups = (int)cdiv(h->K, EX_GROUP);
    const long tiles = round_up(n, SK_TILE) / SK_TILE;
    const size_t lds_l = (size_t)cdiv(n_groups, 64) * sizeof(unsigned long long);
    exact_scout_lists_kernel<<<dim3((unsigned)tiles), dim3(64), lds_l, h->stream>>>(keys, nullptr, n, n_groups, ex.glist, ex.gcnt, 1);
    const size_t lds = (size_t)WD_SLOTS * wd_stage_bytes(KS32);
    int per_cu = 1;
    if (int rc = kernel_per_cu(h, (const void*)bmu_bf16_wide_kernel<KS32, E, false, true>, 64 * WD_NW, lds, &per_cu)) return rc;
    bmu_bf16_wide_kernel<KS32, E, false, true><<<dim3((unsigned)tiles, 1), dim3(64 * WD_NW), lds, h->stream>>>(
        (const char*)(sr.Xb_s + s0 * h->dp), n, h->Wst, h->n_stages, best64, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, ExactBound(),
        ex.glist, ex.gcnt, n_groups);
    exact_scout_pos_kernel<<<dim3((unsigned)cdiv(n, 256)), dim3(256), 0, h->stream>>>(best64, n, h->K, sr.lastpos_s + s0, h->ex_perm);
    HIPCHK(h, hipGetLastError());
    return 0;
}
#define SOM_WIDE_DISPATCH(fn, ...) \
    switch (h->n_kchunks) { \
    case 5: return fn<5, E>(__VA_ARGS__); case 6: return fn<6, E>(__VA_ARGS__); case 7: return fn<7, E>(__VA_ARGS__); case 8: return fn<8, E>(__VA_ARGS__); \
    case 9: return fn<9, E>(__VA_ARGS__); case 10: return fn<10, E>(__VA_ARGS__); case 11: return fn<11, E>(__VA_ARGS__); case 12: return fn<12, E>(__VA_ARGS__); \
    case 13: return fn<13, E>(__VA_ARGS__); case 14: return fn<14, E>(__VA_ARGS__); case 15: return fn<15, E>(__VA_ARGS__); case 16: return fn<16, E>(__VA_ARGS__); \
    case 17: return fn<17, E>(__VA_ARGS__); case 18: return fn<18, E>(__VA_ARGS__); case 19: return fn<19, E>(__VA_ARGS__); case 20: return fn<20, E>(__VA_ARGS__); \
    case 21: return fn<21, E>(__VA_ARGS__); case 22: return fn<22, E>(__VA_ARGS__); case 23: return fn<23, E>(__VA_ARGS__); case 24: return fn<24, E>(__VA_ARGS__); \
    case 25: return fn<25, E>(__VA_ARGS__); }
template <class E>
int exact_wide_scout_nearest(som_handle* h, const __bf16* Ximg, long n, unsigned long long* best64, int* g_out) {
    SOM_WIDE_DISPATCH(exact_wide_scout_nearest_ks, h, Ximg, n, best64, g_out)
    return fail(h, "exact: no wide scout instance for this input_len");
}
template <class E>
int exact_wide_scout_pick(som_handle* h, som_handle::ExactScratch::SortedRows& sr, long s0, long n, const int* keys, unsigned long long* best64) {
    SOM_WIDE_DISPATCH(exact_wide_scout_pick_ks, h, sr, s0, n, keys, best64)
    return fail(h, "exact: no wide scout instance for this input_len");
}

// one pass's plan on the sorted rows: the float32 score of every row's last BMU, the rows' thresholds, the wide kernel in its
// PLAN mode over the centroid image, the tiles' lists
template <int KS32, class E>
int exact_wide_plan_ks(som_handle* h, som_handle::ExactScratch::SortedRows& sr, long s0, long n, const float* xmax2, const ExactBound& eb) {
    auto& ex = h->ex;
    const auto& c0 = ex.cen[0];
    const int n_groups = (int)cdiv(h->K, EX_GROUP);
    const long np = round_up(n, SK_TILE);
    const long tiles = np / SK_TILE;
    // (sr.lastpos_s holds the sorted rows' last BMUs as UNIT ids here; the seed itself is not used beyond 128 features)
    exact_seed_kernel<<<dim3((unsigned)cdiv(n * 16, 256)), dim3(256), 0, h->stream>>>(
        sr.Xf_s + s0 * h->D, n, h->D, h->W, h->wsq, h->K, sr.lastpos_s + s0, sr.xsq_s + s0, sr.xerr_s + s0, h->wmax2, xmax2, h->wmax2 + 1, eb, ex.seed, ex.tq);
    wide_plan_rows_kernel<<<dim3((unsigned)cdiv(n, 256)), dim3(256), 0, h->stream>>>(n, sr.xsq_s + s0, sr.xerr_s + s0, ex.tq, xmax2, c0.cmax2, h->wmax2,
                                                                                  h->wmax2 + 1, eb, ex.skip_mode == 3 ? 1 : 0, sr.seed_s + s0, sr.sU_s + s0);
    HIPCHK(h, hipMemsetAsync(ex.need, 0, (size_t)tiles * c0.n_cstages * sizeof(unsigned long long), h->stream));
    auto kern = bmu_bf16_wide_kernel<KS32, E, false, false, true>;
    const size_t lds = (size_t)WD_SLOTS * wd_stage_bytes(KS32) + (size_t)c0.n_cstages * sizeof(unsigned long long);
    int per_cu = 1;
    if (int rc = kernel_per_cu(h, (const void*)kern, 64 * WD_NW, lds, &per_cu)) return rc;
    const long slots = (long)per_cu * (h->n_cus > 0 ? h->n_cus : 256);
    int parts = (int)std::max<long>(1, std::min<long>({cdiv(2 * slots, tiles), 8L, (long)c0.n_img_stages}));
    bmu_bf16_wide_kernel<KS32, E, false, false, true><<<dim3((unsigned)tiles, (unsigned)parts), dim3(64 * WD_NW), lds, h->stream>>>(
        (const char*)(sr.Xb_s + s0 * h->dp), n, c0.Cst, c0.n_img_stages, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, ExactBound(),
        nullptr, nullptr, 0, sr.seed_s + s0, sr.sU_s + s0, ex.need, c0.n_cstages);
    exact_lists_kernel<<<dim3((unsigned)tiles), dim3(64), 0, h->stream>>>(ex.need, c0.n_cstages, nullptr, n_groups, ex.glist, ex.gcnt, ex.tile_counts, ex.tlist, ex.tcnt);
    // (... and the listed screen's work queue: the lists -- counted in 16-unit blocks, four to a group -- cut into items)
    exact_list_totals_kernel<<<dim3(1), dim3(1024), 0, h->stream>>>(ex.tile_counts, tiles, ex.ctr + 2 * n_groups + 3, ex.ctr + 2 * n_groups + 4,
                                                                  ex.item_queue ? (ex.screen_slots > 0 ? ex.screen_slots : ex.item_slots) : 0,
                                                                  ex.item_queue ? ex.items + 8 : nullptr, (int*)ex.items, (int*)ex.items + 1,
                                                                  ex.item_len_pct);
    HIPCHK(h, hipGetLastError());
    return 0;
}
template <class E>
int exact_wide_plan(som_handle* h, som_handle::ExactScratch::SortedRows& sr, long s0, long n, const float* xmax2, const ExactBound& eb) {
    switch (h->n_kchunks) {
#define SOM_WIDE_CASE(k) case k: return exact_wide_plan_ks<k, E>(h, sr, s0, n, xmax2, eb);
    SOM_WIDE_CASE(5) SOM_WIDE_CASE(6) SOM_WIDE_CASE(7) SOM_WIDE_CASE(8) SOM_WIDE_CASE(9) SOM_WIDE_CASE(10)
    SOM_WIDE_CASE(11) SOM_WIDE_CASE(12) SOM_WIDE_CASE(13) SOM_WIDE_CASE(14) SOM_WIDE_CASE(15) SOM_WIDE_CASE(16)
    SOM_WIDE_CASE(17) SOM_WIDE_CASE(18) SOM_WIDE_CASE(19) SOM_WIDE_CASE(20) SOM_WIDE_CASE(21) SOM_WIDE_CASE(22)
    SOM_WIDE_CASE(23) SOM_WIDE_CASE(24) SOM_WIDE_CASE(25)
#undef SOM_WIDE_CASE
    }
    return fail(h, "exact: no wide plan instance for this input_len");
}

template <int KG>
int exact_rescore_kg(som_handle* h, const float* X, int n_groups, int deint) {
    auto& ex = h->ex;
    auto kern = exact_rescore_mfma_kernel<KG>;
    const size_t lds = (size_t)fr_stage_bytes(KG);
    int per_cu = 1;
    if (int rc = kernel_per_cu(h, (const void*)kern, 256, lds, &per_cu)) return rc;
    // (twice the resident slots: the runs of tiles are uneven -- partial tiles, idle waves -- and finer runs balance them)
    const long grid = std::min<long>(ex.max_tiles, (long)h->ex.grid_mult * per_cu * (h->n_cus > 0 ? h->n_cus : 256));
    kern<<<dim3((unsigned)grid), dim3(256), lds, h->stream>>>(X, h->D, h->Wfst, h->K, ex.tile_tab, ex.ctr + 2 * n_groups + 1, ex.plist,
                                                             h->best64, h->ex_perm, nullptr, h->ex_sub44 ? 1 : 0, deint);
    return 0;
}

// the refinement pass over a sorted pass's candidate pairs (bmu_exact.hpp): tiles -> refined minima -> lists compacted in place
template <int KS32, class E>
int exact_refine_ks(som_handle* h, som_handle::ExactScratch::SortedRows& sr, long r0, long n, const float* xmax2, const ExactBound& eb) {
    auto& ex = h->ex;
    const int n_groups = (int)cdiv(h->K, EX_GROUP);
    int* gcount = ex.ctr; int* fb_count = ex.ctr + 2 * n_groups;
    int* n_tiles = fb_count + 1; int* overflow = fb_count + 2;
    exact_tiles_kernel<<<dim3(1), dim3(1024), 0, h->stream>>>(gcount, n_groups, ex.stride, ex.stride * ex.pairs, ex.tile_tab, n_tiles,
                                                             overflow, nullptr, nullptr, fb_count + 5);
    uint32_t* rowmin2 = (uint32_t*)ex.rowarg;              // (round 1's scratch: unused in the one-round scheme)
    HIPCHK(h, hipMemsetAsync(rowmin2, 0xFF, (size_t)n * sizeof(uint32_t), h->stream));
    const size_t lds = (size_t)k16_stage_bytes(KS32) + (size_t)K16_T * KS32 * 1024;
    int per_cu = 1;
    if (int rc = kernel_per_cu(h, (const void*)exact_refine_kernel<KS32, E>, 256, lds, &per_cu)) return rc;
    const long grid = std::min<long>(ex.max_tiles, (long)h->ex.grid_mult * per_cu * (h->n_cus > 0 ? h->n_cus : 256));
    exact_refine_kernel<KS32, E><<<dim3((unsigned)grid), dim3(256), lds, h->stream>>>(
        sr.Xb_s + r0 * h->dp, sr.Xl_s + r0 * h->dp, h->Wst, h->Wst_lo, ex.tile_tab, n_tiles, ex.plist, ex.gmin, rowmin2);
    exact_thr2_kernel<<<dim3((unsigned)cdiv(n, 256)), dim3(256), 0, h->stream>>>(rowmin2, n, sr.xsq_s + r0, sr.xerr_s + r0, h->wmax2, xmax2,
                                                                               h->wmax2 + 1, eb);
    exact_select2_kernel<<<dim3((unsigned)n_groups), dim3(256), 0, h->stream>>>(ex.plist, ex.gmin, ex.stride, gcount, rowmin2, fb_count + 6);
    HIPCHK(h, hipGetLastError());
    return 0;
}
template <class E>
int exact_refine(som_handle* h, som_handle::ExactScratch::SortedRows& sr, long r0, long n, const float* xmax2, const ExactBound& eb) {
    switch (h->ks32) {
    case 1: return exact_refine_ks<1, E>(h, sr, r0, n, xmax2, eb);
    case 2: return exact_refine_ks<2, E>(h, sr, r0, n, xmax2, eb);
    case 3: return exact_refine_ks<3, E>(h, sr, r0, n, xmax2, eb);
    case 4: return exact_refine_ks<4, E>(h, sr, r0, n, xmax2, eb);
    }
    return fail(h, "exact: the refinement pass supports input_len <= 128");
}

// the lists' entries from gstart on -> tiles -> float32 scores merged into best64 (the pass's slice of the merge keys)
// (sorted_copy: X is a sorted pass's float32 copy -- up to 128 features and a multiple of 8 of them: de-interleaved rows,
//  exact_gather_sorted_kernel)
int exact_rescore_round(som_handle* h, const float* X, const float* xsq, unsigned long long* best64, const int* gstart,
                        int* gstart_out, bool sorted_copy = false) {
    auto& ex = h->ex;
    const int deint = (sorted_copy && !h->wide && (h->D & 7) == 0) ? 1 : 0;
    const int n_groups = (int)cdiv(h->K, EX_GROUP);
    int* gcount = ex.ctr; int* fb_count = ex.ctr + 2 * n_groups;
    int* n_tiles = fb_count + 1; int* overflow = fb_count + 2;
    // (the pairs the select kernel found go back with the pass's counters -- unless the refinement pass has counted them already)
    exact_tiles_kernel<<<dim3(1), dim3(1024), 0, h->stream>>>(gcount, n_groups, ex.stride, ex.stride * ex.pairs, ex.tile_tab, n_tiles,
                                                             overflow, gstart, gstart_out,
                                                             (gstart == nullptr && gstart_out == nullptr && !ex.refine_live) ? fb_count + 5 : nullptr);
    unsigned long long* saved = h->best64;
    h->best64 = best64;                                   // (exact_rescore_kg reads it from the handle)
    int rc = 0;
    if (h->wide) {
        // beyond 128 features: the float32 tile image, chunk by chunk
        if (!h->Wfimg) { h->best64 = saved; return fail(h, "exact: no float32 tile image"); }
        const bool cosine = h->cfg.distance == SOM_DIST_COSINE;
        const void* kern = cosine ? (const void*)exact_rescore_tiled_kernel<SCORE_COSINE>
                                  : (const void*)exact_rescore_tiled_kernel<SCORE_EUCLID_PART>;
        int per_cu = 1;
        if (int rc2 = kernel_per_cu(h, kern, 256, 0, &per_cu)) { h->best64 = saved; return rc2; }
        const long grid = std::min<long>(ex.max_tiles, (long)h->ex.grid_mult * per_cu * (h->n_cus > 0 ? h->n_cus : 256));
        if (cosine)
            exact_rescore_tiled_kernel<SCORE_COSINE><<<dim3((unsigned)grid), dim3(256), 0, h->stream>>>(
                X, h->D, xsq, h->Wfimg, h->ft_kchunks, h->K, ex.tile_tab, n_tiles, ex.plist, best64, h->ex_perm, nullptr, h->ex_sub44 ? 1 : 0);
        else
            exact_rescore_tiled_kernel<SCORE_EUCLID_PART><<<dim3((unsigned)grid), dim3(256), 0, h->stream>>>(
                X, h->D, xsq, h->Wfimg, h->ft_kchunks, h->K, ex.tile_tab, n_tiles, ex.plist, best64, h->ex_perm, nullptr, h->ex_sub44 ? 1 : 0);
        h->best64 = saved;
        return 0;
    }
    switch (h->fr_kg) {
    case 1: rc = exact_rescore_kg<1>(h, X, n_groups, deint); break;
    case 2: rc = exact_rescore_kg<2>(h, X, n_groups, deint); break;
    case 4: rc = exact_rescore_kg<4>(h, X, n_groups, deint); break;
    case 8: rc = exact_rescore_kg<8>(h, X, n_groups, deint); break;
    case 16: rc = exact_rescore_kg<16>(h, X, n_groups, deint); break;
    default: rc = fail(h, "exact: bad k-group count");
    }
    h->best64 = saved;
    return rc;
}

int build_tables(som_handle* h, double sigma, double eta, int neigh_f64, hipStream_t st);

// X, xsq, Xb, out: the row set's float32 rows, their |x|^2, their hi / lo operand image, the ids to write.
int launch_bmu_exact(som_handle* h, const float* X, long N, const float* xsq, const __bf16* Xb, const float* xmax2, int* out) {
    if (h->capturing) return fail(h, "precision 'exact' reads a counter back per pass: not capturable");
    if (!xsq) return fail(h, "exact: no row norms");
    const float* xerr = exact_err_of(h, xsq);
    if (!xerr) return fail(h, "exact: unknown row-norm buffer");
    auto& ex = h->ex;
    auto& cost = ex.cost;
    if (int rc = exact_reserve(h, N)) return rc;
    if (N > h->best64_cap) {
        (void)hipFree(h->best64);
        h->best64 = nullptr; h->best64_cap = 0;
        if (int rc = dev_alloc(h, &h->best64, (size_t)round_up(N, 1024))) return rc;
        h->best64_cap = round_up(N, 1024);
    }
    if (!cost.have) {
        for (auto& e : cost.ev) HIPCHK(h, hipEventCreate(&e));
        cost.have = true;
    }
    HIPCHK(h, hipEventRecord(cost.ev[0], h->stream));
    const long units = (long)h->n_stages * h->stage_units;
    prep_wsqh_kernel<<<dim3((unsigned)cdiv(std::max(units, N), 256)), dim3(256), 0, h->stream>>>(
        h->wn, h->K, h->wmax2, xmax2, h->Wst, h->n_stages, h->stage_bytes, h->stage_units, h->best64, N, 1);
    const ExactBound eb = exact_bound(h);
    const int n_groups = (int)cdiv(h->K, EX_GROUP);
    const long chunk = std::min(exact_chunk_rows(h), ex.stride);
    const bool two_round = ex.two_round >= 0 ? ex.two_round != 0 : h->wide;
    // block skipping (exact_skip.hpp), one round, <= 128 features, maps of >= 4096 units: a plan needs, per row, SOME unit whose
    // distance bounds the distance to the BMU.  RESIDENT rows from their second epoch on have last epoch's BMU; every other row
    // set (query rows, streamed chunks, a row set's first epoch) -- and resident rows while last epoch's BMUs say little, a
    // schedule's first epochs -- gets a pseudo last BMU from the SCOUT.  The scout pays where a full scan costs more than its
    // own fixed part (some thirty small launches) several times over: 2 N K D flop at the screen's rate against a quarter of a
    // millisecond, i.e. from some 40 000 rows of a 256 x 256 x 128 map on.
    const bool resident = out == h->bmu;
    const bool have_last = resident && h->bmu_valid;
    const bool can_skip = ex.skip_mode > 0 && ex.seed_on && !h->wide && !two_round && (ex.skip_mode > 1 ? n_groups >= 2 : h->K >= 4096);
    const bool scout_ok = can_skip && ex.scout_on && n_groups <= 262144 &&
                          (ex.skip_mode > 1 || policy::rows_worth_a_scout((double)N, (double)h->K, (double)h->D));
    // beyond 128 features (exact_skip_wide.hpp): euclidean, resident rows with last epoch's BMUs, whole 64-unit groups
    const bool wide_can = ex.skip_mode > 0 && ex.seed_on && h->wide && h->cfg.distance == SOM_DIST_EUCLIDEAN &&
                          h->K % EX_GROUP == 0 && (h->n_stages & 1) == 0 && (ex.skip_mode > 1 ? n_groups >= 2 : h->K >= 4096);
    // (the scout there: rows without last BMUs -- queries, streamed chunks, a first epoch -- from the same break-even on)
    const bool wide_scout_ok = wide_can && ex.scout_on && n_groups <= 262144 &&
                               (ex.skip_mode > 1 || policy::rows_worth_a_scout((double)N, (double)h->K, (double)h->D));
    const bool wide_skip = wide_can && (have_last || wide_scout_ok);
    ex.skip_live = (can_skip && (have_last || scout_ok)) || wide_skip;
    // default mode: two launches in a row whose plans kept (nearly) every block -- rows without structure -- are followed
    // by two launches without a plan (the plan costs 4-8 % of a full scan), and so on while the plans stay idle
    if (ex.skip_live && ex.skip_mode == 1) {
        int& cool = resident ? ex.skip_cooldown : ex.tr_cooldown;
        if (cool > 0) { --cool; ex.skip_live = false; }
    }
    auto& sr = ex.srt[resident ? 0 : 1];
    const int64_t run_before = ex.blocks_run, total_before = ex.blocks_total;
    if (ex.skip_live && exact_skip_reserve(h, sr, resident ? N : std::min(N, chunk), ex.stride) != 0) {
        // no memory for the sorted pass's buffers: every block runs, from now on (the ids are the same either way)
        (void)hipGetLastError();
        if (h->debug) std::fprintf(stderr, "[somhip] exact: block skipping off (%s)\n", h->err.c_str());
        h->err.clear();
        ex.skip_live = false; ex.skip_mode = 0;
    }
    // the resident sorted pass: (re-)sort when there is none for these rows, when asked to (SOM_EXACT_RESORT=n: every n-th
    // planned epoch), or when the order has gone stale: while a quarter of the blocks or more still run a sort costs a few
    // percent of the screen it sharpens (the early epochs of a schedule, where rows still travel across the map); later
    // every eighth planned epoch, and a sort that did not pay (the share it left is within 7 % of the stale order's, level 2 on
    // or off in both: the schedule, not the order, moves the share) doubles that interval, up to 64; one that paid resets it.
    // (Measured, tools/resid_probe.py + bound_probe.py: past a schedule's first epochs an order three epochs old runs the same
    // blocks as a fresh one; a trigger on the share's growth fired on the schedule's own late growth, where sorting buys nothing.)
    // A stale order costs speed, never correctness: the plan tests every row of a tile where it sits.
    // A transient row set is sorted by the scout every time (there is nothing to keep).
    bool resort = false, scout = false;
    if (ex.skip_live) {
        const bool fresh = !resident || !have_last || !ex.res_valid || ex.res_rows != (const void*)X || ex.res_n != N;
        if (fresh) resort = true;
        else if (ex.res_every > 0) resort = ex.res_since >= ex.res_every;
        else resort = ex.res_share_last >= 0.25 || ex.res_since >= ex.res_forced;
        // the scout: always where there is no last BMU; with one, in the epochs that sort anyway because much of the map still
        // runs -- there the bound from the current codebook's own centroids is the better one (tools/ucent_probe.py: 0.78 against
        // 0.95 of the blocks in a schedule's second epoch, 0.28 against 0.52 in its third), and the plan takes the better of the
        // two units row by row
        // ... and goes on, sorting the rows by its keys, while its picks still beat last epoch's BMUs by a tenth of the squared
        // distance or more on a quarter of the rows (counted in the plan's prologue) AND halving the screen would still pay for
        // it: (last share) x (measured screen time per block) / 2 against what a scouted launch spends beyond an unscouted one
        // outside its screen (measured; before that: a tenth of a full screen)
        const bool scout_on_wins = scout_ok && have_last && !fresh &&
                                   policy::scout_continues(cost, ex.scout_win_share, ex.res_share_last, (double)n_groups * K16_T / (double)SK_TILE);
        if (scout_on_wins) resort = true;
        scout = scout_ok && (!have_last || (resort && (fresh || ex.res_share_last >= 0.25 || scout_on_wins)));
        if (h->wide) scout = wide_scout_ok && !have_last;    // (beyond 128 features: only where there is no last BMU)
        // level 2 of the plan (the groups' 16-unit sub-blocks) where it pays.  Whether it does is MEASURED each time it runs
        // (both levels' shares come back with the pass's counters): it costs about a tenth of level 1's share of a full scan
        // (four centroids per kept group), it saves the blocks it drops -- on the smooth maps of a schedule's first epochs
        // and on the compact patches of its middle it drops next to nothing, late, when the patches have spread out, more
        // than half.  While it does not pay it is probed again every fourth planned epoch, or at once when level 1's share
        // has moved by half since the last probe.
        bool probe = ex.l1_share_probe < 0.0 || ex.l2_wait <= 0 || ex.l1_share_last > 1.5 * ex.l1_share_probe || ex.l1_share_last < ex.l1_share_probe / 1.5;
        // (a new row set starts like a new engine: level 2 is taken to pay until it has been measured on these rows)
        if (fresh && resident) { ex.l2_pays = true; ex.l1_share_probe = -1.0; }
        if (ex.l1_share_last > 0.9 && ex.l1_share_probe >= 0.0 && !ex.l2_pays) probe = false;   // (nothing for four times the centroids to find)
        ex.l2_live = !h->wide && ex.sub_blocks && (ex.l2_pays || probe || ex.skip_mode >= 2 || !resident) &&
                     2 * (size_t)h->stage_bytes + (size_t)cdiv(n_groups, K16_STAGE_UNITS) * (64 * sizeof(int) + 4 * 8) <= 150 * 1024;   // (its list of kept groups lives in LDS)
        ex.scout_live = scout;
        if (h->wide) { if (int rc = SOM_HALF(h, exact_wide_centroids, h, xmax2)) return rc; }
        else if (int rc = SOM_HALF(h, exact_skip_centroids, h, xmax2)) return rc;
    }
    ex.scout_live = scout;
    // (the forecast from sample tiles: where the scout plans and there is no good recent plan of the same kind to go by)
    bool estimate = ex.skip_live && scout && ex.skip_mode == 1 && (resident || ex.tr_share_last >= 0.5);
    if (estimate) {
        // the cheap question first (exact_scout_rowneed_kernel): 128 sampled rows against the group centroids.  A tile needs at
        // least what its rows need: where a row alone needs more than 0.9 of the groups -- a random codebook, rows without
        // structure -- the launch runs without the scout, the sort and the plan (one small launch and one host wait spent)
        const int n_samples = (int)std::min<long>(128, N);
        HIPCHK(h, hipMemsetAsync(ex.ctr, 0, 2 * sizeof(int), h->stream));
        exact_scout_rowneed_kernel<<<dim3((unsigned)n_samples), dim3(256), (size_t)h->D * sizeof(float), h->stream>>>(
            X, N, h->D, n_samples, ex.cen[0].Cc, ex.cen[0].rg, n_groups, ex.ctr);
        HIPCHK(h, hipMemcpyAsync(ex.fb_count_host, ex.ctr, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        const double f = (double)ex.fb_count_host[0] / std::max(1.0, (double)ex.fb_count_host[1] * n_groups);
        if (h->debug) std::fprintf(stderr, "[somhip] exact scout: a sampled row needs %.4f of the groups\n", f);
        // (... or about as much as when the sample tiles last said no, up to eight launches ago: the same answer without asking them)
        if (ex.scout_f_age < 8) ex.scout_f_age += 1; else ex.scout_f_declined = 0.0;
        ex.scout_f_now = f;
        if (f > 0.9 || (ex.scout_f_declined > 0.0 && f >= 0.9 * ex.scout_f_declined)) {
            // (nearly free: not counted as an idle plan, asked again at the next launch)
            ex.skip_live = false; scout = false; resort = false; ex.scout_live = false; estimate = false; ex.scout_declined += 1;
        }
    }
    if (h->wide) estimate = false;                       // (beyond 128 features: the row sample only, no sample tiles)
    if (estimate && exact_skip_reserve(h, ex.srt[1], 128 * SK_TILE, ex.stride) != 0) { (void)hipGetLastError(); h->err.clear(); estimate = false; }
    ex.share_forecast = resident ? ex.res_share_last : ex.tr_share_last;
    // the refinement pass (bmu_exact.hpp) where it pays: it costs about a third of the float32 re-score of the pairs it is
    // given (it is bound by the same gather of rows) and leaves one to one and a half pairs a row, at two small launches more:
    // worth it from three candidate pairs a row on (the last planned epoch's count) -- the smooth maps of a schedule's middle
    ex.refine_live = ex.skip_live && ex.refine_on && h->Wst_lo != nullptr && (ex.pairs_per_row_last >= 3.0 || ex.skip_mode >= 2);
    if (ex.refine_live && !sr.xl_filled) {
        // the rows' second half image (a quarter of the sorted copies' bytes) exists from the first launch that refines: the
        // order is rebuilt in that launch, so that the gather fills it
        if (sr.Xl_s == nullptr && dev_alloc(h, &sr.Xl_s, (size_t)sr.cap * h->dp) != 0) { (void)hipGetLastError(); h->err.clear(); sr.Xl_s = nullptr; ex.refine_live = false; }
        else resort = true;
    }
    int64_t groups_run = 0, pairs_in = 0, pairs_out = 0, scout_wins = 0;
    // which phases this launch times (the launch as a whole: always)
    const bool planned_at_start = ex.skip_live;
    bool time_phases = !ex.skip_live || cost.since >= 3 || cost.blk_ms == 0.0 || resort || scout;
    double t_total = 0.0, t_screen = 0.0, t_l2 = 0.0, t_sort = 0.0;
    bool l2_timed = false, sort_timed = false, screen_timed = false;
    const double blocks_per_row = (double)n_groups * K16_T / (double)SK_TILE;
    for (long r0 = 0; r0 < N; r0 += chunk) {
        const long n = std::min(chunk, N - r0);
        const long s0 = resident ? r0 : 0;                   // where the pass sits in the sorted copies
        if (r0 > 0) HIPCHK(h, hipEventRecord(cost.ev[0], h->stream));
        // (a pass behind one whose fallback rows went through the float32 kernel: its image back in patch order)
        if (h->wf_patch != h->ex_patch) if (int rc = refresh_codebook_operands(h, true, true)) return rc;
        HIPCHK(h, hipMemsetAsync(ex.ctr, 0, (size_t)(2 * n_groups + 8) * sizeof(int), h->stream));
        // (sorted pass: the screen, the select kernel and the merge keys work on positions of the sorted order)
        const float* p_xsq = xsq + r0; const float* p_xerr = xerr + r0; const float* p_seed = nullptr;
        const __bf16* p_Xb = Xb + r0 * h->dp;
        const float* p_X = X + r0 * h->D;                    // the rows the re-score reads, indexed like the lists' entries
        const int* p_order = nullptr;
        if (ex.skip_live) {
            unsigned long long* best = h->best64 + r0;
            if (scout && h->wide) { if (int rc = SOM_HALF(h, exact_wide_scout_nearest, h, Xb + r0 * h->dp, n, best, ex.scout_g)) return rc; }
            else if (scout)
                if (int rc = SOM_HALF(h, exact_scout_nearest, h, Xb + r0 * h->dp, n, best, ex.scout_g)) return rc;
            if (resort) {
                if (time_phases) { HIPCHK(h, hipEventRecord(cost.ev[5], h->stream)); }
                if (int rc = exact_skip_sortkeys(h, sr, s0, n, out + r0, scout ? ex.scout_g : nullptr)) return rc;
            }
            // Is there anything for the plan to skip?  The scout, the gather and the plan cost a fifth of a full scan: before the
            // pass is committed to them, every stride-th TILE of its sorted order -- up to 128 of the very tiles the plan would see
            // -- goes through gather, pick and plan as a small pass of its own and the executed share comes back (one host wait).
            // The launch runs every block, unsorted, without a plan where the forecast says that is cheaper -- share x (screen
            // time per block under a plan) + (what a scouted launch spends outside its screen) against the last launch without
            // a plan, all MEASURED (before the first measurements: a scouted plan's overhead taken as a fifth of a full screen;
            // with no full scan on record either: declined above 0.8 of the blocks) -- a random codebook, the smooth map of a
            // schedule's second epoch, rows without structure.  The sample also says what level 2 is worth before it runs on
            // the whole pass: it removes (1 - ratio) of a kept group's four blocks at its measured (else: a sixth of the
            // group's screen) cost per kept group.
            const long tiles_all = n / SK_TILE;
            if (estimate && r0 == 0 && tiles_all > 256) {
                auto& ss = ex.srt[1];
                const long st = tiles_all / 128, n_st = std::min<long>(128, tiles_all / st), ns = n_st * SK_TILE;
                int* s_order = ex.sk_keys;                    // (the sort's input keys and row ids: free since the sort)
                int* s_keys = ex.sk_vals;
                exact_sample_tiles_kernel<<<dim3((unsigned)n_st), dim3(SK_TILE), 0, h->stream>>>(sr.order + s0, ex.sk_keys2, st, s_order, s_keys);
                if (int rc = SOM_HALF(h, exact_skip_gather, h, ss, 0L, s_order, X + r0 * h->D, Xb + r0 * h->dp, ns, xsq + r0, xerr + r0, xmax2)) return rc;
                if (int rc = SOM_HALF(h, exact_scout_pick, h, ss, 0L, ns, s_keys, best)) return rc;
                const int* lp2 = nullptr;
                if (have_last) {
                    exact_lastpos_kernel<<<dim3((unsigned)cdiv(ns, 256)), dim3(256), 0, h->stream>>>(out + r0, s_order, h->ex_inv, ns, h->K, ex.scout_g);
                    lp2 = ex.scout_g;                         // (the nearest groups have gone into the sort keys: free)
                }
                if (int rc = SOM_HALF(h, exact_skip_plan, h, ss, 0L, ns, xmax2, eb, lp2)) return rc;
                HIPCHK(h, hipMemcpyAsync(ex.fb_count_host, ex.ctr + 2 * n_groups, 8 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
                HIPCHK(h, hipStreamSynchronize(h->stream));
                // (the sample's count of the scout's wins is not the pass's: cleared again)
                HIPCHK(h, hipMemsetAsync(ex.ctr + 2 * n_groups + 7, 0, sizeof(int), h->stream));
                const double est = (double)ex.fb_count_host[3] / (double)(n_st * n_groups * K16_T);
                const double est1 = (double)ex.fb_count_host[4] / (double)(n_st * n_groups);
                ex.scout_est_last = est;
                const double blk = policy::block_ms(cost, blocks_per_row), over = policy::scouted_overhead(cost);
                const bool decline = !policy::commit_scouted_plan(cost, est, blocks_per_row);
                if (ex.l2_live && est1 > 0.0 && ex.skip_mode == 1) ex.l2_live = policy::level2_from_sample(cost, est, est1, blocks_per_row);
                if (h->debug)
                    std::fprintf(stderr, "[somhip] exact scout: %ld sample tiles of %ld would run %.4f of their blocks (level 1: %.4f) -> %s, level 2 %d "
                                 "[per row: full %.3g us, block %.3g us, overhead %.3g us]\n", n_st, tiles_all, est, est1,
                                 decline ? "no plan" : "plan", ex.l2_live ? 1 : 0, 1e3 * cost.full_total, 1e3 * blk, 1e3 * over);
                if (decline) { ex.scout_f_declined = ex.scout_f_now; ex.scout_f_age = 0; } else ex.scout_f_declined = 0.0;
                if (decline) {
                    ex.skip_live = false; scout = false; resort = false; ex.scout_live = false; ex.refine_live = false; ex.scout_declined += 1;
                    if (resident) ex.res_valid = false;       // (the order was rebuilt, the sorted copies were not)
                    // (a declined plan counts as an idle one: rows without structure are asked less and less often)
                    int& idle = resident ? ex.skip_idle : ex.tr_idle;
                    int& cool = resident ? ex.skip_cooldown : ex.tr_cooldown;
                    int& pause = resident ? ex.skip_pause : ex.tr_pause;
                    if (++idle >= 2) { cool = pause; pause = std::min(2 * pause, 16); }
                }
            }
        }
        // resident rows from their second epoch on: last epoch's BMU of every row caps the screen's keep threshold -- under a
        // plan the plan's prologue forms that seed from the operands it holds (exact_skip.hpp), else exact_seed_kernel
        ex.seed_live = ex.seed_on && !h->wide && have_last;
        if (ex.seed_live && !ex.skip_live) {
            exact_seed_kernel<<<dim3((unsigned)cdiv(n * 16, 256)), dim3(256), 0, h->stream>>>(
                X + r0 * h->D, n, h->D, h->W, h->wsq, h->K, out + r0, xsq + r0, xerr + r0, h->wmax2, xmax2, h->wmax2 + 1, eb, ex.seed);
            p_seed = ex.seed;
        }
        if (ex.skip_live && h->wide) {
            // beyond 128 features: the sorted float32 rows + the tile image built from them, the plan as a mode of the wide kernel
            if (resort) {
                // (no last BMUs: `out` holds nothing yet -- the gather's copy of it is overwritten by the scout's picks below)
                if (int rc = SOM_HALF(h, exact_wide_gather, h, sr, s0, X + r0 * h->D, n, xsq + r0, xerr + r0, have_last ? out + r0 : ex.scout_g, xmax2)) return rc;
                if (time_phases) { HIPCHK(h, hipEventRecord(cost.ev[6], h->stream)); sort_timed = true; }
                if (scout)
                    if (int rc = SOM_HALF(h, exact_wide_scout_pick, h, sr, s0, n, ex.sk_keys2, h->best64 + r0)) return rc;
            } else {
                wide_prev_sorted_kernel<<<dim3((unsigned)cdiv(n, 256)), dim3(256), 0, h->stream>>>(sr.order + s0, n, out + r0, sr.lastpos_s + s0);
            }
            if (int rc = SOM_HALF(h, exact_wide_plan, h, sr, s0, n, xmax2, eb)) return rc;
            p_xsq = sr.xsq_s + s0; p_xerr = sr.xerr_s + s0; p_seed = nullptr; p_Xb = sr.Xb_s + s0 * h->dp; p_order = sr.order + s0;
            p_X = sr.Xf_s + s0 * h->D;
        } else if (ex.skip_live) {
            unsigned long long* best = h->best64 + r0;
            if (resort) {
                if (int rc = SOM_HALF(h, exact_skip_gather, h, sr, s0, sr.order + s0, X + r0 * h->D, Xb + r0 * h->dp, n, xsq + r0, xerr + r0, xmax2)) return rc;
                if (time_phases) { HIPCHK(h, hipEventRecord(cost.ev[6], h->stream)); sort_timed = true; }
                if (sr.Xl_s != nullptr && r0 + n >= N) sr.xl_filled = true;
            }
            const int* lastpos2 = nullptr;
            if (scout) {
                if (int rc = SOM_HALF(h, exact_scout_pick, h, sr, s0, n, ex.sk_keys2, best)) return rc;
                if (have_last) {
                    // (the rows' real last BMUs beside the scout's picks: the plan's prologue keeps the better unit; sk_vals: free since the sort)
                    exact_lastpos_kernel<<<dim3((unsigned)cdiv(n, 256)), dim3(256), 0, h->stream>>>(out + r0, sr.order + s0, h->ex_inv, n, h->K, ex.sk_vals);
                    lastpos2 = ex.sk_vals;
                }
            } else {
                exact_lastpos_kernel<<<dim3((unsigned)cdiv(n, 256)), dim3(256), 0, h->stream>>>(out + r0, sr.order + s0, h->ex_inv, n, h->K, sr.lastpos_s + s0);
            }
            if (int rc = SOM_HALF(h, exact_skip_plan, h, sr, s0, n, xmax2, eb, lastpos2, time_phases && ex.l2_live)) return rc;
            if (time_phases && ex.l2_live) l2_timed = true;
            p_xsq = sr.xsq_s + s0; p_xerr = sr.xerr_s + s0; p_seed = sr.seed_s + s0; p_Xb = sr.Xb_s + s0 * h->dp; p_order = sr.order + s0;
            p_X = sr.Xf_s + s0 * h->D;
        }
        {
            Timed ts(h, SOM_K_SCREEN);
            if (time_phases) { HIPCHK(h, hipEventRecord(cost.ev[1], h->stream)); }
            // (the lists the screen walks: dense 16-unit tiles up to 128 features, whole groups beyond)
            if (int rc = SOM_HALF(h, exact_screen_ks, h, p_Xb, n, h->best64 + r0, p_xsq, p_xerr, xmax2, eb, p_seed,
                                  ex.skip_live ? (h->wide ? ex.glist : ex.tlist) : nullptr,
                                  ex.skip_live ? (h->wide ? ex.gcnt : ex.tcnt) : nullptr)) return rc;
            if (time_phases) { HIPCHK(h, hipEventRecord(cost.ev[2], h->stream)); screen_timed = true; }
        }
        const dim3 sel_grid((unsigned)cdiv(n, 64)), sel_block(64 * EX_SCAN_SPLIT);
        unsigned long long* best = h->best64 + r0;
        // two rounds beyond 128 features, where a (row, group) pair costs 64 x D flop AND a gather of the row's D floats
        // (configs[4]: 92.8 -> 87.8 ms per epoch); one round up to 128 features, where the three launches more cost more than
        // the pairs they save (256 x 256 x 128, 1 Mi rows: 15.5 vs 15.8 ms; 65 536 rows: +3 % in every map state)
        if (two_round) {
            // round 1: every row against the group that holds its screen minimum; round 2: the groups within the ONE-unit
            // bound of that float32 score (exact_select_kernel<true>: 20-32 % fewer pairs than the one-round scheme on
            // smooth maps, up to one pair per row more on random ones)
            // (under a plan -- exact_skip_wide.hpp -- rows are sorted positions: p_X, p_xsq, p_xerr; the select kernel walks the lists)
            exact_first_kernel<<<dim3((unsigned)cdiv(n, 256)), dim3(256), 0, h->stream>>>(
                best, n, n_groups, ex.stride, p_xsq, h->wmax2, xmax2, eb, p_xerr, h->wmax2 + 1, ex.plist, ex.ctr, ex.rowarg);
            if (int rc = exact_rescore_round(h, p_X, p_xsq, best, nullptr, ex.ctr + n_groups, ex.skip_live)) return rc;
            exact_select_kernel<true><<<sel_grid, sel_block, 0, h->stream>>>(
                ex.gmin, ex.gflags, ex.stride, n_groups, n, best, p_xsq, h->wmax2, xmax2, eb, p_xerr, h->wmax2 + 1, ex.plist,
                ex.ctr, ex.rowcnt, ex.rowarg, nullptr, ex.skip_live ? ex.glist : nullptr, ex.skip_live ? ex.gcnt : nullptr, SK_TILE);
            if (int rc = exact_rescore_round(h, p_X, p_xsq, best, ex.ctr + n_groups, nullptr, ex.skip_live)) return rc;
        } else {
            exact_select_kernel<false><<<sel_grid, sel_block, 0, h->stream>>>(
                ex.gmin, ex.gflags, ex.stride, n_groups, n, best, p_xsq, h->wmax2, xmax2, eb, p_xerr, h->wmax2 + 1, ex.plist,
                ex.ctr, ex.rowcnt, nullptr, p_seed, ex.skip_live ? ex.glist : nullptr, ex.skip_live ? ex.gcnt : nullptr, SK_TILE);
            if (ex.refine_live)
                if (int rc = SOM_HALF(h, exact_refine, h, sr, s0, n, xmax2, eb)) return rc;
            if (int rc = exact_rescore_round(h, p_X, xsq + r0, best, nullptr, nullptr, ex.skip_live)) return rc;
        }
        exact_finalize_kernel<<<dim3((unsigned)cdiv(n, 256)), dim3(256), 0, h->stream>>>(
            best, n, h->K, ex.ctr + 2 * n_groups + 2, out + r0, ex.fb_list, ex.ctr + 2 * n_groups, p_order);
        HIPCHK(h, hipGetLastError());
        // rows the scheme could not settle (normally none): the float32 kernel itself
        HIPCHK(h, hipMemcpyAsync(ex.fb_count_host, ex.ctr + 2 * n_groups, 8 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipEventRecord(cost.ev[7], h->stream));
        if (h->early.armed && !h->early.done && out == h->bmu && r0 + n >= N) {
            // the last pass of a resident epoch: the host waits for the counter only (an event behind the copy); what the
            // update needs besides the BMUs is queued behind it and runs while the host wakes up
            if (!ex.fb_ready) HIPCHK(h, hipEventCreateWithFlags(&ex.fb_ready, hipEventDisableTiming));
            HIPCHK(h, hipEventRecord(ex.fb_ready, h->stream));
            HIPCHK(h, hipMemsetAsync(h->SC, 0, (size_t)h->K * (h->D1p + 1) * sizeof(float), h->stream));
            if (int rc = build_tables(h, h->early.sigma, h->early.eta, h->early.neigh_f64, h->stream)) return rc;
            h->early.done = true;
            HIPCHK(h, hipEventSynchronize(ex.fb_ready));
        } else {
            HIPCHK(h, hipStreamSynchronize(h->stream));
        }
        {
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, cost.ev[0], cost.ev[7]) == hipSuccess) t_total += ms;
            if (screen_timed && hipEventElapsedTime(&ms, cost.ev[1], cost.ev[2]) == hipSuccess) t_screen += ms;
            if (l2_timed && hipEventElapsedTime(&ms, cost.ev[3], cost.ev[4]) == hipSuccess) t_l2 += ms;
            if (sort_timed && hipEventElapsedTime(&ms, cost.ev[5], cost.ev[6]) == hipSuccess) t_sort += ms;
            (void)hipGetLastError();
        }
        const int n_fb = ex.fb_count_host[0];
        ex.rows_total += n; ex.rows_fallback += n_fb; ex.chunks += 1;
        // (counted in 16-unit blocks: four per (256-row tile, group))
        ex.blocks_total += cdiv(n, SK_TILE) * n_groups * K16_T;
        ex.blocks_run += ex.skip_live ? ex.fb_count_host[3] : cdiv(n, SK_TILE) * n_groups * K16_T;
        groups_run += ex.skip_live ? ex.fb_count_host[4] : cdiv(n, SK_TILE) * n_groups;
        pairs_in += ex.fb_count_host[5];
        if (ex.refine_live) pairs_out += ex.fb_count_host[6];
        if (ex.skip_live && scout && have_last) scout_wins += ex.fb_count_host[7];
        if (n_fb < 0 || n_fb > n) return fail(h, "exact: fallback counter out of range");
        if (n_fb > 0) {
            if (n_fb > ex.fb_cap) {
                (void)hipFree(ex.fbX); (void)hipFree(ex.fb_ids);
                ex.fbX = nullptr; ex.fb_ids = nullptr; ex.fb_cap = 0;
                const long cap = round_up(n_fb, 1024);
                if (int rc = dev_alloc(h, &ex.fbX, (size_t)cap * h->D)) return rc;
                if (int rc = dev_alloc(h, &ex.fb_ids, (size_t)cap)) return rc;
                ex.fb_cap = cap;
            }
            // the float32 kernel names units by their place in its image: the units' own order for it
            if (h->wf_patch) if (int rc = refresh_codebook_operands(h, true, false)) return rc;
            exact_gather_rows_kernel<<<dim3((unsigned)cdiv((long)n_fb * h->D, 256)), dim3(256), 0, h->stream>>>(
                X + r0 * h->D, ex.fb_list, n_fb, h->D, ex.fbX);
            // (its part merge may reuse best64[0 .. n_fb): rows this pass has already settled)
            if (h->cfg.distance == SOM_DIST_COSINE) {
                // (|x|^2 of the gathered rows in NumPy's order, into the head of the pass's spent minima)
                float* fsq = (float*)ex.gmin;
                row_sq_f32_kernel<<<dim3((unsigned)cdiv(n_fb, 256)), dim3(256), 0, h->stream>>>(ex.fbX, n_fb, h->D, fsq);
                if (int rc = launch_bmu_f32_any<SCORE_COSINE>(h, ex.fbX, n_fb, fsq, ex.fb_ids)) return rc;
            } else if (int rc = launch_bmu_f32_any<SCORE_EUCLID_PART>(h, ex.fbX, n_fb, nullptr, ex.fb_ids)) return rc;
            exact_scatter_ids_kernel<<<dim3((unsigned)cdiv(n_fb, 256)), dim3(256), 0, h->stream>>>(ex.fb_ids, ex.fb_list, n_fb,
                                                                                                out + r0);
            HIPCHK(h, hipGetLastError());
        }
    }
    // what this launch cost, per row
    if (N > 0 && t_total > 0.0) {
        if (!ex.skip_live) {
            cost.full_total = t_total / (double)N;
            if (screen_timed && t_screen > 0.0) cost.full_screen = t_screen / (double)N;
        } else {
            cost.plan_total = t_total / (double)N;
            if (screen_timed && t_screen > 0.0) {
                (scout ? cost.plan_over_scout : cost.plan_over) = (t_total - t_screen) / (double)N;
                const double run = (double)(ex.blocks_run - run_before);
                if (run > 0.0) cost.blk_ms = t_screen / run;
                cost.since = 0;
            } else {
                cost.since += 1;
            }
            if (sort_timed && t_sort > 0.0) cost.sort_ms = t_sort / (double)N;
        }
    }
    (void)planned_at_start;
    if (ex.skip_live && ex.blocks_total > total_before) {
        const double share = (double)(ex.blocks_run - run_before) / (double)(ex.blocks_total - total_before);
        const double l1_share = (double)groups_run * K16_T / (double)(ex.blocks_total - total_before);
        if (scout) ex.scouted += 1;
        if (resident) ex.scout_win_share = (scout && have_last) ? (double)scout_wins / (double)std::max<long>(N, 1) : 0.0;
        ex.pairs_per_row_last = (double)pairs_in / (double)std::max<long>(N, 1);
        if (ex.refine_live) { ex.pairs_refined_in += pairs_in; ex.pairs_refined_out += pairs_out; }
        if (ex.l2_live && l2_timed && t_l2 > 0.0 && groups_run > 0) {
            cost.l2_ms_group = t_l2 / (double)groups_run;
            cost.l2_ratio = l1_share > 0.0 ? share / l1_share : 1.0;
        }
        // an IDLE plan: the launch cost what the last launch without a plan cost (per row; with no such launch on record: it
        // kept more than 0.97 of the blocks).  Rows without structure: two idle plans in a row pause the plan for two launches,
        // the next idle one for four, then eight, sixteen; a plan that pays again resets the pause.
        // (... and ran more than half of the blocks: with most of them proven empty a slow launch is somebody else's kernels on
        //  the card, not an idle plan)
        const bool idle_plan = policy::plan_idle(cost, share);
        if (!resident) {
            ex.tr_planned += 1;
            ex.tr_share_last = share;
            if (h->debug)
                std::fprintf(stderr, "[somhip] exact plan (transient, %ld rows): share %.4f level-1 %.4f level-2 %d refine %d pairs/row %.2f -> %.2f; %.3f ms (screen %.3f)\n",
                             N, share, l1_share, ex.l2_live ? 1 : 0, ex.refine_live ? 1 : 0, ex.pairs_per_row_last, (double)pairs_out / (double)std::max<long>(N, 1),
                             t_total, t_screen);
            if (ex.skip_mode == 1) {
                if (idle_plan) {
                    if (++ex.tr_idle >= 2) { ex.tr_cooldown = ex.tr_pause; ex.tr_pause = std::min(2 * ex.tr_pause, 16); }
                } else {
                    ex.tr_idle = 0; ex.tr_pause = 2;
                }
            }
            return 0;
        }
        ex.planned += 1;
        if (resort) {
            // a sort that did not pay doubles the wait before the next one, up to 64 epochs; one that paid resets it to eight.  Paid:
            // the blocks it saved against the stale order's share, at the measured screen time per block, over the epochs the order
            // will serve, outweigh the measured sort + gather (before those are measured: the share fell by 7 % or more)
            if (ex.res_valid && ex.res_share_last < 0.25 && ex.res_l2_last == ex.l2_live) {
                const bool paid = policy::sort_paid(cost, ex.res_share_last, share, blocks_per_row, ex.res_forced);
                ex.res_forced = paid ? 8 : std::min(2 * ex.res_forced, 64);
            }
            ex.resorts += 1; ex.res_since = 0; ex.res_share_sort = share; ex.res_l2_sort = ex.l2_live; ex.res_valid = true; ex.res_rows = (const void*)X; ex.res_n = N;
        }
        ex.res_since += 1;
        ex.res_share_last = share; ex.res_l2_last = ex.l2_live;
        ex.l1_share_last = l1_share;
        if (h->debug)
            std::fprintf(stderr, "[somhip] exact plan %ld: share %.4f level-1 %.4f level-2 %d (paid %d) sorted %d scout %d (since %d, next forced at %d) refine %d pairs/row %.2f -> %.2f; "
                         "scout wins %.3f; %.3f ms (screen %.3f, level 2 %.3f, sort %.3f) [block %.3g us, level 2 per kept group %.3g us, ratio %.3f]\n",
                         (long)ex.planned, share, ex.l1_share_last, ex.l2_live ? 1 : 0, ex.l2_pays ? 1 : 0, resort ? 1 : 0, scout ? 1 : 0, ex.res_since, ex.res_forced,
                         ex.refine_live ? 1 : 0, ex.pairs_per_row_last, (double)pairs_out / (double)std::max<long>(N, 1), ex.scout_win_share, t_total, t_screen, t_l2, t_sort,
                         1e3 * cost.blk_ms, 1e3 * cost.l2_ms_group, cost.l2_ratio);
        if (ex.l2_live) {
            // level 2 pays where the blocks it removes from a kept group -- (1 - ratio) of four, at the measured screen time per block
            // -- cost more than its own measured time per kept group (before both are measured: round 4's fitted rule)
            ex.l2_pays = policy::level2_pays(cost, share, ex.l1_share_last);
            ex.l1_share_probe = ex.l1_share_last;
            ex.l2_wait = 4;
        } else {
            ex.l2_wait -= 1;
        }
        if (ex.skip_mode == 1) {
            if (idle_plan) {
                if (++ex.skip_idle >= 2) { ex.skip_cooldown = ex.skip_pause; ex.skip_pause = std::min(2 * ex.skip_pause, 16); }
            } else {
                ex.skip_idle = 0; ex.skip_pause = 2;
            }
        }
    }
    return 0;
}

