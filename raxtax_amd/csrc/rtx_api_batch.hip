// The per-batch workspace, the upload (rtx_batch_prefetch / rtx_batch_activate) and the kernel sequence of a sub-batch on the handle's HIP
// stream: order -> kmer_extract -> [pair_union] -> [bounds -> prune -> lists of the live tiles] -> hit_count -> prob -> taxon_prefix + walk.
#include "rtx_index.hpp"

#ifndef RTX_B2_HEAVY_PER_ATILE
#define RTX_B2_HEAVY_PER_ATILE 4u  // two-level bounds pass: a query whose rule asks for more B-tiles per A-tile (of 16) than this goes to the one-level pass (with lo = 0.36 t: 2 / 3 / 4: 80.9 / 79.2 / 78.8 ms per step at configs[2], 6.19 / 6.13 / 6.10 M queries/s at 5 % divergence; lo = 0.33 t and 3: 78.7 ms, 5.84 M)
#endif

namespace rtxi {

int bind(rtx_index *ix) {
    if (!ix) { set_error("null index handle"); return RTX_ERR_INVALID; }
    RTX_HIP(hipSetDevice(ix->device));
    return RTX_OK;
}

int ensure_events(rtx_index *ix, size_t count) {
    while (ix->events.size() < count) {
        hipEvent_t e;
        RTX_HIP(hipEventCreate(&e));
        ix->events.push_back(e);
    }
    return RTX_OK;
}


SubBatch sub_batch_of(rtx_index *ix, uint32_t sb, bool timed) {
    SubBatch b;
    b.sb = sb;
    if (sb < ix->sub_q0.size()) {  // the plan of the run (plan_sub_batches): classes of different sub-batch sizes follow one another
        b.q0 = ix->sub_q0[sb];
        b.nq = ix->sub_nq[sb];
    } else {
        b.q0 = (uint64_t)sb * ix->sub_batch;
        b.nq = (uint32_t)std::min<uint64_t>(ix->sub_batch, ix->n_q - b.q0);
    }
    b.set = ix->staged ? (sb & 1u) : 0u;
    b.s = ix->stream;
    b.timed = timed;
    b.timed_all = timed && ix->stage_timing != 0;
    return b;
}

// The class whose sub-batches are enqueued next: its shape becomes the handle's (the kernel parameters are filled from these fields).
void apply_class(rtx_index *ix, uint32_t c) {
    const rtx_index::BatchClass &k = ix->cls[c];
    ix->diet_used = k.diet;
    ix->cnt_rows_cur = k.diet ? k.cnt_rows : k.sub_batch;
    ix->tmax = k.tmax;
    ix->kstride = k.kstride;
    ix->rstride = k.rstride;
    ix->hstride = k.hstride;
    ix->planes = k.planes;
    ix->sub_batch = k.sub_batch;
    ix->use_tables = k.use_tables;
    ix->pair_used = k.pair;
    ix->prune_used = k.prune;
    ix->rec_used = k.rec;
    ix->cur_cls = (int)c;
}

// The sub-batches of the run, class after class (positions of the processing order: the class leads the sort key, order_batch).
int plan_sub_batches(rtx_index *ix) {
    ix->sub_q0.clear();
    ix->sub_nq.clear();
    ix->sub_cls.clear();
    uint64_t pos = 0;
    for (uint32_t c = 0; c < ix->n_cls; c++) {  // positions: the classes in the order of their sort rank
        ix->cls[c].pos0 = pos;
        pos += ix->cls[c].n;
    }
    if (pos != ix->n_q) { set_error("internal: the length classes hold %llu of %llu queries", (unsigned long long)pos, (unsigned long long)ix->n_q); return RTX_ERR_STATE; }
    // execution: the side classes first (a few long reads: their slow, latency-bound back halves then run beside the front halves of the
    // bulk instead of behind everything), then the others -- the last sub-batch of a run, the one the taps read, belongs to the bulk
    for (int pass = 0; pass < 2; pass++)
        for (uint32_t c = 0; c < ix->n_cls; c++) {
            rtx_index::BatchClass &k = ix->cls[c];
            if (k.side != (pass == 0)) continue;
            k.sb0 = (uint32_t)ix->sub_q0.size();
            for (uint64_t a = 0; a < k.n; a += k.sub_batch) {
                ix->sub_q0.push_back(k.pos0 + a);
                ix->sub_nq.push_back((uint32_t)std::min<uint64_t>(k.sub_batch, k.n - a));
                ix->sub_cls.push_back((uint8_t)c);
            }
            k.n_sub = (uint32_t)ix->sub_q0.size() - k.sb0;
        }
    ix->n_sub_total = (uint32_t)ix->sub_q0.size();
    return RTX_OK;
}

// Counts of a sub-batch between hit_count and taxon_prefix.  With 10 bit planes (t <= 1023) they travel packed,
// 10 bits per reference: [B][npad] low bytes, then [B][npad / 8] u16 with the two high bits of eight references
// each; otherwise [B][npad] u16.  Both live in the same allocation (sized for the format in use).
// Rows: one per query of the sub-batch -- or, behind tile pruning with the records path (BatchClass::diet), sub_batch >> diet_shift rows that
// prune_kernel hands to the queries that take the dense epilogues (HitParams::cnt_row): at configs[2] 83 of the 125 GB of two scratch sets were
// counts that 99 % of the queries never wrote.  The recounting taps (rtx_api_debug.hip) count a sub-batch in full: a row per query again.
constexpr uint32_t kDietMinRows = 1024;
uint32_t diet_rows(const rtx_index *ix, uint32_t B) { return std::min<uint32_t>(B, std::max<uint32_t>(kDietMinRows, B >> ix->diet_shift)); }
uint32_t counts_rows_layout(const rtx_index *ix) { return ix->diet_used && !ix->dbg_full_run && !ix->dbg_full ? ix->cnt_rows_cur : ix->sub_batch; }
uint8_t *counts_lo(rtx_index *ix, rtx_index::Scratch &sc) { return reinterpret_cast<uint8_t *>(sc.d_counts.p); }
uint16_t *counts_hi(rtx_index *ix, rtx_index::Scratch &sc) {
    return reinterpret_cast<uint16_t *>(reinterpret_cast<uint8_t *>(sc.d_counts.p) + (size_t)counts_rows_layout(ix) * ix->npad);
}
size_t counts_elems(const rtx_index *ix, uint64_t B) {  // u16 elements of d_counts
    return ix->packed() ? (size_t)B * ix->npad * 5 / 8 : (size_t)B * ix->npad;
}
int ensure_full_counts(rtx_index *ix, rtx_index::Scratch &sc) { return sc.d_counts.alloc(counts_elems(ix, ix->sub_batch)); }

hipEvent_t stage_event(rtx_index *ix, const SubBatch &b, int stage, int which) {
    return ix->events[((size_t)b.sb * RTX_NUM_STAGES + stage) * 2 + which];
}

// group 1: kmer_extract + hit_count -> counts, per-shard histogram
static KmerParams kmer_params(rtx_index *ix, const SubBatch &b) {
    rtx_index::Scratch &sc = ix->sc[b.set];
    KmerParams kp{};
    kp.bases = ix->d_bases.p;
    kp.base_off = ix->in[ix->cur_in].d_base_off.p;
    kp.q0 = b.q0;
    kp.perm = ix->d_perm.p;
    kp.row_of = ix->d_row_of.p;
    kp.list_len = ix->d_list_len.p;
    kp.row_len = ix->d_row_len.p;
    kp.zero_row = ix->n_rows;
    kp.kmers = sc.d_kmers.p;
    kp.kstride = ix->kstride;
    kp.seginfo = ix->d_seginfo.p;
    kp.seg_stride = ix->seg_stride;
    kp.segcls = ix->d_segcls.p;
    kp.cls_stride = ix->cls_stride;
    kp.ntiles = ix->ntiles;
    kp.seg_dbits = ix->d_seg_dbits.p;
    kp.seg_sbits = ix->d_seg_sbits.p;
    kp.seg_sbase = ix->d_seg_sbase.p;
    kp.seg_blocks = ix->seg_blocks;
    kp.rows = sc.d_rows.p;
    kp.rstride = ix->rstride;
    kp.dmask = sc.d_dmask.p;
    kp.srows = sc.d_srows.p;
    kp.nsparse = sc.d_nsparse.p;
    kp.t = sc.d_t.p;
    kp.nrows = sc.d_nrows.p;
    kp.hq = ix->d_hq.p;
    kp.t_all = ix->d_t_all.p;
    kp.nrows_all = ix->d_nrows_all.p;
    kp.hist = sc.d_hist.p;  // zeroed by kmer_extract for hit_count's global atomics
    kp.hstride = ix->hstride;
    return kp;
}

int enqueue_kmer(rtx_index *ix, const SubBatch &b, hipStream_t s) {
    KmerParams kp = kmer_params(ix, b);
    // with tile pruning the per-tile lists wait until the live tiles are known (enqueue_hit); databases of few tiles build
    // their lists in one pass per tile whatever is live
    kp.mode = ix->prune_used && !ix->dbg_full_run && ix->seg_blocks ? 1u : 0u;
    if (b.timed_all) RTX_HIP(hipEventRecord(stage_event(ix, b, RTX_STAGE_KMER_EXTRACT, 0), s));
    launch_kmer_extract(s, kp, b.nq);
    if (b.timed_all) RTX_HIP(hipEventRecord(stage_event(ix, b, RTX_STAGE_KMER_EXTRACT, 1), s));
    return RTX_OK;
}

// part 0: everything.  A reference shard that prunes stops in the middle for the exchange of the best blocks: part 1 = up to the
// candidates (bounds pass, prune_kernel phase 1), part 2 = the rest (prune_kernel phase 2, lists of the live tiles, counting).
int enqueue_hit(rtx_index *ix, const SubBatch &b, uint32_t flags, hipStream_t s, int part, hipStream_t s_mid) {
    rtx_index::Scratch &sc = ix->sc[b.set];
    ix->last_set = b.set;
    HitParams hp{};
    hp.bitmap = ix->d_bitmap.p;
    hp.stride_bytes = ix->stride_bytes;
    hp.n_rows1 = ix->n_rows + 1;
    hp.n_refs = ix->n_refs;
    hp.ref_base = ix->ref_lo;
    hp.rows = sc.d_rows.p;
    hp.rstride = ix->rstride;
    hp.dmask = sc.d_dmask.p;
    hp.nrows = sc.d_nrows.p;
    hp.zero_row = ix->n_rows;
    hp.srows = sc.d_srows.p;
    hp.nsparse = sc.d_nsparse.p;
    hp.segslots = ix->d_segslots.p;
    hp.ntiles = ix->ntiles;
    hp.t = sc.d_t.p;
    hp.counts = sc.d_counts.p;
    hp.counts_lo = ix->packed() ? counts_lo(ix, sc) : nullptr;  // null: u16 counts
    hp.counts_hi = ix->packed() ? counts_hi(ix, sc) : nullptr;
    hp.npad = ix->npad;
    hp.hist = sc.d_hist.p;
    hp.hstride = ix->hstride;
    hp.tile_max = sc.d_tilemax.p;
    hp.flags = flags;
    hp.q0 = b.q0;
    hp.perm = ix->d_perm.p;
    hp.exact = ExactRef{ix->in[ix->cur_in].d_exact_ids.p, ix->in[ix->cur_in].d_exact_off.p, ix->dev_exact_used ? ix->d_exact_grp.p : nullptr, ix->d_em_goff.p, ix->d_em_gids.p};
    hp.nq = b.nq;
    hp.group_rows = ix->pair_used ? ix->d_group_rows.p : nullptr;
    hp.group_base = b.sb * ix->groups_per_sub;
    hp.pair_urec = sc.d_urec.p;
    hp.pair_nu = sc.d_nu.p;
    hp.pair_ustride = 2u * ix->rstride;
    hp.live = nullptr;
    hp.live_words = 0;
    hp.items = nullptr;
    hp.n_items = nullptr;
    hp.prune_thr = nullptr;
    const bool prune = ix->prune_used && !ix->dbg_full_run;
    if (ix->pair_used && part != 2) {
        if (b.timed_all) RTX_HIP(hipEventRecord(stage_event(ix, b, RTX_STAGE_PAIR_UNION, 0), s));
        launch_pair_union(s, sc.d_rows.p, sc.d_nrows.p, ix->rstride, b.nq, sc.d_urec.p, sc.d_nu.p, 2u * ix->rstride);
        if (b.timed_all) RTX_HIP(hipEventRecord(stage_event(ix, b, RTX_STAGE_PAIR_UNION, 1), s));
    }
    if (b.timed && !prune) RTX_HIP(hipEventRecord(stage_event(ix, b, RTX_STAGE_HIT_COUNT, 0), s));
    if (part != 0 && !prune) { set_error("internal: a split run without tile pruning"); return RTX_ERR_STATE; }
    if (prune) {
        // (1) the queries against the union bitmap: every row dense, no lists, packed counts (bounds per block of references)
        HitParams up = hp;
        up.bitmap = ix->d_ubitmap.p;
        up.stride_bytes = ix->u_stride_bytes;
        up.n_refs = ix->u_nblocks;
        up.dmask = nullptr;    // (a union bitmap is read densely: the kernel takes every row of the query, no masks, no sparse lists)
        up.nsparse = nullptr;
        up.ntiles = ix->u_ntiles;
        up.counts = nullptr;  // nothing is stored per block: the epilogue keeps the largest bound per tile and the best block
        up.counts_lo = nullptr;
        up.counts_hi = nullptr;
        up.hist = nullptr;
        up.tile_max = nullptr;
        up.bounds_tile_ub = sc.d_tile_ub.p;
        up.bounds_tile_stride = ix->ntiles;
        up.bounds_ntiles = ix->ntiles;
        up.bounds_best = sc.d_best_key.p;
        up.flags = 0;
        up.group_base = hp.group_base + ix->n_groups_run;  // work accounting apart from the counting proper
        if (part != 2) {
            if (b.timed) RTX_HIP(hipEventRecord(stage_event(ix, b, RTX_STAGE_TILE_BOUNDS, 0), s));
            // whole-database handles: the two-level pass (rtx_bounds2.hip); reference shards and RTX_OPT_TWO_LEVEL_BOUNDS = 0: blocks of 64 throughout
            ix->two_level_used = part == 0 && ix->two_level_opt && ix->d_abitmap.p && ix->d_bbitmap.p && ix->n_refs == ix->n_total && ix->planes <= kBounds2MaxPlanes;
            if (ix->two_level_used) {
                Bounds2Params bp{};
                bp.abitmap = ix->d_abitmap.p;
                bp.bbitmap = ix->d_bbitmap.p;
                bp.n_rows1 = ix->n_rows + 1;
                bp.n_atiles = ix->n_atiles;
                bp.ntiles = ix->ntiles;
                bp.zero_row = ix->n_rows;
                bp.pair_urec = sc.d_urec.p;
                bp.pair_nu = sc.d_nu.p;
                bp.pair_ustride = 2u * ix->rstride;
                bp.nq = b.nq;
                bp.t = sc.d_t.p;
                bp.tile_ub = sc.d_tile_ub.p;
                bp.tile_ub_stride = ix->ntiles;
                bp.best_key = sc.d_best_key.p;
                bp.delta_ct = ix->b2_delta[0];
                bp.delta_cm = ix->b2_delta[1];
                bp.delta_lo = ix->b2_delta[2];
                bp.delta_hi = ix->b2_delta[3];
                bp.group_rows = up.group_rows;
                bp.group_base = up.group_base;
                // queries whose rule asks for more than five B-tiles per A-tile go to the one-level pass (5 folds of level B ~ half of it)
                const bool heavy_ok = sc.d_heavy.p && sc.d_heavy_items.p && sc.d_heavy_items.n >= (size_t)((b.nq + 1u) / 2u) * ix->u_ntiles + 9u;
                bp.heavy = heavy_ok ? sc.d_heavy.p : nullptr;
                bp.heavy_max = std::max<uint32_t>(1u, std::min<uint32_t>(RTX_B2_HEAVY_PER_ATILE * ix->n_atiles, ix->n_btiles / 2u));
                launch_bounds2(s, bp, b.nq, ix->planes, up, ix->u_ntiles, heavy_ok ? sc.d_heavy_items.p : nullptr);
            } else {
                RTX_HIP(hipMemsetAsync(sc.d_best_key.p, 0, (size_t)b.nq * 4, s));  // the waves of a query's union tiles meet in an atomicMax
                launch_hit_count_pair_bounds(s, up, b.nq, ix->u_ntiles, ix->planes);  // the union of the pair's rows serves both passes
            }
            if (b.timed) RTX_HIP(hipEventRecord(stage_event(ix, b, RTX_STAGE_TILE_BOUNDS, 1), s));
        }
        if (s_mid && s_mid != s && part == 0) {  // RTX_OPT_OVERLAP = 2: threshold, lists and counting go on on a stream of their own
            RTX_HIP(hipEventRecord(ix->ev_mid[b.sb], s));
            RTX_HIP(hipStreamWaitEvent(s_mid, ix->ev_mid[b.sb], 0));
            s = s_mid;
        }
        if (b.timed && part != 1) RTX_HIP(hipEventRecord(stage_event(ix, b, RTX_STAGE_TILE_PRUNE, 0), s));
        // (2) bounds per tile, a lower bound of the best hit, the threshold, the live tiles of every pair
        PruneParams pr{};
        pr.tile_ub = sc.d_tile_ub.p;
        pr.best_key = sc.d_best_key.p;
        pr.tile_ub_stride = ix->ntiles;
        pr.ntiles = ix->ntiles;
        pr.nq = b.nq;
        pr.n_refs = ix->n_refs;
        pr.n_total = ix->n_total;
        pr.ref_base = ix->ref_lo;
        pr.phase = (uint32_t)part;
        pr.best = part ? sc.d_best.p : nullptr;
        pr.bitmap = ix->d_bitmap.p;
        pr.cbitmap = part == 0 && (ix->two_level_used || (ix->planes > kBounds2MaxPlanes && ix->two_level_opt && ix->n_refs == ix->n_total)) ? reinterpret_cast<const uint2 *>(ix->d_cbitmap.p) : nullptr;  // (with the two-level pass: RTX_OPT_TWO_LEVEL_BOUNDS = 0 is the round-4 path whole)
        pr.n_rows1 = ix->n_rows + 1;
        pr.stride_bytes = ix->stride_bytes;
        pr.rows = sc.d_rows.p;
        pr.rstride = ix->rstride;
        pr.nrows = sc.d_nrows.p;
        pr.t = sc.d_t.p;
        pr.flags = flags;
        pr.q0 = b.q0;
        pr.perm = ix->d_perm.p;
        pr.exact = hp.exact;
        pr.lnfact = ix->d_lnfact.p;
        pr.inv = ix->d_inv.p;
        pr.nlf = std::min<uint32_t>(kLnFactLen, ix->tmax + ix->tmax / 2 + 2);  // (pruning runs with tmax <= 2047: 12 KB at t <= 1023, 25 KB at most)
        pr.hist = sc.d_hist.p;
        pr.hstride = ix->hstride;
        pr.live = sc.d_live.p;
        pr.live_words = (ix->ntiles + 31u) / 32u + 1u;
        pr.pair_live = sc.d_items.p + (size_t)((b.nq + 1u) / 2u) * ix->ntiles + 9u;
        pr.thr_out = sc.d_prune_thr.p;
        pr.i1_out = sc.d_prune_i1.p;
        pr.stats = ix->d_prune_stats.p;
        pr.detail = ix->debug_taps && ix->d_prune_detail.n >= (size_t)b.nq * kPruneDetailWords ? ix->d_prune_detail.p : nullptr;
        // the records path: whole-database handles whose walk rides in the prefix launch (enqueue_prob_prefix starts records_tail_kernel there)
        const bool records = part == 0 && ix->rec_used && sc.d_rec.p != nullptr;
        const RecordRef rr{sc.d_rec_nslots.p, sc.d_rec_slots.p, sc.d_rec_cnt.p, sc.d_rec.p, std::min<uint32_t>(ix->rec_opt, kRecMaxSlots), ix->rec_seg_len, ix->d_flags.p};
        if (records) { pr.rec = rr; pr.rec_max_slots = rr.stride; }
        if (ix->diet_used && !(records && sc.d_cnt_row.p)) { set_error("internal: the counts buffer is on its diet without the records path"); return RTX_ERR_STATE; }
        if (records && ix->diet_used && sc.d_cnt_row.p) {  // the rows of the counts buffer are handed out with the decision about the records path
            RTX_HIP(hipMemsetAsync(sc.d_cnt_cursor.p, 0, 4, s));
            pr.cnt_row = sc.d_cnt_row.p;
            pr.cnt_cursor = sc.d_cnt_cursor.p;
            pr.cnt_cap = ix->cnt_rows_cur;
            pr.flags_out = ix->d_flags.p;
            hp.cnt_row = sc.d_cnt_row.p;
        }
        ProbTables tb{ix->d_tab_cmf.p, ix->d_tab_ratio.p, ix->d_tab_off.p, ix->d_tab_moff.p, ix->d_tab_ilo.p, ix->d_tab_sat.p, ix->tab_tmax};
        launch_prune(s, pr, tb, b.nq);
        if (part == 1) { RTX_HIP(hipGetLastError()); return RTX_OK; }  // the caller exchanges RTX_BUF_BEST, then part 2
        // (2b) second stage of the bounds (whole-database handles with a fine union bitmap): the pairs that are left many live tiles are
        // counted against the union bitmap over blocks of 8 references, which takes the tiles without a block above the threshold off
        // their lists (fine_epilogue); prune_kernel's number of live tiles per pair is brought up to date for the list below
        if (part == 0 && ix->fine_opt && ix->d_fbitmap.p && ix->pair_used && sc.d_fine_items.p) {
            HitParams fp = hp;
            fp.bitmap = ix->d_fbitmap.p;
            fp.stride_bytes = ix->f_stride_bytes;
            fp.n_refs = ix->f_nblocks;
            fp.ntiles = ix->f_ntiles;
            fp.counts = nullptr;
            fp.counts_lo = nullptr;
            fp.counts_hi = nullptr;
            fp.tile_max = nullptr;
            fp.flags = 0;
            fp.group_rows = nullptr;  // (its rows are not part of the work accounting of the roofline: reported through its own counters)
            fp.live = sc.d_live.p;
            fp.live_words = pr.live_words;
            fp.prune_thr = sc.d_prune_thr.p;
            fp.fine_n_refs = ix->n_refs;
            fp.fine_ref_ntiles = ix->ntiles;
            fp.fine_stats = ix->d_prune_stats.p + 2 * kPruneStatCopies * 8;
            const size_t cap_f = (size_t)((b.nq + 1u) / 2u) * ix->f_ntiles;
            launch_fine_bounds(s, fp, b.nq, ix->ntiles, ix->f_ntiles, pr.pair_live, sc.d_fine_items.p + cap_f + 9u, sc.d_fine_items.p, sc.d_fine_items.p + cap_f, ix->planes);
        }
        // (3) tiles that are not counted keep a largest count of 0: taxon_prefix leaves them out
        RTX_HIP(hipMemsetAsync(sc.d_tilemax.p, 0, (size_t)b.nq * ix->ntiles * 2, s));
        hp.live = sc.d_live.p;
        hp.live_words = pr.live_words;
        hp.prune_thr = sc.d_prune_thr.p;
        if (records) hp.rec = rr;
        if (ix->pair_used) {  // the grid of the counting pass walks the live (pair, tile) blocks instead of all of them
            const size_t np = (b.nq + 1u) / 2u, cap = np * ix->ntiles;
            launch_live_items(s, sc.d_live.p, pr.live_words, pr.pair_live, b.nq, ix->ntiles, sc.d_items.p + cap + 9u + np, sc.d_items.p, sc.d_items.p + cap);
            hp.items = sc.d_items.p;
            hp.n_items = sc.d_items.p + cap;
        }
        if (ix->seg_blocks) {  // (4) the row lists of the live tiles (kmer_extract left them out)
            KmerParams kp = kmer_params(ix, b);
            kp.mode = 2u;
            kp.live = sc.d_live.p;
            kp.live_words = pr.live_words;
            launch_kmer_extract(s, kp, b.nq);
        }
        if (b.timed) {
            RTX_HIP(hipEventRecord(stage_event(ix, b, RTX_STAGE_TILE_PRUNE, 1), s));
            RTX_HIP(hipEventRecord(stage_event(ix, b, RTX_STAGE_HIT_COUNT, 0), s));
        }
    }
    if (ix->pair_used) launch_hit_count_pair(s, hp, b.nq, ix->ntiles, ix->planes);
    else launch_hit_count(s, hp, b.nq, ix->ntiles, ix->planes);
    if (b.timed) RTX_HIP(hipEventRecord(stage_event(ix, b, RTX_STAGE_HIT_COUNT, 1), s));
    ix->hit_stream = s;
    return RTX_OK;
}

int enqueue_count(rtx_index *ix, const SubBatch &b, uint32_t flags, hipStream_t s_mid) {
    int rc = enqueue_kmer(ix, b, b.s);
    return rc ? rc : enqueue_hit(ix, b, flags, b.s, 0, s_mid);
}

static WalkParams walk_params(rtx_index *ix, const SubBatch &b, const double *prefix);
static int reset_sub_alloc(rtx_index *ix, hipStream_t s);

// group 2: prob table from the (whole-database) histogram + prefix sums over this handle's references; with
// fuse_walk (whole database on this handle) the taxonomy walk of group 3 runs inside the prefix kernel
int enqueue_prob_prefix(rtx_index *ix, const SubBatch &b, bool fuse_walk, bool prob_only) {
    rtx_index::Scratch &sc = ix->sc[b.set];
    hipStream_t s = b.s;
    ProbParams pp{};
    pp.t = sc.d_t.p;
    pp.hist = sc.d_hist.p;
    pp.hstride = ix->hstride;
    pp.tmax = ix->tmax;
    pp.n1max = ix->tmax / 2 + 1;
    pp.lnfact = ix->d_lnfact.p;
    pp.n_refs = ix->n_total;
    pp.q0 = b.q0;
    pp.table_z = sc.d_table_z.p;
    pp.z = ix->d_z.p;
    pp.gs = ix->d_gs.p;
    pp.status = ix->d_status.p;
    pp.ndist = ix->d_ndist.p;
    pp.prune_thr = ix->prune_used && !ix->dbg_full_run ? sc.d_prune_thr.p : nullptr;
    pp.prune_i1 = pp.prune_thr ? sc.d_prune_i1.p : nullptr;
    if (b.timed_all) RTX_HIP(hipEventRecord(stage_event(ix, b, RTX_STAGE_PROB_TABLE, 0), s));
    if (ix->use_tables) {
        ProbTables tb{ix->d_tab_cmf.p, ix->d_tab_ratio.p, ix->d_tab_off.p, ix->d_tab_moff.p,
                      ix->d_tab_ilo.p, ix->d_tab_sat.p, ix->tab_tmax};
        launch_prob_order(s, sc.d_t.p, b.nq, sc.d_order.p);
        pp.order = sc.d_order.p;
        launch_prob_lookup(s, pp, tb, b.nq);
    } else {
        if (ix->cur_cls >= 0 && ix->cls[ix->cur_cls].huge) {  // its arrays do not fit LDS: a stretch of global memory per query
            pp.gstride = (uint32_t)((prob_table_lds_bytes(ix->tmax) + 7) / 8);
            if (ix->d_prob_scratch.n < (size_t)b.nq * pp.gstride) { set_error("internal: scratch of prob_table too small"); return RTX_ERR_STATE; }
            pp.gscratch = ix->d_prob_scratch.p;
        }
        launch_prob_table(s, pp, b.nq);
    }
    if (b.timed_all) RTX_HIP(hipEventRecord(stage_event(ix, b, RTX_STAGE_PROB_TABLE, 1), s));
    if (prob_only) return RTX_OK;  // debug taps after a pruned run: counts and table again, the result rows stay

    PrefixParams fp{};
    fp.status = ix->d_status.p;
    fp.t = sc.d_t.p;
    fp.tz_in_lds = (size_t)ix->hstride * 8 <= 16 * 1024 ? 1u : 0u;
    fp.q0 = b.q0;
    fp.counts = sc.d_counts.p;
    fp.counts_lo = counts_lo(ix, sc);
    fp.counts_hi = counts_hi(ix, sc);
    fp.packed = ix->packed() ? 1u : 0u;
    fp.npad = ix->npad;
    fp.table_z = sc.d_table_z.p;
    fp.hstride = ix->hstride;
    fp.n_refs = ix->n_refs;
    fp.bnd_bits = ix->d_bnd_bits.p;
    fp.bnd_rank = ix->d_bnd_rank.p;
    fp.prefix = sc.d_prefix.p;
    fp.n_bnd = ix->n_bnd_local;
    fp.tile_max = ix->tile_skip ? sc.d_tilemax.p : nullptr;
    fp.ntiles = ix->ntiles;
    fp.prune_thr = ix->prune_used && !ix->dbg_full_run ? sc.d_prune_thr.p : nullptr;
    fp.prune_stats = fp.prune_thr ? ix->d_prune_stats.p + kPruneStatCopies * 8 : nullptr;
    fp.fuse_walk = fuse_walk ? 1u : 0u;
    const bool records = fuse_walk && fp.prune_thr && ix->rec_used && sc.d_rec.p != nullptr;
    fp.rec_nslots = records ? sc.d_rec_nslots.p : nullptr;
    fp.cnt_row = records && ix->diet_used && sc.d_cnt_row.p ? sc.d_cnt_row.p : nullptr;
    if (fuse_walk) {
        fp.walk = walk_params(ix, b, sc.d_prefix.p);
        int rc_r = fp.walk.sub_alloc ? reset_sub_alloc(ix, s) : RTX_OK;
        if (rc_r) return rc_r;
    }
    if (b.timed_all) RTX_HIP(hipEventRecord(stage_event(ix, b, RTX_STAGE_TAXON_PREFIX, 0), s));
    launch_taxon_prefix(s, fp, b.nq);
    if (records) {  // the queries on the records path: prefix sums from their records, the walk from LDS (rtx_records.hip)
        TailParams tp{};
        tp.rec = RecordRef{sc.d_rec_nslots.p, sc.d_rec_slots.p, sc.d_rec_cnt.p, sc.d_rec.p, std::min<uint32_t>(ix->rec_opt, kRecMaxSlots), ix->rec_seg_len, ix->d_flags.p};
        tp.t = sc.d_t.p;
        tp.table_z = sc.d_table_z.p;
        tp.hstride = ix->hstride;
        tp.bnd_bits = ix->d_bnd_bits.p;
        tp.bnd_rank = ix->d_bnd_rank.p;
        tp.prefix = sc.d_prefix.p;
        if (fp.cnt_row) { tp.cnt_cursor = sc.d_cnt_cursor.p; tp.cnt_cap = ix->cnt_rows_cur; tp.flags_out = ix->d_flags.p; }
        tp.n_bnd = ix->n_bnd_local;
        tp.nq = b.nq;
        tp.walk = fp.walk;
        tp.prefix_stats = fp.prune_stats;
        tp.stats = ix->d_prune_stats.p ? ix->d_prune_stats.p + 3 * kPruneStatCopies * 8 : nullptr;
        launch_records_tail(s, tp, b.nq);
    }
    if (b.timed_all) RTX_HIP(hipEventRecord(stage_event(ix, b, RTX_STAGE_TAXON_PREFIX, 1), s));
    return RTX_OK;
}

// The final result arrays (finalise_kernel): the per-query fields of the batch in input order, as many rows as the arena holds
// The result state of a batch in two sets (rtx_index::ResultSet): the members of the handle <-> `alt`
void swap_result_sets(rtx_index *ix) {
    rtx_index::ResultSet &a = ix->alt;
    swap_buf(ix->d_status, a.d_status); swap_buf(ix->d_t_all, a.d_t_all); swap_buf(ix->d_nrows_all, a.d_nrows_all); swap_buf(ix->d_n_rows, a.d_n_rows);
    swap_buf(ix->d_flags, a.d_flags); swap_buf(ix->d_ndist, a.d_ndist); swap_buf(ix->d_gs, a.d_gs); swap_buf(ix->d_z, a.d_z);
    swap_buf(ix->d_hq, a.d_hq); swap_buf(ix->d_row_start, a.d_row_start); swap_buf(ix->d_cursor, a.d_cursor); swap_buf(ix->d_arena, a.d_arena);
    swap_buf(ix->d_fin_t, a.d_fin_t); swap_buf(ix->d_fin_row_count, a.d_fin_row_count); swap_buf(ix->d_fin_lineage, a.d_fin_lineage);
    swap_buf(ix->d_fin_node, a.d_fin_node); swap_buf(ix->d_fin_depth, a.d_fin_depth); swap_buf(ix->d_fin_status, a.d_fin_status);
    swap_buf(ix->d_fin_depth8, a.d_fin_depth8); swap_buf(ix->d_fin_hund, a.d_fin_hund); swap_buf(ix->d_fin_gs, a.d_fin_gs);
    swap_buf(ix->d_fin_local, a.d_fin_local); swap_buf(ix->d_fin_conf, a.d_fin_conf); swap_buf(ix->d_fin_row_begin, a.d_fin_row_begin);
    swap_buf(ix->d_fin_cursor, a.d_fin_cursor); swap_buf(ix->d_perm, a.d_perm); swap_buf(ix->d_iperm, a.d_iperm); swap_buf(ix->d_exact_grp, a.d_exact_grp);
    std::swap(ix->fin_cap, a.fin_cap); std::swap(ix->arena_cap, a.arena_cap); std::swap(ix->side_base, a.side_base);
    swap_buf(ix->h_flags, a.h_flags); swap_buf(ix->h_fin_sub, a.h_fin_sub); swap_buf(ix->h_cursor_sub, a.h_cursor_sub);
    ix->ev_sub.swap(a.ev_sub);
    std::swap(ix->ev_exact, a.ev_exact); std::swap(ix->ev_flags, a.ev_flags);
}

// The per-query arrays, the cursors, the arena and the final arrays of the current set, for a batch of n_queries
int alloc_result_set(rtx_index *ix, uint64_t n_queries) {
    int rc;
    if ((rc = ix->d_status.alloc(n_queries)) || (rc = ix->d_t_all.alloc(n_queries)) || (rc = ix->d_nrows_all.alloc(n_queries)) ||
        (rc = ix->d_n_rows.alloc(n_queries)) || (rc = ix->d_gs.alloc(n_queries)) || (rc = ix->d_z.alloc(n_queries)) ||
        (rc = ix->d_hq.alloc(n_queries)) || (rc = ix->d_row_start.alloc(n_queries)) || (rc = ix->d_ndist.alloc(n_queries)) ||
        (rc = ix->d_cursor.alloc(2)) || (rc = ix->d_flags.alloc(1)))
        return rc;
    const uint64_t want_arena = n_queries * 10 + 4096;  // eight rows per query + two for what the sub-allocators leave unused (walk_params)
    if (ix->arena_cap < want_arena) {
        if ((rc = ix->d_arena.alloc(want_arena))) return rc;
        ix->arena_cap = want_arena;
    }
    return alloc_final(ix, n_queries);
}

// A run that may be followed by a run-ahead (RTX_OPT_RUN_AHEAD) leaves out the wait of the handle's stream for the stream of its back
// halves; whoever needs the handle's stream to cover the whole run asks for it here.
int settle_join(rtx_index *ix) {
    if (ix->join_pending && ix->join_ev) RTX_HIP(hipStreamWaitEvent(ix->stream, ix->join_ev, 0));
    ix->join_pending = false;
    return RTX_OK;
}

int alloc_final(rtx_index *ix, uint64_t n_queries) {
    int rc;
    if ((rc = ix->d_fin_t.alloc(n_queries)) || (rc = ix->d_fin_status.alloc(n_queries)) || (rc = ix->d_fin_gs.alloc(n_queries)) ||
        (rc = ix->d_fin_row_begin.alloc(n_queries)) || (rc = ix->d_fin_row_count.alloc(n_queries)) || (rc = ix->d_fin_cursor.alloc(2)))
        return rc;
    const uint64_t rows = ix->arena_cap, D = ix->fin_D;
    if ((rc = ix->d_fin_lineage.alloc(rows)) || (rc = ix->d_fin_node.alloc(rows)) || (rc = ix->d_fin_depth.alloc(rows)) || (rc = ix->d_fin_depth8.alloc(rows)) ||
        (rc = ix->d_fin_local.alloc(rows)) || (rc = ix->d_fin_conf.alloc(rows * D)) || (rc = ix->d_fin_hund.alloc(rows * D)))
        return rc;
    ix->fin_cap = rows;
    return RTX_OK;
}

// Behind the walks of a sub-batch: its rows sorted, finished and laid out for the host's view (rtx_finalise.hip).  The launches of a run
// follow one another (the bulk's on the stream of the back halves, the side classes' behind the join: enqueue_batch), so the rows of a
// sub-batch are [cursor before its launch, cursor behind it) -- h_fin_sub keeps the latter for the streamed download.
int enqueue_finalise(rtx_index *ix, const SubBatch &b, hipStream_t s) {
    FinaliseParams fp{};
    fp.status = ix->d_status.p;
    fp.t_all = ix->d_t_all.p;
    fp.gs = ix->d_gs.p;
    fp.n_rows = ix->d_n_rows.p;
    fp.row_start = ix->d_row_start.p;
    fp.arena = ix->d_arena.p;
    fp.arena_cap = ix->arena_cap;
    fp.perm = ix->d_perm.p;
    fp.q0 = b.q0;
    fp.nq = b.nq;
    fp.node_depth = ix->d_node_depth.p;
    fp.node_sig0 = ix->d_node_sig0.p;
    fp.node_begin = ix->d_node_begin.p;
    fp.node_eb = ix->d_node_eb.p;
    fp.D = ix->fin_D;
    fp.o_t = ix->d_fin_t.p;
    fp.o_status = ix->d_fin_status.p;
    fp.o_gs = ix->d_fin_gs.p;
    fp.o_row_begin = ix->d_fin_row_begin.p;
    fp.o_row_count = ix->d_fin_row_count.p;
    fp.r_lineage = ix->d_fin_lineage.p;
    fp.r_node = ix->d_fin_node.p;
    fp.r_depth = ix->d_fin_depth.p;
    fp.r_depth8 = ix->d_fin_depth8.p;
    fp.r_hund = ix->d_fin_hund.p;
    fp.r_local = ix->d_fin_local.p;
    fp.r_conf = ix->d_fin_conf.p;
    fp.row_cap = ix->fin_cap;
    fp.fin_cursor = ix->d_fin_cursor.p;
    fp.flags_out = ix->d_flags.p;
    launch_finalise(s, fp);
    if (ix->stream_dl && b.sb < ix->h_fin_sub.size())
        RTX_HIP(hipMemcpyAsync(&ix->h_fin_sub[b.sb], ix->d_fin_cursor.p, 8, hipMemcpyDeviceToHost, s));
    return RTX_OK;
}

// group 3: taxonomy walk over prefix sums covering the WHOLE database ([nq][n_bnd], device)
static WalkParams walk_params(rtx_index *ix, const SubBatch &b, const double *prefix) {
    WalkParams wp{};
    wp.status = ix->d_status.p;
    wp.q0 = b.q0;
    wp.prefix = prefix;
    wp.n_bnd = ix->n_bnd;
    wp.rec = ix->d_noderec.p;
    wp.arena = ix->d_arena.p;
    const bool side = ix->cur_cls >= 0 && ix->cls[ix->cur_cls].side;  // its rows go to the top of the arena through a cursor of their own
    wp.arena_cap = side ? ix->arena_cap : ix->side_base;
    wp.arena_cursor = ix->d_cursor.p + (side ? 1 : 0);
    // (launches of a few thousand walks do not contend, and the rows the sub-allocators leave unused must stay within the arena's
    // allowance of two rows per query: at most kWalkSubAllocs * kWalkChunkRows = 8192 per launch of kWalkSubMinQueries or more)
    wp.sub_alloc = !side && b.nq >= kWalkSubMinQueries && ix->arena_cap < (1ull << 32) ? ix->d_sub_alloc.p : nullptr;
    wp.n_rows = ix->d_n_rows.p;
    wp.row_start = ix->d_row_start.p;
    wp.flags_out = ix->d_flags.p;
    return wp;
}
// the sub-allocators of the result arena start empty in every launch that walks (WalkParams::sub_alloc)
static int reset_sub_alloc(rtx_index *ix, hipStream_t s) {
    if (ix->d_sub_alloc.p) RTX_HIP(hipMemsetAsync(ix->d_sub_alloc.p, 0, (size_t)kWalkSubAllocs * kWalkSubStride * 8, s));
    return RTX_OK;
}

int enqueue_walk(rtx_index *ix, const SubBatch &b, const double *prefix, hipStream_t s) {
    const WalkParams wp = walk_params(ix, b, prefix);
    int rc_r = wp.sub_alloc ? reset_sub_alloc(ix, s) : RTX_OK;
    if (rc_r) return rc_r;
    if (b.timed_all) RTX_HIP(hipEventRecord(stage_event(ix, b, RTX_STAGE_LINEAGE_WALK, 0), s));
    launch_lineage_walk(s, wp, b.nq);
    if (b.timed_all) RTX_HIP(hipEventRecord(stage_event(ix, b, RTX_STAGE_LINEAGE_WALK, 1), s));
    return RTX_OK;
}

// Processing order of the uploaded batch: related queries next to each other (rtx_cluster.hip), or input order.
int order_batch(rtx_index *ix, bool cluster) {
    const uint32_t n = (uint32_t)ix->n_q;
    int rc;
    ix->perm_cur ^= 1u;  // (the other set may still be read by the download of the batch before this one)
    if ((rc = ix->d_perm.alloc(n)) || (rc = ix->d_iperm.alloc(n)) || (rc = ix->h_perm_now().resize(n)) || (rc = ix->h_inv_now().resize(n))) return rc;
    const bool multi = ix->n_cls > 1;       // several length classes: the class leads the key, whatever orders the queries inside a class
    const bool sketch = cluster && n > 2;
    if (sketch || multi) {
        if ((rc = ix->d_skey_in.alloc(n)) || (rc = ix->d_skey_out.alloc(n)) || (rc = ix->d_sidx.alloc(n))) return rc;
        size_t tmp = 0;
        if (cluster_sort(ix->stream, nullptr, &tmp, ix->d_skey_in.p, ix->d_skey_out.p, ix->d_sidx.p, ix->d_perm.p, n, multi)) {
            set_error("radix sort: size query failed");
            return RTX_ERR_HIP;
        }
        if (ix->d_sort_tmp.n < tmp && (rc = ix->d_sort_tmp.alloc(tmp + 256))) return rc;
        if (sketch) {
            launch_sketch(ix->stream, ix->d_bases.p, ix->in[ix->cur_in].d_base_off.p, n, ix->d_skey_in.p, ix->d_sidx.p);
            if (ix->d_loc_table.p && ix->locator_opt)
                launch_locator(ix->stream, ix->d_bases.p, ix->in[ix->cur_in].d_base_off.p, n, ix->d_loc_table.p, ix->n_total, ix->d_skey_in.p);
        }
        if (multi) launch_class_keys(ix->stream, ix->d_skey_in.p, ix->in[ix->cur_in].d_base_off.p, n, ix->key_lim, !sketch, ix->d_sidx.p);
        tmp = ix->d_sort_tmp.n;
        if (cluster_sort(ix->stream, ix->d_sort_tmp.p, &tmp, ix->d_skey_in.p, ix->d_skey_out.p, ix->d_sidx.p, ix->d_perm.p, n, multi)) {
            set_error("radix sort of the query sketches failed");
            return RTX_ERR_HIP;
        }
        launch_invert_perm(ix->stream, ix->d_perm.p, n, ix->d_iperm.p);
    } else {
        launch_identity_perm(ix->stream, n, ix->d_perm.p, ix->d_iperm.p);
    }
    RTX_HIP(hipGetLastError());
    RTX_HIP(hipMemcpyAsync(ix->h_perm_now().data(), ix->d_perm.p, (size_t)n * 4, hipMemcpyDeviceToHost, ix->stream));
    RTX_HIP(hipMemcpyAsync(ix->h_inv_now().data(), ix->d_iperm.p, (size_t)n * 4, hipMemcpyDeviceToHost, ix->stream));
    return RTX_OK;
}

int begin_run(rtx_index *ix, uint32_t *n_sub_out, bool *timed_out, bool cluster) {
    int rc_p = plan_sub_batches(ix);
    if (rc_p) return rc_p;
    const uint32_t n_sub = ix->n_sub_total;
    const bool timed = n_sub <= 4096;
    if (timed) {
        int rc = ensure_events(ix, (size_t)n_sub * RTX_NUM_STAGES * 2);
        if (rc) return rc;
    }
    const bool ev_all = timed && ix->stage_timing;
    if (ev_all) RTX_HIP(hipEventRecord(ix->events[(size_t)RTX_STAGE_ORDER * 2], ix->stream));  // sub-batch 0
    int rc_o = order_batch(ix, cluster);
    if (rc_o) return rc_o;
    if (ev_all) RTX_HIP(hipEventRecord(ix->events[(size_t)RTX_STAGE_ORDER * 2 + 1], ix->stream));
    {   // the result arena: the bulk's rows from 0 on, the side classes' (kWalkMaxRows per query at most) at its top
        uint64_t n_side_q = 0;
        for (uint32_t c = 0; c < ix->n_cls; c++)
            if (ix->cls[c].side) n_side_q += ix->cls[c].n;
        const uint64_t side_rows = n_side_q ? n_side_q * kWalkMaxRows + 64 : 0;
        if (side_rows + ix->n_q * 2 + 1024 > ix->arena_cap) {  // (size_workspace has made room; a sharded or debug run has no side class)
            int rc_a = ix->d_arena.alloc(ix->arena_cap + side_rows);
            if (rc_a) return rc_a;
            ix->arena_cap += side_rows;
        }
        if (ix->fin_cap < ix->arena_cap || ix->d_fin_t.n < ix->n_q) {
            int rc_f = alloc_final(ix, ix->n_q);
            if (rc_f) return rc_f;
        }
        ix->side_base = ix->arena_cap - side_rows;
        int rc_h = ix->h_side_base.resize(2);
        if (rc_h) return rc_h;
        ix->h_side_base[0] = 0;
        ix->h_side_base[1] = ix->side_base;
        RTX_HIP(hipMemcpyAsync(ix->d_cursor.p, ix->h_side_base.data(), 2 * sizeof(unsigned long long), hipMemcpyHostToDevice, ix->stream));
    }
    RTX_HIP(hipMemsetAsync(ix->d_flags.p, 0, sizeof(uint32_t), ix->stream));
    RTX_HIP(hipMemsetAsync(ix->d_fin_cursor.p, 0, 2 * sizeof(unsigned long long), ix->stream));  // the final rows of this run start at 0
    // tile pruning: the pair kernel, the memoised tables (their ln cmf rows give the threshold), taxon_prefix skipping tiles by
    // their largest count, the whole database on this handle
    // a whole-database handle driven by rtx_batch_run, or a reference shard that was asked to (RTX_OPT_SHARD_PRUNE: the caller then
    // drives rtx_shard_bounds and exchanges the best blocks); never a k-mer shard (its counts are partial sums)
    const bool whole = ix->n_refs == ix->n_total && !ix->staged;
    const bool shard = ix->staged && ix->shard_prune_opt && ix->n_refs != ix->n_total;
    auto scratch_ok = [&](const rtx_index::Scratch &sc, uint32_t B) {  // sized at the upload / rtx_shard_begin (alloc_scratch_set) for this sub-batch size
        return sc.d_tile_ub.p != nullptr && sc.d_tile_ub.n >= (size_t)B * ix->ntiles && sc.d_best_key.n >= B && sc.d_prune_thr.n >= B &&
               sc.d_live.n >= (size_t)(B + 1u) * ((ix->ntiles + 31u) / 32u + 1u) && sc.d_best.n >= (size_t)B * kPruneBestWords &&
               sc.d_items.n >= (size_t)((B + 1u) / 2u) * (ix->ntiles + 2u) + 9u;
    };
    bool any_pair = false, any_prune = false;
    uint32_t b_max = 1;
    for (uint32_t c = 0; c < ix->n_cls; c++) {  // what every class of the batch runs through
        rtx_index::BatchClass &k = ix->cls[c];
        // two neighbours per wave only pays when neighbours are related: with the processing order on
        k.pair = ix->pair_opt && cluster && k.planes <= 11 && ix->n_q > 1 && k.rstride <= 4096;
        const rtx_index::Scratch &s0 = ix->sc[k.side ? kSideSet : 0u];  // (a side class runs through the set of its own)
        if (whole && ix->rec_opt != 0u && ix->pruning())  // (segments that have grown since the workspace was sized: RecordRef::seg_len)
            for (uint32_t j = 0; j < 4u; j++) {
                rtx_index::Scratch &sc = ix->sc[j];
                const size_t need_r = (size_t)k.sub_batch * std::min<uint32_t>(ix->rec_opt, kRecMaxSlots) * ix->rec_seg_len;
                if ((j == kSideSet) != k.side || sc.d_kmers.p == nullptr || sc.d_rec.p == nullptr || sc.d_rec.n >= need_r) continue;
                RTX_HIP(hipStreamSynchronize(ix->stream));
                if (sc.d_rec.alloc(need_r)) sc.d_rec.release();
            }
        k.prune = ix->pruning() && k.pair && k.use_tables && ix->tile_skip && ix->d_ubitmap.p && (whole || shard) &&
                  scratch_ok(s0, k.sub_batch) && (!ix->staged || scratch_ok(ix->sc[1], k.sub_batch));
        k.rec = k.prune && whole && ix->rec_opt != 0u && ix->n_bnd_local == ix->n_bnd && s0.d_rec.p != nullptr &&
                s0.d_rec.n >= (size_t)k.sub_batch * std::min<uint32_t>(ix->rec_opt, kRecMaxSlots) * ix->rec_seg_len;
        // the diet of the counts buffer: only with the records path (the queries without rows are exactly those on it), in every set the class may run through
        auto diet_ok = [&](const rtx_index::Scratch &sc) { return sc.d_rec.p != nullptr && sc.d_cnt_row.p != nullptr && sc.d_cnt_cursor.p != nullptr && sc.d_cnt_row.n >= k.sub_batch; };
        k.diet = k.rec && diet_rows(ix, k.sub_batch) < k.sub_batch && diet_ok(s0);
        if (k.diet && !k.side)
            for (uint32_t j = 1; j <= 2u; j++)
                if (ix->sc[j].d_kmers.p != nullptr && !diet_ok(ix->sc[j])) k.diet = false;
        k.cnt_rows = k.diet ? diet_rows(ix, k.sub_batch) : k.sub_batch;
        {   // room for that many rows in the sets the class runs through (a class that was sized for the diet and runs without it, a diet that has grown)
            const size_t need = ix->packed_opt && k.planes <= 10 ? (size_t)k.cnt_rows * ix->npad * 5 / 8 : (size_t)k.cnt_rows * ix->npad;
            const size_t need_p = (size_t)k.cnt_rows * ix->n_bnd_local;  // (the rows of the boundary prefix sums go with those of the counts)
            for (uint32_t j = 0; j < 4u; j++) {
                rtx_index::Scratch &sc = ix->sc[j];
                if ((j == kSideSet) != k.side || sc.d_kmers.p == nullptr || (sc.d_counts.n >= need && sc.d_prefix.n >= need_p)) continue;
                RTX_HIP(hipStreamSynchronize(ix->stream));  // (nothing of an earlier run may still read the old buffers)
                int rc_c = sc.d_counts.alloc(need);
                if (!rc_c) rc_c = sc.d_prefix.alloc(need_p);
                if (rc_c) return rc_c;
            }
        }
        any_pair = any_pair || k.pair;
        any_prune = any_prune || k.prune;
        b_max = std::max(b_max, k.sub_batch);
    }
    ix->sub_batch_max = b_max;
    ix->any_prune = any_prune;
    ix->groups_per_sub = (b_max + 1u) / 2u;
    ix->dbg_full = false;
    if (any_prune) {
        int rc_s = ix->d_prune_stats.alloc(kPruneStatCopies * 32);
        if (!rc_s && ix->debug_taps) rc_s = ix->d_prune_detail.alloc((size_t)b_max * kPruneDetailWords);
        if (rc_s) return rc_s;
        RTX_HIP(hipMemsetAsync(ix->d_prune_stats.p, 0, kPruneStatCopies * 256, ix->stream));
    }
    if (any_pair) {
        ix->n_groups_run = n_sub * ix->groups_per_sub;
        int rc_g = ix->d_group_rows.alloc((size_t)2 * n_sub * ix->groups_per_sub);  // second half: the bounds pass of the tile pruning
        if (rc_g) return rc_g;
        RTX_HIP(hipMemsetAsync(ix->d_group_rows.p, 0, (size_t)2 * n_sub * ix->groups_per_sub * 4, ix->stream));
    }
    ix->n_sub_last = timed ? n_sub : 0;
    ix->overlap_used = 0;  // scratch sets in use beside each other (RTX_OPT_OVERLAP)
    // (only behind tile pruning: where every tile is counted the counting pass lives on its rows staying in L2, and the sweeps of a back half
    // beside it cost more than they hide: 990 against 955 ms per 1 M queries at configs[2] with RTX_OPT_TILE_PRUNE = 0)
    if (ix->overlap_opt != 0u && whole && n_sub >= 2 && !ix->shared_device && any_prune && ix->sc[1].d_kmers.p != nullptr && scratch_ok(ix->sc[1], b_max) == scratch_ok(ix->sc[0], b_max)) {
        ix->overlap_used = 2;
        if (ix->overlap_opt >= 2u && n_sub >= 3 && ix->sc[2].d_kmers.p != nullptr && scratch_ok(ix->sc[2], b_max) == scratch_ok(ix->sc[0], b_max)) ix->overlap_used = 3;
    }
    if (ix->n_cls) apply_class(ix, 0);
    if (ix->dev_exact_used) {  // Tree.sequences.get for every query of the batch (raxtax.rs:42), part of the run
        ExactParams xp{ix->d_bases.p, ix->in[ix->cur_in].d_base_off.p, (uint32_t)ix->n_q, ix->d_em_table.p, ix->em_bits, ix->d_em_rep_off.p,
                       ix->d_em_rep_bytes.p, ix->d_exact_grp.p, ix->em_hash_mask};
        const bool ev = timed && ix->stage_timing;
        if (ev) RTX_HIP(hipEventRecord(ix->events[(size_t)RTX_STAGE_EXACT_MATCH * 2], ix->stream));  // sub-batch 0
        launch_exact_match(ix->stream, xp);
        if (ev) RTX_HIP(hipEventRecord(ix->events[(size_t)RTX_STAGE_EXACT_MATCH * 2 + 1], ix->stream));
        if (!ix->ev_exact) RTX_HIP(hipEventCreateWithFlags(&ix->ev_exact, hipEventDisableTiming));
        RTX_HIP(hipEventRecord(ix->ev_exact, ix->stream));
    }
    *n_sub_out = n_sub;
    *timed_out = timed;
    return RTX_OK;
}

// Enqueues every kernel of the uploaded batch (whole-database handle), sub-batch after sub-batch on the handle's stream.
// (Round 2 also offered side streams -- the two small latency-bound kernels, or prob/prefix/walk of sub-batch i, beside the
// counting of sub-batch i + 1: no gain on MI355X in any arrangement, hit_count holds every wave slot of the chip; DESIGN.md
// section 3.  Removed in round 3.)
int enqueue_batch(rtx_index *ix, uint32_t flags) {
    if (ix->n_refs != ix->n_total) {
        set_error("this handle holds a reference shard: drive it with rtx_shard_count/_prob/_walk");
        return RTX_ERR_STATE;
    }
    uint32_t n_sub = 0;
    bool timed = false;
    int rc;
    // (a run that left its join out, and no run-ahead in progress: the kernels of this run write what its back halves may still read)
    if (ix->join_pending && !ix->hold_join && (rc = settle_join(ix))) return rc;
    rc = begin_run(ix, &n_sub, &timed, ix->cluster != 0);
    if (rc) return rc;
    ix->stream_dl = false;
    if (n_sub <= 4096) {  // per sub-batch: completion event (+ cursor snapshot) for the streamed download
        if (!ix->copy_stream) RTX_HIP(hipStreamCreateWithFlags(&ix->copy_stream, hipStreamNonBlocking));
        while (ix->ev_sub.size() < n_sub) {
            hipEvent_t e;
            RTX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ix->ev_sub.push_back(e);
        }
        if ((rc = ix->h_cursor_sub.resize(n_sub)) || (rc = ix->h_fin_sub.resize(n_sub))) return rc;
        ix->n_sub_run = n_sub;
        ix->stream_dl = true;
    }
    // the walk rides inside the prefix kernel (the stage time of lineage_walk is then part of taxon_prefix)
    const bool fuse = ix->n_bnd_local == ix->n_bnd;
    // RTX_OPT_OVERLAP: the back half of sub-batch k (prob_lookup, taxon_prefix / records tail: chains of dependent round trips) on a second
    // stream beside the front half of sub-batch k + 1 (bounds and counting: VALU and L1 rate); two scratch sets alternate (sb & 1), a
    // front half waits for the back half that last used its set.
    const uint32_t nsets = ix->overlap_used;  // 0: one stream; 2: two stages, two scratch sets; 3: front | threshold + counting | back
    const bool overlap = nsets != 0u;
    if (overlap) {
        if (!ix->stream2) RTX_HIP(hipStreamCreateWithFlags(&ix->stream2, hipStreamNonBlocking));
        if (nsets == 3u && !ix->stream3) RTX_HIP(hipStreamCreateWithFlags(&ix->stream3, hipStreamNonBlocking));
        while (ix->ev_front.size() < n_sub) {
            hipEvent_t e, f, g;
            RTX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            RTX_HIP(hipEventCreateWithFlags(&f, hipEventDisableTiming));
            RTX_HIP(hipEventCreateWithFlags(&g, hipEventDisableTiming));
            ix->ev_front.push_back(e);
            ix->ev_back.push_back(f);
            ix->ev_mid.push_back(g);
        }
    }
    uint32_t n_side = 0;
    for (uint32_t c = 0; c < ix->n_cls; c++)
        if (ix->cls[c].side) n_side += ix->cls[c].n_sub;
    // RTX_OPT_RUN_AHEAD: a run of this shape (two streams, no side class) leaves out the join at its end -- the next chunk may be enqueued
    // behind its last FRONT half (rtx_batch_download_then_run) -- and, enqueued that way itself (hold_join: the result sets have been
    // swapped), drops the join of the run before it: its first front halves wait for the scratch sets only.
    const bool ra = ix->run_ahead_opt != 0u && nsets == 2u && n_side == 0u && ix->stream_dl && n_sub >= 2u;  // (RTX_OPT_STAGE_TIMING: the stage events are then those of whichever run recorded them last)
    if (ix->join_pending) {  // (hold_join)
        if (ra) ix->join_pending = false;
        else if ((rc = settle_join(ix))) return rc;
    }
    if (ra)
        for (uint32_t k = 0; k < 2u; k++)
            if (!ix->ev_set_free[k]) RTX_HIP(hipEventCreateWithFlags(&ix->ev_set_free[k], hipEventDisableTiming));
    for (uint32_t sb = 0; sb < n_sub; sb++) {
        if ((int)ix->sub_cls[sb] != ix->cur_cls) apply_class(ix, ix->sub_cls[sb]);  // the next length class: its planes, strides, kernels
        SubBatch b = sub_batch_of(ix, sb, timed);
        const bool side = ix->cls[ix->sub_cls[sb]].side;
        if (side) {  // (the side classes come first: sub-batches 0 .. n_side - 1, one after the other through their own set)
            b.set = kSideSet;
            if (overlap) {
                // ... and as a whole on the stream of the BACK halves, in front of them: a few waves per kernel, every one a long chain of
                // round trips (6 000 rows through one wave, the recurrence of prob_table) -- 3 ms for ten reads of 1 .. 8 kb -- that hides
                // under the bulk's first front half.  (Until the library asked for eight hardware queues this was a stream of its own that
                // SHARED a queue: 12.0 ms per batch of 131 072 barcodes + ten long reads; with a queue of its own every launch of the chain
                // queued for a slot on a full device and the batch took 13.0; in front of the bulk on the handle's stream: 14.7.)
                // Their result rows go to the top of the arena through a cursor of their own (walk_params); they are finalised first.
                if (sb == 0) {
                    RTX_HIP(hipEventRecord(ix->ev_mid[0], ix->stream));  // the processing order and the exact matches are in place
                    RTX_HIP(hipStreamWaitEvent(ix->stream2, ix->ev_mid[0], 0));
                }
                b.s = ix->stream2;
            }
        } else if (overlap) {
            const uint32_t m = sb - n_side;  // among the sub-batches of the bulk
            b.set = m % nsets;
            if (m >= nsets) RTX_HIP(hipStreamWaitEvent(ix->stream, ix->ev_back[sb - nsets], 0));  // the scratch set is free again
            else if (ra && ix->set_busy[b.set]) RTX_HIP(hipStreamWaitEvent(ix->stream, ix->ev_set_free[b.set], 0));  // ... of the run before (run-ahead)
        }
        if ((rc = enqueue_count(ix, b, flags, nsets == 3u && !side && n_side == 0 ? ix->stream3 : nullptr))) return rc;
        if (overlap && !side) {
            RTX_HIP(hipEventRecord(ix->ev_front[sb], ix->hit_stream));
            b.s = ix->stream2;
            RTX_HIP(hipStreamWaitEvent(b.s, ix->ev_front[sb], 0));
        }
        if ((rc = enqueue_prob_prefix(ix, b, fuse))) return rc;
        if (!fuse && (rc = enqueue_walk(ix, b, ix->sc[b.set].d_prefix.p, b.s))) return rc;
        if (fuse && b.timed_all) {  // keeps rtx_batch_stage_times whole: an empty interval
            RTX_HIP(hipEventRecord(stage_event(ix, b, RTX_STAGE_LINEAGE_WALK, 0), b.s));
            RTX_HIP(hipEventRecord(stage_event(ix, b, RTX_STAGE_LINEAGE_WALK, 1), b.s));
        }
        // the rows of the sub-batch finalised on the device (rtx_finalise.hip).  The launches of a run follow one another -- a sub-batch's rows
        // are then a range of the final arrays: the side classes' first, then the bulk's, on the stream of the back halves
        if ((rc = enqueue_finalise(ix, b, b.s))) return rc;
        if (ix->stream_dl) {
            RTX_HIP(hipMemcpyAsync(&ix->h_cursor_sub[sb], ix->d_cursor.p + (side ? 1 : 0), 8, hipMemcpyDeviceToHost, b.s));
            RTX_HIP(hipEventRecord(ix->ev_sub[sb], b.s));
        }
        if (overlap) RTX_HIP(hipEventRecord(ix->ev_back[sb], b.s));
        if (ra) {
            RTX_HIP(hipEventRecord(ix->ev_set_free[b.set], b.s));
            ix->set_busy[b.set] = true;
        }
    }
    if (ra) {  // the join is left to whoever needs it (settle_join)
        ix->join_pending = true;
        ix->join_ev = ix->ev_back[n_sub - 1];
    } else if (overlap && n_sub) {  // a wait for the handle's stream covers all of them
        RTX_HIP(hipStreamWaitEvent(ix->stream, ix->ev_back[n_sub - 1], 0));
        if (n_side && n_side < n_sub) RTX_HIP(hipStreamWaitEvent(ix->stream, ix->ev_back[n_side - 1], 0));  // (a batch of side classes alone)
    }
    {   // the flags of the run, behind its last kernel (the handle's stream has joined the others): the download finds them in page-locked memory
        int rc_f = ix->h_flags.resize(1);
        if (rc_f) return rc_f;
        const hipStream_t fs = ra ? ix->stream2 : ix->stream;  // (run-ahead: behind the last back half -- every front half lies before it)
        RTX_HIP(hipMemcpyAsync(ix->h_flags.data(), ix->d_flags.p, 4, hipMemcpyDeviceToHost, fs));
        if (!ix->ev_flags) RTX_HIP(hipEventCreateWithFlags(&ix->ev_flags, hipEventDisableTiming));
        RTX_HIP(hipEventRecord(ix->ev_flags, fs));
    }
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}

// Builds (once per handle and tmax) the memoised cmf tables used by prob_lookup_kernel.
constexpr uint32_t kProbTablesMaxT = (uint32_t)2047;  // (= kMidClassMaxT: 0.74 GB at t = 651, 9 GB at t = 1 493, 23 GB at t = 2 047)
int ensure_prob_tables(rtx_index *ix, uint32_t tmax, bool *usable) {
    *usable = false;
    if (ix->prob_mode == 1 || tmax < 2) return RTX_OK;
    if (tmax > kProbTablesMaxT) {
        if (ix->prob_mode == 2) { set_error("prob tables need t <= %u (got %u)", kProbTablesMaxT, tmax); return RTX_ERR_TOO_LONG; }
        return RTX_OK;
    }
    if (ix->tab_tmax >= tmax) { *usable = true; return RTX_OK; }
    const uint32_t T = tmax;
    std::vector<uint64_t> off(T + 1, 0);
    std::vector<uint32_t> moff(T + 1, 0);
    uint64_t run = 0;
    uint32_t mrun = 0;
    for (uint32_t t = 2; t <= T; t++) {
        off[t] = run;
        moff[t] = mrun;
        run += (uint64_t)t * (t / 2 + 1);
        mrun += t;
    }
    int rc;
    size_t free_b = 0, total_b = 0;
    RTX_HIP(hipMemGetInfo(&free_b, &total_b));
    if (run * 16 > free_b / 2) {  // keep at least half of the free HBM for the batch workspace
        if (ix->prob_mode == 2) { set_error("prob tables (%llu bytes) do not fit", (unsigned long long)(run * 16)); return RTX_ERR_OOM; }
        return RTX_OK;
    }
    if ((rc = ix->d_tab_cmf.alloc(run)) || (rc = ix->d_tab_ratio.alloc(run)) || (rc = ix->d_tab_off.alloc(T + 1)) ||
        (rc = ix->d_tab_moff.alloc(T + 1)) || (rc = ix->d_tab_ilo.alloc(mrun)) || (rc = ix->d_tab_sat.alloc(mrun)))
        return rc;
    RTX_HIP(hipMemcpy(ix->d_tab_off.p, off.data(), (T + 1) * 8, hipMemcpyHostToDevice));
    RTX_HIP(hipMemcpy(ix->d_tab_moff.p, moff.data(), (T + 1) * 4, hipMemcpyHostToDevice));
    ProbTables tb{ix->d_tab_cmf.p, ix->d_tab_ratio.p, ix->d_tab_off.p, ix->d_tab_moff.p, ix->d_tab_ilo.p, ix->d_tab_sat.p, T};
    launch_prob_tables_build(ix->stream, tb, ix->d_lnfact.p, ix->d_inv.p);
    RTX_HIP(hipGetLastError());
    RTX_HIP(hipStreamSynchronize(ix->stream));
    ix->tab_tmax = T;
    *usable = true;
    return RTX_OK;
}

// Queries per kernel launch: larger sub-batches amortise launch tails (measured: 4096 -> 8192 queries saves
// 5 % of a step at N = 50k).
constexpr uint32_t kMaxSubBatch = 65536;
// default: fewer, larger launches save the drain/fill between the kernels of a sub-batch (N = 50k, per 100k queries: 10 000:
// 20.4 ms, 14 286: 21.3, 25 000: 20.3, 50 000: 22.0).  A large database gains from more queries per launch -- every tile's
// bitmap region is fetched once per launch and XCD, whatever the number of queries (N = 500k, per 1M queries: 10 240: 1 094 ms,
// 16 384: 1 072, 24 576: 1 070, 32 768: 1 098).  With the kernels of the end of round 2: N = 50k, per 100k queries: 10 000: 19.55 ms,
// 20 000: 19.08, 25 000: 21.1 (the last sub-batch's host work is no longer hidden), 50 000: 20.7; N = 500k, per 1 M queries:
// 8 192: 971 ms, 16 384: 953, 32 768: 964, 65 536: 995
// With tile pruning a sub-batch is far less work and the fixed cost of its nine launches counts: N = 500k, per 1 M queries:
// 8 192: 191 ms, 16 384: 171, 32 768: 159.5, 49 152: 158.1, 65 536: 157.8
// Round 4 (stages for every kernel, 113 ms per step): 32 768: 113.1 ms, 49 152: 111.0, 65 536: 109.7 -- fewer launches, fewer tails.
constexpr uint32_t kDefaultSubBatch = 20480, kDefaultSubBatchLarge = 16384, kDefaultSubBatchPruned = 65536;


// ---- length classes ----------------------------------------------------------------------------------------------------------
// class of a query by its length (t <= length - 7, known on the host when the batch is staged):
//   0: t <= 255    8 bit planes, pair kernel, memoised tables, tile pruning
//   1: t <= 1023  10 bit planes, the same
//   2: t <= 2047  11 bit planes, the same with u16 counts (round 6: full-length 16S -- until then every read beyond 1 030 bases fell to class 3's
//                 one-query-per-wave kernel without pruning: 0.26 M reads of 1 500 bases per second against 12.9 M barcodes); fewer than
//                 kMinMidClass of them in a batch ride with class 3 (the tables of this class are gigabytes, built for the batch's longest read)
//   3: longer     12 / 16 planes, one query per wave, prob_table_kernel with its arrays in LDS (t up to ~ 6 600)
//   4: up to t = 65 535 (raxtax.rs:56): 16 planes, the histogram of hit_count and the arrays of prob_table in global memory
constexpr size_t kProbTableLdsLimit = 160 * 1024 - 512;
constexpr uint64_t kMidClassMaxT = 2047;

uint32_t length_class(uint64_t len) {
    const uint64_t t = len >= 8 ? len - 7 : 1;
    if (t <= 255) return 0;
    if (t <= 1023) return 1;
    if (t <= kMidClassMaxT) return 2;
    return t <= 65535 && prob_table_lds_bytes((uint32_t)t) <= kProbTableLdsLimit ? 3u : 4u;
}
uint64_t class3_max_len() {  // the longest query of class 3: where prob_table's arrays still fit LDS
    static uint64_t cached = 0;
    if (!cached) {
        uint64_t lo = kMidClassMaxT + 8, hi = 65535 + 7;
        while (lo < hi) { const uint64_t mid = (lo + hi + 1) / 2; if (length_class(mid) <= 3u) lo = mid; else hi = mid - 1; }
        cached = lo;
    }
    return cached;
}
constexpr uint64_t kSideMaxQueries = 2048;  // a class of at most this many queries (and a 64th of the bulk) runs as a side class
constexpr uint64_t kMinShortClass = 4096;  // fewer queries of t <= 255 than this ride with the t <= 1023 class (same results: 8 or 10 planes hold their counts)
constexpr uint64_t kMinMidClass = 4096;    // fewer queries of 1024 <= t <= 2047 than this ride with the longer reads (one query per wave, the recurrence kernel)

// mid: the class of 1024 <= t <= 2047 proper (eleven planes: pair kernel, pruning, tables); reads of that length that ride with the longer
// ones (too few of them, a reference shard) keep the twelve-plane form of the one-query-per-wave kernel and the recurrence kernel
static void shape_class(rtx_index::BatchClass &k, uint64_t n, uint64_t max_len, bool mid = false) {
    // t <= min(len - 7, 65536): the reference asserts on the DISTINCT k-mers of a read (raxtax.rs:56, utils.rs:27-40), so a read of any
    // length is served; one that really holds all 65 536 8-mers is reported per query (RTX_Q_ALL_KMERS, kmer_extract_kernel)
    const uint64_t tmax = std::min<uint64_t>(max_len >= 8 ? max_len - 7 : 1, 65535);
    k = rtx_index::BatchClass();
    k.n = n;
    k.max_len = max_len;
    k.tmax = (uint32_t)tmax;
    k.kstride = (uint32_t)align_up(tmax, 8);
    k.rstride = (uint32_t)align_up(tmax, 64) + 64;  // row list padded to whole 64-row chunks
    k.hstride = (uint32_t)align_up(tmax + 1, 8);
    k.planes = tmax <= 255 ? 8 : (tmax <= 1023 ? 10 : (tmax <= kMidClassMaxT && mid ? 11 : (tmax <= 4095 ? 12 : 16)));  // (8 / 10 / 11: the pair kernel)
    k.huge = prob_table_lds_bytes((uint32_t)tmax) > kProbTableLdsLimit;
}

static int size_workspace(rtx_index *ix, uint64_t n_queries);

// Sizes and allocates the per-batch workspace for the staged batch: n_queries queries, cls_n[c] of them in length class c (the longest of
// which has cls_max[c] bases).
int prepare_workspace(rtx_index *ix, uint64_t n_queries, const uint64_t cls_n_in[5], const uint64_t cls_max_in[5]) {
    uint64_t cn[5], cm[5];
    for (int c = 0; c < 5; c++) { cn[c] = cls_n_in[c]; cm[c] = cls_max_in[c]; }
    if (ix->n_refs != ix->n_total) {  // a reference / k-mer shard: the exchange buffers of rtx_shard_* have one row stride -- one class
        uint64_t mx = 0;
        for (int c = 0; c < 5; c++) mx = std::max(mx, cm[c]);
        return prepare_workspace_single(ix, n_queries, mx >= 8 ? mx - 7 : 1, mx);
    }
    // a handful of short reads among barcodes ride with them (the results do not depend on the number of planes)
    if (cn[0] && cn[1] && cn[0] < kMinShortClass) { cn[1] += cn[0]; cm[1] = std::max(cm[1], cm[0]); cn[0] = 0; cm[0] = 0; }
    // a few reads of 1 031 .. 2 054 bases (among barcodes, or alone) ride with the longer ones: the pruned path of their own class needs
    // memoised tables of gigabytes (t^3), built for the longest read -- not for a handful of queries.  (No pair kernel, no pruning: no such class.)
    if (cn[2] && ((cn[2] < kMinMidClass && ix->prob_mode != 2) || !ix->pair_opt || !ix->pruning() || ix->prob_mode == 1)) { cn[3] += cn[2]; cm[3] = std::max(cm[3], cm[2]); cn[2] = 0; cm[2] = 0; }
    // ... and a handful of reads of a few kilobases with the longer ones still (one set of launches; the global-memory forms compute the same values)
    if (cn[3] && cn[4] && cn[3] + cn[4] <= kSideMaxQueries) { cn[4] += cn[3]; cm[4] = std::max(cm[4], cm[3]); cn[3] = 0; cm[3] = 0; }
    const uint64_t key[14] = {n_queries, cn[0], cn[1], cn[2] << 32 | cn[3], cn[4], cm[0], cm[1], cm[2] << 32 | cm[3], cm[4], ix->sub_batch_req,
                              (uint64_t)ix->packed_opt | (uint64_t)ix->pair_opt << 1 | (uint64_t)ix->pruning() << 2 | (uint64_t)ix->shard_prune_opt << 3 | (uint64_t)ix->fine_opt << 4 |
                                  (uint64_t)(ix->prob_mode & 3) << 5 | (uint64_t)ix->rec_opt << 8 | (uint64_t)ix->overlap_opt << 16 | (uint64_t)ix->min_subs << 20,
                              (uint64_t)ix->n_bnd_local, ix->shared_device ? 1u : 0u, 0};
    // A batch of the shape of the last one under the same options (the chunks of rtx_raxtax): everything below would come out the same --
    // and hipMemGetInfo alone costs a good part of a millisecond between two chunks, with the device idle
    if (ix->ws_valid && std::memcmp(key, ix->ws_key, sizeof key) == 0 && !ix->staged) {
        ix->n_q = n_queries;
        return RTX_OK;
    }
    ix->ws_valid = false;
    // (reads beyond 65 542 bases belong to the last class; t is what kmer_extract finds, at most 65 535 -- or the query is flagged)
    ix->n_cls = 0;
    // the longest query each class may hold: the sort rank of a query is the number of class boundaries its length exceeds
    const uint64_t cls_len[4] = {255 + 7, 1023 + 7, kMidClassMaxT + 7, class3_max_len()};
    for (int c = 0; c < 4; c++) ix->key_lim[c] = ~0ull;
    int last = -1;
    for (int c = 0; c < 5; c++) {
        if (!cn[c]) continue;
        if (last >= 0) ix->key_lim[ix->n_cls - 1] = cls_len[last];  // a boundary between two classes that both exist
        shape_class(ix->cls[ix->n_cls], cn[c], cm[c], c == 2);
        ix->n_cls++;
        last = c;
    }
    if (ix->n_cls == 0) { set_error("internal: a batch without queries"); return RTX_ERR_INVALID; }
    ix->n_q = n_queries;
    int rc = size_workspace(ix, n_queries);
    if (rc) return rc;
    std::memcpy(ix->ws_key, key, sizeof key);
    ix->ws_valid = true;
    return RTX_OK;
}

// One class whatever the lengths: reference shards (rtx_shard_*: one row stride for the exchange buffers), rtx_debug_evaluate.
int prepare_workspace_single(rtx_index *ix, uint64_t n_queries, uint64_t tmax, uint64_t max_len) {
    ix->ws_valid = false;
    tmax = std::min<uint64_t>(tmax, 65535);  // (shape_class)
    ix->n_cls = 1;
    for (int c = 0; c < 4; c++) ix->key_lim[c] = ~0ull;
    shape_class(ix->cls[0], n_queries, tmax + 7);
    ix->cls[0].max_len = max_len;
    ix->n_q = n_queries;
    return size_workspace(ix, n_queries);
}

// per-query scratch bytes of a class (what a sub-batch of it costs per query)
static uint64_t class_per_q(const rtx_index *ix, const rtx_index::BatchClass &k) {
    const bool packed = ix->packed_opt && k.planes <= 10;
    return (uint64_t)k.kstride * 2 + (uint64_t)k.rstride * 12 + 4 + (uint64_t)ix->ntiles * (k.rstride / 8 + ((kSegMaxSparseRows + 1) * 4 + 10)) + (packed ? ix->npad * 5 / 4 : ix->npad * 2) +
           (uint64_t)k.hstride * 12 + (uint64_t)ix->n_bnd_local * 8 + 64 + (k.huge ? prob_table_lds_bytes(k.tmax) : 0) +
           // + the scratch of the tile pruning: tile bounds, thresholds, live masks, best blocks, the lists of live blocks
           (k.will_prune ? (uint64_t)ix->ntiles * 2 + 12 + (ix->ntiles + 31u) / 32u * 2u + 2u + kPruneBestWords * 4 +
                               ((uint64_t)ix->ntiles + 2u) * 2u  /* the list of live (pair, tile) blocks: 4 bytes per pair and tile */ +
                               (ix->d_fbitmap.p ? (uint64_t)ix->f_ntiles * 2u + 1u : 0u) /* the items of the fine bounds pass */ +
                               (ix->rec_opt && ix->n_refs == ix->n_total ? (uint64_t)std::min<uint32_t>(ix->rec_opt, kRecMaxSlots) * 32768u + kRecMaxSlots * 6u + 2u : 0u) /* record segments */
                           : 0);
}

static int size_workspace(rtx_index *ix, uint64_t n_queries) {
    int rc;
    // the memoised probability tables serve every class with t <= 1023: built for the longest of them
    uint32_t tab_t = 0;
    for (uint32_t c = 0; c < ix->n_cls; c++) {
        if (ix->cls[c].tmax > kProbTablesMaxT && ix->prob_mode == 2) { set_error("prob tables need t <= %u (got %u)", kProbTablesMaxT, ix->cls[c].tmax); return RTX_ERR_TOO_LONG; }
        if (ix->cls[c].tmax <= kProbTablesMaxT && ix->cls[c].planes <= 11) tab_t = std::max(tab_t, ix->cls[c].tmax);  // (not for mid-length reads that ride with the longer ones)
    }
    bool tables = false;
    if (tab_t && (rc = ensure_prob_tables(ix, tab_t, &tables))) return rc;
    for (uint32_t c = 0; c < ix->n_cls; c++) ix->cls[c].use_tables = tables && ix->cls[c].tmax >= 2 && ix->cls[c].tmax <= kProbTablesMaxT && ix->cls[c].planes <= 11;
    // ---- per-query results
    if ((rc = alloc_result_set(ix, n_queries))) return rc;
    if ((rc = ix->d_sub_alloc.alloc((size_t)kWalkSubAllocs * kWalkSubStride))) return rc;
    const uint64_t want_arena = n_queries * 10 + 4096;  // (alloc_result_set)
    // ---- sub-batch scratch, sized against free HBM: every class gets the sub-batch size its own shape allows, the buffers the largest
    // product over the classes (a class runs after the other through the same buffers)
    // RTX_OPT_OVERLAP: two (three) scratch sets -- not for a handle that shares its device with another one driven beside it (rtx_raxtax_multi):
    // two sets of 60 GB each per handle (N = 500k, sub-batches of 65 536) left the second handle of a device a sliver of HBM and sub-batches of
    // a few thousand queries: a tenth of the speed
    uint32_t n_sets = ix->n_refs == ix->n_total && !ix->shared_device ? 1u + std::min<uint32_t>(ix->overlap_opt, 2u) : 1u;
    size_t free_b = 0, total_b = 0;
    RTX_HIP(hipMemGetInfo(&free_b, &total_b));
    uint64_t held = 0;  // scratch already held by this handle is reusable
    for (const auto &sc : ix->sc) held += sc.d_counts.n * 2 + sc.d_prefix.n * 8 + sc.d_rec.n * 4 + sc.d_srows.n * 4;
    const uint64_t budget = (uint64_t)((free_b + held) * 0.6);
    uint64_t worst = 0;
    for (uint32_t c = 0; c < ix->n_cls; c++) {
        rtx_index::BatchClass &k = ix->cls[c];
        k.will_prune = ix->pruning() && ix->d_ubitmap.p && ix->pair_opt && ix->ntiles >= RTX_PRUNE_MIN_TILES && k.tmax <= kProbTablesMaxT && (ix->n_refs == ix->n_total || ix->shard_prune_opt);  // begin_run decides
        const uint64_t per_q = class_per_q(ix, k);
        uint32_t B = ix->sub_batch_req;
        if (B == 0) {
            B = (uint32_t)std::min<uint64_t>(k.will_prune ? kDefaultSubBatchPruned : ix->ntiles >= 16 ? kDefaultSubBatchLarge : kDefaultSubBatch,
                                             std::max<uint64_t>(64, budget / per_q));
            if (k.huge) B = std::min<uint32_t>(B, 1024u);
            // at least four sub-batches per batch (of 16 384 queries or more): the records of a finished sub-batch are copied and finalised on
            // the host while the next ones run, and what is left when the device is done is the last sub-batch -- a chunk of 131 072 queries
            // (rtx_raxtax) in two halves left 7 ms of host work exposed on real barcodes (ten result rows per query)
            if (k.will_prune && k.n < (uint64_t)ix->min_subs * B) B = (uint32_t)std::max<uint64_t>(16384, (k.n + ix->min_subs - 1) / ix->min_subs);
        }
        if (B > kMaxSubBatch) B = kMaxSubBatch;
        B = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(B, k.n));
        k.sub_batch = B;
        worst = std::max<uint64_t>(worst, (uint64_t)B * per_q);
    }
    // side classes: a handful of queries of another length beside the bulk (ten 16S reads in a file of barcodes)
    {
        uint32_t bulk = 0;
        for (uint32_t c = 1; c < ix->n_cls; c++)
            if (ix->cls[c].n > ix->cls[bulk].n) bulk = c;
        worst = 0;
        for (uint32_t c = 0; c < ix->n_cls; c++) {
            ix->cls[c].side = ix->n_refs == ix->n_total && c != bulk && ix->cls[c].n <= kSideMaxQueries && ix->cls[c].n * 64u <= ix->cls[bulk].n;
            if (!ix->cls[c].side) worst = std::max<uint64_t>(worst, (uint64_t)ix->cls[c].sub_batch * class_per_q(ix, ix->cls[c]));
        }
    }
    {   // room for the rows of the side classes at the top of the arena (begin_run: side_base)
        uint64_t n_side_q = 0;
        for (uint32_t c = 0; c < ix->n_cls; c++)
            if (ix->cls[c].side) n_side_q += ix->cls[c].n;
        const uint64_t want2 = want_arena + (n_side_q ? n_side_q * kWalkMaxRows + 64 : 0);
        if (ix->arena_cap < want2) {
            if ((rc = ix->d_arena.alloc(want2))) return rc;
            ix->arena_cap = want2;
        }
    }
    // the further sets of the overlap never shrink a sub-batch: they are taken only while they fit the budget beside the first
    while (n_sets > 1u && (uint64_t)n_sets * worst > budget) n_sets--;
    ix->staged = false;
    if ((rc = plan_sub_batches(ix))) return rc;
    apply_class(ix, ix->n_cls - 1u);
    if ((rc = alloc_scratch_set(ix, 0))) return rc;
    {
        bool any_side = false;
        for (uint32_t c = 0; c < ix->n_cls; c++) any_side = any_side || ix->cls[c].side;
        if (any_side) { if ((rc = alloc_scratch_set(ix, kSideSet))) return rc; }
        else ix->sc[kSideSet].release_all();
    }
    for (uint32_t k = 1; k <= 2u; k++) {  // (without the further sets the run stays on one stream)
        if (k < n_sets && ix->n_sub_total > 1) {
            if (alloc_scratch_set(ix, k)) ix->sc[k].d_kmers.release();
        } else if (ix->n_refs == ix->n_total) {
            ix->sc[k].release_all();  // none wanted on a whole-database handle: given back (a shard keeps its second set for rtx_shard_begin)
        }
    }
    return RTX_OK;
}

// One set of sub-batch scratch, every buffer sized for the class that needs most of it.
int alloc_scratch_set(rtx_index *ix, uint32_t k) {
    int rc;
    rtx_index::Scratch &sc = ix->sc[k];
    size_t n_kmers = 0, n_rows = 0, n_dmask = 0, n_counts = 0, n_hist = 0, n_urec = 0, b_max = 1, n_probscr = 0, r_max = 1;
    for (uint32_t c = 0; c < ix->n_cls; c++) {
        const rtx_index::BatchClass &kc = ix->cls[c];
        if (kc.side != (k == kSideSet)) continue;  // set 3 serves the side classes, the others the bulk
        const size_t B = kc.sub_batch;
        b_max = std::max(b_max, B);
        n_kmers = std::max(n_kmers, B * kc.kstride);
        n_rows = std::max(n_rows, B * kc.rstride);
        n_dmask = std::max(n_dmask, B * ix->ntiles * (kc.rstride / 64));
        // (a class that will prune with the records path starts with a fraction of the rows: begin_run enlarges the buffer if the run turns out otherwise)
        const bool diet = kc.will_prune && ix->rec_opt != 0u && ix->n_refs == ix->n_total && ix->n_bnd_local == ix->n_bnd;
        const size_t R = diet ? diet_rows(ix, (uint32_t)B) : B;
        n_counts = std::max(n_counts, ix->packed_opt && kc.planes <= 10 ? R * ix->npad * 5 / 8 : R * ix->npad);
        r_max = std::max(r_max, R);
        n_hist = std::max(n_hist, B * kc.hstride);
        n_urec = std::max(n_urec, ((B + 1u) / 2u) * 2u * kc.rstride);
        if (kc.huge) n_probscr = std::max(n_probscr, B * ((prob_table_lds_bytes(kc.tmax) + 7) / 8));
    }
    const size_t B = b_max;
    if ((rc = sc.d_kmers.alloc(n_kmers)) || (rc = sc.d_rows.alloc(n_rows)) || (rc = sc.d_dmask.alloc(n_dmask)) ||
        (rc = sc.d_nsparse.alloc(B * ix->ntiles)) || (rc = sc.d_srows.alloc(B * ix->ntiles * (kSegMaxSparseRows + 1))) ||
        (rc = sc.d_t.alloc(B)) || (rc = sc.d_nrows.alloc(B)) || (rc = sc.d_counts.alloc(n_counts)) ||
        (rc = sc.d_hist.alloc(n_hist)) || (rc = sc.d_table_z.alloc(n_hist)) ||
        (rc = sc.d_prefix.alloc(r_max * ix->n_bnd_local)) || (rc = sc.d_order.alloc(B)) ||
        (rc = sc.d_tilemax.alloc(B * ix->ntiles)) ||
        (rc = sc.d_urec.alloc(n_urec)) || (rc = sc.d_nu.alloc((B + 1u) / 2u)))
        return rc;
    if (n_probscr && (rc = ix->d_prob_scratch.alloc(std::max(n_probscr, ix->d_prob_scratch.n)))) return rc;
    if (ix->pruning() && ix->d_ubitmap.p && (ix->n_refs == ix->n_total || ix->shard_prune_opt)) {
        if ((rc = sc.d_tile_ub.alloc(B * ix->ntiles)) || (rc = sc.d_best_key.alloc(B)) || (rc = sc.d_prune_thr.alloc(B)) || (rc = sc.d_prune_i1.alloc(B)) || (rc = sc.d_best.alloc(B * kPruneBestWords)) || (rc = sc.d_live.alloc((B + 1u) * ((ix->ntiles + 31u) / 32u + 1u))) ||
            (rc = sc.d_items.alloc(((B + 1u) / 2u) * (ix->ntiles + 2u) + 9u)))
            return rc;
        if (ix->d_abitmap.p && (sc.d_heavy.alloc(B + 1u) || sc.d_heavy_items.alloc(((B + 1u) / 2u) * ix->u_ntiles + 9u))) { sc.d_heavy.release(); sc.d_heavy_items.release(); }
        // (a failed allocation of the fine pass's lists only switches the pass off: enqueue_hit tolerates a null pointer)
        if (ix->d_fbitmap.p && sc.d_fine_items.alloc(((B + 1u) / 2u) * ix->f_ntiles + 9u + ix->f_ntiles)) sc.d_fine_items.release();
        if (ix->rec_opt && ix->n_refs == ix->n_total) {  // the records path; without its buffers the run takes the dense epilogues
            const size_t slots = std::min<uint32_t>(ix->rec_opt, kRecMaxSlots);
            if (sc.d_rec_nslots.alloc(B) || sc.d_rec_slots.alloc(B * kRecMaxSlots) || sc.d_rec_cnt.alloc(B * kRecMaxSlots) ||
                sc.d_rec.alloc(B * slots * ix->rec_seg_len) || sc.d_cnt_row.alloc(B) || sc.d_cnt_cursor.alloc(4))
                sc.d_rec.release();
        }
    }
    return RTX_OK;
}

}  // namespace rtxi

extern "C" {

// Stages a batch in the input set that is NOT the current one: validation, bases packed two per byte into pinned memory (threads of
// the library's budget), offsets and exact-match ids beside them, asynchronous H2D on a stream of its own.  The batch that is running
// (or whose results are being downloaded) is not touched: rtx_raxtax stages chunk c + 1 while chunk c is classified.
int rtx_batch_prefetch(rtx_index *ix, uint64_t n_queries, const uint8_t *bases, const uint64_t *base_off,
                       const uint32_t *exact_ids, const uint64_t *exact_off) {
    int rc = bind(ix);
    if (rc) return rc;
    if (n_queries == 0 || !base_off || (!bases && base_off[n_queries])) {
        set_error("rtx_batch_upload: invalid argument");
        return RTX_ERR_INVALID;
    }
    rtx_index::Inputs &in = ix->in[ix->cur_in ^ 1u];
    if (!ix->h2d_stream) RTX_HIP(hipStreamCreateWithFlags(&ix->h2d_stream, hipStreamNonBlocking));
    if (!in.ready) RTX_HIP(hipEventCreateWithFlags(&in.ready, hipEventDisableTiming));
    if (in.recorded) RTX_HIP(hipEventSynchronize(in.ready));  // the last transfer out of this set's pinned buffers (long done, as a rule)
    in.staged = false;
    if (ix->ev_activated) RTX_HIP(hipStreamWaitEvent(ix->h2d_stream, ix->ev_activated, 0));  // the batch that read this set has run
    uint64_t max_len = 0;
    uint64_t cls_n[5] = {0, 0, 0, 0, 0}, cls_max[5] = {0, 0, 0, 0, 0};  // the length classes of the batch (length_class)
    {
        const uint64_t lim3 = class3_max_len();
        for (uint64_t q = 0; q < n_queries; q++) {
            if (base_off[q + 1] < base_off[q]) { set_error("base_off not monotone at query %llu", (unsigned long long)q); return RTX_ERR_INVALID; }
            const uint64_t len = base_off[q + 1] - base_off[q];
            max_len = std::max(max_len, len);
            const int c = len <= 262 ? 0 : (len <= 1030 ? 1 : (len <= kMidClassMaxT + 7 ? 2 : (len <= lim3 ? 3 : 4)));
            cls_n[c]++;
            cls_max[c] = std::max(cls_max[c], len);
        }
    }
    const uint64_t total = base_off[n_queries] - base_off[0];
    uint64_t n_exact = 0;
    if (exact_off) {
        if (exact_off[0] != 0) { set_error("exact_off[0] must be 0"); return RTX_ERR_INVALID; }
        n_exact = exact_off[n_queries];
        for (uint64_t q = 0; q < n_queries; q++)
            if (exact_off[q + 1] < exact_off[q]) { set_error("exact_off not monotone"); return RTX_ERR_INVALID; }
        if (n_exact && !exact_ids) { set_error("exact_ids is null"); return RTX_ERR_INVALID; }
        for (uint64_t i = 0; i < n_exact; i++)
            if (exact_ids[i] >= ix->n_total) { set_error("exact id %u out of range", exact_ids[i]); return RTX_ERR_INVALID; }
    }
    const uint64_t n_packed = (total + 1) / 2;
    if ((rc = in.h_packed.resize(total + 64)) || (rc = in.h_base_off.resize(n_queries + 1)) || (rc = in.d_packed.alloc(total + 64)) ||
        (rc = in.d_base_off.alloc(n_queries + 1)) || (rc = in.d_exact_off.alloc(n_queries + 1)) || (rc = in.d_exact_ids.alloc(n_exact + 1)))
        return rc;
    for (uint64_t q = 0; q <= n_queries; q++) in.h_base_off[q] = base_off[q] - base_off[0];
    // two bases per byte; a byte above 15 is no code of parser.rs:11-34 -- such a batch travels as it is (the kernels see the caller's bytes)
    in.packed = total == 0 || rtx::pack_nibbles_mt(bases + base_off[0], total, in.h_packed.data(), rtx::host_threads(8u));
    if (!in.packed) std::memcpy(in.h_packed.data(), bases + base_off[0], total);
    RTX_HIP(hipMemcpyAsync(in.d_packed.p, in.h_packed.data(), in.packed ? n_packed : total, hipMemcpyHostToDevice, ix->h2d_stream));
    RTX_HIP(hipMemcpyAsync(in.d_base_off.p, in.h_base_off.data(), (n_queries + 1) * 8, hipMemcpyHostToDevice, ix->h2d_stream));
    if (exact_off) {
        if ((rc = in.h_exact_off.resize(n_queries + 1)) || (rc = in.h_exact_ids.resize(n_exact + 1))) return rc;
        std::memcpy(in.h_exact_off.data(), exact_off, (n_queries + 1) * 8);
        if (n_exact) std::memcpy(in.h_exact_ids.data(), exact_ids, n_exact * 4);
        RTX_HIP(hipMemcpyAsync(in.d_exact_off.p, in.h_exact_off.data(), (n_queries + 1) * 8, hipMemcpyHostToDevice, ix->h2d_stream));
        if (n_exact) RTX_HIP(hipMemcpyAsync(in.d_exact_ids.p, in.h_exact_ids.data(), n_exact * 4, hipMemcpyHostToDevice, ix->h2d_stream));
    } else {
        RTX_HIP(hipMemsetAsync(in.d_exact_off.p, 0, (n_queries + 1) * 8, ix->h2d_stream));
    }
    RTX_HIP(hipEventRecord(in.ready, ix->h2d_stream));
    in.recorded = true;
    in.n_q = n_queries;
    in.total = total;
    in.max_len = max_len;
    for (int c = 0; c < 5; c++) { in.cls_n[c] = cls_n[c]; in.cls_max[c] = cls_max[c]; }
    in.n_exact = n_exact;
    in.has_exact = exact_off != nullptr;
    in.staged = true;
    return RTX_OK;
}

// The staged batch becomes the current one: the handle's stream waits for the transfer (the host does not), the workspace is sized
// for the batch (options that shape it are read here), the bases are unpacked.  The batch before it must have been downloaded.
int rtx_batch_activate(rtx_index *ix) {
    int rc = bind(ix);
    if (rc) return rc;
    rtx_index::Inputs &in = ix->in[ix->cur_in ^ 1u];
    if (!in.staged) { set_error("rtx_batch_activate without a staged batch (rtx_batch_prefetch)"); return RTX_ERR_STATE; }
    ix->uploaded = ix->ran = ix->synced = false;
    if ((rc = prepare_workspace(ix, in.n_q, in.cls_n, in.cls_max))) return rc;
    ix->sum_query_bytes = in.total;
    if ((rc = ix->d_bases.alloc(in.total + 64))) return rc;
    RTX_HIP(hipStreamWaitEvent(ix->stream, in.ready, 0));
    if (in.packed) {
        rtx::launch_unpack_nibbles(ix->stream, in.d_packed.p, ix->d_bases.p, in.total, in.total + 64);
    } else {
        RTX_HIP(hipMemcpyAsync(ix->d_bases.p, in.d_packed.p, in.total, hipMemcpyDeviceToDevice, ix->stream));
        RTX_HIP(hipMemsetAsync(ix->d_bases.p + in.total, 0, 64, ix->stream));
    }
    ix->dev_exact_used = !in.has_exact && ix->dev_exact_opt && ix->d_em_table.p && ix->n_refs == ix->n_total;
    if (ix->dev_exact_used && (rc = ix->d_exact_grp.alloc(in.n_q))) return rc;
    if (!ix->ev_activated) RTX_HIP(hipEventCreateWithFlags(&ix->ev_activated, hipEventDisableTiming));
    RTX_HIP(hipEventRecord(ix->ev_activated, ix->stream));
    in.staged = false;
    ix->cur_in ^= 1u;
    ix->uploaded = true;
    return RTX_OK;
}

int rtx_pack_bases(const uint8_t *bases, uint64_t n_bases, uint8_t *packed) {
    if ((!bases || !packed) && n_bases) { set_error("rtx_pack_bases: null argument"); return RTX_ERR_INVALID; }
    return rtx::pack_nibbles_mt(bases, n_bases, packed, rtx::host_threads(8u)) ? 1 : 0;
}

int rtx_batch_upload(rtx_index *ix, uint64_t n_queries, const uint8_t *bases, const uint64_t *base_off,
                     const uint32_t *exact_ids, const uint64_t *exact_off) {
    if (ix) ix->uploaded = ix->ran = ix->synced = false;
    int rc = rtx_batch_prefetch(ix, n_queries, bases, base_off, exact_ids, exact_off);
    return rc ? rc : rtx_batch_activate(ix);
}

int rtx_batch_run(rtx_index *ix, uint32_t flags) {
    int rc = bind(ix);
    if (rc) return rc;
    if (!ix->uploaded) { set_error("rtx_batch_run before rtx_batch_upload"); return RTX_ERR_STATE; }
    ix->last_flags = flags;
    ix->synced = false;
    rc = enqueue_batch(ix, flags);
    ix->ran = rc == RTX_OK;
    return rc;
}

int rtx_batch_sync(rtx_index *ix) {
    int rc = bind(ix);
    if (rc) return rc;
    if (!ix->ran) { set_error("rtx_batch_sync before rtx_batch_run"); return RTX_ERR_STATE; }
    if ((rc = settle_join(ix))) return rc;
    RTX_HIP(hipStreamSynchronize(ix->stream));
    ix->synced = true;
    return RTX_OK;
}

}  // extern "C"
