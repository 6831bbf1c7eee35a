// Stage times, work counters and the parity / debug taps (full vectors of the last sub-batch; not used on the fast path).
#include "rtx_index.hpp"

extern "C" {

int rtx_batch_stage_times(rtx_index *ix, float ms[RTX_NUM_STAGES], uint32_t launches[RTX_NUM_STAGES]) {
    int rc = bind(ix);
    if (rc) return rc;
    if (!ix->synced) { set_error("rtx_batch_stage_times: batch not synchronised"); return RTX_ERR_STATE; }
    for (int s = 0; s < RTX_NUM_STAGES; s++) { ms[s] = 0.f; launches[s] = 0; }
    for (uint32_t sb = 0; sb < ix->n_sub_last; sb++) {
        const rtx_index::BatchClass &kc = ix->cls[sb < ix->sub_cls.size() ? ix->sub_cls[sb] : 0u];  // which events its class recorded
        for (int s = 0; s < RTX_NUM_STAGES; s++) {
            if (s == RTX_STAGE_EXACT_MATCH) {
                if (sb != 0 || !ix->dev_exact_used || !ix->stage_timing) continue;  // one launch per run
            } else if (s == RTX_STAGE_ORDER) {
                if (sb != 0 || !ix->stage_timing) continue;  // once per run
            } else if (s == RTX_STAGE_PAIR_UNION) {
                if (!kc.pair || !ix->stage_timing) continue;
            } else if (s == RTX_STAGE_TILE_BOUNDS || s == RTX_STAGE_TILE_PRUNE ? !kc.prune : (s != RTX_STAGE_HIT_COUNT && !ix->stage_timing)) continue;  // events were not recorded
            float t = 0.f;
            RTX_HIP(hipEventElapsedTime(&t, ix->events[((size_t)sb * RTX_NUM_STAGES + s) * 2],
                                        ix->events[((size_t)sb * RTX_NUM_STAGES + s) * 2 + 1]));
            ms[s] += t;
            launches[s]++;
        }
    }
    return RTX_OK;
}

int rtx_batch_work(rtx_index *ix, uint64_t *sum_hits, uint64_t *sum_query_bytes, uint64_t *bitmap_bytes_read) {
    if (!ix) { set_error("null index handle"); return RTX_ERR_INVALID; }
    int rc = bind(ix);
    if (rc) return rc;
    if (!ix->ran) { set_error("rtx_batch_work before rtx_batch_run"); return RTX_ERR_STATE; }
    if ((rc = ix->h_hq.resize(ix->n_q)) || (rc = ix->h_nrows_all.resize(ix->n_q))) return rc;
    RTX_HIP(hipStreamSynchronize(ix->stream));
    RTX_HIP(hipMemcpy(ix->h_hq.data(), ix->d_hq.p, ix->n_q * 8, hipMemcpyDeviceToHost));
    RTX_HIP(hipMemcpy(ix->h_nrows_all.data(), ix->d_nrows_all.p, ix->n_q * 4, hipMemcpyDeviceToHost));
    uint64_t h = 0, b = 0;
    const uint64_t row_bytes = ((ix->n_refs + 7) / 8 + ix->ntiles - 1) / ix->ntiles;  // per dense segment (nrows counts segments)
    for (uint64_t q = 0; q < ix->n_q; q++) h += ix->h_hq[q];
    // per sub-batch: a class on the pair kernel loaded its rows once per pair of queries (the union rows every wave counted: d_group_rows),
    // the others once per query (kmer_extract's number of dense segments)
    const uint32_t n_sub = ix->n_sub_total;
    const size_t ng = (size_t)n_sub * ix->groups_per_sub;
    std::vector<uint32_t> gr;
    bool any_pair = false, any_prune = false;
    for (uint32_t c = 0; c < ix->n_cls; c++) { any_pair = any_pair || ix->cls[c].pair; any_prune = any_prune || ix->cls[c].prune; }
    if (any_pair) {
        gr.resize(2 * ng);
        RTX_HIP(hipMemcpy(gr.data(), ix->d_group_rows.p, gr.size() * 4, hipMemcpyDeviceToHost));
    }
    // (the two-level pass counts load instructions: a KiB each, several short rows)
    const uint64_t urow_bytes = ix->two_level_used ? 1024u : (ix->u_ntiles ? ((ix->u_nblocks + 7) / 8 + ix->u_ntiles - 1) / ix->u_ntiles : 0);
    for (uint32_t sb = 0; sb < n_sub; sb++) {
        const rtx_index::BatchClass &kc = ix->cls[ix->sub_cls[sb]];
        if (kc.pair) {
            for (size_t g = (size_t)sb * ix->groups_per_sub; g < (size_t)(sb + 1) * ix->groups_per_sub; g++) {
                b += (uint64_t)gr[g] * row_bytes;
                if (kc.prune) b += (uint64_t)gr[ng + g] * urow_bytes;  // + the rows of the union bitmap the bounds pass loaded (the same kernel)
            }
        } else {
            for (uint64_t pos = ix->sub_q0[sb]; pos < ix->sub_q0[sb] + ix->sub_nq[sb]; pos++) b += (uint64_t)ix->h_nrows_all[pos] * row_bytes;
        }
    }
    (void)any_prune;
    if (sum_hits) *sum_hits = h;
    if (sum_query_bytes) *sum_query_bytes = ix->sum_query_bytes;
    if (bitmap_bytes_read) *bitmap_bytes_read = b;
    return RTX_OK;
}

// bitmap_bytes_read of rtx_batch_work split by launch kind: the counting of the (live) tiles of the database, and the bounds pass of the
// tile pruning on the union bitmap (0 if the run did not prune)
int rtx_batch_work_split(rtx_index *ix, uint64_t *live_bytes, uint64_t *bounds_bytes) {
    uint64_t total = 0;
    int rc = rtx_batch_work(ix, nullptr, nullptr, &total);
    if (rc) return rc;
    uint64_t bounds = 0;
    {
        const uint32_t n_sub = ix->n_sub_total;
        const size_t ng = (size_t)n_sub * ix->groups_per_sub;
        bool any = false;
        for (uint32_t c = 0; c < ix->n_cls; c++) any = any || (ix->cls[c].pair && ix->cls[c].prune);
        if (any) {
            std::vector<uint32_t> gr(ng);
            RTX_HIP(hipMemcpy(gr.data(), ix->d_group_rows.p + ng, ng * 4, hipMemcpyDeviceToHost));
            const uint64_t urow_bytes = ix->two_level_used ? 1024u : ((ix->u_nblocks + 7) / 8 + ix->u_ntiles - 1) / ix->u_ntiles;
            for (uint32_t sb = 0; sb < n_sub; sb++)
                if (ix->cls[ix->sub_cls[sb]].prune)
                    for (size_t g = (size_t)sb * ix->groups_per_sub; g < (size_t)(sb + 1) * ix->groups_per_sub; g++) bounds += (uint64_t)gr[g] * urow_bytes;
        }
    }
    if (live_bytes) *live_bytes = total - bounds;
    if (bounds_bytes) *bounds_bytes = bounds;
    return RTX_OK;
}

int rtx_batch_prob_work(rtx_index *ix, uint64_t *sum_grid_points, uint64_t *sum_distinct_counts) {
    if (!ix) { set_error("null index handle"); return RTX_ERR_INVALID; }
    int rc = bind(ix);
    if (rc) return rc;
    if (!ix->ran) { set_error("rtx_batch_prob_work before rtx_batch_run"); return RTX_ERR_STATE; }
    RTX_HIP(hipStreamSynchronize(ix->stream));
    std::vector<uint32_t> nd(ix->n_q), tt(ix->n_q);
    RTX_HIP(hipMemcpy(nd.data(), ix->d_ndist.p, ix->n_q * 4, hipMemcpyDeviceToHost));
    RTX_HIP(hipMemcpy(tt.data(), ix->d_t_all.p, ix->n_q * 4, hipMemcpyDeviceToHost));
    uint64_t g = 0, d = 0;
    for (uint64_t q = 0; q < ix->n_q; q++) {
        g += (uint64_t)nd[q] * (tt[q] / 2 + 1);  // D_q (n_q + 1), n_q = t_q / 2 (raxtax.rs:57)
        d += nd[q];
    }
    if (sum_grid_points) *sum_grid_points = g;
    if (sum_distinct_counts) *sum_distinct_counts = d;
    return RTX_OK;
}

// ---- debug taps -------------------------------------------------------------------------
static int debug_recount_full(rtx_index *ix);
static int debug_slot_as_run(rtx_index *ix, uint64_t query, uint32_t *slot) {  // the scratch as the run left it (no recount)
    int rc = bind(ix);
    if (rc) return rc;
    if (!ix->synced) { set_error("debug tap: batch not synchronised"); return RTX_ERR_STATE; }
    const uint64_t last0 = ix->n_sub_total && ix->sub_q0.size() == ix->n_sub_total ? ix->sub_q0[ix->n_sub_total - 1] : 0;  // the last sub-batch of the run (of its last length class)
    if (query >= ix->n_q) { set_error("debug tap: query %llu out of range", (unsigned long long)query); return RTX_ERR_INVALID; }
    const uint64_t pos = ix->h_inv_now()[query];  // position in the processing order (valid once the stream is synchronised)
    if (pos < last0) { set_error("debug tap: query %llu not in the last sub-batch", (unsigned long long)query); return RTX_ERR_INVALID; }
    *slot = (uint32_t)(pos - last0);
    return RTX_OK;
}
static int debug_slot(rtx_index *ix, uint64_t query, uint32_t *slot) {
    int rc = debug_slot_as_run(ix, query, slot);
    return rc ? rc : debug_recount_full(ix);
}

// After a pruned run the scratch of the last sub-batch holds the counts of the live tiles only and a histogram with the
// uncounted references lumped into bin 0: the taps promise the full vectors, so the sub-batch is counted again in full
// (k-mers, hit counts, histogram, probability table; the result rows of the run are not touched).
static int debug_recount_full(rtx_index *ix) {
    if (!ix->prune_used || ix->dbg_full) return RTX_OK;
    const uint32_t n_sub = ix->n_sub_total;
    SubBatch b = sub_batch_of(ix, n_sub - 1, false);
    b.set = ix->last_set;
    RTX_HIP(hipStreamSynchronize(ix->stream));
    int rc = ensure_full_counts(ix, ix->sc[b.set]);  // the recount writes the counts of EVERY query of the sub-batch (the run's buffer may be on its diet)
    if (rc) return rc;
    ix->dbg_full_run = true;
    rc = enqueue_kmer(ix, b, ix->stream);
    if (!rc) rc = enqueue_hit(ix, b, ix->last_flags, ix->stream);
    if (!rc) rc = enqueue_prob_prefix(ix, b, false, true);
    ix->dbg_full_run = false;
    if (rc) return rc;
    RTX_HIP(hipStreamSynchronize(ix->stream));
    ix->dbg_full = true;
    return RTX_OK;
}

// u16 counts of one slot of the last sub-batch on the device (unpacked into a scratch row if they are packed)
static int debug_counts_u16(rtx_index *ix, uint32_t slot, const uint16_t **out) {
    rtx_index::Scratch &sc = ix->sc[ix->last_set];
    if (ix->diet_used && !ix->dbg_full && sc.d_cnt_row.p) {  // as the run left them: the query's row of the counts buffer (HitParams::cnt_row)
        uint32_t row = 0;
        RTX_HIP(hipMemcpy(&row, sc.d_cnt_row.p + slot, 4, hipMemcpyDeviceToHost));
        if (row == 0xFFFFFFFFu) { set_error("debug tap: the query holds no row of the counts buffer (records path)"); return RTX_ERR_STATE; }
        slot = row;
    }
    if (!ix->packed()) { *out = sc.d_counts.p + (size_t)slot * ix->npad; return RTX_OK; }
    int rc = ix->d_counts_dbg.alloc(ix->npad);
    if (rc) return rc;
    launch_counts_unpack(ix->stream, counts_lo(ix, sc) + (size_t)slot * ix->npad, counts_hi(ix, sc) + (size_t)slot * (ix->npad >> 3), ix->npad,
                         ix->d_counts_dbg.p);
    RTX_HIP(hipStreamSynchronize(ix->stream));
    *out = ix->d_counts_dbg.p;
    return RTX_OK;
}

int rtx_debug_kmers(rtx_index *ix, uint64_t query, uint16_t *kmers, uint32_t *t) {
    uint32_t slot;
    int rc = debug_slot(ix, query, &slot);
    if (rc) return rc;
    uint32_t tt = 0;
    RTX_HIP(hipMemcpy(&tt, ix->sc[ix->last_set].d_t.p + slot, 4, hipMemcpyDeviceToHost));
    if (t) *t = tt;
    if (kmers && tt) RTX_HIP(hipMemcpy(kmers, ix->sc[ix->last_set].d_kmers.p + (size_t)slot * ix->kstride, std::min(tt, ix->kstride) * 2, hipMemcpyDeviceToHost));
    return RTX_OK;
}

int rtx_debug_hit_counts(rtx_index *ix, uint64_t query, uint16_t *counts) {
    uint32_t slot;
    int rc = debug_slot(ix, query, &slot);
    if (rc) return rc;
    const uint16_t *src = nullptr;
    if ((rc = debug_counts_u16(ix, slot, &src))) return rc;
    RTX_HIP(hipMemcpy(counts, src, ix->n_refs * 2, hipMemcpyDeviceToHost));
    return RTX_OK;
}

int rtx_debug_prob_table(rtx_index *ix, uint64_t query, double *table_over_z, double *z) {
    uint32_t slot;
    int rc = debug_slot(ix, query, &slot);
    if (rc) return rc;
    uint32_t tt = 0;
    RTX_HIP(hipMemcpy(&tt, ix->sc[ix->last_set].d_t.p + slot, 4, hipMemcpyDeviceToHost));
    std::vector<uint32_t> hist(tt + 1);
    RTX_HIP(hipMemcpy(hist.data(), ix->sc[ix->last_set].d_hist.p + (size_t)slot * ix->hstride, (tt + 1) * 4, hipMemcpyDeviceToHost));
    RTX_HIP(hipMemcpy(table_over_z, ix->sc[ix->last_set].d_table_z.p + (size_t)slot * ix->hstride, (tt + 1) * 8, hipMemcpyDeviceToHost));
    for (uint32_t m = 0; m <= tt; m++)
        if (!hist[m]) table_over_z[m] = 0.0;  // entries of absent counts are never written
    if (z) RTX_HIP(hipMemcpy(z, ix->d_z.p + ix->h_inv_now()[query], 8, hipMemcpyDeviceToHost));
    return RTX_OK;
}

// table / Z of a query as the PRUNED run computed it (entries of the counts up to the threshold are 0), its Z and its threshold
int rtx_debug_pruned_prob_table(rtx_index *ix, uint64_t query, double *table_over_z, double *z, uint32_t *threshold) {
    uint32_t slot;
    int rc = debug_slot_as_run(ix, query, &slot);
    if (rc) return rc;
    if (!ix->prune_used) { set_error("rtx_debug_pruned_prob_table: the last run did not prune"); return RTX_ERR_STATE; }
    if (ix->dbg_full) { set_error("rtx_debug_pruned_prob_table: another tap has recounted the sub-batch in full"); return RTX_ERR_STATE; }
    rtx_index::Scratch &sc = ix->sc[ix->last_set];
    uint32_t tt = 0;
    uint16_t thr = 0;
    RTX_HIP(hipMemcpy(&tt, sc.d_t.p + slot, 4, hipMemcpyDeviceToHost));
    RTX_HIP(hipMemcpy(&thr, sc.d_prune_thr.p + slot, 2, hipMemcpyDeviceToHost));
    std::vector<uint32_t> hist(tt + 1);
    RTX_HIP(hipMemcpy(hist.data(), sc.d_hist.p + (size_t)slot * ix->hstride, (tt + 1) * 4, hipMemcpyDeviceToHost));
    RTX_HIP(hipMemcpy(table_over_z, sc.d_table_z.p + (size_t)slot * ix->hstride, (tt + 1) * 8, hipMemcpyDeviceToHost));
    for (uint32_t m = 0; m <= tt; m++)
        if (!hist[m] || m <= thr) table_over_z[m] = 0.0;  // entries of absent counts are never written; up to the threshold: 0 by construction
    if (z) RTX_HIP(hipMemcpy(z, ix->d_z.p + ix->h_inv_now()[query], 8, hipMemcpyDeviceToHost));
    if (threshold) *threshold = thr;
    return RTX_OK;
}

// The last sub-batch exactly as the run left it -- no recount: the counts hit_count wrote for the tiles it visited, which tiles
// those were, the histogram as prune_kernel (bin 0: the references never counted) and hit_count (every counted reference) left it,
// the query's threshold and i* + 1.  The parity tests hold THIS against the oracle (the recounting taps prove the unpruned kernel).
int rtx_debug_run_counts(rtx_index *ix, uint64_t query, uint16_t *counts, uint8_t *tile_live, uint32_t *hist, uint32_t *threshold,
                         uint32_t *i1) {
    uint32_t slot;
    int rc = debug_slot_as_run(ix, query, &slot);
    if (rc) return rc;
    if (ix->dbg_full) { set_error("rtx_debug_run_counts: another tap has recounted the sub-batch in full"); return RTX_ERR_STATE; }
    rtx_index::Scratch &sc = ix->sc[ix->last_set];
    const uint32_t nt = ix->ntiles;
    std::vector<uint8_t> live(nt, 1);
    uint16_t thr = 0, i1v = 0;
    if (ix->prune_used) {
        const uint32_t lw = (nt + 31u) / 32u + 1u;
        std::vector<uint32_t> words(lw);
        RTX_HIP(hipMemcpy(words.data(), sc.d_live.p + (size_t)slot * lw, lw * 4, hipMemcpyDeviceToHost));
        for (uint32_t T = 0; T < nt; T++) live[T] = (uint8_t)((words[T >> 5] >> (T & 31u)) & 1u);
        RTX_HIP(hipMemcpy(&thr, sc.d_prune_thr.p + slot, 2, hipMemcpyDeviceToHost));
        RTX_HIP(hipMemcpy(&i1v, sc.d_prune_i1.p + slot, 2, hipMemcpyDeviceToHost));
    }
    uint16_t n_seg = 0;  // > 0: the query took the records path -- its counts are the records of its segments
    if (ix->prune_used && ix->rec_used && sc.d_rec_nslots.p) RTX_HIP(hipMemcpy(&n_seg, sc.d_rec_nslots.p + slot, 2, hipMemcpyDeviceToHost));
    if (counts && n_seg) {
        // visited tiles: the count of every reference above the threshold, 0 for the others (the run never wrote those); unvisited: 0xFFFF
        const uint32_t stride = std::min<uint32_t>(ix->rec_opt, kRecMaxSlots);
        uint16_t tiles[kRecMaxSlots];
        uint32_t cnts[kRecMaxSlots];
        RTX_HIP(hipMemcpy(tiles, sc.d_rec_slots.p + (size_t)slot * kRecMaxSlots, sizeof tiles, hipMemcpyDeviceToHost));
        RTX_HIP(hipMemcpy(cnts, sc.d_rec_cnt.p + (size_t)slot * kRecMaxSlots, sizeof cnts, hipMemcpyDeviceToHost));
        for (uint64_t r = 0; r < ix->n_refs; r++) counts[r] = live[r >> 13] ? 0u : 0xFFFFu;
        std::vector<uint32_t> seg(ix->rec_seg_len);
        for (uint32_t k = 0; k < n_seg && k < stride; k++) {
            const uint32_t c = std::min<uint32_t>(cnts[k], ix->rec_seg_len);
            if (!c) continue;
            RTX_HIP(hipMemcpy(seg.data(), sc.d_rec.p + ((size_t)slot * stride + k) * ix->rec_seg_len, (size_t)c * 4, hipMemcpyDeviceToHost));
            uint32_t prev = 0;
            for (uint32_t i = 0; i < c; i++) {
                const uint32_t rl = seg[i] & 8191u;
                const uint64_t r = (uint64_t)tiles[k] * 8192u + rl;
                if (r >= ix->n_refs || (i && rl <= prev) || !live[tiles[k]]) { set_error("rtx_debug_run_counts: malformed record %u of segment %u of query %llu", i, k, (unsigned long long)query); return RTX_ERR_STATE; }
                counts[r] = (uint16_t)(seg[i] >> 13);
                prev = rl;
            }
        }
    } else if (counts) {
        const uint16_t *src = nullptr;
        if ((rc = debug_counts_u16(ix, slot, &src))) return rc;
        RTX_HIP(hipMemcpy(counts, src, ix->n_refs * 2, hipMemcpyDeviceToHost));
        for (uint32_t T = 0; T < nt; T++)
            if (!live[T]) {  // never written by this run: whatever an earlier sub-batch left there
                const uint64_t lo = (uint64_t)T * 8192u, hi = std::min<uint64_t>(lo + 8192u, ix->n_refs);
                for (uint64_t r = lo; r < hi; r++) counts[r] = 0xFFFFu;
            }
    }
    if (tile_live) std::memcpy(tile_live, live.data(), nt);
    if (hist) {
        uint32_t tt = 0;
        RTX_HIP(hipMemcpy(&tt, sc.d_t.p + slot, 4, hipMemcpyDeviceToHost));
        RTX_HIP(hipMemcpy(hist, sc.d_hist.p + (size_t)slot * ix->hstride, (size_t)(tt + 1) * 4, hipMemcpyDeviceToHost));
    }
    if (threshold) *threshold = thr;
    if (i1) *i1 = i1v;
    return RTX_OK;
}

// number of record segments of a query of the last sub-batch (RTX_OPT_RECORDS): 0 = it took the dense epilogues
int rtx_debug_run_mode(rtx_index *ix, uint64_t query, uint32_t *n_segments) {
    uint32_t slot;
    int rc = debug_slot_as_run(ix, query, &slot);
    if (rc) return rc;
    if (!n_segments) { set_error("null argument"); return RTX_ERR_INVALID; }
    uint16_t n_seg = 0;
    rtx_index::Scratch &sc = ix->sc[ix->last_set];
    if (ix->prune_used && ix->rec_used && sc.d_rec_nslots.p) RTX_HIP(hipMemcpy(&n_seg, sc.d_rec_nslots.p + slot, 2, hipMemcpyDeviceToHost));
    *n_segments = n_seg;
    return RTX_OK;
}

// prune_kernel's view of a query of the last sub-batch (RTX_OPT_DEBUG_TAPS = 1 before the run): kPruneDetailWords words,
// PruneParams::detail
int rtx_debug_prune_detail(rtx_index *ix, uint64_t query, uint32_t *out) {
    uint32_t slot;
    int rc = debug_slot_as_run(ix, query, &slot);
    if (rc) return rc;
    if (!out) { set_error("null argument"); return RTX_ERR_INVALID; }
    if (!ix->prune_used || !ix->debug_taps || ix->d_prune_detail.n < (size_t)(slot + 1) * kPruneDetailWords) {
        set_error("rtx_debug_prune_detail: the last run did not prune, or RTX_OPT_DEBUG_TAPS was off");
        return RTX_ERR_STATE;
    }
    RTX_HIP(hipMemcpy(out, ix->d_prune_detail.p + (size_t)slot * kPruneDetailWords, kPruneDetailWords * 4, hipMemcpyDeviceToHost));
    return RTX_OK;
}

int rtx_debug_tile_bounds(rtx_index *ix, uint64_t query, uint16_t *tile_ub) {
    uint32_t slot;
    int rc = debug_slot_as_run(ix, query, &slot);
    if (rc) return rc;
    if (!tile_ub) { set_error("null argument"); return RTX_ERR_INVALID; }
    rtx_index::Scratch &sc = ix->sc[ix->last_set];
    if (!ix->prune_used || sc.d_tile_ub.n < (size_t)(slot + 1) * ix->ntiles) { set_error("rtx_debug_tile_bounds: the last run did not prune"); return RTX_ERR_STATE; }
    RTX_HIP(hipMemcpy(tile_ub, sc.d_tile_ub.p + (size_t)slot * ix->ntiles, (size_t)ix->ntiles * 2, hipMemcpyDeviceToHost));
    return RTX_OK;
}

int rtx_debug_prune_stats(rtx_index *ix, uint64_t *out) {
    if (!ix || !out) { set_error("null argument"); return RTX_ERR_INVALID; }
    std::memset(out, 0, 128);
    if (!ix->any_prune || !ix->d_prune_stats.p) return RTX_OK;
    RTX_HIP(hipStreamSynchronize(ix->stream));
    unsigned long long h[kPruneStatCopies * 32];
    RTX_HIP(hipMemcpy(h, ix->d_prune_stats.p, sizeof(h), hipMemcpyDeviceToHost));
    for (uint32_t c = 0; c < kPruneStatCopies; c++) {
        for (uint32_t k = 0; k < 8; k++) out[k] += h[c * 8 + k];
        for (uint32_t k = 0; k < 2; k++) out[8 + k] += h[(kPruneStatCopies + c) * 8 + k];
        for (uint32_t k = 0; k < 3; k++) out[10 + k] += h[(2 * kPruneStatCopies + c) * 8 + k];  // the fine bounds pass
        out[13] += h[(3 * kPruneStatCopies + c) * 8 + 0];  // the records path: records,
        out[14] += h[(3 * kPruneStatCopies + c) * 8 + 1];  // queries,
        out[15] += h[(3 * kPruneStatCopies + c) * 8 + 3];  // queries whose boundary entries did not fit LDS (slow path)
    }
    return RTX_OK;
}

int rtx_batch_sub_batch(const rtx_index *ix, uint32_t *sub_batch, uint32_t *n_sub) {
    if (!ix) { set_error("null index handle"); return RTX_ERR_INVALID; }
    if (!ix->uploaded || ix->sub_batch == 0 || ix->n_cls == 0) { set_error("rtx_batch_sub_batch: no batch has been uploaded"); return RTX_ERR_STATE; }
    if (sub_batch) *sub_batch = ix->cls[ix->n_cls - 1].sub_batch;
    if (n_sub) *n_sub = ix->n_sub_total;
    return RTX_OK;
}

// the last sub-batch of the uploaded batch (the one the taps can read): positions [first, first + n) of the processing order
int rtx_batch_last_sub_batch(const rtx_index *ix, uint64_t *first, uint32_t *n) {
    if (!ix) { set_error("null index handle"); return RTX_ERR_INVALID; }
    if (!ix->uploaded || ix->n_sub_total == 0 || ix->sub_q0.size() != ix->n_sub_total) { set_error("rtx_batch_last_sub_batch: no batch has been uploaded"); return RTX_ERR_STATE; }
    if (first) *first = ix->sub_q0[ix->n_sub_total - 1];
    if (n) *n = ix->sub_nq[ix->n_sub_total - 1];
    return RTX_OK;
}

// length classes of the uploaded batch (rtx_index.hpp: BatchClass): out[c] = {queries, longest query, sub-batch size, bit planes} for c < *n_classes (at most 4)
int rtx_batch_classes(const rtx_index *ix, uint32_t *n_classes, uint64_t out[20]) {
    if (!ix || !n_classes || !out) { set_error("null argument"); return RTX_ERR_INVALID; }
    if (!ix->uploaded) { set_error("rtx_batch_classes: no batch has been uploaded"); return RTX_ERR_STATE; }
    *n_classes = ix->n_cls;
    for (uint32_t c = 0; c < ix->n_cls && c < 5u; c++) {
        out[c * 4 + 0] = ix->cls[c].n;
        out[c * 4 + 1] = ix->cls[c].max_len;
        out[c * 4 + 2] = ix->cls[c].sub_batch;
        out[c * 4 + 3] = (uint64_t)ix->cls[c].planes | (ix->cls[c].use_tables ? 1ull << 8 : 0) | (ix->cls[c].pair ? 1ull << 9 : 0) | (ix->cls[c].prune ? 1ull << 10 : 0) |
                         (ix->cls[c].rec ? 1ull << 11 : 0) | (ix->cls[c].huge ? 1ull << 12 : 0);
    }
    return RTX_OK;
}

int rtx_debug_order(rtx_index *ix, uint32_t *perm) {
    if (!ix || !perm) { set_error("null argument"); return RTX_ERR_INVALID; }
    if (!ix->ran || ix->h_perm_now().n < ix->n_q) { set_error("rtx_debug_order: no batch has been run"); return RTX_ERR_STATE; }
    RTX_HIP(hipStreamSynchronize(ix->stream));
    std::memcpy(perm, ix->h_perm_now().data(), (size_t)ix->n_q * 4);
    return RTX_OK;
}

int rtx_debug_probs(rtx_index *ix, uint64_t query, double *probs) {
    uint32_t slot;
    int rc = debug_slot(ix, query, &slot);
    if (rc) return rc;
    if ((rc = ix->d_probs_dbg.alloc(ix->n_refs))) return rc;
    const uint16_t *src = nullptr;
    if ((rc = debug_counts_u16(ix, slot, &src))) return rc;
    launch_probs_expand(ix->stream, src, ix->sc[ix->last_set].d_table_z.p + (size_t)slot * ix->hstride, ix->n_refs, ix->d_probs_dbg.p);
    RTX_HIP(hipStreamSynchronize(ix->stream));
    RTX_HIP(hipMemcpy(probs, ix->d_probs_dbg.p, ix->n_refs * 8, hipMemcpyDeviceToHost));
    return RTX_OK;
}

// Runs taxon_prefix + lineage_walk + host finalisation on a caller-supplied probability vector
// (Lineage::new(label, tree, probs).evaluate(), lineage.rs:61-112), so that the reference's
// lineage KATs (lineage.rs:192-334) pin the device walk directly.  n_refs <= 65535.
int rtx_debug_evaluate(rtx_index *ix, const double *probs, rtx_result_view *out) {
    int rc = bind(ix);
    if (rc) return rc;
    if (!probs || !out) { set_error("null argument"); return RTX_ERR_INVALID; }
    const uint64_t N = ix->n_refs;
    if (N > 65535) { set_error("rtx_debug_evaluate supports at most 65535 references"); return RTX_ERR_INVALID; }
    ix->uploaded = ix->ran = ix->synced = false;
    if ((rc = prepare_workspace_single(ix, 1, std::max<uint64_t>(N, 8), 0))) return rc;
    ix->last_set = 0;
    ix->sum_query_bytes = 0;
    ix->stream_dl = false;
    if ((rc = order_batch(ix, false))) return rc;
    if ((rc = ix->sc[0].d_counts.alloc(ix->npad))) return rc;  // u16 format here whatever the batch format would be
    std::vector<uint16_t> counts(ix->npad, 0);
    for (uint64_t r = 0; r < N; r++) counts[r] = (uint16_t)r;  // count_r = r, table[r] = probs[r]
    double gs = 0.0;
    for (uint64_t r = 0; r < N; r++) { const double d = probs[r] - 1.0 / (double)N; gs += d * d; }
    gs = std::sqrt(gs);
    const uint8_t ok = RTX_Q_OK;
    hipStream_t s = ix->stream;
    RTX_HIP(hipMemcpy(ix->sc[ix->last_set].d_counts.p, counts.data(), ix->npad * 2, hipMemcpyHostToDevice));
    RTX_HIP(hipMemcpy(ix->sc[ix->last_set].d_table_z.p, probs, N * 8, hipMemcpyHostToDevice));
    RTX_HIP(hipMemcpy(ix->d_status.p, &ok, 1, hipMemcpyHostToDevice));
    RTX_HIP(hipMemcpy(ix->d_gs.p, &gs, 8, hipMemcpyHostToDevice));
    {   // taxon_prefix scans table[0 .. t] of the slot for the smallest count with a probability: the pseudo-query's
        // "counts" are 0 .. N-1 (the slot's t was left to whatever the allocation held: an out-of-bounds scan)
        const uint32_t t_pseudo = (uint32_t)N - 1u;
        RTX_HIP(hipMemcpy(ix->sc[ix->last_set].d_t.p, &t_pseudo, 4, hipMemcpyHostToDevice));
    }
    RTX_HIP(hipMemset(ix->d_t_all.p, 0, 4));
    RTX_HIP(hipMemset(ix->d_nrows_all.p, 0, 4));
    RTX_HIP(hipMemset(ix->d_hq.p, 0, 8));
    RTX_HIP(hipMemset(ix->d_z.p, 0, 8));
    RTX_HIP(hipMemset(ix->d_cursor.p, 0, 16));  // both cursors: the download reads the side classes' one as well (left to the allocation it sized the host arrays by garbage)
    RTX_HIP(hipMemset(ix->d_flags.p, 0, 4));
    RTX_HIP(hipMemset(ix->d_fin_cursor.p, 0, 16));
    PrefixParams fp{};
    fp.status = ix->d_status.p;
    fp.t = ix->sc[ix->last_set].d_t.p;
    fp.tz_in_lds = 0;  // the pseudo-query's "counts" index the probability vector directly
    fp.q0 = 0;
    fp.counts = ix->sc[ix->last_set].d_counts.p;
    fp.npad = ix->npad;
    fp.table_z = ix->sc[ix->last_set].d_table_z.p;
    fp.hstride = ix->hstride;
    fp.n_refs = N;
    fp.bnd_bits = ix->d_bnd_bits.p;
    fp.bnd_rank = ix->d_bnd_rank.p;
    fp.prefix = ix->sc[ix->last_set].d_prefix.p;
    fp.n_bnd = ix->n_bnd_local;
    launch_taxon_prefix(s, fp, 1);
    WalkParams wp{};
    wp.status = ix->d_status.p;
    wp.q0 = 0;
    wp.prefix = ix->sc[ix->last_set].d_prefix.p;
    wp.n_bnd = ix->n_bnd;
    wp.rec = ix->d_noderec.p;
    wp.arena = ix->d_arena.p;
    wp.arena_cap = ix->arena_cap;
    wp.arena_cursor = ix->d_cursor.p;
    wp.n_rows = ix->d_n_rows.p;
    wp.row_start = ix->d_row_start.p;
    wp.flags_out = ix->d_flags.p;
    launch_lineage_walk(s, wp, 1);
    {
        SubBatch b{};
        b.sb = 0; b.nq = 1; b.set = 0; b.q0 = 0; b.s = s;
        if ((rc = enqueue_finalise(ix, b, s))) return rc;
    }
    RTX_HIP(hipGetLastError());
    ix->n_sub_last = 0;
    ix->ran = true;
    ix->last_flags = 0;
    return rtx_batch_download(ix, out);
}

}  // extern "C"
