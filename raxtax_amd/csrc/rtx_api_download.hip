// Result download: the device finalises the rows of every sub-batch (rtx_finalise.hip: sort lineage.rs:91-93, expected vectors and local
// signal lineage.rs:95-102, final layout); the rows of finished sub-batches are copied into the page-locked arrays of the view while
// later sub-batches still run -- the host copies, it computes nothing; exact matches as the device found them; rtx_classify_batch.
#include "rtx_index.hpp"

namespace rtxi {

// The per-node tables of finalise_kernel (rtx_finalise.hip): depth, begin of the node's range (the lineage a row reports, lineage.rs:105),
// and the expected side of the local signal (lineage.rs:95-98,137-139): the level it starts at and per level the expected share
// |range of the ancestor| / N over the sum of those shares (rtx_math.hpp: fin_node_expected) -- everything of a row's local signal
// that depends on the node alone.
int node_tables(rtx_index *ix) {
    const FlatNodes &f = ix->nodes;
    const uint32_t D = std::max(1u, f.max_depth), nn = f.size();
    ix->fin_D = D;
    std::vector<uint8_t> depth(nn), sig0(nn);
    std::vector<double> eb((size_t)nn * D, 0.0);
    std::vector<uint32_t> size(D);
    const double N = (double)ix->n_total;
    for (uint32_t v = 0; v < nn; v++) {
        depth[v] = (uint8_t)f.depth[v];
        uint32_t anc = v;
        for (int d = (int)f.depth[v] - 1; d >= 0; d--) {
            size[d] = f.end[anc] - f.begin[anc];
            anc = f.parent[anc];
        }
        sig0[v] = (uint8_t)fin_node_expected(size.data(), f.depth[v], N, eb.data() + (size_t)v * D);
    }
    int rc;
    if ((rc = ix->d_node_depth.alloc(nn)) || (rc = ix->d_node_sig0.alloc(nn)) || (rc = ix->d_node_begin.alloc(nn)) || (rc = ix->d_node_eb.alloc((size_t)nn * D))) return rc;
    RTX_HIP(hipMemcpy(ix->d_node_depth.p, depth.data(), nn, hipMemcpyHostToDevice));
    RTX_HIP(hipMemcpy(ix->d_node_sig0.p, sig0.data(), nn, hipMemcpyHostToDevice));
    RTX_HIP(hipMemcpy(ix->d_node_begin.p, f.begin.data(), (size_t)nn * 4, hipMemcpyHostToDevice));
    RTX_HIP(hipMemcpy(ix->d_node_eb.p, eb.data(), (size_t)nn * D * 8, hipMemcpyHostToDevice));
    return RTX_OK;
}

// Host arrays of a download: the per-query fields for nq queries, the row arrays for at least `rows` rows with the first `keep` preserved
static int size_host_results(rtx_index *ix, rtx_index::HostRes &hr, uint64_t nq, uint64_t rows, uint64_t keep) {
    int rc;
    if ((rc = hr.h_t.resize(nq)) || (rc = hr.h_status.resize(nq)) || (rc = hr.h_gs.resize(nq)) || (rc = hr.v_row_begin.resize(nq)) ||
        (rc = hr.v_row_count.resize(nq)))
        return rc;
    const uint64_t D = ix->fin_D;
    if ((rc = hr.v_row_lineage.grow_keep(rows, keep)) || (rc = hr.v_row_node.grow_keep(rows, keep)) || (rc = hr.v_row_depth.grow_keep(rows, keep)) ||
        (rc = hr.v_row_depth8.grow_keep(rows, keep)) || (rc = hr.v_row_local.grow_keep(rows, keep)) || (rc = hr.v_row_conf.grow_keep(rows * D, keep * D)) ||
        (rc = hr.v_row_hund.grow_keep(rows * D, keep * D)))
        return rc;
    return RTX_OK;
}

// D2H of the final rows [r0, r1) on stream cs (asynchronous): the device laid them out as the view wants them
static int copy_rows(rtx_index *ix, rtx_index::HostRes &hr, uint64_t r0, uint64_t r1, hipStream_t cs) {
    if (r1 <= r0) return RTX_OK;
    const uint64_t n = r1 - r0, D = ix->fin_D;
    RTX_HIP(hipMemcpyAsync(hr.v_row_lineage.data() + r0, ix->d_fin_lineage.p + r0, n * 4, hipMemcpyDeviceToHost, cs));
    RTX_HIP(hipMemcpyAsync(hr.v_row_node.data() + r0, ix->d_fin_node.p + r0, n * 4, hipMemcpyDeviceToHost, cs));
    RTX_HIP(hipMemcpyAsync(hr.v_row_depth.data() + r0, ix->d_fin_depth.p + r0, n * 4, hipMemcpyDeviceToHost, cs));
    RTX_HIP(hipMemcpyAsync(hr.v_row_depth8.data() + r0, ix->d_fin_depth8.p + r0, n, hipMemcpyDeviceToHost, cs));
    RTX_HIP(hipMemcpyAsync(hr.v_row_local.data() + r0, ix->d_fin_local.p + r0, n * 8, hipMemcpyDeviceToHost, cs));
    RTX_HIP(hipMemcpyAsync(hr.v_row_conf.data() + r0 * D, ix->d_fin_conf.p + r0 * D, n * D * 8, hipMemcpyDeviceToHost, cs));
    RTX_HIP(hipMemcpyAsync(hr.v_row_hund.data() + r0 * D, ix->d_fin_hund.p + r0 * D, n * D, hipMemcpyDeviceToHost, cs));
    return RTX_OK;
}
// ... and of the per-query fields (input order: complete when the last sub-batch has been finalised)
static int copy_queries(rtx_index *ix, rtx_index::HostRes &hr, uint64_t nq, hipStream_t cs) {
    RTX_HIP(hipMemcpyAsync(hr.h_t.data(), ix->d_fin_t.p, nq * 4, hipMemcpyDeviceToHost, cs));
    RTX_HIP(hipMemcpyAsync(hr.h_status.data(), ix->d_fin_status.p, nq, hipMemcpyDeviceToHost, cs));
    RTX_HIP(hipMemcpyAsync(hr.h_gs.data(), ix->d_fin_gs.p, nq * 8, hipMemcpyDeviceToHost, cs));
    RTX_HIP(hipMemcpyAsync(hr.v_row_begin.data(), ix->d_fin_row_begin.p, nq * 8, hipMemcpyDeviceToHost, cs));
    RTX_HIP(hipMemcpyAsync(hr.v_row_count.data(), ix->d_fin_row_count.p, nq * 4, hipMemcpyDeviceToHost, cs));
    return RTX_OK;
}

// Streamed download: while later sub-batches are still running, the records of every finished one are copied
// (copy_stream) and finalised on the calling thread, so that only the last sub-batch is left once the device is
// done.  *done = false: not applicable (batch already complete: the bulk path with its threads is faster) or the
// arena overflowed (the bulk path repeats the run).
// the exact matches the device found belong to the download (same alternation as the result sets)
static int fetch_exact_groups(rtx_index *ix, uint64_t nq) {
    rtx_index::HostExact &hx = ix->host_exact[ix->res_set];
    hx.valid = hx.csr_valid = false;
    if (ix->dev_exact_used) {
        hx.grp.resize(nq);
        if (ix->ev_exact) RTX_HIP(hipEventSynchronize(ix->ev_exact));  // the lookup ran at the head of the batch (a re-run: of the re-run)
        RTX_HIP(hipMemcpy(hx.grp.data(), ix->d_exact_grp.p, nq * 4, hipMemcpyDeviceToHost));
        hx.valid = true;
    }
    return RTX_OK;
}

// rtx_batch_download_then_run: the staged batch becomes the current one and is enqueued -- called when the last records of the batch
// being downloaded have left the device, while its last sub-batch is still to be finalised on the host
static int run_staged(rtx_index *ix, uint32_t flags, bool *ran_next) {
    *ran_next = false;
    if (!ix->in[ix->cur_in ^ 1u].staged) return RTX_OK;
    int rc = rtx_batch_activate(ix);
    if (!rc) rc = rtx_batch_run(ix, flags);
    *ran_next = rc == RTX_OK;
    return rc;
}

// What a run asked for that it did not have (its flags; the arena cursor it reached): more rows of counts, longer record segments, a larger
// arena with its final arrays -- for the CURRENT result set.  The caller repeats the (deterministic) run.
static int grow_for_flags(rtx_index *ix, uint32_t flags, unsigned long long cursor, uint64_t nq) {
    int rc;
    if (flags & 4u) {  // more queries took the dense epilogues than the counts buffer had rows (HitParams::cnt_row): twice the rows
        if (ix->diet_shift == 0u) { set_error("the counts buffer ran out of rows without being on its diet (internal error)"); return RTX_ERR_HIP; }
        ix->diet_shift--;
    }
    if (flags & 8u) {  // a query left a tile more records than a segment holds (RecordRef::seg_len): twice the length
        if (ix->rec_seg_len >= 8192u) { set_error("a record segment of a whole tile overflowed (internal error)"); return RTX_ERR_HIP; }
        ix->rec_seg_len *= 2u;
    }
    if ((flags & 12u) && !(flags & 1u)) return RTX_OK;
    // arena too small: grow to what this run asked for
    // (+ what the sub-allocators of the walk may leave unused on top of this run's share: their holes differ from run to run, and an
    // arena cut to this run's cursor overflowed again on every other step of the same batch -- 3.3 instead of 4.5 M queries/s on real barcodes)
    // and at least half as much again as the arena that overflowed: walks that find their sub-allocator's piece used up at the
    // same moment each take a fresh one, so the holes of a launch are not bounded by the number of sub-allocators
    const uint64_t want = std::max<uint64_t>(cursor + 4096 + (uint64_t)(ix->n_sub_run ? ix->n_sub_run : 1u) * kWalkSubAllocs * kWalkChunkRows,
                                             ix->arena_cap + ix->arena_cap / 2) + (ix->arena_cap - ix->side_base);  // (+ the side classes' region)
    if ((rc = ix->d_arena.alloc(want))) return rc;
    ix->arena_cap = want;
    return alloc_final(ix, nq);  // (the final arrays hold as many rows as the arena)
}

// RTX_OPT_RUN_AHEAD: the staged batch is enqueued while the batch being downloaded is still on the device.  The result state of that batch
// moves to rtx_index::alt (swap_result_sets), the next batch writes the other set; its first front half starts behind this batch's last
// FRONT half (the handle's stream), its scratch sets wait for the back halves that last used them (enqueue_batch).
static int run_ahead(rtx_index *ix, uint32_t flags) {
    rtx_index::Inputs &nx = ix->in[ix->cur_in ^ 1u];
    swap_result_sets(ix);
    int rc = alloc_result_set(ix, nx.n_q);
    if (!rc && ix->arena_cap < ix->alt.arena_cap) {  // what the other set has grown to, this one will need as well (and would find out by an abandoned run-ahead)
        rc = ix->d_arena.alloc(ix->alt.arena_cap);
        if (!rc) {
            ix->arena_cap = ix->alt.arena_cap;
            rc = alloc_final(ix, nx.n_q);
        }
    }
    ix->hold_join = true;
    bool ran = false;
    if (!rc) rc = run_staged(ix, flags, &ran);
    ix->hold_join = false;
    if (!rc) ix->n_run_ahead++;
    return rc;
}

// Streamed download.  Under RTX_OPT_RUN_AHEAD (`ahead` below) the staged batch is enqueued FIRST (run_ahead): two batches are then on the device,
// and the host has a whole batch's time to come back with the one after.  From there on the members of the handle are the next batch's; this
// batch's state comes back under its names for the rest of the call (the copies are asynchronous: the pointers are read when they are
// enqueued) and leaves again at its end.  An overflow of this batch found now -- with the next one on the device already -- ends in
// RTX_RETRY_CHUNK: both streams drained, the buffers grown, no batch on the handle; the caller runs this chunk again and stages the next one
// anew (host_raxtax.cpp).  Happens while a handle's buffers find their size (its first chunks), not in steady state.
static int download_streamed(rtx_index *ix, rtx_index::HostRes &hr, bool *done, uint64_t *nrows_out, bool then_run, uint32_t next_flags, bool *ran_next) {
    *done = false;
    const uint32_t n_sub = ix->n_sub_run;
    if (!ix->stream_dl || n_sub < 2 || ix->sub_q0.size() != n_sub) return RTX_OK;
    uint32_t n_side = 0;  // the side classes' sub-batches come first in the plan (and on the stream of the back halves): they are waited for LAST, their rows left with the first range
    while (n_side < n_sub && ix->cls[ix->sub_cls[n_side]].side) n_side++;
    const bool ahead = then_run && ix->run_ahead_opt != 0u && ix->join_pending && n_side == 0 && ix->in[ix->cur_in ^ 1u].staged;
    // (a batch that is complete already takes the bulk path -- unless the next one is about to be enqueued ahead: the device comes first)
    if (!ahead && hipEventQuery(ix->ev_sub[n_sub - 1]) == hipSuccess && hipEventQuery(ix->ev_sub[0]) == hipSuccess) return RTX_OK;
    const uint64_t nq = ix->n_q;
    int rc = size_host_results(ix, hr, nq, nq + nq / 4 + 64, 0);
    if (rc) return rc;
    if ((rc = fetch_exact_groups(ix, nq))) return rc;  // (now, beside the kernels: 4 MB per 1 M queries that used to cross at the tail, with the device idle)
    struct Restore {  // this batch's result state under the handle's names while `on`
        rtx_index *ix; bool on;
        ~Restore() { if (on) swap_result_sets(ix); }
    } restore{ix, false};
    if (ahead) {
        if ((rc = run_ahead(ix, next_flags))) return rc;
        *ran_next = true;
        swap_result_sets(ix);
        restore.on = true;
    }
    uint64_t prev = 0;
    uint32_t redo = 0;  // what sends this batch round again (bits of d_flags)
    for (uint32_t k = 0; k < n_sub && !redo; k++) {
        const uint32_t sb = k + n_side < n_sub ? k + n_side : k + n_side - n_sub;
        RTX_HIP(hipEventSynchronize(ix->ev_sub[sb]));
        const bool side = !ahead && ix->cls[ix->sub_cls[sb]].side;  // its rows lie in the side region of the arena, behind its own cursor (ahead: no side class; the plan is the next batch's by now)
        if (ix->h_cursor_sub[sb] > (side ? ix->arena_cap : ix->side_base)) { redo = 1u; break; }  // arena overflow
        const uint64_t cur = ix->h_fin_sub[sb];  // final rows [prev, cur): this sub-batch's (the finalise launches follow one another)
        if (cur > ix->fin_cap) { redo = 1u; break; }
        // (cur <= prev: a side class that ran in front of the bulk on the one stream -- its rows left with the first range)
        if (cur > std::min<uint64_t>({hr.v_row_lineage.cap, hr.v_row_depth8.cap, hr.v_row_local.cap, hr.v_row_conf.cap / ix->fin_D, hr.v_row_hund.cap / ix->fin_D}))
            RTX_HIP(hipStreamSynchronize(ix->copy_stream));  // an array is about to move: the copies into it have to have landed
        if (cur > prev && ((rc = size_host_results(ix, hr, nq, cur, prev)) || (rc = copy_rows(ix, hr, prev, cur, ix->copy_stream)))) return rc;
        prev = std::max(prev, cur);
        if (k + 1 == n_sub) {  // the last records leave the device
            if ((rc = copy_queries(ix, hr, nq, ix->copy_stream))) return rc;
            RTX_HIP(hipStreamSynchronize(ix->copy_stream));
            if (!ahead) {
                if ((rc = settle_join(ix))) return rc;
                RTX_HIP(hipStreamSynchronize(ix->stream));
            }
            // the run's flags have arrived (copied behind its last kernel -- under RTX_OPT_RUN_AHEAD on the stream of the back halves, which the
            // handle's stream joins in FRONT of that copy: the event behind the copy is what says so)
            if (ix->ev_flags) RTX_HIP(hipEventSynchronize(ix->ev_flags));
            uint32_t flags = ix->h_flags.size() ? ix->h_flags[0] : 0u;  // (copied behind the run's last kernel: enqueue_batch)
            if (ahead && ix->run_ahead_opt == 2u && (ix->n_run_ahead & 1u)) flags |= 1u;  // (test aid: as if the arena had overflowed)
            if (flags & 12u) { redo = flags & 13u; break; }  // the rows of the counts buffer ran out, or a record segment was too short (whatever else such a run flagged)
            if (flags & 2u) { set_error("lineage walk exceeded its row/depth bounds (internal error)"); return RTX_ERR_HIP; }
            if (flags & 1u) { redo = 1u; break; }  // arena overflow
            if (!ahead && then_run && (rc = run_staged(ix, next_flags, ran_next))) return rc;  // the device is free for the next batch
        }
    }
    if (redo && !ahead) return RTX_OK;  // the bulk path repeats the run
    if (redo) {  // this batch has to run again, and the next one is on the device: drain, grow, hand both back to the caller
        RTX_HIP(hipStreamSynchronize(ix->copy_stream));
        RTX_HIP(hipStreamSynchronize(ix->stream));
        if (ix->stream2) RTX_HIP(hipStreamSynchronize(ix->stream2));
        uint32_t flags = 0;
        unsigned long long both[2] = {0, 0};
        RTX_HIP(hipMemcpy(&flags, ix->d_flags.p, 4, hipMemcpyDeviceToHost));
        RTX_HIP(hipMemcpy(both, ix->d_cursor.p, 16, hipMemcpyDeviceToHost));
        restore.on = false;  // (this batch's set stays the current one: it is the one that grows)
        if ((rc = grow_for_flags(ix, (flags & 13u) | redo, both[0], nq))) return rc;
        ix->join_pending = false;
        ix->uploaded = ix->ran = false;
        ix->synced = true;
        ix->in[0].staged = ix->in[1].staged = false;
        ix->ws_valid = false;
        ix->res_set ^= 1u;  // (the repeated download takes this host set again: the other one holds the view of the chunk before)
        ix->n_run_ahead_retry++;
        *ran_next = false;
        return RTX_RETRY_CHUNK;
    }
    if (!*ran_next) ix->synced = true;  // (else: the next batch is running)
    *nrows_out = prev;
    *done = true;
    return RTX_OK;
}


}  // namespace rtxi

extern "C" {

static int download_impl(rtx_index *ix, rtx_result_view *out, bool then_run, uint32_t next_flags) {
    int rc = bind(ix);
    if (rc) return rc;
    if (!ix->ran || !out) { set_error("rtx_batch_download before rtx_batch_run"); return RTX_ERR_STATE; }
    const uint64_t nq = ix->n_q;
    ix->res_set ^= 1u;
    rtx_index::HostRes &hr = ix->host_res[ix->res_set];
    bool streamed = false, ran_next = false;
    uint64_t nrows = 0;
    if ((rc = download_streamed(ix, hr, &streamed, &nrows, then_run, next_flags, &ran_next))) return rc;
    if (!streamed) {
        // (a streamed download that gave up half way -- an overflow: the run is repeated -- may have copies in flight into this result set)
        if (ix->copy_stream) RTX_HIP(hipStreamSynchronize(ix->copy_stream));
        unsigned long long cursor = 0, cursor_side = 0;
        for (int attempt = 0;; attempt++) {
            if ((rc = settle_join(ix))) return rc;
            RTX_HIP(hipStreamSynchronize(ix->stream));
            ix->synced = true;
            uint32_t flags = 0;
            unsigned long long both[2] = {0, 0};
            RTX_HIP(hipMemcpy(&flags, ix->d_flags.p, 4, hipMemcpyDeviceToHost));
            RTX_HIP(hipMemcpy(both, ix->d_cursor.p, 16, hipMemcpyDeviceToHost));
            cursor = both[0];
            cursor_side = both[1];
            if ((flags & 2u) && !(flags & 12u)) { set_error("lineage walk exceeded its row/depth bounds (internal error)"); return RTX_ERR_HIP; }
            if (!(flags & 13u)) break;
            if (attempt >= 8) { set_error("result arena / counts rows overflow persists"); return RTX_ERR_HIP; }
            if ((rc = grow_for_flags(ix, flags, cursor, nq))) return rc;
            if ((flags & 12u) && !(flags & 1u)) {
                if ((rc = enqueue_batch(ix, ix->last_flags))) return rc;
                continue;
            }
            if (ix->n_refs != ix->n_total) {  // a sharded run is driven by the caller: ask it to repeat
                set_error("result arena overflow: repeat the sharded run (the arena has been enlarged)");
                return RTX_ERR_STATE;
            }
            if ((rc = enqueue_batch(ix, ix->last_flags))) return rc;
        }
        unsigned long long fin = 0;
        RTX_HIP(hipMemcpy(&fin, ix->d_fin_cursor.p, 8, hipMemcpyDeviceToHost));
        if (fin > ix->fin_cap) { set_error("final result arrays overflowed without a flag (internal error)"); return RTX_ERR_HIP; }
        (void)cursor_side;
        if ((rc = size_host_results(ix, hr, nq, fin, 0)) || (rc = copy_rows(ix, hr, 0, fin, ix->stream)) || (rc = copy_queries(ix, hr, nq, ix->stream))) return rc;
        RTX_HIP(hipStreamSynchronize(ix->stream));
        nrows = fin;
    }
    {   // the first download of a handle: the other result set (the two alternate, a view stays valid until the second-next
        // download) is allocated now, so that the second batch does not pay for its page-locked allocations
        rtx_index::HostRes &other = ix->host_res[ix->res_set ^ 1u];
        if (other.h_t.empty() && other.v_row_lineage.empty() && (rc = size_host_results(ix, other, nq, hr.v_row_lineage.size(), 0))) return rc;
    }
    if (!streamed) {  // (the streamed path has fetched them before it let the next batch onto the device)
        if ((rc = fetch_exact_groups(ix, nq))) return rc;
        if (then_run && (rc = run_staged(ix, next_flags, &ran_next))) return rc;
    }
    out->n_queries = (uint32_t)nq;
    out->n_rows = nrows;
    out->t = hr.h_t.data();
    out->status = hr.h_status.data();
    out->global_signal = hr.h_gs.data();
    out->row_begin = reinterpret_cast<const uint64_t *>(hr.v_row_begin.data());
    out->row_count = hr.v_row_count.data();
    out->row_lineage = hr.v_row_lineage.data();
    out->row_node = hr.v_row_node.data();
    out->row_depth = hr.v_row_depth.data();
    out->row_conf = hr.v_row_conf.data();
    out->row_local_signal = hr.v_row_local.data();
    out->row_conf_stride = ix->fin_D;
    out->row_depth_u8 = hr.v_row_depth8.data();
    out->row_conf_hundredths = hr.v_row_hund.data();
    return RTX_OK;
}

int rtx_batch_download(rtx_index *ix, rtx_result_view *out) { return download_impl(ix, out, false, 0); }

// Download of the current batch; as soon as its last records have left the device the STAGED batch (rtx_batch_prefetch) is activated and
// run with `flags` -- the host finalises the last sub-batch of this batch while the device already classifies the next one.  Without a
// staged batch: rtx_batch_download.
int rtx_batch_download_then_run(rtx_index *ix, rtx_result_view *out, uint32_t flags) { return download_impl(ix, out, true, flags); }

int rtx_index_has_exact_lookup(const rtx_index *index) { return index && index->d_em_table.p && index->dev_exact_opt ? 1 : 0; }

// Tree.sequences.get(query) for every query of the last download, as the device found it: CSR over the queries
int rtx_batch_exact_matches(rtx_index *ix, const uint64_t **exact_off, const uint32_t **exact_ids) {
    if (!ix || !exact_off || !exact_ids) { set_error("null argument"); return RTX_ERR_INVALID; }
    rtx_index::HostExact &hx = ix->host_exact[ix->res_set];
    if (!hx.valid) { set_error("rtx_batch_exact_matches: the last download has no device lookup (ids were passed in, or no table)"); return RTX_ERR_STATE; }
    if (!hx.csr_valid) {
        const size_t nq = hx.grp.size();
        hx.off.assign(nq + 1, 0);
        for (size_t q = 0; q < nq; q++) {
            const uint32_t g = hx.grp[q];
            hx.off[q + 1] = hx.off[q] + (g == 0xFFFFFFFFu ? 0u : ix->h_em_goff[g + 1] - ix->h_em_goff[g]);
        }
        hx.ids.resize(hx.off[nq] + 1);
        for (size_t q = 0; q < nq; q++) {
            const uint32_t g = hx.grp[q];
            if (g != 0xFFFFFFFFu) std::copy(ix->h_em_gids.begin() + ix->h_em_goff[g], ix->h_em_gids.begin() + ix->h_em_goff[g + 1], hx.ids.begin() + hx.off[q]);
        }
        hx.csr_valid = true;
    }
    *exact_off = hx.off.data();
    *exact_ids = hx.ids.data();
    return RTX_OK;
}

int rtx_classify_batch(rtx_index *index, uint64_t n_queries, const uint8_t *bases, const uint64_t *base_off,
                       const uint32_t *exact_ids, const uint64_t *exact_off, uint32_t flags, rtx_result_view *out) {
    int rc = rtx_batch_upload(index, n_queries, bases, base_off, exact_ids, exact_off);
    if (rc) return rc;
    if ((rc = rtx_batch_run(index, flags))) return rc;
    return rtx_batch_download(index, out);
}

}  // extern "C"
