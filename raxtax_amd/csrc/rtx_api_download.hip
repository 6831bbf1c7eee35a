// Result download: the records of finished sub-batches are copied and finalised on the host (sort lineage.rs:91-93, expected vectors
// and local signal lineage.rs:95-102) while later sub-batches still run; exact matches as the device found them; rtx_classify_batch.
#include "rtx_index.hpp"

namespace rtxi {

// lineage.rs:91-110 for the rows of one query: expected vectors, stable descending sort by confidence vector, local signal
// (utils.rs:91-105).  The device hands a row over as {node, confidence per level in hundredths}; everything that depends on the node
// alone -- depth, the expected vector (|range| / N per level, lineage.rs:137-139), the level the local signal starts at
// (lineage.rs:95-98) -- is tabulated once per handle (node_tables), so that a row costs a handful of loads: real barcodes return ten
// rows per query where the synthetic workload returns one, and the finalisation must keep up with the device there too.
double euclidean_distance_l1(const double *a, const double *b, uint32_t n) {  // utils.rs:91-105
    if (n == 0) return 0.0;
    double a_sum = 0.0, b_sum = 0.0;
    for (uint32_t i = 0; i < n; i++) a_sum += a[i];
    for (uint32_t i = 0; i < n; i++) b_sum += b[i];
    double s = 0.0;
    for (uint32_t i = 0; i < n; i++) {
        const double d = a[i] / a_sum - b[i] / b_sum;
        s += d * d;
    }
    return std::sqrt(s);
}

void node_tables(rtx_index *ix) {  // expd[node][d], local-signal start per node
    const FlatNodes &f = ix->nodes;
    const uint32_t D = std::max(1u, f.max_depth), nn = f.size();
    ix->h_node_stride = D;
    ix->h_node_expd.assign((size_t)nn * D, 0.0);
    ix->h_node_sig0.assign(nn, 0);
    const double N = (double)ix->n_total;
    for (uint32_t v = 0; v < nn; v++) {
        const uint32_t depth = f.depth[v];
        double *e = ix->h_node_expd.data() + (size_t)v * D;
        uint32_t anc = v;
        for (int d = (int)depth - 1; d >= 0; d--) {
            e[d] = (double)(f.end[anc] - f.begin[anc]) / N;
            anc = f.parent[anc];
        }
        uint32_t s0 = depth ? depth - 1 : 0;  // lineage.rs:95-98: the first level whose expected share is below 1, else the last
        for (uint32_t d = 0; d < depth; d++)
            if (1.0 > e[d]) { s0 = d; break; }
        ix->h_node_sig0[v] = (uint8_t)s0;
    }
}

// Host finalisation of the queries at positions [pa, pb) of the processing order; their rows go to
// [row_base, ...) of the host row arrays in that order.
void finalise_range(rtx_index *ix, uint64_t pa, uint64_t pb, uint64_t row_base) {
    rtx_index::HostRes &hr = ix->host_res[ix->res_set];
    const FlatNodes &f = ix->nodes;
    const uint32_t D = ix->h_node_stride;
    std::vector<uint32_t> ord;
    uint64_t o = row_base;
    for (uint64_t pos = pa; pos < pb; pos++) {
        const uint64_t q = ix->dl_perm[pos];  // the device records are in processing order
        hr.h_t[q] = ix->hs_t[pos];
        hr.h_status[q] = ix->hs_status[pos];
        hr.h_gs[q] = ix->hs_gs[pos];
        const uint32_t nr = ix->h_n_rows[pos];
        hr.v_row_begin[q] = o;
        hr.v_row_count[q] = nr;
        const DevRow *src = ix->h_arena.data() + ix->h_row_start[pos];
        ord.resize(nr);
        for (uint32_t r = 0; r < nr; r++) ord[r] = r;
        if (nr > 1) {
            // stable, descending by confidence vector, a shorter prefix smaller (lineage.rs:91-93): the hundredths order like the values
            std::stable_sort(ord.begin(), ord.end(), [&](uint32_t x, uint32_t y) {  // true: x comes first = y < x
                const uint32_t dx = f.depth[src[x].node], dy = f.depth[src[y].node], n = std::min(dx, dy);
                const int c = n ? std::memcmp(src[y].k, src[x].k, n) : 0;  // bytes compare like the numbers they hold
                return c ? c < 0 : dy < dx;
            });
        }
        for (uint32_t r = 0; r < nr; r++, o++) {
            const DevRow &h = src[ord[r]];
            const uint32_t depth = f.depth[h.node];
            hr.v_row_lineage[o] = f.begin[h.node];
            hr.v_row_node[o] = h.node;
            hr.v_row_depth[o] = depth;
            double *c = hr.v_row_conf.data() + o * RTX_MAX_DEPTH;  // (entries from the deepest lineage of the tree on are never written: zero since the resize)
            for (uint32_t d = 0; d < D; d++) c[d] = d < depth ? (double)h.k[d] / 100.0 : 0.0;  // == round(x*100)/100, lineage.rs:128-129
            const uint32_t s = ix->h_node_sig0[h.node];
            hr.v_row_local[o] = depth ? euclidean_distance_l1(c + s, ix->h_node_expd.data() + (size_t)h.node * D + s, depth - s) : 0.0;
        }
    }
}

// Finalises positions [pa, pb) on up to nt threads; returns the number of rows they produced.
uint64_t finalise_mt(rtx_index *ix, uint64_t pa, uint64_t pb, uint64_t row_base, unsigned nt) {
    nt = rtx::host_threads(nt);  // this process's share of the host's CPUs (cgroup quota, ranks per host)
    nt = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(nt, (pb - pa) / 4096));  // a thread start costs what 2 000 single-row queries cost
    std::vector<uint64_t> cut(nt + 1), base(nt + 1, row_base);
    for (unsigned i = 0; i <= nt; i++) cut[i] = pa + (pb - pa) * i / nt;
    for (unsigned i = 0; i < nt; i++) {
        uint64_t rows = 0;
        for (uint64_t pos = cut[i]; pos < cut[i + 1]; pos++) rows += ix->h_n_rows[pos];
        base[i + 1] = base[i] + rows;
    }
    rtx_index::HostRes &hr = ix->host_res[ix->res_set];
    const uint64_t nrows = base[nt];
    if (hr.v_row_lineage.size() < nrows) {
        // Only the set being written grows: the other one is the view of the previous download, which stays valid (and
        // may be read by the caller's formatting thread) until the second-next download (include/raxtax_hip.h).
        // Growth keeps 25 % headroom so that a batch with a few more rows than the last one does not reallocate.
        const uint64_t want = nrows + nrows / 4 + 64;
        hr.v_row_lineage.resize(want);
        hr.v_row_node.resize(want);
        hr.v_row_depth.resize(want);
        hr.v_row_local.resize(want);
        hr.v_row_conf.resize(want * RTX_MAX_DEPTH);
    }
    if (nt == 1) {
        finalise_range(ix, pa, pb, row_base);
    } else {
        std::vector<std::thread> th;
        for (unsigned i = 0; i < nt; i++) th.emplace_back(finalise_range, ix, cut[i], cut[i + 1], base[i]);
        for (auto &t : th) t.join();
    }
    return nrows - row_base;
}

int size_host_results(rtx_index *ix, rtx_index::HostRes &hr, uint64_t nq, uint64_t arena_rows) {
    int rc;
    if ((rc = ix->hs_status.resize(nq)) || (rc = ix->hs_t.resize(nq)) || (rc = ix->h_n_rows.resize(nq)) || (rc = ix->hs_gs.resize(nq)) ||
        (rc = ix->h_row_start.resize(nq)) || (rc = ix->h_arena.resize(arena_rows ? arena_rows : 1)))
        return rc;
    hr.h_status.resize(nq);
    hr.h_t.resize(nq);
    hr.h_gs.resize(nq);
    hr.v_row_begin.resize(nq);
    hr.v_row_count.resize(nq);
    return RTX_OK;
}

// D2H of the per-query records at positions [q0, q0+n) and of arena rows [r0, r1) on stream cs (asynchronous)
int copy_results(rtx_index *ix, uint64_t q0, uint64_t n, uint64_t r0, uint64_t r1, hipStream_t cs) {
    RTX_HIP(hipMemcpyAsync(ix->hs_status.data() + q0, ix->d_status.p + q0, n, hipMemcpyDeviceToHost, cs));
    RTX_HIP(hipMemcpyAsync(ix->hs_t.data() + q0, ix->d_t_all.p + q0, n * 4, hipMemcpyDeviceToHost, cs));
    RTX_HIP(hipMemcpyAsync(ix->h_n_rows.data() + q0, ix->d_n_rows.p + q0, n * 4, hipMemcpyDeviceToHost, cs));
    RTX_HIP(hipMemcpyAsync(ix->hs_gs.data() + q0, ix->d_gs.p + q0, n * 8, hipMemcpyDeviceToHost, cs));
    RTX_HIP(hipMemcpyAsync(ix->h_row_start.data() + q0, ix->d_row_start.p + q0, n * 8, hipMemcpyDeviceToHost, cs));
    if (r1 > r0) RTX_HIP(hipMemcpyAsync(ix->h_arena.data() + r0, ix->d_arena.p + r0, (r1 - r0) * sizeof(DevRow), hipMemcpyDeviceToHost, cs));
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

static int download_streamed(rtx_index *ix, rtx_index::HostRes &hr, bool *done, uint64_t *nrows_out, bool then_run, uint32_t next_flags, bool *ran_next) {
    *done = false;
    const uint32_t n_sub = ix->n_sub_run;
    if (!ix->stream_dl || n_sub < 2 || ix->sub_q0.size() != n_sub || (hipEventQuery(ix->ev_sub[n_sub - 1]) == hipSuccess && hipEventQuery(ix->ev_sub[0]) == hipSuccess)) return RTX_OK;
    (void)0;
    const uint64_t nq = ix->n_q;
    int rc = size_host_results(ix, hr, nq, ix->arena_cap);
    if (rc) return rc;
    uint64_t prev_main = 0, prev_side = ix->side_base, nrows = 0;
    uint32_t n_side = 0;  // the side classes' sub-batches come first in the plan and run beside the bulk: they are waited for LAST
    while (n_side < n_sub && ix->cls[ix->sub_cls[n_side]].side) n_side++;
    for (uint32_t k = 0; k < n_sub; k++) {
        const uint32_t sb = k + n_side < n_sub ? k + n_side : k + n_side - n_sub;
        RTX_HIP(hipEventSynchronize(ix->ev_sub[sb]));
        const uint64_t cur = ix->h_cursor_sub[sb];
        const bool side = ix->cls[ix->sub_cls[sb]].side;  // its rows lie in the side region of the arena, behind its own cursor
        uint64_t &prev = side ? prev_side : prev_main;
        if (cur > (side ? ix->arena_cap : ix->side_base)) return RTX_OK;  // overflow: bulk path
        const uint64_t q0 = ix->sub_q0[sb], n = ix->sub_nq[sb];  // (classes of different sub-batch sizes follow one another: plan_sub_batches)
        if ((rc = copy_results(ix, q0, n, prev, cur, ix->copy_stream))) return rc;
        RTX_HIP(hipStreamSynchronize(ix->copy_stream));
        if (k + 1 == n_sub) {  // the last records have left the device: it is free for the next batch while the host finalises these
            RTX_HIP(hipStreamSynchronize(ix->stream));
            uint32_t flags = 0;
            RTX_HIP(hipMemcpy(&flags, ix->d_flags.p, 4, hipMemcpyDeviceToHost));
            if (flags & 2u) { set_error("lineage walk exceeded its row/depth bounds (internal error)"); return RTX_ERR_HIP; }
            if (flags & 1u) return RTX_OK;  // arena overflow: the bulk path repeats the run
            if ((rc = fetch_exact_groups(ix, nq))) return rc;
            if (then_run && (rc = run_staged(ix, next_flags, ran_next))) return rc;
        }
        // one thread finalises 8192 queries in ~1.4 ms, about what the device needs for the next sub-batch: with a short
        // last sub-batch the host would still be busy with the one before it when the device is done
        nrows += finalise_mt(ix, q0, q0 + n, nrows, k + 1 == n_sub ? 16 : 8);
        prev = cur;
    }
    if (!*ran_next) ix->synced = true;  // (else: the next batch is running)
    *nrows_out = nrows;
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
    ix->dl_perm = ix->h_perm_now().data();
    rtx_index::HostRes &hr = ix->host_res[ix->res_set];
    bool streamed = false, ran_next = false;
    uint64_t nrows = 0;
    if ((rc = download_streamed(ix, hr, &streamed, &nrows, then_run, next_flags, &ran_next))) return rc;
    if (!streamed) {
        unsigned long long cursor = 0, cursor_side = 0;
        for (int attempt = 0;; attempt++) {
            RTX_HIP(hipStreamSynchronize(ix->stream));
            ix->synced = true;
            uint32_t flags = 0;
            unsigned long long both[2] = {0, 0};
            RTX_HIP(hipMemcpy(&flags, ix->d_flags.p, 4, hipMemcpyDeviceToHost));
            RTX_HIP(hipMemcpy(both, ix->d_cursor.p, 16, hipMemcpyDeviceToHost));
            cursor = both[0];
            cursor_side = both[1];
            if (flags & 2u) { set_error("lineage walk exceeded its row/depth bounds (internal error)"); return RTX_ERR_HIP; }
            if (!(flags & 1u)) break;
            if (attempt >= 5) { set_error("result arena overflow persists"); return RTX_ERR_HIP; }
            // arena too small: grow to what this run asked for and repeat the (deterministic) run
            // (+ what the sub-allocators of the walk may leave unused on top of this run's share: their holes differ from run to run, and an
            // arena cut to this run's cursor overflowed again on every other step of the same batch -- 3.3 instead of 4.5 M queries/s on real barcodes)
            // and at least half as much again as the arena that overflowed: walks that find their sub-allocator's piece used up at the
            // same moment each take a fresh one, so the holes of a launch are not bounded by the number of sub-allocators
            const uint64_t want = std::max<uint64_t>(cursor + 4096 + (uint64_t)(ix->n_sub_run ? ix->n_sub_run : 1u) * kWalkSubAllocs * kWalkChunkRows,
                                                     ix->arena_cap + ix->arena_cap / 2) + (ix->arena_cap - ix->side_base);  // (+ the side classes' region)
            if ((rc = ix->d_arena.alloc(want))) return rc;
            ix->arena_cap = want;
            if (ix->n_refs != ix->n_total) {  // a sharded run is driven by the caller: ask it to repeat
                set_error("result arena overflow: repeat the sharded run (the arena has been enlarged)");
                return RTX_ERR_STATE;
            }
            if ((rc = enqueue_batch(ix, ix->last_flags))) return rc;
        }
        const bool side_rows = cursor_side > ix->side_base && ix->side_base < ix->arena_cap;  // rows of side classes at the top of the arena
        if ((rc = size_host_results(ix, hr, nq, side_rows ? (uint64_t)cursor_side : (uint64_t)cursor)) || (rc = copy_results(ix, 0, nq, 0, cursor, ix->stream))) return rc;
        if (side_rows) RTX_HIP(hipMemcpyAsync(ix->h_arena.data() + ix->side_base, ix->d_arena.p + ix->side_base, (cursor_side - ix->side_base) * sizeof(DevRow), hipMemcpyDeviceToHost, ix->stream));
        RTX_HIP(hipStreamSynchronize(ix->stream));
        nrows = finalise_mt(ix, 0, nq, 0, nq < 4096 ? 1 : 16);
    }
    {   // the first download of a handle: the other result set (the two alternate, a view stays valid until the second-next
        // download) is sized and touched now, so that the second batch does not pay for its page faults (60 ms at 1M queries)
        rtx_index::HostRes &other = ix->host_res[ix->res_set ^ 1u];
        if (other.h_t.empty() && other.v_row_lineage.empty()) {
            other.h_t.resize(hr.h_t.size());
            other.h_status.resize(hr.h_status.size());
            other.h_gs.resize(hr.h_gs.size());
            other.v_row_begin.resize(hr.v_row_begin.size());
            other.v_row_count.resize(hr.v_row_count.size());
            other.v_row_lineage.resize(hr.v_row_lineage.size());
            other.v_row_node.resize(hr.v_row_node.size());
            other.v_row_depth.resize(hr.v_row_depth.size());
            other.v_row_conf.resize(hr.v_row_conf.size());
            other.v_row_local.resize(hr.v_row_local.size());
        }
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
    out->row_begin = hr.v_row_begin.data();
    out->row_count = hr.v_row_count.data();
    out->row_lineage = hr.v_row_lineage.data();
    out->row_node = hr.v_row_node.data();
    out->row_depth = hr.v_row_depth.data();
    out->row_conf = hr.v_row_conf.data();
    out->row_local_signal = hr.v_row_local.data();
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
