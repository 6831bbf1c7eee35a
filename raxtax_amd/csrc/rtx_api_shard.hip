// The staged path of a sharded database (BASELINE.json configs[4]): the kernels of a sub-batch in groups, the exchanges of the caller in between.
#include "rtx_index.hpp"

extern "C" {

// ---- reference-sharded database (BASELINE.json configs[4], SURVEY.md 8e mode B) ------------------
// Every rank holds the bitmaps of a contiguous range of references and classifies the SAME queries.
// Per sub-batch the caller alternates library stages with two exchanges (RCCL through
// torch.distributed in raxtax_amd/sharded.py):
//   rtx_shard_count  -> all-reduce(sum) of the histograms  (RTX_BUF_HIST,  [nq][hstride] u32)
//   rtx_shard_prob   -> all-gather of the local prefix sums (RTX_BUF_PREFIX, [nq][n_bnd_local] f64), offset
//                       by the running shard totals and concatenated into [nq][n_bnd]
//   rtx_shard_walk(prefix_global)
int rtx_shard_begin(rtx_index *ix, uint32_t *n_sub_batches, uint32_t *sub_batch) {
    int rc = bind(ix);
    if (rc) return rc;
    if (!ix->uploaded) { set_error("rtx_shard_begin before rtx_batch_upload"); return RTX_ERR_STATE; }
    if (!ix->staged) {  // second scratch set: sub-batch sb + 1 may be counted while sub-batch sb is exchanged
        if ((rc = alloc_scratch_set(ix, 1))) return rc;
        ix->staged = true;
    }
    uint32_t n_sub = 0;
    bool timed = false;
    // shards must agree on the order: input order, or -- for shards that prune (the pair kernel needs neighbours that are related) --
    // the min-hash order, which is a function of the queries alone (no locator on a shard: stable radix sort of the sketch keys)
    if ((rc = begin_run(ix, &n_sub, &timed, ix->shard_prune_opt && ix->prune_opt && ix->d_ubitmap.p && ix->n_refs != ix->n_total && ix->cluster))) return rc;
    ix->ran = true;
    ix->synced = false;
    ix->last_flags = 0;
    if (n_sub_batches) *n_sub_batches = n_sub;
    if (sub_batch) *sub_batch = ix->sub_batch;
    return RTX_OK;
}

static int shard_sb(rtx_index *ix, uint32_t sb, SubBatch *b) {
    int rc = bind(ix);
    if (rc) return rc;
    if (!ix->ran) { set_error("rtx_shard_* before rtx_shard_begin"); return RTX_ERR_STATE; }
    const uint32_t n_sub = ix->n_sub_total;
    if (sb >= n_sub) { set_error("sub-batch %u out of range (%u)", sb, n_sub); return RTX_ERR_INVALID; }
    *b = sub_batch_of(ix, sb, ix->n_sub_last != 0);
    ix->synced = false;
    return RTX_OK;
}

int rtx_shard_prunes(const rtx_index *ix) { return ix && ix->ran && ix->staged && ix->prune_used ? 1 : 0; }

// A pruning shard, first half of the counting of a sub-batch: k-mers, bounds against the union bitmap of this shard, its candidate for
// the best block of the database (RTX_BUF_BEST).  The caller keeps per query the candidate with the largest bound over all shards
// (ties: the lowest shard) in every shard's buffer, then rtx_shard_count.
int rtx_shard_bounds(rtx_index *ix, uint32_t sb, uint32_t flags) {
    SubBatch b;
    int rc = shard_sb(ix, sb, &b);
    if (rc) return rc;
    if (!ix->prune_used) { set_error("rtx_shard_bounds: this run does not prune (rtx_shard_prunes)"); return RTX_ERR_STATE; }
    ix->last_flags = flags;
    if ((rc = enqueue_kmer(ix, b, b.s))) return rc;
    return enqueue_hit(ix, b, flags, b.s, 1);
}

int rtx_shard_count(rtx_index *ix, uint32_t sb, uint32_t flags) {
    SubBatch b;
    int rc = shard_sb(ix, sb, &b);
    if (rc) return rc;
    ix->last_flags = flags;
    if (ix->prune_used) return enqueue_hit(ix, b, flags, b.s, 2);  // after rtx_shard_bounds and the exchange of RTX_BUF_BEST
    return enqueue_count(ix, b, flags);
}

int rtx_shard_prob(rtx_index *ix, uint32_t sb) {
    SubBatch b;
    int rc = shard_sb(ix, sb, &b);
    if (rc) return rc;
    return enqueue_prob_prefix(ix, b, false);
}

int rtx_shard_walk(rtx_index *ix, uint32_t sb, const double *prefix_global) {
    SubBatch b;
    int rc = shard_sb(ix, sb, &b);
    if (rc) return rc;
    if (!prefix_global) { set_error("rtx_shard_walk: null prefix"); return RTX_ERR_INVALID; }
    if ((rc = enqueue_walk(ix, b, prefix_global, b.s))) return rc;
    return enqueue_finalise(ix, b, b.s);
}

int rtx_shard_info(const rtx_index *ix, uint64_t *ref_lo, uint64_t *ref_hi, uint32_t *n_bnd_global, uint32_t *n_bnd_local,
                   uint32_t *first_bnd) {
    if (!ix) { set_error("null index handle"); return RTX_ERR_INVALID; }
    if (ref_lo) *ref_lo = ix->ref_lo;
    if (ref_hi) *ref_hi = ix->ref_lo + ix->n_refs;
    if (n_bnd_global) *n_bnd_global = ix->n_bnd;
    if (n_bnd_local) *n_bnd_local = ix->n_bnd_local;
    if (first_bnd) *first_bnd = ix->bnd_first;
    return RTX_OK;
}

int rtx_device_buffer(rtx_index *ix, int which, void **ptr, uint64_t *row_stride_elems) {
    if (!ix || !ptr) { set_error("null argument"); return RTX_ERR_INVALID; }
    if (!ix->uploaded) { set_error("rtx_device_buffer before rtx_batch_upload"); return RTX_ERR_STATE; }
    rtx_index::Scratch &sc = ix->sc[0];
    switch (which) {
        case RTX_BUF_HIST: *ptr = sc.d_hist.p; if (row_stride_elems) *row_stride_elems = ix->hstride; return RTX_OK;
        case RTX_BUF_PREFIX: *ptr = sc.d_prefix.p; if (row_stride_elems) *row_stride_elems = ix->n_bnd_local; return RTX_OK;
        default: break;
    }
    set_error("rtx_device_buffer: unknown buffer %d", which);
    return RTX_ERR_INVALID;
}

int rtx_shard_buffer(rtx_index *ix, uint32_t sb, int which, void **ptr, uint64_t *row_stride_elems) {
    if (!ix || !ptr) { set_error("null argument"); return RTX_ERR_INVALID; }
    if (!ix->uploaded) { set_error("rtx_shard_buffer before rtx_batch_upload"); return RTX_ERR_STATE; }
    rtx_index::Scratch &sc = ix->sc[ix->staged ? (sb & 1u) : 0u];
    switch (which) {
        case RTX_BUF_HIST: *ptr = sc.d_hist.p; if (row_stride_elems) *row_stride_elems = ix->hstride; return RTX_OK;
        case RTX_BUF_PREFIX: *ptr = sc.d_prefix.p; if (row_stride_elems) *row_stride_elems = ix->n_bnd_local; return RTX_OK;
        case RTX_BUF_BEST:
            if (!sc.d_best.p) { set_error("RTX_BUF_BEST: the handle does not prune"); return RTX_ERR_STATE; }
            *ptr = sc.d_best.p;
            if (row_stride_elems) *row_stride_elems = kPruneBestWords;
            return RTX_OK;
        case RTX_BUF_COUNTS:
            if (ix->packed()) { set_error("RTX_BUF_COUNTS needs u16 counts (RTX_OPT_PACKED_COUNTS = 0)"); return RTX_ERR_STATE; }
            *ptr = sc.d_counts.p;
            if (row_stride_elems) *row_stride_elems = ix->npad;
            return RTX_OK;
        default: break;
    }
    set_error("rtx_shard_buffer: unknown buffer %d", which);
    return RTX_ERR_INVALID;
}

int rtx_index_stream(rtx_index *ix, void **hip_stream) {
    if (!ix || !hip_stream) { set_error("null argument"); return RTX_ERR_INVALID; }
    *hip_stream = (void *)ix->stream;
    return RTX_OK;
}

// k-mer-sharded database (SURVEY.md 8e mode A): the counts of a sub-batch have been all-reduced over the ranks;
// the histogram of prob.rs:13-19 is rebuilt from them (the one hit_count wrote covered this rank's k-mers only)
int rtx_shard_rehist(rtx_index *ix, uint32_t sb) {
    SubBatch b;
    int rc = shard_sb(ix, sb, &b);
    if (rc) return rc;
    if (ix->packed()) { set_error("rtx_shard_rehist needs u16 counts (RTX_OPT_PACKED_COUNTS = 0)"); return RTX_ERR_STATE; }
    rtx_index::Scratch &sc = ix->sc[b.set];
    launch_rehist(b.s, sc.d_counts.p, ix->npad, ix->n_refs, sc.d_t.p, sc.d_hist.p, ix->hstride, sc.d_tilemax.p, ix->ntiles, b.nq);
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}

}  // extern "C"
