// C-ABI implementation of the device side of libraxtax_hip.so: index upload/re-encoding,
// batch workspace, kernel sequencing on the handle's HIP stream, result download and the
// host finalisation (sort lineage.rs:91-93, local signal lineage.rs:95-102).
// There is deliberately no CPU fallback in this file: without a gfx950 device every entry
// point returns RTX_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "rtx_internal.hpp"
#include "rtx_kernels.hpp"
#include "rtx_math.hpp"

using namespace rtx;

namespace {

constexpr uint32_t kEmptyRow = 0xFFFFFFFFu;
constexpr uint32_t kLnFactLen = 98320;  // covers t + n - 1 for every t <= 65535

#define RTX_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return e_ == hipErrorOutOfMemory ? RTX_ERR_OOM : RTX_ERR_HIP;                      \
        }                                                                                      \
    } while (0)

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    int alloc(size_t count) {
        if (count <= n && p) return RTX_OK;
        release();
        if (count == 0) count = 1;
        hipError_t e = hipMalloc((void **)&p, count * sizeof(T));
        if (e != hipSuccess) {
            p = nullptr;
            set_error("hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
            return RTX_ERR_OOM;
        }
        n = count;
        return RTX_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    ~DevBuf() { release(); }
};

// Pinned host array (hipHostMalloc): D2H copies of the result records run at PCIe rate and asynchronously.
template <class T>
struct PinBuf {
    T *p = nullptr;
    size_t cap = 0, n = 0;
    int resize(size_t count) {
        if (count > cap) {
            if (p) (void)hipHostFree(p);
            p = nullptr;
            const size_t want = count + count / 4 + 16;
            if (hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocDefault) != hipSuccess) {
                p = nullptr;
                cap = n = 0;
                set_error("hipHostMalloc(%zu bytes) failed", want * sizeof(T));
                return RTX_ERR_OOM;
            }
            cap = want;
        }
        n = count;
        return RTX_OK;
    }
    T *data() { return p; }
    const T *data() const { return p; }
    size_t size() const { return n; }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
    ~PinBuf() { if (p) (void)hipHostFree(p); }
};

// statrs 0.16 `ln_factorial` (the reference's ln_binomial, prob.rs:5,20,117,143): ln of a cached
// f64 factorial up to 170, Lanczos ln_gamma (g = 10.900511, 11 terms) above.
double statrs_ln_gamma(double x) {
    static const double dk[11] = {2.48574089138753565546e-5, 1.05142378581721974210,  -3.45687097222016235469,
                                  4.51227709466894823700,    -2.98285225323576655721, 1.05639711577126713077,
                                  -1.95428773191645869583e-1, 1.70970543404441224307e-2,
                                  -5.71926117404305781283e-4, 4.63399473359905636708e-6,
                                  -2.71994908488607703910e-9};
    const double r = 10.900511, ln_2_sqrt_e_over_pi = 0.6207822376352452223455184457816472122518527279025978;
    double s = dk[0];
    for (int i = 1; i < 11; i++) s += dk[i] / (x + (double)i - 1.0);
    return std::log(s) + ln_2_sqrt_e_over_pi + (x - 0.5) * std::log((x - 0.5 + r) / M_E);
}
void fill_ln_factorial(std::vector<double> &lf) {
    lf.resize(kLnFactLen);
    double f = 1.0;
    for (uint32_t x = 0; x < kLnFactLen; x++) {
        if (x <= 170) {
            if (x > 0) f *= (double)x;
            lf[x] = std::log(f);
        } else {
            lf[x] = statrs_ln_gamma((double)x + 1.0);
        }
    }
}

inline uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

}  // namespace

struct rtx_index {
    int device = -1;
    hipStream_t stream = nullptr;
    uint64_t n_refs = 0;    // references held by this handle (the whole database, or one shard of it)
    uint64_t n_total = 0;   // references of the whole database (Tree.num_tips)
    uint32_t ref_lo = 0;    // first global reference id of this shard
    uint32_t n_bnd_local = 0, bnd_first = 0;  // boundaries in (ref_lo, ref_hi] + 1; global index of ref_lo
    const double *ext_prefix = nullptr;       // sharded mode: assembled global prefix handed to the walk
    // ---- index proper
    uint32_t n_rows = 0;        // non-empty posting lists
    uint32_t stride_bytes = 0;  // bytes per bitmap row over all tiles (multiple of 1024)
    uint64_t npad = 0;          // references per padded row (= stride_bytes * 8)
    uint32_t ntiles = 0;        // 8192-reference tiles
    DevBuf<uint32_t> d_bitmap, d_row_of, d_list_len;
    // segment classes (rtx_segments.hip): class / sparse slot of every (row, tile) segment, slots of 32 local ids
    DevBuf<uint32_t> d_seginfo, d_seg_sbase;
    DevBuf<unsigned long long> d_seg_dbits, d_seg_sbits;
    uint32_t seg_blocks = 0;  // > 0: kmer_extract uses the bit tables (many tiles)
    DevBuf<uint16_t> d_segslots;
    uint64_t n_seg_slots = 0;
    uint32_t seg_stride = 0;
    DevBuf<double> d_lnfact, d_inv;
    // ---- memoised prob tables (t <= 1023), built lazily for the largest tmax seen
    int prob_mode = 0;  // 0 auto, 1 recurrence kernel only, 2 tables (error if they do not fit)
    uint32_t tab_tmax = 0;
    DevBuf<double> d_tab_cmf, d_tab_ratio;
    DevBuf<uint64_t> d_tab_off;
    DevBuf<uint32_t> d_tab_moff;
    DevBuf<uint16_t> d_tab_ilo, d_tab_sat;
    bool use_tables = false;
    // ---- taxonomy
    FlatNodes nodes;
    std::vector<uint32_t> bnd;  // sorted unique range endpoints
    uint32_t n_bnd = 0;
    DevBuf<uint4> d_noderec;  // {blo, bhi, first_child, n_children | type << 30} per node (lineage_walk)
    DevBuf<uint32_t> d_bnd_rank;
    DevBuf<uint8_t> d_bnd_bits;
    // ---- exact-match lookup on the device (rtx_exact.hip): the distinct reference sequences ("groups") in a hash table
    uint32_t dev_exact_opt = 1;       // RTX_OPT_DEVICE_EXACT
    uint32_t em_groups = 0, em_bits = 0;
    uint64_t em_hash_mask = ~0ull;    // RTX_DEFAULT_EXACT_HASH_MASK at creation (tests: a weak hash, so that probes collide)
    DevBuf<uint2> d_em_table;         // [2^em_bits] {tag, group + 1}
    DevBuf<uint64_t> d_em_rep_off;    // [groups + 1]
    DevBuf<uint8_t> d_em_rep_bytes;   // the distinct sequences
    DevBuf<uint32_t> d_em_goff, d_em_gids;   // ids of group g: gids[goff[g] .. goff[g + 1]), ascending (tree.rs:109-112)
    std::vector<uint32_t> h_em_goff, h_em_gids;  // host copies: the ids behind the groups the device reports
    DevBuf<uint32_t> d_exact_grp;     // [n_q] group of every query of the batch (0xFFFFFFFF: none)
    bool dev_exact_used = false;      // the uploaded batch came without ids: the device looks them up (every rtx_batch_run)
    struct HostExact {                // per host result set: the groups of a download and, on demand, the CSR of their ids
        std::vector<uint32_t> grp;
        std::vector<uint64_t> off;
        std::vector<uint32_t> ids;
        bool csr_valid = false, valid = false;
    } host_exact[2];
    // ---- batch inputs
    uint64_t n_q = 0;
    bool uploaded = false, ran = false, synced = false;
    uint32_t last_flags = 0;
    // ---- processing order of the batch (rtx_cluster.hip): perm[position] = query, inv[query] = position
    uint32_t cluster = 1;  // RTX_OPT_CLUSTER
    uint32_t packed_opt = 1;  // RTX_OPT_PACKED_COUNTS
    uint32_t tile_skip = 1;   // RTX_OPT_TILE_SKIP: taxon_prefix reads only the tiles that hold a reference with p >= 1e-30
    uint32_t pair_opt = 1;    // RTX_OPT_HIT_PAIR
    uint32_t prune_opt = 1;   // RTX_OPT_TILE_PRUNE: hit_count visits only the tiles that can hold a reference with any probability (rtx_prune.hip)
    bool prune_used = false;  // the last run pruned
    bool dbg_full = false;    // ... and the debug taps have recounted the last sub-batch in full since
    bool dbg_full_run = false;  // (the recount in progress: enqueue_hit leaves the pruning out)
    DevBuf<uint32_t> d_ubitmap;  // union bitmap: one column per block of 2^kPruneShift references, tile-major like d_bitmap
    uint32_t u_stride_bytes = 0, u_ntiles = 0;
    uint64_t u_nblocks = 0;
    DevBuf<unsigned long long> d_prune_stats;
    uint32_t shard_prune_opt = 0;  // RTX_OPT_SHARD_PRUNE: a reference shard prunes with the threshold of the whole database (rtx_shard_bounds)
    uint32_t debug_taps = 0;     // RTX_OPT_DEBUG_TAPS: prune_kernel leaves its view of every query (rtx_debug_prune_detail)
    DevBuf<uint32_t> d_prune_detail;  // [sub_batch][kPruneDetailWords]
    uint32_t locator_opt = 1; // RTX_OPT_LOCATOR: the sort key of the processing order is led by the query's position in the database
    DevBuf<uint32_t> d_loc_table;  // 12-mer -> lowest reference position (rtx_cluster.hip); only when built from sequences
    bool pair_used = false;   // the last run went through hit_count_pair_kernel
    DevBuf<uint32_t> d_group_rows;
    uint32_t n_groups_run = 0;  // groups of the whole batch (n_sub * groups_per_sub): the second half of d_group_rows starts there
    uint32_t groups_per_sub = 0;
    bool packed() const { return packed_opt && planes <= 10; }
    DevBuf<uint64_t> d_skey_in, d_skey_out;
    DevBuf<uint32_t> d_sidx, d_perm, d_iperm;
    DevBuf<uint8_t> d_sort_tmp;
    PinBuf<uint32_t> h_perm, h_inv;
    DevBuf<uint8_t> d_bases;  // the current batch, one byte per base (what the kernels read): unpacked from the staged transfer at activation
    // Two input sets: a batch is STAGED (rtx_batch_prefetch: bases packed two per byte into pinned memory, offsets, exact-match ids;
    // asynchronous H2D on h2d_stream) while the batch before it runs out of the other set, and becomes the current one at
    // rtx_batch_activate.  rtx_batch_upload = prefetch + activate.
    struct Inputs {
        DevBuf<uint8_t> d_packed;          // bases two per byte (or raw, one per byte, if a byte above 15 was seen)
        DevBuf<uint64_t> d_base_off, d_exact_off;
        DevBuf<uint32_t> d_exact_ids;
        PinBuf<uint8_t> h_packed;
        PinBuf<uint64_t> h_base_off, h_exact_off;
        PinBuf<uint32_t> h_exact_ids;
        uint64_t n_q = 0, total = 0, max_len = 0, n_exact = 0;
        bool packed = true, has_exact = false, staged = false, recorded = false;
        hipEvent_t ready = nullptr;        // its transfer has arrived
    } in[2];
    uint32_t cur_in = 0;               // the set of the current (activated) batch
    hipStream_t h2d_stream = nullptr;
    hipEvent_t ev_activated = nullptr; // on the handle's stream, behind everything that was enqueued before the current batch was activated:
                                       // the kernels that read the OTHER input set have run when it fires (a transfer into that set waits for it)
    uint64_t sum_query_bytes = 0;
    uint32_t kstride = 0, rstride = 0, hstride = 0, tmax = 0;
    int planes = 10;
    // ---- sub-batch scratch: two sets -- a staged (reference-sharded) run alternates between them, so that the exchange of
    // one sub-batch can overlap with the counting of the next; a whole-database handle uses set 0 only
    uint32_t sub_batch_req = 0, sub_batch = 0;
    struct Scratch {
        DevBuf<uint16_t> d_kmers, d_counts, d_tilemax;
        DevBuf<uint32_t> d_rows, d_t, d_nrows, d_hist, d_order, d_srows, d_nsparse;
        DevBuf<unsigned long long> d_dmask;
        DevBuf<double> d_table_z, d_prefix;
        DevBuf<uint2> d_urec;   // hit_count_pair_kernel: union row lists of the pairs of the sub-batch
        DevBuf<uint32_t> d_nu;
        // tile pruning: the queries counted against the union bitmap (every row dense: constant masks) leave the largest bound of
        // every tile and the best block (bounds_epilogue); thresholds and the live tiles per pair (prune_kernel)
        DevBuf<unsigned long long> d_uones;
        DevBuf<uint32_t> d_uzero, d_live, d_best_key;
        DevBuf<uint32_t> d_items;  // [pairs x tiles] the (pair, tile) blocks with a live query | [1] their number | [8] queue per XCD | [pairs] live tiles per pair | [pairs] offsets
        DevBuf<uint16_t> d_tile_ub, d_prune_thr, d_prune_i1;
        DevBuf<uint32_t> d_best;  // [B][kPruneBestWords] reference shards: the candidate for the best block of the database
    } sc[2];
    bool staged = false;  // driven with rtx_shard_*: sub-batch sb works in scratch set sb & 1, so that the exchange of one
                          // sub-batch (RCCL, on the caller's stream) can overlap with the counting of the next
    uint32_t last_set = 0;  // scratch set of the last sub-batch (debug taps)
    DevBuf<double> d_probs_dbg;
    DevBuf<uint16_t> d_counts_dbg;
    // ---- per-query results
    DevBuf<uint8_t> d_status;
    DevBuf<uint32_t> d_t_all, d_nrows_all, d_n_rows, d_flags, d_ndist;
    DevBuf<double> d_gs, d_z;
    DevBuf<unsigned long long> d_hq, d_row_start, d_cursor;
    DevBuf<DevRow> d_arena;
    uint64_t arena_cap = 0;
    // ---- timing
    std::vector<hipEvent_t> events;  // 2 per (sub-batch, stage)
    uint32_t n_sub_last = 0;
    // ---- host results
    // two alternating sets: the view of download c stays valid while batch c+1 runs and is downloaded
    struct HostRes {
        std::vector<uint32_t> v_row_lineage, v_row_node, v_row_depth;
        std::vector<uint32_t> h_t;
        std::vector<uint8_t> h_status;
        std::vector<double> v_row_conf, v_row_local;
        std::vector<double> h_gs;
        std::vector<uint64_t> v_row_begin;  // by query; the rows themselves are in processing order
        std::vector<uint32_t> v_row_count;
    } host_res[2];
    // D2H staging of the per-query records, indexed by position in the processing order
    PinBuf<uint32_t> hs_t;
    PinBuf<uint8_t> hs_status;
    PinBuf<double> hs_gs;
    uint32_t res_set = 0;
    PinBuf<uint32_t> h_nrows_all, h_n_rows;
    // streamed download: per sub-batch a snapshot of the arena cursor + an event; rtx_batch_download copies and
    // finalises finished sub-batches on `copy_stream` while later ones are still running
    std::vector<hipEvent_t> ev_sub;
    PinBuf<unsigned long long> h_cursor_sub;
    hipStream_t copy_stream = nullptr;
    uint32_t n_sub_run = 0;
    bool stream_dl = false;
    PinBuf<unsigned long long> h_hq, h_row_start;
    uint32_t stage_timing = 0;  // 0: HIP events around hit_count only; 1: around every kernel
    PinBuf<DevRow> h_arena;
    // per node: expected vector and first level of the local signal (node_tables; finalise_range)
    std::vector<double> h_node_expd;
    std::vector<uint8_t> h_node_sig0;
    uint32_t h_node_stride = 1;

    ~rtx_index() {
        for (auto e : events) (void)hipEventDestroy(e);
        for (auto e : ev_sub) (void)hipEventDestroy(e);
        for (auto &i : in)
            if (i.ready) (void)hipEventDestroy(i.ready);
        if (ev_activated) (void)hipEventDestroy(ev_activated);
        if (h2d_stream) (void)hipStreamDestroy(h2d_stream);
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

namespace {

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

// One sub-batch = three groups of kernels.  A whole-database handle runs them back to back; a
// reference-sharded handle (config 5) stops after each group for the exchange with the other shards.
struct SubBatch {
    uint32_t sb, nq, set;
    uint64_t q0;
    hipStream_t s;   // main stream
    bool timed;      // HIP events around hit_count (the roofline kernel)
    bool timed_all;  // ... and around every other kernel (RTX_OPT_STAGE_TIMING)
};

SubBatch sub_batch_of(rtx_index *ix, uint32_t sb, bool timed) {
    SubBatch b;
    b.sb = sb;
    b.q0 = (uint64_t)sb * ix->sub_batch;
    b.nq = (uint32_t)std::min<uint64_t>(ix->sub_batch, ix->n_q - b.q0);
    b.set = ix->staged ? (sb & 1u) : 0u;
    b.s = ix->stream;
    b.timed = timed;
    b.timed_all = timed && ix->stage_timing != 0;
    return b;
}

// Counts of a sub-batch between hit_count and taxon_prefix.  With 10 bit planes (t <= 1023) they travel packed,
// 10 bits per reference: [B][npad] low bytes, then [B][npad / 8] u16 with the two high bits of eight references
// each; otherwise [B][npad] u16.  Both live in the same allocation (sized for the format in use).
uint8_t *counts_lo(rtx_index *ix, rtx_index::Scratch &sc) { return reinterpret_cast<uint8_t *>(sc.d_counts.p); }
uint16_t *counts_hi(rtx_index *ix, rtx_index::Scratch &sc) {
    return reinterpret_cast<uint16_t *>(reinterpret_cast<uint8_t *>(sc.d_counts.p) + (size_t)ix->sub_batch * ix->npad);
}
size_t counts_elems(const rtx_index *ix, uint64_t B) {  // u16 elements of d_counts
    return ix->packed() ? (size_t)B * ix->npad * 5 / 8 : (size_t)B * ix->npad;
}

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
    kp.zero_row = ix->n_rows;
    kp.kmers = sc.d_kmers.p;
    kp.kstride = ix->kstride;
    kp.seginfo = ix->d_seginfo.p;
    kp.seg_stride = ix->seg_stride;
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
int enqueue_hit(rtx_index *ix, const SubBatch &b, uint32_t flags, hipStream_t s, int part = 0) {
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
        up.dmask = sc.d_uones.p;
        up.nsparse = sc.d_uzero.p;
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
            RTX_HIP(hipMemsetAsync(sc.d_best_key.p, 0, (size_t)b.nq * 4, s));  // the waves of a query's union tiles meet in an atomicMax
            launch_hit_count_pair_bounds(s, up, b.nq, ix->u_ntiles);  // the union of the pair's rows serves both passes
            if (b.timed) RTX_HIP(hipEventRecord(stage_event(ix, b, RTX_STAGE_TILE_BOUNDS, 1), s));
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
        pr.hist = sc.d_hist.p;
        pr.hstride = ix->hstride;
        pr.live = sc.d_live.p;
        pr.live_words = (ix->ntiles + 31u) / 32u + 1u;
        pr.pair_live = sc.d_items.p + (size_t)((b.nq + 1u) / 2u) * ix->ntiles + 9u;
        pr.thr_out = sc.d_prune_thr.p;
        pr.i1_out = sc.d_prune_i1.p;
        pr.stats = ix->d_prune_stats.p;
        pr.detail = ix->debug_taps && ix->d_prune_detail.n >= (size_t)b.nq * kPruneDetailWords ? ix->d_prune_detail.p : nullptr;
        ProbTables tb{ix->d_tab_cmf.p, ix->d_tab_ratio.p, ix->d_tab_off.p, ix->d_tab_moff.p, ix->d_tab_ilo.p, ix->d_tab_sat.p, ix->tab_tmax};
        launch_prune(s, pr, tb, b.nq);
        if (part == 1) { RTX_HIP(hipGetLastError()); return RTX_OK; }  // the caller exchanges RTX_BUF_BEST, then part 2
        // (3) tiles that are not counted keep a largest count of 0: taxon_prefix leaves them out
        RTX_HIP(hipMemsetAsync(sc.d_tilemax.p, 0, (size_t)b.nq * ix->ntiles * 2, s));
        hp.live = sc.d_live.p;
        hp.live_words = pr.live_words;
        hp.prune_thr = sc.d_prune_thr.p;
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
    if (ix->pair_used) launch_hit_count_pair(s, hp, b.nq, ix->ntiles);
    else launch_hit_count(s, hp, b.nq, ix->ntiles, ix->planes);
    if (b.timed) RTX_HIP(hipEventRecord(stage_event(ix, b, RTX_STAGE_HIT_COUNT, 1), s));
    return RTX_OK;
}

int enqueue_count(rtx_index *ix, const SubBatch &b, uint32_t flags) {
    int rc = enqueue_kmer(ix, b, b.s);
    return rc ? rc : enqueue_hit(ix, b, flags, b.s);
}

static WalkParams walk_params(rtx_index *ix, const SubBatch &b, const double *prefix);

// group 2: prob table from the (whole-database) histogram + prefix sums over this handle's references; with
// fuse_walk (whole database on this handle) the taxonomy walk of group 3 runs inside the prefix kernel
int enqueue_prob_prefix(rtx_index *ix, const SubBatch &b, bool fuse_walk, bool prob_only = false) {
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
    if (fuse_walk) fp.walk = walk_params(ix, b, sc.d_prefix.p);
    if (b.timed_all) RTX_HIP(hipEventRecord(stage_event(ix, b, RTX_STAGE_TAXON_PREFIX, 0), s));
    launch_taxon_prefix(s, fp, b.nq);
    if (b.timed_all) RTX_HIP(hipEventRecord(stage_event(ix, b, RTX_STAGE_TAXON_PREFIX, 1), s));
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
    wp.arena_cap = ix->arena_cap;
    wp.arena_cursor = ix->d_cursor.p;
    wp.n_rows = ix->d_n_rows.p;
    wp.row_start = ix->d_row_start.p;
    wp.flags_out = ix->d_flags.p;
    return wp;
}

int enqueue_walk(rtx_index *ix, const SubBatch &b, const double *prefix, hipStream_t s) {
    const WalkParams wp = walk_params(ix, b, prefix);
    if (b.timed_all) RTX_HIP(hipEventRecord(stage_event(ix, b, RTX_STAGE_LINEAGE_WALK, 0), s));
    launch_lineage_walk(s, wp, b.nq);
    if (b.timed_all) RTX_HIP(hipEventRecord(stage_event(ix, b, RTX_STAGE_LINEAGE_WALK, 1), s));
    return RTX_OK;
}

// Processing order of the uploaded batch: related queries next to each other (rtx_cluster.hip), or input order.
int order_batch(rtx_index *ix, bool cluster) {
    const uint32_t n = (uint32_t)ix->n_q;
    int rc;
    if ((rc = ix->d_perm.alloc(n)) || (rc = ix->d_iperm.alloc(n)) || (rc = ix->h_perm.resize(n)) || (rc = ix->h_inv.resize(n))) return rc;
    if (cluster && n > 2) {
        if ((rc = ix->d_skey_in.alloc(n)) || (rc = ix->d_skey_out.alloc(n)) || (rc = ix->d_sidx.alloc(n))) return rc;
        size_t tmp = 0;
        if (cluster_sort(ix->stream, nullptr, &tmp, ix->d_skey_in.p, ix->d_skey_out.p, ix->d_sidx.p, ix->d_perm.p, n)) {
            set_error("radix sort: size query failed");
            return RTX_ERR_HIP;
        }
        if (ix->d_sort_tmp.n < tmp && (rc = ix->d_sort_tmp.alloc(tmp + 256))) return rc;
        launch_sketch(ix->stream, ix->d_bases.p, ix->in[ix->cur_in].d_base_off.p, n, ix->d_skey_in.p, ix->d_sidx.p);
        if (ix->d_loc_table.p && ix->locator_opt)
            launch_locator(ix->stream, ix->d_bases.p, ix->in[ix->cur_in].d_base_off.p, n, ix->d_loc_table.p, ix->n_total, ix->d_skey_in.p);
        tmp = ix->d_sort_tmp.n;
        if (cluster_sort(ix->stream, ix->d_sort_tmp.p, &tmp, ix->d_skey_in.p, ix->d_skey_out.p, ix->d_sidx.p, ix->d_perm.p, n)) {
            set_error("radix sort of the query sketches failed");
            return RTX_ERR_HIP;
        }
        launch_invert_perm(ix->stream, ix->d_perm.p, n, ix->d_iperm.p);
    } else {
        launch_identity_perm(ix->stream, n, ix->d_perm.p, ix->d_iperm.p);
    }
    RTX_HIP(hipGetLastError());
    RTX_HIP(hipMemcpyAsync(ix->h_perm.data(), ix->d_perm.p, (size_t)n * 4, hipMemcpyDeviceToHost, ix->stream));
    RTX_HIP(hipMemcpyAsync(ix->h_inv.data(), ix->d_iperm.p, (size_t)n * 4, hipMemcpyDeviceToHost, ix->stream));
    return RTX_OK;
}

int begin_run(rtx_index *ix, uint32_t *n_sub_out, bool *timed_out, bool cluster) {
    const uint32_t n_sub = (uint32_t)((ix->n_q + ix->sub_batch - 1) / ix->sub_batch);
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
    RTX_HIP(hipMemsetAsync(ix->d_cursor.p, 0, sizeof(unsigned long long), ix->stream));
    RTX_HIP(hipMemsetAsync(ix->d_flags.p, 0, sizeof(uint32_t), ix->stream));
    // two neighbours per wave only pays when neighbours are related: with the processing order on
    ix->pair_used = ix->pair_opt && cluster && ix->planes <= 10 && ix->n_q > 1 && ix->rstride <= 4096;
    ix->groups_per_sub = (ix->sub_batch + 1u) / 2u;
    // tile pruning: the pair kernel, the memoised tables (their ln cmf rows give the threshold), taxon_prefix skipping tiles by
    // their largest count, the whole database on this handle
    // a whole-database handle driven by rtx_batch_run, or a reference shard that was asked to (RTX_OPT_SHARD_PRUNE: the caller then
    // drives rtx_shard_bounds and exchanges the best blocks); never a k-mer shard (its counts are partial sums)
    const bool whole = ix->n_refs == ix->n_total && !ix->staged;
    const bool shard = ix->staged && ix->shard_prune_opt && ix->n_refs != ix->n_total;
    auto scratch_ok = [&](const rtx_index::Scratch &sc) {  // sized at the upload / rtx_shard_begin (alloc_scratch_set) for this sub-batch size
        return sc.d_tile_ub.p != nullptr && sc.d_tile_ub.n >= (size_t)ix->sub_batch * ix->ntiles && sc.d_best_key.n >= ix->sub_batch && sc.d_prune_thr.n >= ix->sub_batch &&
               sc.d_live.n >= (size_t)(ix->sub_batch + 1u) * ((ix->ntiles + 31u) / 32u + 1u) && sc.d_best.n >= (size_t)ix->sub_batch * kPruneBestWords &&
               sc.d_items.n >= (size_t)((ix->sub_batch + 1u) / 2u) * (ix->ntiles + 2u) + 9u;
    };
    ix->prune_used = ix->prune_opt && ix->pair_used && ix->use_tables && ix->tile_skip && ix->d_ubitmap.p && (whole || shard) &&
                     scratch_ok(ix->sc[0]) && (!ix->staged || scratch_ok(ix->sc[1]));
    ix->dbg_full = false;
    if (ix->prune_used) {
        int rc_s = ix->d_prune_stats.alloc(kPruneStatCopies * 16);
        if (!rc_s && ix->debug_taps) rc_s = ix->d_prune_detail.alloc((size_t)ix->sub_batch * kPruneDetailWords);
        if (rc_s) return rc_s;
        RTX_HIP(hipMemsetAsync(ix->d_prune_stats.p, 0, kPruneStatCopies * 128, ix->stream));
    }
    if (ix->pair_used) {
        ix->n_groups_run = n_sub * ix->groups_per_sub;
        int rc_g = ix->d_group_rows.alloc((size_t)2 * n_sub * ix->groups_per_sub);  // second half: the bounds pass of the tile pruning
        if (rc_g) return rc_g;
        RTX_HIP(hipMemsetAsync(ix->d_group_rows.p, 0, (size_t)2 * n_sub * ix->groups_per_sub * 4, ix->stream));
    }
    ix->n_sub_last = timed ? n_sub : 0;
    if (ix->dev_exact_used) {  // Tree.sequences.get for every query of the batch (raxtax.rs:42), part of the run
        ExactParams xp{ix->d_bases.p, ix->in[ix->cur_in].d_base_off.p, (uint32_t)ix->n_q, ix->d_em_table.p, ix->em_bits, ix->d_em_rep_off.p,
                       ix->d_em_rep_bytes.p, ix->d_exact_grp.p, ix->em_hash_mask};
        const bool ev = timed && ix->stage_timing;
        if (ev) RTX_HIP(hipEventRecord(ix->events[(size_t)RTX_STAGE_EXACT_MATCH * 2], ix->stream));  // sub-batch 0
        launch_exact_match(ix->stream, xp);
        if (ev) RTX_HIP(hipEventRecord(ix->events[(size_t)RTX_STAGE_EXACT_MATCH * 2 + 1], ix->stream));
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
    int rc = begin_run(ix, &n_sub, &timed, ix->cluster != 0);
    if (rc) return rc;
    ix->stream_dl = false;
    if (n_sub <= 4096) {  // per sub-batch: completion event (+ cursor snapshot) for the streamed download
        if (!ix->copy_stream) RTX_HIP(hipStreamCreateWithFlags(&ix->copy_stream, hipStreamNonBlocking));
        while (ix->ev_sub.size() < n_sub) {
            hipEvent_t e;
            RTX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ix->ev_sub.push_back(e);
        }
        if ((rc = ix->h_cursor_sub.resize(n_sub))) return rc;
        ix->n_sub_run = n_sub;
        ix->stream_dl = true;
    }
    // the walk rides inside the prefix kernel (the stage time of lineage_walk is then part of taxon_prefix)
    const bool fuse = ix->n_bnd_local == ix->n_bnd;
    for (uint32_t sb = 0; sb < n_sub; sb++) {
        const SubBatch b = sub_batch_of(ix, sb, timed);
        if ((rc = enqueue_count(ix, b, flags))) return rc;
        if ((rc = enqueue_prob_prefix(ix, b, fuse))) return rc;
        if (!fuse && (rc = enqueue_walk(ix, b, ix->sc[b.set].d_prefix.p, b.s))) return rc;
        if (fuse && b.timed_all) {  // keeps rtx_batch_stage_times whole: an empty interval
            RTX_HIP(hipEventRecord(stage_event(ix, b, RTX_STAGE_LINEAGE_WALK, 0), b.s));
            RTX_HIP(hipEventRecord(stage_event(ix, b, RTX_STAGE_LINEAGE_WALK, 1), b.s));
        }
        if (ix->stream_dl) {
            RTX_HIP(hipMemcpyAsync(&ix->h_cursor_sub[sb], ix->d_cursor.p, 8, hipMemcpyDeviceToHost, b.s));
            RTX_HIP(hipEventRecord(ix->ev_sub[sb], b.s));
        }
    }
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}

// Builds (once per handle and tmax) the memoised cmf tables used by prob_lookup_kernel.
constexpr uint32_t kProbTablesMaxT = 1023;
int ensure_prob_tables(rtx_index *ix) {
    ix->use_tables = false;
    if (ix->prob_mode == 1 || ix->tmax < 2) return RTX_OK;
    if (ix->tmax > kProbTablesMaxT) {
        if (ix->prob_mode == 2) { set_error("prob tables need t <= %u (got %u)", kProbTablesMaxT, ix->tmax); return RTX_ERR_TOO_LONG; }
        return RTX_OK;
    }
    if (ix->tab_tmax >= ix->tmax) { ix->use_tables = true; return RTX_OK; }
    const uint32_t T = ix->tmax;
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
    ix->use_tables = true;
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
#ifndef RTX_PRUNE_MIN_TILES
#define RTX_PRUNE_MIN_TILES 4  // tiles of 8192 references from which on the tile pruning is worth its bounds pass (configs[1], 7 tiles: 5.6 -> 7.8 M queries/s; it was 8 until the bounds pass lost its stores)
#endif
constexpr uint32_t kDefaultSubBatch = 20480, kDefaultSubBatchLarge = 16384, kDefaultSubBatchPruned = 32768;

int alloc_scratch_set(rtx_index *ix, uint32_t k);

// Sizes and allocates the per-batch workspace for n_queries queries of at most tmax k-mers.
int prepare_workspace(rtx_index *ix, uint64_t n_queries, uint64_t tmax, uint64_t max_len) {
    int rc;
    if (tmax > 65535) { set_error("query with up to %llu k-mers: raxtax.rs:56 asserts t fits u16", (unsigned long long)tmax); return RTX_ERR_TOO_LONG; }
    if (prob_table_lds_bytes((uint32_t)tmax) > 160 * 1024 - 512) {
        set_error("query of %llu bases needs %zu bytes of LDS in prob_table (limit 160 KiB)", (unsigned long long)max_len,
                  prob_table_lds_bytes((uint32_t)tmax));
        return RTX_ERR_TOO_LONG;
    }
    ix->tmax = (uint32_t)tmax;
    ix->kstride = (uint32_t)align_up(tmax, 8);
    ix->rstride = (uint32_t)align_up(tmax, 64) + 64;  // row list padded to whole 64-row chunks
    ix->hstride = (uint32_t)align_up(tmax + 1, 8);
    ix->planes = tmax <= 1023 ? 10 : (tmax <= 4095 ? 12 : 16);
    ix->n_q = n_queries;
    if ((rc = ensure_prob_tables(ix))) return rc;
    // ---- per-query results
    if ((rc = ix->d_status.alloc(n_queries)) || (rc = ix->d_t_all.alloc(n_queries)) || (rc = ix->d_nrows_all.alloc(n_queries)) ||
        (rc = ix->d_n_rows.alloc(n_queries)) || (rc = ix->d_gs.alloc(n_queries)) || (rc = ix->d_z.alloc(n_queries)) ||
        (rc = ix->d_hq.alloc(n_queries)) || (rc = ix->d_row_start.alloc(n_queries)) || (rc = ix->d_ndist.alloc(n_queries)))
        return rc;
    const uint64_t want_arena = n_queries * 8 + 4096;
    if (ix->arena_cap < want_arena) {
        if ((rc = ix->d_arena.alloc(want_arena))) return rc;
        ix->arena_cap = want_arena;
    }
    // ---- sub-batch scratch, sized against free HBM
    const bool will_prune = ix->prune_opt && ix->d_ubitmap.p && ix->pair_opt && ix->ntiles >= RTX_PRUNE_MIN_TILES && tmax <= 1023 && (ix->n_refs == ix->n_total || ix->shard_prune_opt);  // begin_run decides
    const uint64_t per_q = (uint64_t)ix->kstride * 2 + (uint64_t)ix->rstride * 12 + 4 + (uint64_t)ix->ntiles * (ix->rstride / 8 + ((kSegMaxSparseRows + 1) * 4 + 10)) + (ix->packed() ? ix->npad * 5 / 4 : ix->npad * 2) + (uint64_t)ix->hstride * 12 +
                           (uint64_t)ix->n_bnd_local * 8 + 64 +
                           // + the scratch of the tile pruning: counts against the union bitmap, constant masks, its histogram, thresholds, live masks
                           (will_prune ? (uint64_t)ix->u_ntiles * (ix->rstride / 8 + 4) + (uint64_t)ix->ntiles * 2 + 12 + (ix->ntiles + 31u) / 32u * 2u + 2u + kPruneBestWords * 4 +
                                          ((uint64_t)ix->ntiles + 2u) * 2u  /* the list of live (pair, tile) blocks: 4 bytes per pair and tile */ : 0);
    uint32_t B = ix->sub_batch_req;
    if (B == 0) {
        size_t free_b = 0, total_b = 0;
        RTX_HIP(hipMemGetInfo(&free_b, &total_b));
        // scratch already held by this handle is reusable
        const uint64_t held = (ix->sc[0].d_counts.n + ix->sc[1].d_counts.n) * 2 + (ix->sc[0].d_prefix.n + ix->sc[1].d_prefix.n) * 8;
        const uint64_t budget = (uint64_t)((free_b + held) * 0.6);
        B = (uint32_t)std::min<uint64_t>(will_prune ? kDefaultSubBatchPruned : ix->ntiles >= 16 ? kDefaultSubBatchLarge : kDefaultSubBatch,
                                         std::max<uint64_t>(64, budget / per_q));
    }
    if (B > kMaxSubBatch) B = kMaxSubBatch;
    B = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(B, n_queries));
    ix->sub_batch = B;
    ix->staged = false;
    return alloc_scratch_set(ix, 0);
}

int alloc_scratch_set(rtx_index *ix, uint32_t k) {
    int rc;
    const uint32_t B = ix->sub_batch;
    {
        rtx_index::Scratch &sc = ix->sc[k];
        if ((rc = sc.d_kmers.alloc((size_t)B * ix->kstride)) || (rc = sc.d_rows.alloc((size_t)B * ix->rstride)) || (rc = sc.d_dmask.alloc((size_t)B * ix->ntiles * (ix->rstride / 64))) ||
            (rc = sc.d_nsparse.alloc((size_t)B * ix->ntiles)) || (rc = sc.d_srows.alloc((size_t)B * ix->ntiles * (kSegMaxSparseRows + 1))) ||
            (rc = sc.d_t.alloc(B)) || (rc = sc.d_nrows.alloc(B)) || (rc = sc.d_counts.alloc(counts_elems(ix, B))) ||
            (rc = sc.d_hist.alloc((size_t)B * ix->hstride)) || (rc = sc.d_table_z.alloc((size_t)B * ix->hstride)) ||
            (rc = sc.d_prefix.alloc((size_t)B * ix->n_bnd_local)) || (rc = sc.d_order.alloc(B)) ||
            (rc = sc.d_tilemax.alloc((size_t)B * ix->ntiles)) ||
            (rc = sc.d_urec.alloc((size_t)((B + 1u) / 2u) * 2u * ix->rstride)) || (rc = sc.d_nu.alloc((B + 1u) / 2u)))
            return rc;
        if (ix->prune_opt && ix->d_ubitmap.p && (ix->n_refs == ix->n_total || ix->shard_prune_opt)) {
            const size_t mw = (size_t)B * ix->u_ntiles * (ix->rstride / 64);
            const bool fresh = sc.d_uones.n < mw;
            if ((rc = sc.d_uones.alloc(mw)) || (rc = sc.d_uzero.alloc((size_t)B * ix->u_ntiles)) ||
                (rc = sc.d_tile_ub.alloc((size_t)B * ix->ntiles)) || (rc = sc.d_best_key.alloc(B)) || (rc = sc.d_prune_thr.alloc(B)) || (rc = sc.d_prune_i1.alloc(B)) || (rc = sc.d_best.alloc((size_t)B * kPruneBestWords)) || (rc = sc.d_live.alloc((size_t)(B + 1u) * ((ix->ntiles + 31u) / 32u + 1u))) ||
                (rc = sc.d_items.alloc((size_t)((B + 1u) / 2u) * (ix->ntiles + 2u) + 9u)))
                return rc;
            if (fresh) RTX_HIP(hipMemsetAsync(sc.d_uones.p, 0xFF, sc.d_uones.n * 8, ix->stream));
            RTX_HIP(hipMemsetAsync(sc.d_uzero.p, 0, sc.d_uzero.n * 4, ix->stream));
        }
    }
    return RTX_OK;
}

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
        const uint64_t q = ix->h_perm[pos];  // the device records are in processing order
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

}  // namespace

extern "C" {

int rtx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}


// Everything of index creation except the bitmap: device checks, stream, taxonomy, tables.
static int create_common(int device, uint64_t n_total, uint64_t ref_lo, uint64_t ref_hi, const uint64_t *cuts,
                         uint32_t n_cuts, uint32_t n_nodes, const uint32_t *node_begin, const uint32_t *node_end,
                         const uint32_t *node_first_child, const uint32_t *node_n_children, const uint8_t *node_type,
                         rtx_index **out) {
    if (!out || !node_begin || !node_end || !node_first_child || !node_n_children || !node_type || n_total == 0 ||
        n_total > 0xFFFFFFFFull || ref_lo >= ref_hi || ref_hi > n_total) {
        set_error("rtx_index_create: invalid argument");
        return RTX_ERR_INVALID;
    }
    const uint64_t n_refs = ref_hi - ref_lo;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        set_error("no usable HIP device (requested %d of %d); libraxtax_hip has no CPU fallback", device, ndev);
        return RTX_ERR_NO_DEVICE;
    }
    RTX_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    RTX_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; this library carries gfx950 (MI355X) code objects only", device, prop.gcnArchName);
        return RTX_ERR_NO_DEVICE;
    }
    auto ix = new rtx_index();
    ix->device = device;
    ix->n_refs = n_refs;
    ix->n_total = n_total;
    ix->ref_lo = (uint32_t)ref_lo;
    int rc = RTX_OK;
    auto fail = [&](int code) { delete ix; return code; };
    if (!derive_flat_nodes(n_total, n_nodes, node_begin, node_end, node_first_child, node_n_children, node_type, ix->nodes))
        return fail(RTX_ERR_INVALID);
    if (ix->nodes.max_depth > RTX_MAX_DEPTH) {
        set_error("lineage depth %u exceeds RTX_MAX_DEPTH=%u", ix->nodes.max_depth, RTX_MAX_DEPTH);
        return fail(RTX_ERR_DEPTH);
    }
    node_tables(ix);
    if (hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking) != hipSuccess) {
        set_error("hipStreamCreate failed");
        return fail(RTX_ERR_HIP);
    }
    // ---- taxonomy boundaries
    {
        std::vector<uint32_t> b;
        b.reserve(2 * (size_t)n_nodes + 2);
        b.push_back(0);
        b.push_back((uint32_t)n_total);
        b.push_back((uint32_t)ref_lo);
        b.push_back((uint32_t)ref_hi);
        for (uint32_t c = 0; c < n_cuts; c++) {  // shard cut points: identical boundary lists on every rank
            if (cuts[c] > n_total) { set_error("shard cut %llu beyond n_refs", (unsigned long long)cuts[c]); return fail(RTX_ERR_INVALID); }
            b.push_back((uint32_t)cuts[c]);
        }
        for (uint32_t v = 0; v < n_nodes; v++) { b.push_back(ix->nodes.begin[v]); b.push_back(ix->nodes.end[v]); }
        std::sort(b.begin(), b.end());
        b.erase(std::unique(b.begin(), b.end()), b.end());
        ix->bnd = std::move(b);
        ix->n_bnd = (uint32_t)ix->bnd.size();
        std::vector<uint32_t> blo(n_nodes), bhi(n_nodes);
        for (uint32_t v = 0; v < n_nodes; v++) {
            blo[v] = (uint32_t)(std::lower_bound(ix->bnd.begin(), ix->bnd.end(), ix->nodes.begin[v]) - ix->bnd.begin());
            bhi[v] = (uint32_t)(std::lower_bound(ix->bnd.begin(), ix->bnd.end(), ix->nodes.end[v]) - ix->bnd.begin());
        }
        // flags / ranks over the LOCAL references: boundary position p in (ref_lo, ref_hi] belongs to
        // local reference p - 1 - ref_lo; local boundary 0 is ref_lo itself
        const size_t nchunk = (size_t)((n_refs + 7) / 8);
        std::vector<uint8_t> bits(nchunk, 0);
        std::vector<uint32_t> rank(nchunk, 0);
        ix->bnd_first = (uint32_t)(std::lower_bound(ix->bnd.begin(), ix->bnd.end(), (uint32_t)ref_lo) - ix->bnd.begin());
        ix->n_bnd_local = 1;
        for (uint32_t j = 1; j < ix->n_bnd; j++) {
            if (ix->bnd[j] <= ref_lo || ix->bnd[j] > ref_hi) continue;
            const uint32_t r = ix->bnd[j] - 1 - (uint32_t)ref_lo;
            bits[r >> 3] |= (uint8_t)(1u << (r & 7u));
            ix->n_bnd_local++;
        }
        uint32_t run = 1;
        for (size_t c = 0; c < nchunk; c++) {
            rank[c] = run;
            run += (uint32_t)__builtin_popcount(bits[c]);
        }
        std::vector<uint4> noderec(n_nodes);
        for (uint32_t v = 0; v < n_nodes; v++) {
            if (ix->nodes.n_children[v] >= (1u << 30)) { set_error("node with 2^30 or more children"); return fail(RTX_ERR_INVALID); }
            noderec[v] = make_uint4(blo[v], bhi[v], ix->nodes.first_child[v], ix->nodes.n_children[v] | ((uint32_t)ix->nodes.type[v] << 30));
        }
        if ((rc = ix->d_noderec.alloc(n_nodes)) || (rc = ix->d_bnd_bits.alloc(nchunk)) || (rc = ix->d_bnd_rank.alloc(nchunk)))
            return fail(rc);
        hipError_t e = hipSuccess;
        auto up = [&](void *d, const void *h, size_t bytes) { if (e == hipSuccess) e = hipMemcpy(d, h, bytes, hipMemcpyHostToDevice); };
        up(ix->d_noderec.p, noderec.data(), (size_t)n_nodes * sizeof(uint4));
        up(ix->d_bnd_bits.p, bits.data(), nchunk);
        up(ix->d_bnd_rank.p, rank.data(), nchunk * 4);
        if (e != hipSuccess) { set_error("taxonomy upload failed: %s", hipGetErrorString(e)); return fail(RTX_ERR_HIP); }
    }
    // ---- ln-factorial table
    {
        std::vector<double> lf;
        fill_ln_factorial(lf);
        if ((rc = ix->d_lnfact.alloc(lf.size()))) return fail(rc);
        if (hipMemcpy(ix->d_lnfact.p, lf.data(), lf.size() * 8, hipMemcpyHostToDevice) != hipSuccess) {
            set_error("lnfact upload failed");
            return fail(RTX_ERR_HIP);
        }
        std::vector<double> inv(lf.size(), 0.0);
        for (size_t x = 1; x < inv.size(); x++) inv[x] = 1.0 / (double)x;
        if ((rc = ix->d_inv.alloc(inv.size()))) return fail(rc);
        if (hipMemcpy(ix->d_inv.p, inv.data(), inv.size() * 8, hipMemcpyHostToDevice) != hipSuccess) {
            set_error("reciprocal table upload failed");
            return fail(RTX_ERR_HIP);
        }
    }
    ix->stride_bytes = (uint32_t)align_up((n_refs + 7) / 8, 1024);  // whole tiles: the bitmap is stored tile by tile
    ix->npad = (uint64_t)ix->stride_bytes * 8;
    ix->ntiles = (ix->stride_bytes + 1023) / 1024;
    if ((rc = ix->d_cursor.alloc(1)) || (rc = ix->d_flags.alloc(1))) return fail(rc);
    *out = ix;
    return RTX_OK;
}

static bool prepare_union_bitmap(rtx_index *ix);

// Hash table of the distinct reference sequences for the device exact-match lookup (rtx_exact.hip).  `groups`: per distinct
// sequence the ids of the references that have it, ascending (Tree.sequences, tree.rs:109-112); group order = order of the first
// id, so that the table is the same however the caller's map iterates.  An aid like the locator: if it cannot be built (memory) the
// handle works without it and callers pass the ids of Tree.sequences.get themselves.
static uint64_t g_em_hash_mask = ~0ull;  // RTX_DEFAULT_EXACT_HASH_MASK (rtx_set_default_option)
static uint64_t em_hash_bytes(const uint8_t *s, uint64_t len) {
    uint64_t sum = 0;
    for (uint64_t j = 0; j * 8 < len; j++) {
        uint64_t w = 0;
        std::memcpy(&w, s + 8 * j, (size_t)std::min<uint64_t>(8, len - 8 * j));
        sum += em_mix_word(w, j);
    }
    return em_finish(sum, len);
}
static void build_exact_table(rtx_index *ix, const uint8_t *seq_bytes, const uint64_t *seq_off,
                              std::vector<const std::vector<uint32_t> *> &groups) {
    if (ix->n_refs != ix->n_total || groups.empty()) return;
    std::sort(groups.begin(), groups.end(), [](const std::vector<uint32_t> *a, const std::vector<uint32_t> *b) { return (*a)[0] < (*b)[0]; });
    const uint32_t G = (uint32_t)groups.size();
    uint32_t bits = 4;
    while ((1ull << bits) < 2ull * G) bits++;
    std::vector<uint64_t> rep_off(G + 1, 0);
    std::vector<uint32_t> goff(G + 1, 0), gids;
    gids.reserve(ix->n_total);
    for (uint32_t g = 0; g < G; g++) {
        const uint32_t rep = (*groups[g])[0];
        rep_off[g + 1] = rep_off[g] + (seq_off[rep + 1] - seq_off[rep]);
        gids.insert(gids.end(), groups[g]->begin(), groups[g]->end());
        goff[g + 1] = (uint32_t)gids.size();
    }
    std::vector<uint8_t> rep_bytes(rep_off[G] + 16, 0);
    std::vector<uint64_t> hashes(G);
    {
        const unsigned nt = rtx::host_threads(8u);
        std::vector<std::thread> th;
        for (unsigned k = 0; k < nt; k++)
            th.emplace_back([&, k] {
                for (uint32_t g = (uint32_t)((uint64_t)G * k / nt); g < (uint32_t)((uint64_t)G * (k + 1) / nt); g++) {
                    const uint32_t rep = (*groups[g])[0];
                    const uint64_t len = seq_off[rep + 1] - seq_off[rep];
                    std::memcpy(rep_bytes.data() + rep_off[g], seq_bytes + seq_off[rep], (size_t)len);
                    hashes[g] = em_hash_bytes(seq_bytes + seq_off[rep], len) & g_em_hash_mask;
                }
            });
        for (auto &t : th) t.join();
    }
    std::vector<uint2> table((size_t)1 << bits, make_uint2(0u, 0u));
    const uint32_t mask = (1u << bits) - 1u;
    for (uint32_t g = 0; g < G; g++) {
        uint32_t slot = em_slot(hashes[g], bits);
        while (table[slot].y) slot = (slot + 1u) & mask;
        table[slot] = make_uint2(em_tag(hashes[g]), g + 1u);
    }
    hipError_t e = hipSuccess;
    if (ix->d_em_table.alloc(table.size()) || ix->d_em_rep_off.alloc(G + 1) || ix->d_em_rep_bytes.alloc(rep_bytes.size()) ||
        ix->d_em_goff.alloc(G + 1) || ix->d_em_gids.alloc(gids.size() + 1))
        e = hipErrorOutOfMemory;
    auto up = [&](void *d, const void *h, size_t bytes) { if (e == hipSuccess && bytes) e = hipMemcpy(d, h, bytes, hipMemcpyHostToDevice); };
    up(ix->d_em_table.p, table.data(), table.size() * sizeof(uint2));
    up(ix->d_em_rep_off.p, rep_off.data(), (G + 1) * 8);
    up(ix->d_em_rep_bytes.p, rep_bytes.data(), rep_bytes.size());
    up(ix->d_em_goff.p, goff.data(), (G + 1) * 4);
    up(ix->d_em_gids.p, gids.data(), gids.size() * 4);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        ix->d_em_table.release(); ix->d_em_rep_off.release(); ix->d_em_rep_bytes.release(); ix->d_em_goff.release(); ix->d_em_gids.release();
        return;
    }
    ix->em_groups = G;
    ix->em_bits = bits;
    ix->em_hash_mask = g_em_hash_mask;
    ix->h_em_goff = std::move(goff);
    ix->h_em_gids = std::move(gids);
}
// ... from the sequences alone (rtx_index_create_from_sequences): what Tree::new's map would hold
static void build_exact_table_from_sequences(rtx_index *ix, uint64_t n_refs, const uint8_t *seq_bytes, const uint64_t *seq_off) {
    std::unordered_map<std::string_view, std::vector<uint32_t>, BytesHash> map;
    map.reserve(n_refs * 2);
    for (uint64_t i = 0; i < n_refs; i++)
        map[std::string_view((const char *)seq_bytes + seq_off[i], (size_t)(seq_off[i + 1] - seq_off[i]))].push_back((uint32_t)i);
    std::vector<const std::vector<uint32_t> *> groups;
    groups.reserve(map.size());
    for (const auto &kv : map) groups.push_back(&kv.second);
    build_exact_table(ix, seq_bytes, seq_off, groups);
}

static int create_from_csr(int device, uint64_t n_total, uint64_t ref_lo, uint64_t ref_hi, const uint64_t *cuts,
                           uint32_t n_cuts, const uint64_t *offsets, const uint32_t *postings, uint32_t n_nodes,
                           const uint32_t *node_begin, const uint32_t *node_end, const uint32_t *node_first_child,
                           const uint32_t *node_n_children, const uint8_t *node_type, rtx_index **out);

int rtx_index_create(int device, uint64_t n_refs, const uint64_t *offsets, const uint32_t *postings,
                     uint32_t n_nodes, const uint32_t *node_begin, const uint32_t *node_end,
                     const uint32_t *node_first_child, const uint32_t *node_n_children, const uint8_t *node_type,
                     rtx_index **out) {
    if (!offsets) { set_error("rtx_index_create: offsets is null"); return RTX_ERR_INVALID; }
    if (offsets[RTX_NUM_KMERS] && !postings) { set_error("rtx_index_create: postings is null"); return RTX_ERR_INVALID; }
    return create_from_csr(device, n_refs, 0, n_refs, nullptr, 0, offsets, postings, n_nodes, node_begin, node_end,
                           node_first_child, node_n_children, node_type, out);
}

int rtx_index_create_shard(int device, uint64_t n_refs_total, uint64_t ref_lo, uint64_t ref_hi, const uint64_t *shard_cuts,
                           uint32_t n_cuts, const uint64_t *offsets, const uint32_t *postings, uint32_t n_nodes,
                           const uint32_t *node_begin, const uint32_t *node_end, const uint32_t *node_first_child,
                           const uint32_t *node_n_children, const uint8_t *node_type, rtx_index **out) {
    if (!offsets) { set_error("rtx_index_create_shard: offsets is null"); return RTX_ERR_INVALID; }
    if (offsets[RTX_NUM_KMERS] && !postings) { set_error("rtx_index_create_shard: postings is null"); return RTX_ERR_INVALID; }
    return create_from_csr(device, n_refs_total, ref_lo, ref_hi, shard_cuts, n_cuts, offsets, postings, n_nodes, node_begin,
                           node_end, node_first_child, node_n_children, node_type, out);
}

// RTX_DEFAULT_SEGMENT_CLASSES (rtx_set_default_option): 0 = every segment is read densely (A/B measurements)
static uint64_t g_seg_classes = 1;

// Classifies every (row, tile) segment of the finished bitmap as empty / dense / sparse and writes the slots of the
// sparse ones (rtx_segments.hip).  Slots are numbered in (row, tile) order: deterministic.
static int build_segments(rtx_index *ix) {
    const uint32_t n_rows1 = ix->n_rows + 1, nt = ix->ntiles;
    const uint32_t ss = (nt + 3u) & ~3u;  // seginfo rows padded to whole uint4
    ix->seg_stride = ss;
    const size_t n = (size_t)n_rows1 * nt;
    const bool sparse_on = g_seg_classes != 0, empty_on = g_seg_classes != 0;
    DevBuf<uint16_t> d_pop;
    int rc;
    if ((rc = d_pop.alloc(n)) || (rc = ix->d_seginfo.alloc((size_t)n_rows1 * ss))) return rc;
    launch_seg_popcount(ix->stream, ix->d_bitmap.p, ix->stride_bytes, n_rows1, nt, d_pop.p);
    RTX_HIP(hipGetLastError());
    RTX_HIP(hipStreamSynchronize(ix->stream));
    std::vector<uint16_t> pop(n);
    RTX_HIP(hipMemcpy(pop.data(), d_pop.p, n * 2, hipMemcpyDeviceToHost));
    std::vector<uint32_t> info((size_t)n_rows1 * ss, 0u);
    uint64_t slots = 0;
    const bool last_full = ix->stride_bytes % 1024u == 0;  // hit_count's byte counters and row images want 64-lane tiles
    for (uint32_t r = 0; r < n_rows1; r++)
        for (uint32_t t = 0; t < nt; t++) {
            const uint32_t c = pop[(size_t)r * nt + t];
            uint32_t &o = info[(size_t)r * ss + t];
            const bool full_tile = t + 1 < nt || last_full;
            if (c == 0) o = empty_on ? 0u : 1u;
            else if (c <= kSegSparseMax && sparse_on && full_tile) o = (uint32_t)(2 + slots++);
            else o = 1u;
        }
    if (slots > 0x7FFFFFF0ull) { set_error("too many sparse segments"); return RTX_ERR_INVALID; }
    // many tiles: the classes as bit tables per block of 64 tiles (kmer_extract transposes 64 rows x 64 tiles at a time)
    ix->seg_blocks = nt > 12 ? (nt + 63) / 64 : 0;
    if (ix->seg_blocks) {
        const uint32_t nb = ix->seg_blocks;
        std::vector<unsigned long long> dbits((size_t)n_rows1 * nb, 0), sbits((size_t)n_rows1 * nb, 0);
        std::vector<uint32_t> sbase((size_t)n_rows1 * nb, 0);
        for (uint32_t r = 0; r < n_rows1; r++)
            for (uint32_t b = 0; b < nb; b++) {
                bool first = true;
                for (uint32_t t = b * 64; t < nt && t < b * 64 + 64; t++) {
                    const uint32_t o = info[(size_t)r * ss + t];
                    if (o == 1u) dbits[(size_t)r * nb + b] |= 1ull << (t & 63u);
                    else if (o >= 2u) {
                        sbits[(size_t)r * nb + b] |= 1ull << (t & 63u);
                        if (first) { sbase[(size_t)r * nb + b] = o - 2u; first = false; }
                    }
                }
            }
        if ((rc = ix->d_seg_dbits.alloc(dbits.size())) || (rc = ix->d_seg_sbits.alloc(sbits.size())) || (rc = ix->d_seg_sbase.alloc(sbase.size()))) return rc;
        RTX_HIP(hipMemcpy(ix->d_seg_dbits.p, dbits.data(), dbits.size() * 8, hipMemcpyHostToDevice));
        RTX_HIP(hipMemcpy(ix->d_seg_sbits.p, sbits.data(), sbits.size() * 8, hipMemcpyHostToDevice));
        RTX_HIP(hipMemcpy(ix->d_seg_sbase.p, sbase.data(), sbase.size() * 4, hipMemcpyHostToDevice));
    }
    ix->n_seg_slots = slots;
    if ((rc = ix->d_segslots.alloc((slots ? slots : 1) * kSegSlotEntries))) return rc;
    RTX_HIP(hipMemset(ix->d_segslots.p, 0xFF, (slots ? slots : 1) * kSegSlotEntries * 2));
    RTX_HIP(hipMemcpy(ix->d_seginfo.p, info.data(), info.size() * 4, hipMemcpyHostToDevice));
    if (slots) {
        launch_seg_emit(ix->stream, ix->d_bitmap.p, ix->stride_bytes, n_rows1, nt, ix->d_seginfo.p, ss, ix->d_segslots.p);
        RTX_HIP(hipGetLastError());
        RTX_HIP(hipStreamSynchronize(ix->stream));
    }
    return RTX_OK;
}

static int create_from_csr(int device, uint64_t n_total, uint64_t ref_lo, uint64_t ref_hi, const uint64_t *cuts,
                           uint32_t n_cuts, const uint64_t *offsets, const uint32_t *postings, uint32_t n_nodes,
                           const uint32_t *node_begin, const uint32_t *node_end, const uint32_t *node_first_child,
                           const uint32_t *node_n_children, const uint8_t *node_type, rtx_index **out) {
    rtx_index *ix = nullptr;
    int rc = create_common(device, n_total, ref_lo, ref_hi, cuts, n_cuts, n_nodes, node_begin, node_end, node_first_child,
                           node_n_children, node_type, &ix);
    if (rc) return rc;
    auto fail = [&](int code) { delete ix; return code; };
    // ---- bitmap index: one row of (local) n_refs bits per posting list that is non-empty in this shard
    {
        std::vector<uint32_t> row_of(RTX_NUM_KMERS, kEmptyRow);
        std::vector<uint32_t> list_len(RTX_NUM_KMERS, 0);
        uint32_t nr = 0;
        for (uint32_t k = 0; k < RTX_NUM_KMERS; k++) {
            if (offsets[k + 1] < offsets[k]) { set_error("offsets not monotone at k-mer %u", k); return fail(RTX_ERR_INVALID); }
            const uint64_t l0 = offsets[k + 1] - offsets[k];
            if (l0 > n_total) { set_error("posting list %u longer than n_refs", k); return fail(RTX_ERR_INVALID); }
            // lists are sorted (tree.rs:134-137): the shard's part is a contiguous run
            const uint32_t *b = postings + offsets[k], *e = postings + offsets[k + 1];
            const uint64_t l = l0 ? (uint64_t)(std::lower_bound(b, e, (uint32_t)ref_hi) - std::lower_bound(b, e, (uint32_t)ref_lo)) : 0;
            list_len[k] = (uint32_t)l;
            if (l) row_of[k] = nr++;
        }
        ix->n_rows = nr;
        const size_t words = (size_t)(nr + 1) * (ix->stride_bytes / 4);
        const uint64_t total = offsets[RTX_NUM_KMERS];
        DevBuf<uint64_t> d_off;
        DevBuf<uint32_t> d_post;
        if ((rc = ix->d_bitmap.alloc(words)) || (rc = ix->d_row_of.alloc(RTX_NUM_KMERS)) ||
            (rc = ix->d_list_len.alloc(RTX_NUM_KMERS)) || (rc = d_off.alloc(RTX_NUM_KMERS + 1)) ||
            (rc = d_post.alloc(total)))
            return fail(rc);
        hipError_t e = hipMemset(ix->d_bitmap.p, 0, words * 4);
        if (e == hipSuccess) e = hipMemcpy(ix->d_row_of.p, row_of.data(), RTX_NUM_KMERS * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(ix->d_list_len.p, list_len.data(), RTX_NUM_KMERS * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_off.p, offsets, (RTX_NUM_KMERS + 1) * 8, hipMemcpyHostToDevice);
        if (e == hipSuccess && total) e = hipMemcpy(d_post.p, postings, total * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            launch_bitmap_build(ix->stream, d_off.p, d_post.p, ix->d_row_of.p, ix->d_bitmap.p, ix->stride_bytes / 4, nr + 1,
                                (uint32_t)ref_lo, (uint32_t)ref_hi);
            e = hipStreamSynchronize(ix->stream);
        }
        if (e != hipSuccess) { set_error("bitmap build failed: %s", hipGetErrorString(e)); return fail(RTX_ERR_HIP); }
        if (prepare_union_bitmap(ix)) {
            launch_bitmap_build(ix->stream, d_off.p, d_post.p, ix->d_row_of.p, ix->d_ubitmap.p, ix->u_stride_bytes / 4, nr + 1, (uint32_t)ref_lo,
                                (uint32_t)ref_hi, kPruneShift);
            if (hipStreamSynchronize(ix->stream) != hipSuccess) { (void)hipGetLastError(); ix->d_ubitmap.release(); }
        }
    }
    if ((rc = build_segments(ix))) return fail(rc);
    *out = ix;
    return RTX_OK;
}

// Union bitmap of the tile pruning (rtx_prune.hip): the bitmap of the database with one column per block of 2^kPruneShift
// references.  Same rows as d_bitmap.  Only for whole databases of some size (8 tiles or more); a failure to allocate
// leaves the handle without it (no pruning).  Sizes first, then one of the two builders below fills it.
static bool prepare_union_bitmap(rtx_index *ix) {
    if (ix->ntiles < RTX_PRUNE_MIN_TILES) return false;  // (a reference shard gets one too: it prunes with the threshold of the whole database, rtx_shard_bounds)
    ix->u_nblocks = (ix->n_refs + (1ull << kPruneShift) - 1) >> kPruneShift;
    // the best-block key of the bounds pass packs the block into 20 bits (bounds_epilogue, prune_kernel): beyond 2^20 blocks (67 M
    // references on one handle) block ids would alias and the threshold would come from the wrong block -- such a handle counts every tile
    if (ix->u_nblocks > 0xFFFFFull) return false;
    ix->u_ntiles = (uint32_t)((ix->u_nblocks + 8191) / 8192);
    ix->u_stride_bytes = ix->u_ntiles * 1024u;
    const size_t words = (size_t)(ix->n_rows + 1) * (ix->u_stride_bytes / 4);
    if (ix->d_ubitmap.alloc(words)) { ix->d_ubitmap.release(); return false; }
    // on the handle's stream: the builder kernel that follows must not start before the zeroes are in (a hipMemset on the null
    // stream is not ordered with a non-blocking stream)
    if (hipMemsetAsync(ix->d_ubitmap.p, 0, words * 4, ix->stream) != hipSuccess) { (void)hipGetLastError(); ix->d_ubitmap.release(); return false; }
    return true;
}

// Locator table of the processing order (rtx_cluster.hip) from the reference sequences already on the device.
// A scheduling aid only: if it cannot be built (memory) the handle works without it.
static void build_locator(rtx_index *ix, const uint8_t *d_seq, const uint64_t *d_off, uint64_t n_refs) {
    if (ix->n_refs != ix->n_total || n_refs < 256) return;  // whole-database handles of some size only
    DevBuf<uint32_t> d_cnt;
    if (ix->d_loc_table.alloc(kLocTableEntries) || d_cnt.alloc(kLocTableEntries)) { ix->d_loc_table.release(); return; }
    hipError_t e = hipMemsetAsync(ix->d_loc_table.p, 0xFF, (size_t)kLocTableEntries * 4, ix->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_cnt.p, 0, (size_t)kLocTableEntries * 4, ix->stream);
    if (e == hipSuccess) {
        launch_loc_mark(ix->stream, d_seq, d_off, n_refs, ix->d_loc_table.p, d_cnt.p);
        launch_loc_finish(ix->stream, ix->d_loc_table.p, d_cnt.p);
        e = hipStreamSynchronize(ix->stream);
    }
    if (e != hipSuccess) { (void)hipGetLastError(); ix->d_loc_table.release(); }
}

static thread_local bool g_from_tree = false;  // rtx_index_create_from_tree -> _from_sequences on the same thread

// Index build on the GPU from the encoded reference sequences in lineage-sorted order
// (the k-mer map of Tree::new, tree.rs:114-123,134-137, without ever materialising posting lists).
int rtx_index_create_from_sequences(int device, uint64_t n_refs, const uint8_t *seq_bytes, const uint64_t *seq_off,
                                    uint32_t n_nodes, const uint32_t *node_begin, const uint32_t *node_end,
                                    const uint32_t *node_first_child, const uint32_t *node_n_children,
                                    const uint8_t *node_type, rtx_index **out) {
    if (!seq_off || (!seq_bytes && n_refs && seq_off[n_refs])) { set_error("rtx_index_create_from_sequences: null sequences"); return RTX_ERR_INVALID; }
    rtx_index *ix = nullptr;
    int rc = create_common(device, n_refs, 0, n_refs, nullptr, 0, n_nodes, node_begin, node_end, node_first_child,
                           node_n_children, node_type, &ix);
    if (rc) return rc;
    auto fail = [&](int code) { delete ix; return code; };
    const uint64_t total = seq_off[n_refs] - seq_off[0];
    DevBuf<uint8_t> d_seq;
    DevBuf<uint64_t> d_off;
    DevBuf<uint32_t> d_present;
    if ((rc = d_seq.alloc(total + 16)) || (rc = d_off.alloc(n_refs + 1)) || (rc = d_present.alloc(2048)) ||
        (rc = ix->d_row_of.alloc(RTX_NUM_KMERS)) || (rc = ix->d_list_len.alloc(RTX_NUM_KMERS)))
        return fail(rc);
    std::vector<uint64_t> off0(n_refs + 1);
    for (uint64_t i = 0; i <= n_refs; i++) off0[i] = seq_off[i] - seq_off[0];
    hipError_t e = hipMemcpy(d_seq.p, seq_bytes + seq_off[0], total, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_off.p, off0.data(), (n_refs + 1) * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(d_present.p, 0, 2048 * 4);
    if (e == hipSuccess) {
        launch_ref_kmer_mark(ix->stream, d_seq.p, d_off.p, n_refs, d_present.p);
        e = hipStreamSynchronize(ix->stream);
    }
    std::vector<uint32_t> present(2048, 0), row_of(RTX_NUM_KMERS, kEmptyRow);
    if (e == hipSuccess) e = hipMemcpy(present.data(), d_present.p, 2048 * 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { set_error("k-mer marking failed: %s", hipGetErrorString(e)); return fail(RTX_ERR_HIP); }
    uint32_t nr = 0;
    for (uint32_t k = 0; k < RTX_NUM_KMERS; k++)
        if (present[k >> 5] & (1u << (k & 31u))) row_of[k] = nr++;
    ix->n_rows = nr;
    const size_t words = (size_t)(nr + 1) * (ix->stride_bytes / 4);
    if ((rc = ix->d_bitmap.alloc(words))) return fail(rc);
    e = hipMemset(ix->d_bitmap.p, 0, words * 4);
    if (e == hipSuccess) e = hipMemcpy(ix->d_row_of.p, row_of.data(), RTX_NUM_KMERS * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        launch_ref_bitmap_set(ix->stream, d_seq.p, d_off.p, n_refs, ix->d_row_of.p, ix->d_bitmap.p, ix->stride_bytes / 4, nr + 1);
        launch_row_popcount(ix->stream, ix->d_row_of.p, ix->d_bitmap.p, ix->stride_bytes / 4, nr + 1, ix->d_list_len.p);
        e = hipStreamSynchronize(ix->stream);
    }
    if (e != hipSuccess) { set_error("bitmap build from sequences failed: %s", hipGetErrorString(e)); return fail(RTX_ERR_HIP); }
    if ((rc = build_segments(ix))) return fail(rc);
    build_locator(ix, d_seq.p, d_off.p, n_refs);
    if (!g_from_tree) build_exact_table_from_sequences(ix, n_refs, seq_bytes, seq_off);  // (from a tree: its map is reused, below)
    if (prepare_union_bitmap(ix)) {
        launch_ref_bitmap_set(ix->stream, d_seq.p, d_off.p, n_refs, ix->d_row_of.p, ix->d_ubitmap.p, ix->u_stride_bytes / 4, nr + 1, kPruneShift);
        if (hipStreamSynchronize(ix->stream) != hipSuccess) { (void)hipGetLastError(); ix->d_ubitmap.release(); }
    }
    *out = ix;
    return RTX_OK;
}

static void exact_table_from_tree(rtx_index *ix, const rtx_tree *tree) {
    if (tree->seq_off.size() != tree->num_tips + 1) return;
    std::vector<const std::vector<uint32_t> *> groups;
    groups.reserve(tree->sequences.size());
    for (const auto &kv : tree->sequences)
        if (!kv.second.empty()) groups.push_back(&kv.second);
    build_exact_table(ix, tree->seq_bytes.data(), tree->seq_off.data(), groups);
}

int rtx_index_create_from_tree(int device, const rtx_tree *tree, rtx_index **out) {
    if (!tree || !out) { set_error("null argument"); return RTX_ERR_INVALID; }
    const FlatNodes &f = tree->flat;
    if (tree->csr_off.empty()) {  // tree built without the host k-mer map: build the bitmaps on the GPU
        g_from_tree = true;   // (this thread's call below: the exact-match table comes from the tree's map, not from a second pass over the sequences)
        const int rc = rtx_index_create_from_sequences(device, tree->num_tips, tree->seq_bytes.data(), tree->seq_off.data(), f.size(),
                                                       f.begin.data(), f.end.data(), f.first_child.data(), f.n_children.data(),
                                                       f.type.data(), out);
        g_from_tree = false;
        if (rc == RTX_OK) exact_table_from_tree(*out, tree);
        return rc;
    }
    int rc = rtx_index_create(device, tree->num_tips, tree->csr_off.data(), tree->postings.data(), f.size(), f.begin.data(),
                              f.end.data(), f.first_child.data(), f.n_children.data(), f.type.data(), out);
    if (rc != RTX_OK) return rc;
    exact_table_from_tree(*out, tree);
    // the tree holds the sequences (Tree.sequences, for the exact-match lookup): the locator table of the processing order
    const uint64_t n = tree->num_tips;
    if (n >= 256 && tree->seq_off.size() == n + 1) {
        rtx_index *ix = *out;
        const uint64_t total = tree->seq_off[n] - tree->seq_off[0];
        DevBuf<uint8_t> d_seq;
        DevBuf<uint64_t> d_off;
        if (!d_seq.alloc(total + 16) && !d_off.alloc(n + 1)) {
            std::vector<uint64_t> off0(n + 1);
            for (uint64_t i = 0; i <= n; i++) off0[i] = tree->seq_off[i] - tree->seq_off[0];
            hipError_t e = hipMemcpy(d_seq.p, tree->seq_bytes.data() + tree->seq_off[0], total, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(d_off.p, off0.data(), (n + 1) * 8, hipMemcpyHostToDevice);
            if (e == hipSuccess) build_locator(ix, d_seq.p, d_off.p, n);
            else (void)hipGetLastError();
        }
    }
    return RTX_OK;
}

void rtx_index_destroy(rtx_index *index) {
    if (!index) return;
    (void)hipSetDevice(index->device);
    delete index;
}

uint64_t rtx_index_num_refs(const rtx_index *index) { return index ? index->n_total : 0; }
uint64_t rtx_index_device_bytes(const rtx_index *index) {
    if (!index) return 0;
    return index->d_seg_dbits.n * 8 + index->d_seg_sbits.n * 8 + index->d_seg_sbase.n * 4 + index->d_seginfo.n * 4 + index->d_segslots.n * 2 + index->d_bitmap.n * 4 + index->d_row_of.n * 4 + index->d_list_len.n * 4 + index->d_loc_table.n * 4 + index->d_ubitmap.n * 4 + index->d_em_table.n * 8 + index->d_em_rep_off.n * 8 + index->d_em_rep_bytes.n + index->d_em_goff.n * 4 + index->d_em_gids.n * 4 + index->d_lnfact.n * 8 +
           index->d_noderec.n * 16 + index->d_bnd_bits.n + index->d_bnd_rank.n * 4;
}
int rtx_index_set_batch(rtx_index *index, uint32_t sub_batch) {
    if (!index) { set_error("null index handle"); return RTX_ERR_INVALID; }
    index->uploaded = index->ran = index->synced = false;  // as RTX_OPT_SUB_BATCH: read at the next upload
    index->sub_batch_req = sub_batch;
    return RTX_OK;
}

int rtx_set_default_option(int option, uint64_t value) {
    if (option == RTX_DEFAULT_SEGMENT_CLASSES) { g_seg_classes = value ? 1 : 0; return RTX_OK; }
    if (option == RTX_DEFAULT_EXACT_HASH_MASK) { g_em_hash_mask = value ? value : ~0ull; return RTX_OK; }
    set_error("rtx_set_default_option: unknown option %d", option);
    return RTX_ERR_INVALID;
}

int rtx_index_set_option(rtx_index *index, int option, uint64_t value) {
    if (!index) { set_error("null index handle"); return RTX_ERR_INVALID; }
    // Options that shape the per-batch workspace (count format, scratch of the tile pruning, sub-batch size, probability tables) are
    // read when a batch is uploaded: setting one of them drops the uploaded batch, so that the next rtx_batch_run cannot work on
    // buffers sized for another layout (it fails with RTX_ERR_STATE until the batch is uploaded again).
    switch (option) {
        case RTX_OPT_SUB_BATCH: case RTX_OPT_PACKED_COUNTS: case RTX_OPT_HIT_PAIR: case RTX_OPT_TILE_PRUNE: case RTX_OPT_PROB_MODE:
            index->uploaded = index->ran = index->synced = false;
            break;
        default: break;
    }
    switch (option) {
        case RTX_OPT_SUB_BATCH: index->sub_batch_req = (uint32_t)value; return RTX_OK;
        case RTX_OPT_STAGE_TIMING:
            index->stage_timing = value ? 1u : 0u;
            return RTX_OK;
        case RTX_OPT_DEBUG_TAPS:
            index->debug_taps = value ? 1u : 0u;
            return RTX_OK;
        case RTX_OPT_SHARD_PRUNE:
            index->uploaded = index->ran = index->synced = false;
            index->shard_prune_opt = value ? 1u : 0u;
            return RTX_OK;
        case RTX_OPT_DEVICE_EXACT:
            index->uploaded = index->ran = index->synced = false;  // decided at the upload
            index->dev_exact_opt = value ? 1u : 0u;
            return RTX_OK;
        case RTX_OPT_CLUSTER:
            index->cluster = value ? 1u : 0u;
            return RTX_OK;
        case RTX_OPT_PACKED_COUNTS:
            index->packed_opt = value ? 1u : 0u;
            return RTX_OK;
        case RTX_OPT_HIT_PAIR:
            index->pair_opt = value ? 1u : 0u;
            return RTX_OK;
        case RTX_OPT_TILE_SKIP:
            index->tile_skip = value ? 1u : 0u;
            return RTX_OK;
        case RTX_OPT_LOCATOR:
            index->locator_opt = value ? 1u : 0u;
            return RTX_OK;
        case RTX_OPT_TILE_PRUNE:
            index->prune_opt = value ? 1u : 0u;
            return RTX_OK;
        case RTX_OPT_PROB_MODE:
            if (value > 2) break;
            index->prob_mode = (int)value;
            return RTX_OK;
        default: break;
    }
    set_error("rtx_index_set_option: unknown option %d / value %llu", option, (unsigned long long)value);
    return RTX_ERR_INVALID;
}

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
    for (uint64_t q = 0; q < n_queries; q++) {
        if (base_off[q + 1] < base_off[q]) { set_error("base_off not monotone at query %llu", (unsigned long long)q); return RTX_ERR_INVALID; }
        max_len = std::max(max_len, base_off[q + 1] - base_off[q]);
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
    // t <= min(len - 7, 65536); t == 65536 would trip the u16 assert of raxtax.rs:56
    const uint64_t tmax = in.max_len >= 8 ? in.max_len - 7 : 1;
    if ((rc = prepare_workspace(ix, in.n_q, tmax, in.max_len))) return rc;
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
    RTX_HIP(hipStreamSynchronize(ix->stream));
    ix->synced = true;
    return RTX_OK;
}

// Finalises positions [pa, pb) on up to nt threads; returns the number of rows they produced.
static uint64_t finalise_mt(rtx_index *ix, uint64_t pa, uint64_t pb, uint64_t row_base, unsigned nt) {
    nt = rtx::host_threads(nt);  // this process's share of the host's CPUs (cgroup quota, ranks per host)
    if (pb - pa < 1024) nt = 1;
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

static int size_host_results(rtx_index *ix, rtx_index::HostRes &hr, uint64_t nq, uint64_t arena_rows) {
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
static int copy_results(rtx_index *ix, uint64_t q0, uint64_t n, uint64_t r0, uint64_t r1, hipStream_t cs) {
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
static int download_streamed(rtx_index *ix, rtx_index::HostRes &hr, bool *done, uint64_t *nrows_out) {
    *done = false;
    const uint32_t n_sub = ix->n_sub_run;
    if (!ix->stream_dl || n_sub < 2 || hipEventQuery(ix->ev_sub[n_sub - 1]) == hipSuccess) return RTX_OK;
    const uint64_t nq = ix->n_q;
    int rc = size_host_results(ix, hr, nq, ix->arena_cap);
    if (rc) return rc;
    uint64_t prev = 0, nrows = 0;
    for (uint32_t sb = 0; sb < n_sub; sb++) {
        RTX_HIP(hipEventSynchronize(ix->ev_sub[sb]));
        const uint64_t cur = ix->h_cursor_sub[sb];
        if (cur > ix->arena_cap) return RTX_OK;  // overflow: bulk path
        const uint64_t q0 = (uint64_t)sb * ix->sub_batch, n = std::min<uint64_t>(ix->sub_batch, nq - q0);
        if ((rc = copy_results(ix, q0, n, prev, cur, ix->copy_stream))) return rc;
        RTX_HIP(hipStreamSynchronize(ix->copy_stream));
        // one thread finalises 8192 queries in ~1.4 ms, about what the device needs for the next sub-batch: with a short
        // last sub-batch the host would still be busy with the one before it when the device is done
        nrows += finalise_mt(ix, q0, q0 + n, nrows, sb + 1 == n_sub ? 16 : 8);
        prev = cur;
    }
    RTX_HIP(hipStreamSynchronize(ix->stream));
    ix->synced = true;
    uint32_t flags = 0;
    RTX_HIP(hipMemcpy(&flags, ix->d_flags.p, 4, hipMemcpyDeviceToHost));
    if (flags & 2u) { set_error("lineage walk exceeded its row/depth bounds (internal error)"); return RTX_ERR_HIP; }
    if (flags & 1u) return RTX_OK;
    *nrows_out = nrows;
    *done = true;
    return RTX_OK;
}

int rtx_batch_download(rtx_index *ix, rtx_result_view *out) {
    int rc = bind(ix);
    if (rc) return rc;
    if (!ix->ran || !out) { set_error("rtx_batch_download before rtx_batch_run"); return RTX_ERR_STATE; }
    const uint64_t nq = ix->n_q;
    ix->res_set ^= 1u;
    rtx_index::HostRes &hr = ix->host_res[ix->res_set];
    bool streamed = false;
    uint64_t nrows = 0;
    if ((rc = download_streamed(ix, hr, &streamed, &nrows))) return rc;
    if (!streamed) {
        unsigned long long cursor = 0;
        for (int attempt = 0;; attempt++) {
            RTX_HIP(hipStreamSynchronize(ix->stream));
            ix->synced = true;
            uint32_t flags = 0;
            RTX_HIP(hipMemcpy(&flags, ix->d_flags.p, 4, hipMemcpyDeviceToHost));
            RTX_HIP(hipMemcpy(&cursor, ix->d_cursor.p, 8, hipMemcpyDeviceToHost));
            if (flags & 2u) { set_error("lineage walk exceeded its row/depth bounds (internal error)"); return RTX_ERR_HIP; }
            if (!(flags & 1u)) break;
            if (attempt >= 2) { set_error("result arena overflow persists"); return RTX_ERR_HIP; }
            // arena too small: grow to what this run asked for and repeat the (deterministic) run
            const uint64_t want = cursor + 4096;
            if ((rc = ix->d_arena.alloc(want))) return rc;
            ix->arena_cap = want;
            if (ix->n_refs != ix->n_total) {  // a sharded run is driven by the caller: ask it to repeat
                set_error("result arena overflow: repeat the sharded run (the arena has been enlarged)");
                return RTX_ERR_STATE;
            }
            if ((rc = enqueue_batch(ix, ix->last_flags))) return rc;
        }
        if ((rc = size_host_results(ix, hr, nq, cursor)) || (rc = copy_results(ix, 0, nq, 0, cursor, ix->stream))) return rc;
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
    {   // the exact matches the device found belong to this download (same alternation as the result sets)
        rtx_index::HostExact &hx = ix->host_exact[ix->res_set];
        hx.valid = hx.csr_valid = false;
        if (ix->dev_exact_used) {
            hx.grp.resize(nq);
            RTX_HIP(hipMemcpy(hx.grp.data(), ix->d_exact_grp.p, nq * 4, hipMemcpyDeviceToHost));
            hx.valid = true;
        }
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
    const uint32_t n_sub = (uint32_t)((ix->n_q + ix->sub_batch - 1) / ix->sub_batch);
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
    return enqueue_walk(ix, b, prefix_global, b.s);
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

int rtx_batch_stage_times(rtx_index *ix, float ms[RTX_NUM_STAGES], uint32_t launches[RTX_NUM_STAGES]) {
    int rc = bind(ix);
    if (rc) return rc;
    if (!ix->synced) { set_error("rtx_batch_stage_times: batch not synchronised"); return RTX_ERR_STATE; }
    for (int s = 0; s < RTX_NUM_STAGES; s++) { ms[s] = 0.f; launches[s] = 0; }
    for (uint32_t sb = 0; sb < ix->n_sub_last; sb++)
        for (int s = 0; s < RTX_NUM_STAGES; s++) {
            if (s == RTX_STAGE_EXACT_MATCH) {
                if (sb != 0 || !ix->dev_exact_used || !ix->stage_timing) continue;  // one launch per run
            } else if (s == RTX_STAGE_ORDER) {
                if (sb != 0 || !ix->stage_timing) continue;  // once per run
            } else if (s == RTX_STAGE_PAIR_UNION) {
                if (!ix->pair_used || !ix->stage_timing) continue;
            } else if (s == RTX_STAGE_TILE_BOUNDS || s == RTX_STAGE_TILE_PRUNE ? !ix->prune_used : (s != RTX_STAGE_HIT_COUNT && !ix->stage_timing)) continue;  // events were not recorded
            float t = 0.f;
            RTX_HIP(hipEventElapsedTime(&t, ix->events[((size_t)sb * RTX_NUM_STAGES + s) * 2],
                                        ix->events[((size_t)sb * RTX_NUM_STAGES + s) * 2 + 1]));
            ms[s] += t;
            launches[s]++;
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
    for (uint64_t q = 0; q < ix->n_q; q++) {
        h += ix->h_hq[q];
        b += (uint64_t)ix->h_nrows_all[q] * row_bytes;
    }
    if (ix->pair_used) {  // rows were loaded once per pair of queries: the union rows every wave counted
        const uint32_t n_sub = (uint32_t)((ix->n_q + ix->sub_batch - 1) / ix->sub_batch);
        const size_t ng = (size_t)n_sub * ix->groups_per_sub;
        std::vector<uint32_t> gr(2 * ng);
        RTX_HIP(hipMemcpy(gr.data(), ix->d_group_rows.p, gr.size() * 4, hipMemcpyDeviceToHost));
        b = 0;
        for (size_t g = 0; g < ng; g++) b += (uint64_t)gr[g] * row_bytes;
        if (ix->prune_used) {  // + the rows of the union bitmap the bounds pass of the tile pruning loaded (the same kernel)
            const uint64_t urow_bytes = ((ix->u_nblocks + 7) / 8 + ix->u_ntiles - 1) / ix->u_ntiles;
            for (size_t g = ng; g < 2 * ng; g++) b += (uint64_t)gr[g] * urow_bytes;
        }
    }
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
    if (ix->pair_used && ix->prune_used) {
        const uint32_t n_sub = (uint32_t)((ix->n_q + ix->sub_batch - 1) / ix->sub_batch);
        const size_t ng = (size_t)n_sub * ix->groups_per_sub;
        std::vector<uint32_t> gr(ng);
        RTX_HIP(hipMemcpy(gr.data(), ix->d_group_rows.p + ng, ng * 4, hipMemcpyDeviceToHost));
        const uint64_t urow_bytes = ((ix->u_nblocks + 7) / 8 + ix->u_ntiles - 1) / ix->u_ntiles;
        for (size_t g = 0; g < ng; g++) bounds += (uint64_t)gr[g] * urow_bytes;
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
    const uint64_t last0 = (ix->n_q - 1) / ix->sub_batch * ix->sub_batch;
    if (query >= ix->n_q) { set_error("debug tap: query %llu out of range", (unsigned long long)query); return RTX_ERR_INVALID; }
    const uint64_t pos = ix->h_inv[query];  // position in the processing order (valid once the stream is synchronised)
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
    const uint32_t n_sub = (uint32_t)((ix->n_q + ix->sub_batch - 1) / ix->sub_batch);
    SubBatch b = sub_batch_of(ix, n_sub - 1, false);
    b.set = ix->last_set;
    ix->dbg_full_run = true;
    int rc = enqueue_kmer(ix, b, ix->stream);
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
    if (z) RTX_HIP(hipMemcpy(z, ix->d_z.p + ix->h_inv[query], 8, hipMemcpyDeviceToHost));
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
    if (z) RTX_HIP(hipMemcpy(z, ix->d_z.p + ix->h_inv[query], 8, hipMemcpyDeviceToHost));
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
    if (counts) {
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
    std::memset(out, 0, 80);
    if (!ix->prune_used || !ix->d_prune_stats.p) return RTX_OK;
    RTX_HIP(hipStreamSynchronize(ix->stream));
    unsigned long long h[kPruneStatCopies * 16];
    RTX_HIP(hipMemcpy(h, ix->d_prune_stats.p, sizeof(h), hipMemcpyDeviceToHost));
    for (uint32_t c = 0; c < kPruneStatCopies; c++) {
        for (uint32_t k = 0; k < 8; k++) out[k] += h[c * 8 + k];
        for (uint32_t k = 0; k < 2; k++) out[8 + k] += h[(kPruneStatCopies + c) * 8 + k];
    }
    return RTX_OK;
}

int rtx_batch_sub_batch(const rtx_index *ix, uint32_t *sub_batch, uint32_t *n_sub) {
    if (!ix) { set_error("null index handle"); return RTX_ERR_INVALID; }
    if (!ix->uploaded || ix->sub_batch == 0) { set_error("rtx_batch_sub_batch: no batch has been uploaded"); return RTX_ERR_STATE; }
    if (sub_batch) *sub_batch = ix->sub_batch;
    if (n_sub) *n_sub = (uint32_t)((ix->n_q + ix->sub_batch - 1) / ix->sub_batch);
    return RTX_OK;
}

int rtx_debug_order(rtx_index *ix, uint32_t *perm) {
    if (!ix || !perm) { set_error("null argument"); return RTX_ERR_INVALID; }
    if (!ix->ran || ix->h_perm.n < ix->n_q) { set_error("rtx_debug_order: no batch has been run"); return RTX_ERR_STATE; }
    RTX_HIP(hipStreamSynchronize(ix->stream));
    std::memcpy(perm, ix->h_perm.data(), (size_t)ix->n_q * 4);
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
    if ((rc = prepare_workspace(ix, 1, std::max<uint64_t>(N, 8), 0))) return rc;
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
    RTX_HIP(hipMemset(ix->d_cursor.p, 0, 8));
    RTX_HIP(hipMemset(ix->d_flags.p, 0, 4));
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
    RTX_HIP(hipGetLastError());
    ix->n_sub_last = 0;
    ix->ran = true;
    ix->last_flags = 0;
    return rtx_batch_download(ix, out);
}

}  // extern "C"
