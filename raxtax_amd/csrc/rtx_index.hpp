// Internals of the device side of libraxtax_hip.so shared by its translation units (rtx_api_*.hip): the index handle, the buffer
// helpers and the functions that sequence the kernels of a sub-batch.  Not part of the ABI (include/raxtax_hip.h is).
//   rtx_api_index.hip     index creation (bitmaps, segment classes, union bitmap, locator, exact-match table), options
//   rtx_api_batch.hip     per-batch workspace, upload (prefetch / activate), the kernel sequence of a sub-batch, rtx_batch_run
//   rtx_api_download.hip  streamed download of the rows the device finalised (rtx_finalise.hip: sort lineage.rs:91-93, local signal lineage.rs:95-102)
//   rtx_api_shard.hip     the staged path of a sharded database (rtx_shard_*)
//   rtx_api_debug.hip     stage times, work counters, parity / debug taps
// There is deliberately no CPU fallback: without a gfx950 device every entry point returns RTX_ERR_NO_DEVICE.
#pragma once

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

namespace rtxi {


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
    // room for `count` elements with the first `keep` of them preserved (the result rows of the sub-batches already copied); grows by doubling
    int grow_keep(size_t count, size_t keep) {
        if (count <= cap) { n = count; return RTX_OK; }
        T *q = nullptr;
        const size_t want = std::max(count + count / 4 + 16, cap * 2);
        if (hipHostMalloc((void **)&q, want * sizeof(T), hipHostMallocDefault) != hipSuccess) {
            set_error("hipHostMalloc(%zu bytes) failed", want * sizeof(T));
            return RTX_ERR_OOM;
        }
        if (p && keep) std::memcpy(q, p, std::min(keep, cap) * sizeof(T));
        if (p) (void)hipHostFree(p);
        p = q;
        cap = want;
        n = count;
        return RTX_OK;
    }
    T *data() { return p; }
    const T *data() const { return p; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
    ~PinBuf() { if (p) (void)hipHostFree(p); }
};

template <class T>
inline void swap_buf(DevBuf<T> &a, DevBuf<T> &b) { std::swap(a.p, b.p); std::swap(a.n, b.n); }
template <class T>
inline void swap_buf(PinBuf<T> &a, PinBuf<T> &b) { std::swap(a.p, b.p); std::swap(a.cap, b.cap); std::swap(a.n, b.n); }

inline uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

#ifndef RTX_PRUNE_MIN_TILES
#define RTX_PRUNE_MIN_TILES 4  // tiles of 8192 references from which on the tile pruning is worth its bounds pass (configs[1], 7 tiles: 5.6 -> 7.8 M queries/s; it was 8 until the bounds pass lost its stores)
#endif

}  // namespace rtxi
using namespace rtxi;

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
    DevBuf<uint2> d_row_len;    // {row_of, list_len} per k-mer (kmer_extract: one gather instead of two)
    // segment classes (rtx_segments.hip): class / sparse slot of every (row, tile) segment, slots of 32 local ids
    DevBuf<uint32_t> d_seginfo, d_seg_sbase, d_segcls;  // (d_segcls: the classes alone, [tile][row] two bits each)
    uint32_t cls_stride = 0;
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
    // per node, for finalise_kernel (rtx_finalise.hip): depth, begin of its range (= the lineage a row reports), and the expected side of its
    // local signal ([node][fin_D] + the level it starts at; lineage.rs:95-98,137-139, rtx_math.hpp: fin_node_expected)
    DevBuf<uint8_t> d_node_depth, d_node_sig0;
    DevBuf<uint32_t> d_node_begin;
    DevBuf<double> d_node_eb;
    uint32_t fin_D = 1;  // levels of the deepest lineage = stride of the confidence arrays of a view
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
    hipEvent_t ev_exact = nullptr;    // behind exact_match_kernel of the run: the download fetches the groups at its START, beside the kernels, not at its tail
    PinBuf<uint32_t> h_flags;         // the run's flags (d_flags), copied behind its last kernel: the download reads them without a round trip of its own
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
    uint32_t self_sample_opt = 1;  // RTX_OPT_PRUNE_SELF_SAMPLE: the verdict of rtx_index_self_sample is honoured
    bool prune_pays = true;        // ... which is: a sample of the database's own references keeps fewer than kSelfSampleOff of its tiles live
    double self_live = -1.0;       // the share of (query, tile) combinations the sample kept live (-1: no sample was taken)
    bool pruning() const { return prune_opt != 0u && (prune_pays || self_sample_opt == 0u); }  // tile pruning is on for this handle (where the batch allows it)
    bool prune_used = false;  // the last run pruned
    bool dbg_full = false;    // ... and the debug taps have recounted the last sub-batch in full since
    bool dbg_full_run = false;  // (the recount in progress: enqueue_hit leaves the pruning out)
    DevBuf<uint32_t> d_ubitmap;  // union bitmap: one column per block of 2^kPruneShift references, tile-major like d_bitmap
    uint32_t u_stride_bytes = 0, u_ntiles = 0;
    uint64_t u_nblocks = 0;
    // the fine union bitmap (blocks of 2^kFineShift = 8 references; databases of kFineMinTiles tiles or more, whole-database handles):
    // second stage of the bounds for the pairs the first stage leaves many live tiles (rtx_hit_pair.hip: launch_fine_bounds)
    DevBuf<uint32_t> d_fbitmap;
    uint32_t f_stride_bytes = 0, f_ntiles = 0;
    uint64_t f_nblocks = 0;
    uint32_t fine_opt = 1;  // RTX_OPT_FINE_BOUNDS
    // the bounds pass in two levels (rtx_bounds2.hip; whole-database handles): blocks of 256 references for every tile (four rows per load
    // instruction), blocks of 64 only for the B-tiles near the query's largest bound (sixteen rows per load instruction)
    DevBuf<uint32_t> d_abitmap;  // [n_atiles][n_rows + 1][64 words]
    DevBuf<uint8_t> d_bbitmap;   // [n_btiles][n_rows + 1][64 bytes]
    uint32_t n_atiles = 0, n_btiles = 0;
    DevBuf<uint8_t> d_cbitmap;   // the database block by block: [ceil(n_refs / 64)][n_rows + 1][8 bytes] (prune_kernel: exact counts of the best block)
    uint32_t two_level_opt = 1;  // RTX_OPT_TWO_LEVEL_BOUNDS
    uint32_t b2_delta[4] = {283u, 205u, 92u, 128u};  // which B-tiles are refined: c_t, c_m, lo, hi in 1/256 (Bounds2Params): dl = 1.105 t - 0.8 max within [0.36 t, 0.5 t]
    bool two_level_used = false;  // the last run's bounds pass was bounds2_kernel (its work accounting counts load instructions of 1 KiB)
    // The HBM diet of the counts buffer (round 6): a class that prunes with the records path holds sub_batch >> diet_shift rows of counts (at
    // least kDietMinRows); a run in which prune_kernel runs out of rows raises bit 2 of d_flags, the download lowers diet_shift and repeats it.
    uint32_t rec_seg_len = 1024;  // records per segment of the records path (RecordRef::seg_len): doubled, up to 8192, when a run's segment overflows
    uint32_t diet_shift = 3;
    bool diet_used = false;      // the class being enqueued lays its counts out in cnt_rows_cur rows
    uint32_t cnt_rows_cur = 0;
    uint32_t rec_opt = 4;   // RTX_OPT_RECORDS: pruned queries with at most this many live tiles take the records path (0: off; at most kRecMaxSlots)
    uint32_t overlap_opt = 1;  // RTX_OPT_OVERLAP: 1 = back half of sub-batch k on a second stream beside the front half of k + 1 (2: three stages)
    uint32_t overlap_used = 0;  // scratch sets the last run used beside each other (0: one stream)
    hipStream_t stream2 = nullptr, stream3 = nullptr, hit_stream = nullptr;  // (hit_stream: where enqueue_hit launched the counting pass)
    std::vector<hipEvent_t> ev_front, ev_back, ev_mid;  // per sub-batch: front half enqueued (on stream), back half done (on stream2)
    bool rec_used = false;  // the last run offered the records path (whole-database handle that prunes, walk fused)
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
    bool packed() const { return packed_opt && planes <= 10; }  // (11 planes -- reads of 1 031 .. 2 054 bases on the pair kernel -- leave u16 counts)
    DevBuf<uint64_t> d_skey_in, d_skey_out;
    DevBuf<uint32_t> d_sidx, d_perm, d_iperm;
    DevBuf<uint8_t> d_sort_tmp;
    // host copies of the order: two sets -- rtx_batch_download_then_run enqueues the next batch (whose order_batch writes a set) while
    // the last sub-batch of the batch before it is still being finalised from the other
    PinBuf<uint32_t> h_perm_[2], h_inv_[2];
    uint32_t perm_cur = 0;              // the set of the batch that ran last
    const uint32_t *dl_perm = nullptr;  // the order of the batch being downloaded (finalise_range)
    PinBuf<uint32_t> &h_perm_now() { return h_perm_[perm_cur]; }
    PinBuf<uint32_t> &h_inv_now() { return h_inv_[perm_cur]; }
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
        uint64_t cls_n[5] = {0, 0, 0, 0, 0}, cls_max[5] = {0, 0, 0, 0, 0};  // queries and longest query per length class (length_class)
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
    uint32_t min_subs = 4;  // RTX_OPT_MIN_SUB_BATCHES: a pruned batch is cut into at least this many sub-batches (the host finalises one while the next run)
    uint64_t ws_key[14] = {0};  // shape and options the workspace was last prepared for (prepare_workspace)
    bool ws_valid = false;
    // ---- length classes of the batch (round 5; round 6: the class of t <= 2047).  t <= length - 7 decides how a query is counted (8 / 10 / 11 / 12 / 16 bit planes, the pair
    // kernel, tile pruning), how its probabilities are computed (memoised tables up to t = 2047, the recurrence kernel in LDS, the same
    // from global memory for reads of tens of kilobases) and how much scratch it needs.  A batch used to take ALL of that from its longest
    // query: one 1 100-base read in a file of COI barcodes moved every query off the fast path.  Now the class leads the sort key of the
    // processing order, every class is cut into sub-batches of its own shape, and the fields above (tmax, strides, planes, sub_batch,
    // use_tables, pair_used, prune_used, rec_used) are those of the class being enqueued (apply_class) -- after a run: of the last one,
    // which is what the taps of the last sub-batch read.  A reference shard and rtx_debug_evaluate run one class.
    struct BatchClass {
        uint64_t pos0 = 0, n = 0, max_len = 0;  // positions [pos0, pos0 + n) of the processing order
        uint32_t tmax = 0, kstride = 0, rstride = 0, hstride = 0, sub_batch = 0, sb0 = 0, n_sub = 0;
        int planes = 10;
        bool use_tables = false, pair = false, prune = false, rec = false, huge = false, will_prune = false;
        bool side = false;  // a handful of queries beside the bulk of the batch: they run FIRST, through a small scratch set of their own (kSideSet)
        bool diet = false;      // rows of the counts buffer are handed out by prune_kernel (behind tile pruning with the records path: HitParams::cnt_row)
        uint32_t cnt_rows = 0;  // rows of counts a sub-batch of the class lays out (diet: a fraction of sub_batch)
    } cls[5];
    uint32_t n_cls = 0;
    int cur_cls = -1;
    uint64_t key_lim[4] = {~0ull, ~0ull, ~0ull, ~0ull};  // sort rank of a query = the number of these lengths it exceeds
    std::vector<uint64_t> sub_q0;   // per sub-batch of the run: first position,
    std::vector<uint32_t> sub_nq;   // queries,
    std::vector<uint8_t> sub_cls;   // class
    uint32_t n_sub_total = 0, sub_batch_max = 0;
    bool any_prune = false;  // some class of the last run pruned (rtx_debug_prune_stats sums over the run)
    DevBuf<double> d_prob_scratch;  // prob_table_kernel's arrays of a class of very long reads (they do not fit LDS)
    struct Scratch {
        DevBuf<uint16_t> d_kmers, d_counts, d_tilemax;
        DevBuf<uint32_t> d_rows, d_t, d_nrows, d_hist, d_order, d_srows, d_nsparse;
        DevBuf<unsigned long long> d_dmask;
        DevBuf<double> d_table_z, d_prefix;
        DevBuf<uint2> d_urec;   // hit_count_pair_kernel: union row lists of the pairs of the sub-batch
        DevBuf<uint32_t> d_nu;
        // tile pruning: the queries counted against the union bitmap (every row dense) leave the largest bound of
        // every tile and the best block (bounds_epilogue); thresholds and the live tiles per pair (prune_kernel)
        DevBuf<uint32_t> d_live, d_best_key;
        DevBuf<uint32_t> d_items;  // [pairs x tiles] the (pair, tile) blocks with a live query | [1] their number | [8] queue per XCD | [pairs] live tiles per pair | [pairs] offsets
        DevBuf<uint16_t> d_tile_ub, d_prune_thr, d_prune_i1;
        DevBuf<uint32_t> d_best;  // [B][kPruneBestWords] reference shards: the candidate for the best block of the database
        DevBuf<uint8_t> d_heavy;        // two-level bounds pass: [B] queries left to the one-level pass (Bounds2Params::heavy)
        DevBuf<uint32_t> d_heavy_items; // ... and the (pair, union tile) items of that pass: [pairs x u_ntiles] | [9]
        DevBuf<uint32_t> d_fine_items;  // fine bounds pass: [pairs x f_ntiles] items | [9] number + XCD queues | [f_ntiles] cursors
        // the records path (RecordRef, rtx_kernels.hpp): per query the live tiles at prune time, the records of each, their number
        DevBuf<uint16_t> d_rec_nslots, d_rec_slots;
        DevBuf<uint32_t> d_rec_cnt, d_rec;
        DevBuf<uint32_t> d_cnt_row, d_cnt_cursor;  // [B] row of the counts buffer per query | [1] rows handed out (HitParams::cnt_row)
        void release_all() {
            d_kmers.release(); d_counts.release(); d_tilemax.release(); d_rows.release(); d_t.release(); d_nrows.release(); d_hist.release();
            d_order.release(); d_srows.release(); d_nsparse.release(); d_dmask.release(); d_table_z.release(); d_prefix.release(); d_urec.release();
            d_nu.release(); d_live.release(); d_best_key.release(); d_items.release(); d_tile_ub.release(); d_prune_thr.release(); d_prune_i1.release();
            d_best.release(); d_fine_items.release(); d_heavy.release(); d_heavy_items.release(); d_rec_nslots.release(); d_rec_slots.release(); d_rec_cnt.release(); d_rec.release();
            d_cnt_row.release(); d_cnt_cursor.release();
        }
    } sc[4];  // 0 .. 2: the sets that alternate (RTX_OPT_OVERLAP, rtx_shard_*); 3: the set of the side classes (a few long reads among barcodes)
    bool staged = false;  // driven with rtx_shard_*: sub-batch sb works in scratch set sb & 1, so that the exchange of one
                          // sub-batch (RCCL, on the caller's stream) can overlap with the counting of the next
    uint32_t last_set = 0;  // scratch set of the last sub-batch (debug taps)
    DevBuf<double> d_probs_dbg;
    DevBuf<uint16_t> d_counts_dbg;
    // ---- per-query results
    DevBuf<uint8_t> d_status;
    DevBuf<uint32_t> d_t_all, d_nrows_all, d_n_rows, d_flags, d_ndist;
    DevBuf<double> d_gs, d_z;
    DevBuf<unsigned long long> d_hq, d_row_start, d_cursor, d_sub_alloc;  // (d_sub_alloc: WalkParams::sub_alloc)
    DevBuf<DevRow> d_arena;
    // the final result arrays (finalise_kernel): the per-query fields in input order, the rows back to back from 0 on (fin_cap = arena_cap rows)
    DevBuf<uint32_t> d_fin_t, d_fin_row_count, d_fin_lineage, d_fin_node, d_fin_depth;
    DevBuf<uint8_t> d_fin_status, d_fin_depth8, d_fin_hund;
    DevBuf<double> d_fin_gs, d_fin_local, d_fin_conf;
    DevBuf<unsigned long long> d_fin_row_begin, d_fin_cursor;
    uint64_t fin_cap = 0;
    PinBuf<unsigned long long> h_fin_sub;  // per sub-batch: the cursor of the final rows behind its finalise launch
    uint64_t arena_cap = 0;
    uint64_t side_base = 0;  // rows [side_base, arena_cap) take the result rows of the side classes (their walks run beside the bulk's: a cursor of their own, d_cursor[1])
    PinBuf<unsigned long long> h_side_base;
    // ---- timing
    std::vector<hipEvent_t> events;  // 2 per (sub-batch, stage)
    uint32_t n_sub_last = 0;
    // ---- host results
    // two alternating sets: the view of download c stays valid while batch c+1 runs and is downloaded
    // (page-locked: the device's final arrays are copied straight into them, rtx_api_download.hip)
    struct HostRes {
        PinBuf<uint32_t> v_row_lineage, v_row_node, v_row_depth;
        PinBuf<uint8_t> v_row_depth8, v_row_hund;
        PinBuf<uint32_t> h_t;
        PinBuf<uint8_t> h_status;
        PinBuf<double> v_row_conf, v_row_local;
        PinBuf<double> h_gs;
        PinBuf<unsigned long long> v_row_begin;  // by query; the rows themselves are in processing order
        PinBuf<uint32_t> v_row_count;
    } host_res[2];
    uint32_t res_set = 0;
    PinBuf<uint32_t> h_nrows_all;
    // streamed download: per sub-batch a snapshot of the arena cursor + an event; rtx_batch_download copies and
    // finalises finished sub-batches on `copy_stream` while later ones are still running
    std::vector<hipEvent_t> ev_sub;
    PinBuf<unsigned long long> h_cursor_sub;
    hipStream_t copy_stream = nullptr;
    uint32_t n_sub_run = 0;
    bool stream_dl = false;
    PinBuf<unsigned long long> h_hq;
    uint32_t stage_timing = 0;  // 0: HIP events around hit_count only; 1: around every kernel

    // ---- run-ahead (RTX_OPT_RUN_AHEAD, set by rtx_raxtax for its chunks): rtx_batch_download_then_run enqueues the staged batch BEFORE the last
    // sub-batch of the batch being downloaded has finished, so that the front half of the next chunk's first sub-batch runs beside the back
    // half of this chunk's last one (the two streams of RTX_OPT_OVERLAP then never drain between chunks).  Everything a batch writes per query
    // and per row, and what its download reads, exists twice: the members of these names are the CURRENT batch's, `alt` holds the other set
    // (swap_result_sets); the scratch sets are shared (a front half waits for the back half that last used its set: ev_set_free).
    struct ResultSet {
        DevBuf<uint8_t> d_status;
        DevBuf<uint32_t> d_t_all, d_nrows_all, d_n_rows, d_flags, d_ndist;
        DevBuf<double> d_gs, d_z;
        DevBuf<unsigned long long> d_hq, d_row_start, d_cursor;
        DevBuf<DevRow> d_arena;
        DevBuf<uint32_t> d_fin_t, d_fin_row_count, d_fin_lineage, d_fin_node, d_fin_depth;
        DevBuf<uint8_t> d_fin_status, d_fin_depth8, d_fin_hund;
        DevBuf<double> d_fin_gs, d_fin_local, d_fin_conf;
        DevBuf<unsigned long long> d_fin_row_begin, d_fin_cursor;
        DevBuf<uint32_t> d_perm, d_iperm, d_exact_grp;
        uint64_t fin_cap = 0, arena_cap = 0, side_base = 0;
        PinBuf<uint32_t> h_flags;
        PinBuf<unsigned long long> h_fin_sub, h_cursor_sub;
        std::vector<hipEvent_t> ev_sub;
        hipEvent_t ev_exact = nullptr, ev_flags = nullptr;
    } alt;
    hipEvent_t ev_flags = nullptr;   // behind the copy of the run's flags into h_flags
    uint32_t run_ahead_opt = 0;      // RTX_OPT_RUN_AHEAD
    bool join_pending = false;       // the last run left out the join of the handle's stream with the stream of its back halves (settle_join enqueues it)
    hipEvent_t join_ev = nullptr;    // ... which is a wait for this event (the run's last ev_back)
    bool hold_join = false;          // a run-ahead is being enqueued: the join of the batch before it is dropped, not enqueued
    hipEvent_t ev_set_free[3] = {nullptr, nullptr, nullptr};  // behind the back half that last used scratch set k
    bool set_busy[3] = {false, false, false};
    uint64_t n_run_ahead = 0, n_run_ahead_retry = 0;  // chunks enqueued ahead / run-aheads abandoned for an overflow of the chunk before (rtx_index_run_ahead_stats)

    bool shared_device = false;  // rtx_raxtax_multi drives another handle on the same device beside this one: no second stream (begin_run)
    ~rtx_index() {
        for (auto e : events) (void)hipEventDestroy(e);
        for (auto e : ev_sub) (void)hipEventDestroy(e);
        for (auto e : alt.ev_sub) (void)hipEventDestroy(e);
        if (alt.ev_exact) (void)hipEventDestroy(alt.ev_exact);
        if (alt.ev_flags) (void)hipEventDestroy(alt.ev_flags);
        if (ev_flags) (void)hipEventDestroy(ev_flags);
        for (auto e : ev_set_free)
            if (e) (void)hipEventDestroy(e);
        for (auto e : ev_front) (void)hipEventDestroy(e);
        for (auto e : ev_back) (void)hipEventDestroy(e);
        for (auto e : ev_mid) (void)hipEventDestroy(e);
        if (stream2) (void)hipStreamDestroy(stream2);
        if (stream3) (void)hipStreamDestroy(stream3);
        for (auto &i : in)
            if (i.ready) (void)hipEventDestroy(i.ready);
        if (ev_activated) (void)hipEventDestroy(ev_activated);
        if (ev_exact) (void)hipEventDestroy(ev_exact);
        if (h2d_stream) (void)hipStreamDestroy(h2d_stream);
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

namespace rtxi {

// ---- rtx_api_batch.hip
int bind(rtx_index *ix);
int ensure_events(rtx_index *ix, size_t count);

// One sub-batch = three groups of kernels.  A whole-database handle runs them back to back; a
// reference-sharded handle (config 5) stops after each group for the exchange with the other shards.
struct SubBatch {
    uint32_t sb, nq, set;
    uint64_t q0;
    hipStream_t s;   // main stream
    bool timed;      // HIP events around hit_count (the roofline kernel)
    bool timed_all;  // ... and around every other kernel (RTX_OPT_STAGE_TIMING)
};
SubBatch sub_batch_of(rtx_index *ix, uint32_t sb, bool timed);
uint8_t *counts_lo(rtx_index *ix, rtx_index::Scratch &sc);
uint16_t *counts_hi(rtx_index *ix, rtx_index::Scratch &sc);
size_t counts_elems(const rtx_index *ix, uint64_t B);
uint32_t counts_rows_layout(const rtx_index *ix);  // rows the counts buffer is laid out for right now (the diet's, or one per query of the sub-batch)
uint32_t diet_rows(const rtx_index *ix, uint32_t B);
int ensure_full_counts(rtx_index *ix, rtx_index::Scratch &sc);  // a row per query of the current class's sub-batch (the recounting taps)
int grow_diet(rtx_index *ix);  // after a run whose rows ran out
hipEvent_t stage_event(rtx_index *ix, const SubBatch &b, int stage, int which);
int enqueue_kmer(rtx_index *ix, const SubBatch &b, hipStream_t s);
int enqueue_hit(rtx_index *ix, const SubBatch &b, uint32_t flags, hipStream_t s, int part = 0, hipStream_t s_mid = nullptr);
int enqueue_count(rtx_index *ix, const SubBatch &b, uint32_t flags, hipStream_t s_mid = nullptr);
int enqueue_prob_prefix(rtx_index *ix, const SubBatch &b, bool fuse_walk, bool prob_only = false);
int enqueue_walk(rtx_index *ix, const SubBatch &b, const double *prefix, hipStream_t s);
int order_batch(rtx_index *ix, bool cluster);
int begin_run(rtx_index *ix, uint32_t *n_sub_out, bool *timed_out, bool cluster);
int enqueue_batch(rtx_index *ix, uint32_t flags);
int ensure_prob_tables(rtx_index *ix, uint32_t tmax, bool *usable);
uint32_t length_class(uint64_t len);
uint64_t class3_max_len();  // 0: t <= 255, 1: t <= 1023, 2: t <= 2047, 3: longer, prob_table in LDS, 4: longer still
int prepare_workspace(rtx_index *ix, uint64_t n_queries, const uint64_t cls_n[5], const uint64_t cls_max[5]);
int prepare_workspace_single(rtx_index *ix, uint64_t n_queries, uint64_t tmax, uint64_t max_len);  // one class whatever the lengths
void apply_class(rtx_index *ix, uint32_t c);
int plan_sub_batches(rtx_index *ix);
int alloc_scratch_set(rtx_index *ix, uint32_t k);
constexpr uint32_t kSideSet = 3;
// ---- rtx_api_download.hip
int node_tables(rtx_index *ix);  // the per-node tables of finalise_kernel, uploaded at creation
void swap_result_sets(rtx_index *ix);  // (rtx_api_batch.hip) the current batch's result state <-> rtx_index::alt
int alloc_result_set(rtx_index *ix, uint64_t n_queries);  // (rtx_api_batch.hip) the per-query arrays, the arena and the final arrays of the CURRENT set
int settle_join(rtx_index *ix);        // (rtx_api_batch.hip) the handle's stream waits for the back halves of the last run, if that run left the join out
int alloc_final(rtx_index *ix, uint64_t n_queries);  // (rtx_api_batch.hip) the final result arrays: n_queries per-query fields, arena_cap rows
int enqueue_finalise(rtx_index *ix, const SubBatch &b, hipStream_t s);  // (rtx_api_batch.hip) behind the walks of a sub-batch

}  // namespace rtxi
