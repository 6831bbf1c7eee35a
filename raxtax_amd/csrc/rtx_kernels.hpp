// Kernel parameter blocks and launchers (rtx_kernels.hip) used by the C-ABI layer
// (rtx_api_*.hip).  Plain structs passed by value to the kernels.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "raxtax_hip.h"

namespace rtx {

#ifndef RTX_NODE_TYPES_DEFINED
#define RTX_NODE_TYPES_DEFINED
enum NodeType : uint8_t { kInner = 0, kTaxon = 1, kSequence = 2 };  // src/tree.rs:181-186
#endif

constexpr uint32_t kWalkMaxRows = 208;
// segment classes (rtx_segments.hip): a (row, tile) segment with at most kSegSparseMax references is kept as a
// slot of kSegSlotEntries local ids (on the bench workload 31.0 % of the requested segments hold <= 16 references,
// 31.3 % <= 32); a (query, tile) pair takes at most kSegMaxSparseRows such segments through the
// byte counters of hit_count (more are read as dense segments)
constexpr uint32_t kSegSlotEntries = 16, kSegSparseMax = 16, kSegMaxSparseRows = 255;
// unused entry i of a slot: local id kSegPad + 4 (i mod 64) -- 64 different pad words behind the byte counters of hit_count, so that
// the LDS atomics of the unused entries of a wave-instruction do not all hit one address (no compare, no exec mask per atomic)
constexpr uint32_t kSegPad = 8192;
__host__ __device__ inline uint32_t seg_pad(uint32_t i) { return kSegPad + ((i & 63u) << 2); }
// hit_count compacts the dense rows of its (query, tile) into an LDS list of this many row ids (+ padding), in
// several rounds if they do not fit
constexpr uint32_t kKmerFewLive = 8;   // kmer_extract, lists of the live tiles only: up to this many tiles one pass per tile (rtx_kernels.hip)
constexpr uint32_t kHitListCap = 1024;  // >= 200 rows of confidence >= 0.005 + fallback

struct DevRow {  // one result row as the device emits it
    uint32_t node;               // flattened node id
    uint8_t k[RTX_MAX_DEPTH];    // rounded confidence per level, in hundredths
};
static_assert(sizeof(DevRow) == 36, "DevRow layout");

// Exact matches of the queries of a batch (Tree.sequences.get, raxtax.rs:42), two sources: ids the caller looked up on the host
// (CSR exact_off / exact_ids, indexed by query), or the group the device lookup found (rtx_exact.hip: grp[query], the ids of group g are
// gids[goff[g] .. goff[g + 1])).  grp != null selects the second.
struct ExactRef {
    const uint32_t *ids;
    const uint64_t *off;
    const uint32_t *grp, *goff, *gids;
};
__device__ __forceinline__ void exact_range(const ExactRef &x, uint64_t qin, uint64_t &e0, uint64_t &e1, const uint32_t *&ids) {
    if (x.grp) {
        const uint32_t g = x.grp[qin];
        ids = x.gids;
        e0 = e1 = 0;
        if (g != 0xFFFFFFFFu) { e0 = x.goff[g]; e1 = x.goff[g + 1]; }
    } else {
        ids = x.ids;
        e0 = x.off[qin];
        e1 = x.off[qin + 1];
    }
}
struct ExactParams {  // rtx_exact.hip
    const uint8_t *bases;      // the batch (padded behind its end)
    const uint64_t *base_off;
    uint32_t n_q;
    const uint2 *table;        // [2^bits] {tag, group + 1}; group + 1 == 0: empty
    uint32_t bits;
    const uint64_t *rep_off;   // [groups + 1] the distinct reference sequences, concatenated (padded behind the end)
    const uint8_t *rep_bytes;
    uint32_t *grp_out;         // [n_q] group of the query, 0xFFFFFFFF: no reference has its sequence
    uint64_t hash_mask;        // all ones; tests weaken the hash (RTX_DEFAULT_EXACT_HASH_MASK) so that chains and tags collide
};
void launch_exact_match(hipStream_t s, const ExactParams &p);
// rtx_ingest.hip: bases two per byte over PCIe (host packer, device unpacker)
void launch_unpack_nibbles(hipStream_t s, const uint8_t *packed, uint8_t *bases, uint64_t n_bases, uint64_t n_out);
bool pack_nibbles_mt(const uint8_t *in, uint64_t n, uint8_t *out, unsigned nt);

struct KmerParams {
    const uint8_t *bases;
    const uint64_t *base_off;
    uint64_t q0;               // first position of the sub-batch in the processing order
    const uint32_t *perm;      // [n_q] query at every position (rtx_cluster.hip); per-query outputs are by position
    const uint32_t *row_of;    // [65536] bitmap row of a k-mer or 0xFFFFFFFF
    const uint32_t *list_len;  // [65536] posting-list length
    const uint2 *row_len;      // [65536] {row_of, list_len}: what kmer_extract gathers per k-mer
    uint32_t zero_row;
    uint16_t *kmers;  // [B][kstride]
    uint32_t kstride;
    const uint32_t *seginfo;  // [n_rows+1][seg_stride] class of every segment (rtx_segments.hip)
    uint32_t seg_stride;      // ntiles rounded up to a multiple of 4
    uint32_t ntiles;
    // the same classes as bit tables per block of 64 tiles, for databases with many tiles (one transpose per
    // 64 rows x 64 tiles instead of a pass per tile): dense / sparse bits and the slot of the block's first sparse segment
    const unsigned long long *seg_dbits, *seg_sbits;  // [n_rows+1][seg_blocks]
    const uint32_t *seg_sbase;                        // [n_rows+1][seg_blocks]
    uint32_t seg_blocks;                              // ceil(ntiles / 64); 0 = use seginfo
    // the classes alone, tile by tile, two bits per row (0 empty, 1 dense, 2 sparse): the lists of the few tiles that tile pruning leaves
    // a query are built from 16 KB per tile that stay in L1 / L2 (a gather from seginfo pulled one line per row: 0.77 G lines per 1 M queries)
    const uint32_t *segcls;                           // [ntiles][cls_stride]
    uint32_t cls_stride;                              // ceil((n_rows + 1) / 16)
    uint32_t *rows;     // [B][rstride] rows of the query's k-mers (ascending), padded with the zero row to a multiple of 64
    uint32_t rstride;
    unsigned long long *dmask;  // [B][ntiles][rstride/64] per tile: which of those rows have a dense segment there
    uint32_t *srows;    // [B][ntiles][kSegMaxSparseRows + 1] per tile: slots of the sparse segments
    uint32_t *nsparse;  // [B][ntiles]
    uint32_t *t;      // [B]
    uint32_t *nrows;  // [B] dense segments over all tiles (work accounting)
    unsigned long long *hq;  // [n_q]
    uint32_t *t_all;         // [n_q]
    uint32_t *nrows_all;     // [n_q]
    uint32_t *hist;          // [B][hstride] zeroed here
    uint32_t hstride;
    uint32_t mode;           // 0: everything; 1: k-mers and row list; 2: the per-tile lists of the live tiles (after a launch with 1)
    const uint32_t *live;    // mode 2: [B][live_words] a mask per query (rtx_prune.hip)
    uint32_t live_words;
};

// The RECORDS path (round 5).  A pruned query (threshold u) only ever needs its references with a count ABOVE u: everything else is
// a reference without a hit to prob_lookup and to the lineage walk.  For a query that prune_kernel leaves at most kRecMaxSlots live
// tiles, the epilogue of hit_count compares the bit planes with u and writes nothing but (reference, count) RECORDS -- no unpacking of
// 8192 counts, no count stores -- in reference order into the segment of its tile; records_tail_kernel (rtx_records.hip) turns them
// into prefix sums at the handful of taxonomy boundaries they touch and walks the lineage from LDS, in the place of taxon_prefix's
// sweeps over whole tiles.  slot k of query q <-> tile slots[q][k] (ascending), segment rec[(q * stride + k) * 8192 ..], cnt[q][k]
// records, record = local reference (13 bits) | count << 13.
constexpr uint32_t kRecMaxSlots = 16;
struct RecordRef {
    uint16_t *nslots;   // [B] live tiles of the query at prune time; 0: the query takes the dense path
    uint16_t *slots;    // [B][kRecMaxSlots] their tiles, ascending
    uint32_t *cnt;      // [B][kRecMaxSlots] records per segment (zeroed by prune_kernel)
    uint32_t *rec;      // [B][stride][seg_len]
    uint32_t stride;    // segments per query (the largest number of live tiles that still takes the path: RTX_OPT_RECORDS)
    uint32_t seg_len;   // records a segment holds (round 6: 1024 to begin with -- a query of the bench workload leaves 70 in all; a segment that
                        // would overflow raises bit 3 of the run's flags and the host repeats the run with twice the length, up to the 8192 of a tile)
    uint32_t *flags_out;
};

struct HitParams {
    const uint32_t *bitmap;   // tile-major: [tile][n_rows1][256 words] (rtx_math.hpp: bitmap_word)
    uint32_t n_rows1;         // rows per tile region (the last one is all zero)
    uint32_t stride_bytes;    // bytes per row over all tiles (a multiple of 1024)
    uint64_t n_refs;
    uint32_t ref_base;  // global id of local reference 0 (reference-sharded index)
    const uint32_t *rows;     // [B][rstride] (kmer_extract)
    uint32_t rstride;
    const unsigned long long *dmask;  // [B][ntiles][rstride/64]
    const uint32_t *nrows;    // [B] length of the row list
    uint32_t zero_row;
    uint32_t lds_cnt8_off;    // set by the launcher: dword offset of the byte counters in dynamic LDS
    const uint32_t *srows;    // [B][ntiles][kSegMaxSparseRows + 1]
    const uint32_t *nsparse;  // [B][ntiles]
    const uint16_t *segslots; // [n_slots][kSegSlotEntries] local ids of the sparse segments
    uint32_t ntiles;
    const uint32_t *t;
    uint16_t *counts;  // [B][npad] u16 counts (more than 10 bit planes: t > 1023)
    // 10 bit planes (t <= 1023): counts leave the kernel packed, 10 bits per reference instead of 16 --
    // low byte per reference in reference order + the two high bits of eight references per u16 (chunk order)
    uint8_t *counts_lo;   // [B][npad]
    uint16_t *counts_hi;  // [B][npad / 8]
    uint64_t npad;
    uint32_t *hist;  // [B][hstride]
    uint32_t hstride;
    uint16_t *tile_max;  // [B][ntiles] largest count of the tile's references (taxon_prefix skips tiles without any probability) or null
    uint32_t flags;
    uint64_t q0;
    const uint32_t *perm;       // [n_q] query at every position: the exact matches are indexed by query
    ExactRef exact;
    uint32_t nq;           // slots of the sub-batch
    uint32_t *group_rows;  // [pairs of the batch] union rows loaded per pair, summed over the tiles (work accounting) or null
    uint32_t group_base;   // index of the sub-batch's first group in group_rows
    // hit_count_pair_kernel (rtx_hit_pair.hip): two consecutive slots per wave
    const uint2 *pair_urec;   // [pairs][pair_ustride] union of the two row lists (pair_union_kernel)
    const uint32_t *pair_nu;  // [pairs] entries of the union
    uint32_t pair_ustride;
    const uint32_t *live;     // [B][live_words] tiles to count for a query (rtx_prune.hip; the masks of a pair are neighbours) or null: all
    uint32_t live_words;
    const uint16_t *prune_thr;  // [B] threshold of the query (0: none) or null: the counts up to it go to bin 0 of the histogram as one number
    const uint32_t *items;    // [n_items] pair * ntiles + tile of the (pair, tile) blocks with a live query (live_items_kernel: groups of 1024 pairs, tile by tile inside), or null:
    uint32_t *n_items;        //           the grid is pairs x tiles.  With the list the grid is one-dimensional and walks it; n_items[1 .. 8]: the queues of the XCDs
    // the bounds pass of the tile pruning (hit_count_pair_kernel<.., kBounds>: this launch counts against the union bitmap)
    uint16_t *bounds_tile_ub;     // [B][bounds_tile_stride] largest bound of every tile of the database
    uint32_t bounds_tile_stride, bounds_ntiles;
    uint32_t *bounds_best;        // [B] key of the block with the largest bound (bound << 20 | 0xFFFFF - block), zeroed by the caller
    const uint8_t *bounds_heavy;  // [B] behind the two-level pass: only the queries flagged here are folded and leave bounds (Bounds2Params::heavy); or null: all
    // the FINE bounds pass (kBounds == 2): the bitmap is the union bitmap over blocks of 8 references -- one "tile" of it covers eight
    // tiles of the database -- and the launch walks the (pair, fine tile) items of the pairs that the coarse bounds left many live tiles
    uint64_t fine_n_refs;         // references of the database (n_refs counts blocks in a bounds pass)
    uint32_t fine_ref_ntiles;     // tiles of the database
    unsigned long long *fine_stats;  // [kPruneStatCopies][8]: [0] += (query, tile) combinations cleared, [1] += blocks of the fine pass, or null
    RecordRef rec;                // the records path of pruned queries (above); rec.nslots == null: every epilogue is the dense one
    // Rows of the counts buffer (round 6, the HBM diet): behind tile pruning with the records path only the queries that take the DENSE
    // epilogues write counts at all (a dozen of 2 000 on the bench workload), so the buffer holds rows for a fraction of the sub-batch and
    // prune_kernel hands them out -- cnt_row[q] = the row of query q, 0xFFFFFFFF: none (a query on the records path; or the rows ran out:
    // the host enlarges the buffer and repeats the run).  null: row q (every query writes counts).
    const uint32_t *cnt_row;
};
// the bounds pass in two levels (rtx_bounds2.hip): level A over blocks of 256 references for every tile, level B over blocks of 64 for the
// B-tiles (4 tiles of the database) near the query's largest level-A bound; one wave per pair, no atomics
struct Bounds2Params {
    const uint32_t *abitmap;   // [n_atiles][n_rows1][64 words]: bit b of a row = block b of 256 references of the A-tile (2048 blocks = 64 tiles)
    const uint8_t *bbitmap;    // [n_btiles][n_rows1][64 bytes]: bit b = block b of 64 references of the B-tile (512 blocks = 4 tiles)
    uint32_t n_rows1, n_atiles, ntiles, zero_row;
    const uint2 *pair_urec;    // the unions of the pairs' row lists (pair_union_kernel)
    const uint32_t *pair_nu;
    uint32_t pair_ustride, nq;
    const uint32_t *t;         // [B]
    uint16_t *tile_ub;         // [B][tile_ub_stride] out: an upper bound of every count of every tile (level A's, or level B's where refined)
    uint32_t tile_ub_stride;
    uint32_t *best_key;        // [B] out: bound << 20 | (0xFFFFF - block) of the best block of 64 among the refined tiles
    uint32_t delta_ct, delta_cm, delta_lo, delta_hi;  // which B-tiles are refined (in 1/256; bounds2_kernel)
    uint8_t *heavy;            // [B] out: 1 = the query's rule asks for more than heavy_max B-tiles (a query far from its best hit: everything is
    uint32_t heavy_max;        //     "near"): level B is left out for it and the one-level pass over blocks of 64 takes it (launch_bounds2 enqueues it)
    uint32_t *group_rows;      // work accounting: load instructions (1 KiB each) per pair, or null
    uint32_t group_base;
};
struct HitParams;
// hp: the parameters of the one-level pass (launch_hit_count_pair_bounds) for the heavy queries; items: [pairs * u_ntiles + 9] scratch
void launch_bounds2(hipStream_t s, const Bounds2Params &p, uint32_t nq, int planes, const HitParams &hp, uint32_t u_ntiles, uint32_t *items);
constexpr uint32_t kTwoLevelMinTiles = 16;  // databases with fewer tiles keep the one-level bounds pass
void launch_bounds2_build(hipStream_t s, const uint32_t *ubitmap, uint32_t n_rows1, uint32_t u_ntiles, uint8_t *bbitmap, uint32_t *abitmap);
constexpr uint32_t kFineShift = 3;       // blocks of 8 references: the 8 references of one byte of a bitmap row (ref_slot)
constexpr uint32_t kFineMinLive = 4;     // pairs with fewer live tiles than this skip the fine pass (a block of it costs what it can save there)
constexpr uint32_t kFineMinTiles = 16;   // databases with fewer tiles have no fine union bitmap

// tile pruning (rtx_prune.hip)
constexpr uint32_t kPruneStatCopies = 64;  // PruneParams::stats: [copies][8] (prune_kernel) + [copies][8] (taxon_prefix), summed by the reader
#ifndef RTX_PRUNE_SHIFT
#define RTX_PRUNE_SHIFT 6  // bench workload: blocks of 32 give 1.9 live tiles per pair for two tiles of bounds, blocks of 64 a few more for one (182 against 190 ms per 1M queries)
#endif
constexpr uint32_t kPruneShift = RTX_PRUNE_SHIFT;  // the union bitmap has one column per block of 64 references (3 .. 6: a block's references meet in one wave)
static_assert(kPruneShift >= 3 && kPruneShift <= 6, "blocks of 8 .. 64 references");
constexpr uint32_t kPruneBestWords = 66;  // PruneParams::best: {largest bound, 0, exact counts of the 64 references of that block}
struct PruneParams {
    const uint16_t *tile_ub;     // [B][tile_ub_stride] largest bound of every tile (the bounds pass: bounds_epilogue, rtx_hit_common.hpp)
    const uint32_t *best_key;    // [B] bound << 20 | (0xFFFFF - block) of the block with the largest bound, the lowest among equals
    uint32_t tile_ub_stride, ntiles, nq;
    uint64_t n_refs;             // references on this handle
    uint64_t n_total;            // references of the whole database (the N of the threshold)
    uint32_t ref_base;           // global id of local reference 0 (reference-sharded index)
    // A reference shard prunes with the threshold of the WHOLE database: the threshold follows from the best block anywhere.
    // phase 1: bounds per tile, the best local block and the exact counts of its references -> best[q]; nothing else.
    // (the caller keeps, per query, the record of the shard with the largest bound -- an all-gather over the shards)
    // phase 2: the threshold from best[q] as handed back, the live tiles of this shard, its uncounted references into bin 0.
    // phase 0 (whole database on the handle): both at once.
    uint32_t phase;
    uint32_t *best;              // [B][kPruneBestWords] (phases 1 and 2) or null
    const uint2 *cbitmap;     // the database once more BLOCK by block: [block of 64 references][row] 8 bytes, bit j = reference 64 block + j has the
                              // row's k-mer (rtx_prune.hip: the exact counts of the best block in ten loads per lane); or null: from `bitmap`
    const uint32_t *bitmap;   // the database's bitmap: the exact count of one reference
    uint32_t n_rows1, stride_bytes;
    const uint32_t *rows;     // [B][rstride]
    uint32_t rstride;
    const uint32_t *nrows, *t;
    uint32_t flags;
    uint64_t q0;
    const uint32_t *perm;
    ExactRef exact;
    const double *lnfact, *inv;   // ln x!, 1 / x
    uint32_t nlf;                 // entries of lnfact the batch can reach (t + n - 1 < 1.5 tmax + 2): staged in LDS by every workgroup
    uint32_t *hist;           // [B][hstride]: bin 0 receives the references of the tiles that are not counted
    uint32_t hstride;
    uint32_t *live;           // [B][live_words] bit T: tile T is counted for the query
    uint32_t live_words;
    uint32_t *pair_live;      // [pairs] tiles counted for either query of the pair (live_offsets_kernel turns them into the list of blocks) or null
    uint16_t *thr_out;        // [B] the threshold of every query (0: not pruned, every tile is counted)
    uint16_t *i1_out;         // [B] i* + 1 of every query with a threshold: Z holds less than eps = 1e-10 at i <= i* (prob_lookup starts there)
    unsigned long long *stats;  // [kPruneStatCopies][8]: [0] += live tiles, [1] += pairs ... (reporting) or null
    uint32_t *detail;           // [B][kPruneDetailWords] debug tap (RTX_OPT_DEBUG_TAPS) or null: {best block, M, threshold, i* + 1, largest
                                // bound, t, 0, 0, exact counts of the 64 references of the best block}
    RecordRef rec;              // rec.nslots != null: queries with a threshold and at most rec_max_slots live tiles take the records path
    uint32_t rec_max_slots;
    uint32_t *cnt_row;          // [B] out: the row of the counts buffer of every query that takes the dense epilogues (HitParams::cnt_row) or null
    uint32_t *cnt_cursor;       // [1] rows handed out in this launch (zeroed before it)
    uint32_t cnt_cap;           // rows there are
    uint32_t *flags_out;        // bit2: the rows ran out
};
constexpr uint32_t kPruneDetailWords = 72;
struct ProbTables;
void launch_prune(hipStream_t s, const PruneParams &p, const ProbTables &tb, uint32_t nq);
void launch_block_major_build(hipStream_t s, const uint32_t *bitmap, uint32_t n_rows1, uint32_t ntiles, uint32_t stride_bytes, uint8_t *cbitmap);

// memoised cmf / pmf-ratio tables for every (t, m, i), t <= tmax (rtx_prob_tables.hip)
struct ProbTables {
    double *cmf;           // [off[t] + m * (t/2+1) + i]
    double *ratio;         // same indexing: pmf / cmf
    const uint64_t *off;   // [tmax+1]
    const uint32_t *moff;  // [tmax+1] offset of row t in ilo / sat
    uint16_t *ilo;         // [moff[t] + m] first i with ln pmf_m(i) >= -100
    uint16_t *sat;         // [moff[t] + m] first i at which cmf_m has stopped changing
    uint32_t tmax;
};


struct ProbParams {
    const uint32_t *order;  // processing order of the sub-batch (slot indices) or null
    const uint32_t *t;
    const uint32_t *hist;
    uint32_t hstride;
    uint32_t tmax;
    uint32_t n1max;
    const double *lnfact;
    uint64_t n_refs;
    uint64_t q0;
    double *table_z;  // [B][hstride]
    double *z;        // [n_q]
    double *gs;       // [n_q]
    uint8_t *status;  // [n_q]
    uint32_t *ndist;  // [n_q] number of distinct hit counts D_q (work accounting)
    const uint16_t *prune_thr;  // [B] tile pruning: references with a count up to this carry nothing (rtx_prune.hip) or null
    const uint16_t *prune_i1;   // [B] ... and the sums over i may start here (everything below holds less than eps = 1e-10 of Z)
    double *gscratch;           // prob_table_kernel for reads whose arrays do not fit LDS: [B][gstride] doubles of global memory, or null
    uint32_t gstride;
};

constexpr uint32_t kWalkSubAllocs = 128;    // sub-allocators of the result arena (512 walks of a launch of 65 536 share one)
constexpr uint32_t kWalkSubStride = 16;     // u64 words between them (128 bytes)
constexpr uint32_t kWalkSubMinQueries = 4096;  // launches with fewer walks add to the arena's cursor directly
constexpr uint32_t kWalkChunkRows = 64;     // rows a sub-allocator takes from the arena at a time (what is left of a piece stays unused)
struct WalkParams {
    const uint8_t *status;
    uint64_t q0;
    const double *prefix;
    uint32_t n_bnd;
    const uint4 *rec;  // per node {boundary index of its range begin, of its end, first child, n_children | type << 30}
    DevRow *arena;
    unsigned long long arena_cap;
    unsigned long long *arena_cursor;
    // Result rows are placed through kWalkSubAllocs sub-allocators (a line of their own each, word = end << 32 | cursor of a piece of
    // kWalkChunkRows rows taken from arena_cursor): 65 536 walks of a launch adding to ONE address were what bounded taxon_prefix
    // (13.8 ms per 1 M queries, 10.5 with the rows at fixed places).  Zeroed before every launch that walks (rows of a sub-batch lie
    // between the cursor's values before and behind it: the download copies that range); null: every walk adds to arena_cursor itself.
    unsigned long long *sub_alloc;
    uint32_t *n_rows;                // [n_q]
    unsigned long long *row_start;   // [n_q]
    uint32_t *flags_out;             // bit0 arena overflow, bit1 row/depth overflow
};

// finalise_kernel (rtx_finalise.hip): the rows of a sub-batch as the walks left them in the arena -> the final result arrays, on the device
// (lineage.rs:91-110): per query the rows sorted, the local signal computed, the rows laid out back to back in the arrays the host's view
// points into; the per-query fields scattered from the processing order to the input order.  The host copies and does nothing else.
struct FinaliseParams {
    // in, by position of the processing order
    const uint8_t *status;
    const uint32_t *t_all;
    const double *gs;
    const uint32_t *n_rows;
    const unsigned long long *row_start;
    const DevRow *arena;
    unsigned long long arena_cap;
    const uint32_t *perm;  // position -> query
    uint64_t q0;
    uint32_t nq;
    // per node: depth, index of its lineage (begin of its range), the expected side of its local signal (rtx_math.hpp: fin_node_expected)
    const uint8_t *node_depth, *node_sig0;
    const uint32_t *node_begin;
    const double *node_eb;  // [node][D]
    uint32_t D;       // levels of the deepest lineage = stride of the confidence arrays
    // out, by query
    uint32_t *o_t;
    uint8_t *o_status;
    double *o_gs;
    unsigned long long *o_row_begin;
    uint32_t *o_row_count;
    // out, rows: [fin_cursor before the launch, behind it)
    uint32_t *r_lineage, *r_node, *r_depth;
    uint8_t *r_depth8, *r_hund;
    double *r_local, *r_conf;
    unsigned long long row_cap;
    unsigned long long *fin_cursor;
    uint32_t *flags_out;  // bit0: no room (the host enlarges the arena and repeats the run)
};

constexpr int kBounds2MaxPlanes = 11;  // bounds2_kernel (rtx_bounds2.hip) is instantiated for 8, 10 and 11 planes

struct PrefixParams {
    const uint8_t *status;
    const uint32_t *t;     // [B] distinct k-mers per slot (size of the table copy)
    uint32_t tz_in_lds;    // copy table/Z to LDS first (hstride * 8 bytes must fit)
    uint64_t q0;
    const uint16_t *counts;      // u16 format (packed == 0)
    const uint8_t *counts_lo;    // packed format (hit_count with 10 bit planes): low bytes ...
    const uint16_t *counts_hi;   // ... and 2 high bits x 8 references per u16
    uint32_t packed;
    uint64_t npad;
    const double *table_z;
    uint32_t hstride;
    uint64_t n_refs;
    const uint8_t *bnd_bits;    // [ceil(N/8)] bit j: position 8*chunk+j+1 is a boundary
    const uint32_t *bnd_rank;   // [ceil(N/8)] boundary index of the first such position
    double *prefix;             // [B][n_bnd]
    uint32_t n_bnd;
    const uint16_t *tile_max;   // [B][ntiles] largest count per tile of 8192 references (hit_count) or null: every tile is swept
    uint32_t ntiles;
    const uint16_t *prune_thr;  // [B] tile pruning: threshold of the query (> 0: tiles with a largest count of 0 were not counted) or null
    unsigned long long *prune_stats;  // reporting (second half of PruneParams::stats) or null
    uint32_t fuse_walk;         // wave 0 of every workgroup walks its query right after the sweeps (walk.prefix == prefix)
    WalkParams walk;
    const uint16_t *rec_nslots; // [B] records path (RecordRef::nslots): a query with slots is left to records_tail_kernel; or null
    const uint32_t *cnt_row;    // [B] row of the counts buffer per query (HitParams::cnt_row) or null: row q
};


// records_tail_kernel (rtx_records.hip): the queries of a sub-batch on the records path, after prob_lookup
struct TailParams {
    RecordRef rec;
    const uint32_t *t;          // [B]
    const double *table_z;      // [B][hstride] table / Z (prob_lookup)
    uint32_t hstride;
    const uint8_t *bnd_bits;    // as PrefixParams
    const uint32_t *bnd_rank;
    double *prefix;             // [B][n_bnd] scratch row of the query: only the slow path (more boundary intervals than fit LDS) writes it
    uint32_t *cnt_cursor;       // the prefix rows on their diet (HitParams::cnt_row): a query on the slow path takes a row of its own here; null: row q
    uint32_t cnt_cap;
    uint32_t *flags_out;        // bit2: the rows ran out
    uint32_t n_bnd;
    uint32_t nq;
    WalkParams walk;
    unsigned long long *prefix_stats;  // PrefixParams::prune_stats (the same two counters, for the queries this kernel takes) or null
    unsigned long long *stats;  // [kPruneStatCopies][8] reporting: [0] += records, [1] += queries on the path, [2] += boundary entries, [3] += slow-path queries; or null
};
void launch_records_tail(hipStream_t s, const TailParams &p, uint32_t nq);

void launch_bitmap_build(hipStream_t s, const uint64_t *off, const uint32_t *post, const uint32_t *row_of,
                         uint32_t *bitmap, uint32_t stride_words, uint32_t n_rows1, uint32_t ref_lo, uint32_t ref_hi, uint32_t shift = 0);
void launch_ref_kmer_mark(hipStream_t s, const uint8_t *bases, const uint64_t *off, uint64_t n_refs, uint32_t *present);
void launch_ref_bitmap_set(hipStream_t s, const uint8_t *bases, const uint64_t *off, uint64_t n_refs,
                           const uint32_t *row_of, uint32_t *bitmap, uint32_t stride_words, uint32_t n_rows1, uint32_t shift = 0);
void launch_row_len_pack(hipStream_t s, const uint32_t *row_of, const uint32_t *list_len, uint2 *out);
void launch_row_popcount(hipStream_t s, const uint32_t *row_of, const uint32_t *bitmap, uint32_t stride_words, uint32_t n_rows1,
                         uint32_t *list_len);
void launch_kmer_extract(hipStream_t s, const KmerParams &p, uint32_t nq);
void launch_hit_count(hipStream_t s, const HitParams &p, uint32_t nq, uint32_t ntiles, int planes);
void launch_pair_union(hipStream_t s, const uint32_t *rows, const uint32_t *nrows, uint32_t rstride, uint32_t nq, uint2 *urec,
                       uint32_t *nu, uint32_t ustride);
void launch_hit_count_pair(hipStream_t s, const HitParams &p, uint32_t nq, uint32_t ntiles, int planes);  // 8 (every t <= 255) or 10 bit planes
// the list of the live (pair, tile) blocks from the masks and the per-pair numbers prune_kernel left: off = [pairs] scratch
void launch_live_items(hipStream_t s, const uint32_t *live, uint32_t live_words, const uint32_t *pair_live, uint32_t nq, uint32_t ntiles, uint32_t *off,
                       uint32_t *items, uint32_t *n_items);
void launch_hit_count_pair_bounds(hipStream_t s, const HitParams &p, uint32_t nq, uint32_t u_ntiles, int planes);
void launch_hit_count_pair_bounds_items(hipStream_t s, const HitParams &p, uint32_t nq, uint32_t u_ntiles, int planes);  // ... over a list of (pair, union tile) items (p.items)
// the fine bounds pass over the pairs with many live tiles (p: the fine union bitmap, live masks, thresholds, histogram); updates pair_live
void launch_fine_bounds(hipStream_t s, const HitParams &p, uint32_t nq, uint32_t ntiles, uint32_t f_ntiles, uint32_t *pair_live, uint32_t *cnt,
                        uint32_t *items, uint32_t *n_items, int planes);  // ... on the union bitmap: bounds_epilogue
size_t prob_table_lds_bytes(uint32_t tmax);
void launch_prob_table(hipStream_t s, const ProbParams &p, uint32_t nq);
size_t prob_lookup_lds_bytes(uint32_t tmax);
void launch_prob_tables_build(hipStream_t s, const ProbTables &tb, const double *lf, const double *inv);
void launch_prob_order(hipStream_t s, const uint32_t *t, uint32_t nq, uint32_t *order);
void launch_prob_lookup(hipStream_t s, const ProbParams &p, const ProbTables &tb, uint32_t nq);
// segment classes of the index (rtx_segments.hip)
void launch_seg_popcount(hipStream_t s, const uint32_t *bitmap, uint32_t stride_bytes, uint32_t n_rows1, uint32_t ntiles, uint16_t *pop);
void launch_seg_emit(hipStream_t s, const uint32_t *bitmap, uint32_t stride_bytes, uint32_t n_rows1, uint32_t ntiles,
                     const uint32_t *seginfo, uint32_t seg_stride, uint16_t *slots);
// processing order of a batch (rtx_cluster.hip)
void launch_sketch(hipStream_t s, const uint8_t *bases, const uint64_t *off, uint32_t n_q, uint64_t *keys, uint32_t *idx);
void launch_invert_perm(hipStream_t s, const uint32_t *perm, uint32_t n, uint32_t *inv);
// locator (rtx_cluster.hip): 12-mer -> lowest reference position, built from the reference sequences; a query votes
constexpr uint32_t kLocTableEntries = 1u << 24;
void launch_loc_mark(hipStream_t s, const uint8_t *bases, const uint64_t *off, uint64_t n_refs, uint32_t *pos_min, uint32_t *cnt);
void launch_loc_finish(hipStream_t s, uint32_t *pos_min, const uint32_t *cnt);
void launch_locator(hipStream_t s, const uint8_t *bases, const uint64_t *off, uint32_t n_q, const uint32_t *table, uint64_t n_refs,
                    uint64_t *keys);
void launch_identity_perm(hipStream_t s, uint32_t n, uint32_t *perm, uint32_t *inv);
int cluster_sort(hipStream_t s, void *tmp, size_t *tmp_bytes, const uint64_t *keys_in, uint64_t *keys_out, const uint32_t *idx_in,
                 uint32_t *perm_out, size_t n, bool with_class = false);
void launch_class_keys(hipStream_t s, uint64_t *keys, const uint64_t *off, uint32_t n, const uint64_t lim[4], bool from_index, uint32_t *idx);
void launch_taxon_prefix(hipStream_t s, const PrefixParams &p, uint32_t nq);
void launch_lineage_walk(hipStream_t s, const WalkParams &p, uint32_t nq);
void launch_finalise(hipStream_t s, const FinaliseParams &p);
void launch_probs_expand(hipStream_t s, const uint16_t *counts, const double *tz, uint64_t n, double *out);
void launch_rehist(hipStream_t s, const uint16_t *counts, uint64_t npad, uint64_t n_refs, const uint32_t *t, uint32_t *hist,
                   uint32_t hstride, uint16_t *tile_max, uint32_t ntiles, uint32_t nq);
void launch_counts_unpack(hipStream_t s, const uint8_t *lo, const uint16_t *hi, uint64_t n, uint16_t *out);

}  // namespace rtx
