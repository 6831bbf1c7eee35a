/*
 * raxtax_hip.h -- C ABI of libraxtax_hip.so, the MI355X (gfx950) drop-in for the
 * per-query k-mer classification hot path of noahares/raxtax v1.5.0.
 *
 * The reference has no FFI or plugin interface: its seam is the Rust function
 *     raxtax(queries, tree, skip_exact_matches, raw_confidence, chunk_size, sender, tsv)
 * (src/raxtax.rs:14-22, called once from src/main.rs:137-145).  The entry points
 * below are what a Rust `extern "C"` block for that seam binds (INTEGRATION.md
 * shows the binding).  Plain pointers and sizes only; every function returns an
 * int status (RTX_OK or a negative RTX_ERR_*); `rtx_last_error()` returns a
 * message for the calling thread.  The caller owns every host buffer it passes
 * in; the library owns all device memory and every buffer it hands out through
 * a view (valid until the next call on the same handle, or its destruction).
 * One handle per GPU; calls on one handle must be serialised by the caller;
 * distinct handles may be driven from distinct host threads.
 *
 * There is NO CPU fallback: every rtx_index_* / rtx_batch_* / rtx_classify_*
 * call fails with RTX_ERR_NO_DEVICE when no gfx950 device is usable.
 *
 * Sequences are passed as the reference stores them: one byte per base, 4-bit
 * one-hot codes A=1 C=2 G=4 T=8, ambiguity codes = unions (src/parser.rs:11-34).
 * Reference ids are indices into the lineage-sorted order of Tree::new
 * (src/tree.rs:53-54), i.e. indices into `tree.lineages`.
 */
#ifndef RAXTAX_HIP_H
#define RAXTAX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTX_ABI_VERSION 6 /* 2: rtx_result_view.row_begin/row_count replace row_off; 3: exact_off == NULL = look the exact matches up on the device;
                             4: RTX_NUM_STAGES 8 -> 10 (rtx_batch_stage_times writes ten entries), rtx_batch_prefetch / rtx_batch_activate;
                             5: rtx_result_view grows row_conf_stride, row_depth_u8, row_conf_hundredths (the rows are finalised on the device and
                                arrive in their final layout: row_conf is [n_rows][row_conf_stride], no longer [n_rows][RTX_MAX_DEPTH]); the exports
                                and options of round 5 (rtx_index_self_sample, rtx_records_format, options 18-22);
                             6: RTX_OPT_RUN_AHEAD (23), RTX_RETRY_CHUNK from rtx_batch_download_then_run under it, rtx_index_run_ahead_stats */
#define RTX_NUM_KMERS 65536u /* 2 << 15 posting lists, src/tree.rs:52 */
#define RTX_MAX_DEPTH 32u    /* deepest lineage (comma-separated levels) the device walk carries */

/* status codes */
#define RTX_OK 0
#define RTX_ERR_INVALID (-1)    /* bad argument / malformed input arrays            */
#define RTX_ERR_HIP (-2)        /* HIP runtime error (see rtx_last_error)           */
#define RTX_ERR_NO_DEVICE (-3)  /* no usable gfx950 device                          */
#define RTX_ERR_OOM (-4)        /* host or device allocation failed                 */
#define RTX_ERR_PARSE (-5)      /* FASTA / lineage annotation error (parser.rs)     */
#define RTX_ERR_DEPTH (-6)      /* lineage deeper than RTX_MAX_DEPTH: refused by rtx_tree_build* / _parse* / _load_bin, the lineage named in rtx_last_error */
#define RTX_ERR_STATE (-7)      /* call sequence violated (e.g. run before upload)  */
#define RTX_ERR_TOO_LONG (-8)   /* the memoised probability tables were demanded (RTX_OPT_PROB_MODE = 2) for a read they do not cover.  (Until ABI 4 also: a
                                   query longer than 65 542 bases.  The reference asserts on DISTINCT k-mers, raxtax.rs:56, and serves such reads: so does
                                   the library now -- per query, RTX_Q_ALL_KMERS for a read that really holds all 65 536 of them) */
#define RTX_ERR_SENDER (-9)     /* the result sink refused a message (closed channel, raxtax.rs:87)    */
#define RTX_RETRY_CHUNK 1        /* not an error: rtx_batch_download_then_run under RTX_OPT_RUN_AHEAD found the batch being downloaded short of buffer space
                                   AFTER it had enqueued the next one; both have been taken off the handle (buffers grown): run this batch again
                                   (upload, run, download) and stage the next one anew */

/* flags of rtx_classify_batch / rtx_batch_run (src/io.rs:119-121,131-133) */
#define RTX_SKIP_EXACT_MATCHES 1u /* zero the hit counts of exact matches, raxtax.rs:65-68 */
#define RTX_RAW_CONFIDENCE 2u     /* host mirror only: suppress the single-exact-match override, raxtax.rs:73-84 */

/* per-query status in rtx_result_view.status */
#define RTX_Q_OK 0
#define RTX_Q_NO_KMERS 1 /* t == 0, or t == 1 without a full-overlap reference: the reference
                            panics here (prob.rs:21 u64 underflow / prob.rs:162 zip_eq); the
                            library reports the query instead of aborting -- deliberate divergence */

#define RTX_Q_ALL_KMERS 2 /* t == 65 536: the read holds every 8-mer (only reads of 65 543 bases or more can).  The reference
                            asserts that t fits a u16 (raxtax.rs:56) and aborts the run; the library reports this query, returns
                            no rows for it (t = 65 536 in the view) and classifies the rest of the batch */

typedef struct rtx_tree rtx_tree;   /* host mirror of `Tree` (src/tree.rs:36-43)          */
typedef struct rtx_index rtx_index; /* device-resident index + per-GPU batch workspace    */

/* ------------------------------------------------------------------------- */
/* diagnostics                                                                */
/* ------------------------------------------------------------------------- */
int rtx_abi_version(void);
const char *rtx_last_error(void);
int rtx_device_count(void); /* number of visible HIP devices, 0 if none / no driver */

/* ------------------------------------------------------------------------- */
/* Host mirror of the index build (plumbing around the hot path)             */
/* ------------------------------------------------------------------------- */
/* Tree::new(lineages, sequences), src/tree.rs:46-140.  Lineage i is the byte
 * range lineage_bytes[lineage_off[i] .. lineage_off[i+1]) (no terminator);
 * sequence i is seq_bytes[seq_off[i] .. seq_off[i+1]). */
int rtx_tree_build(uint64_t n, const char *lineage_bytes, const uint64_t *lineage_off,
                   const uint8_t *seq_bytes, const uint64_t *seq_off, rtx_tree **out);
/* The same with options.  RTX_TREE_SKIP_KMER_MAP: sort, taxonomy and exact-sequence map only; the k-mer
 * index is then built on the GPU from the sequences by rtx_index_create_from_tree. */
#define RTX_TREE_SKIP_KMER_MAP 1u
int rtx_tree_build_ex(uint64_t n, const char *lineage_bytes, const uint64_t *lineage_off,
                      const uint8_t *seq_bytes, const uint64_t *seq_off, uint32_t flags, rtx_tree **out);
/* parse_reference_fasta_str, src/parser.rs:46-105 */
int rtx_tree_parse_reference_fasta(const char *text, uint64_t len, rtx_tree **out);
/* The same with RTX_TREE_* flags: RTX_TREE_SKIP_KMER_MAP leaves Tree.k_mer_map unbuilt (rtx_index_create_from_tree
 * then builds the bitmaps on the GPU from the sequences); rtx_tree_build_kmer_map builds it later, e.g. on a
 * background thread for the `.bin` cache while queries are already being classified.  Both parsers cut the text
 * at header lines and parse the pieces on several threads. */
int rtx_tree_parse_reference_fasta_ex(const char *text, uint64_t len, uint32_t flags, rtx_tree **out);
int rtx_tree_build_kmer_map(rtx_tree *tree);
/* Tree::save_to_file / Tree::load_from_file (src/tree.rs:147-164): the reference's `.bin` database,
 * bincode 1.3.3 default options.  Files written here are readable by upstream raxtax and vice versa. */
int rtx_tree_save_bin(const rtx_tree *tree, const char *path);
int rtx_tree_load_bin(const char *path, rtx_tree **out);
void rtx_tree_destroy(rtx_tree *tree);
uint64_t rtx_tree_num_tips(const rtx_tree *tree);                  /* Tree.num_tips      */
const char *rtx_tree_lineage(const rtx_tree *tree, uint64_t i);    /* Tree.lineages[i]   */
uint64_t rtx_tree_original_index(const rtx_tree *tree, uint64_t i);/* sorted -> input idx */
/* Tree.k_mer_map as CSR: offsets[65537], postings sorted-unique per list (tree.rs:134-137) */
int rtx_tree_kmer_csr(const rtx_tree *tree, const uint64_t **offsets, const uint32_t **postings);
/* Tree.sequences.get(seq), src/raxtax.rs:42.  Returns the number of ids. */
uint64_t rtx_tree_exact_matches(const rtx_tree *tree, const uint8_t *seq, uint64_t len,
                                const uint32_t **ids);
/* The same lookup for a batch: fills exact_off[n_queries+1] and up to ids_cap ids; returns the
 * total number of ids (call again with a larger buffer if it exceeds ids_cap). */
uint64_t rtx_tree_exact_matches_batch(const rtx_tree *tree, uint64_t n_queries, const uint8_t *bases,
                                      const uint64_t *base_off, uint64_t *exact_off, uint32_t *exact_ids,
                                      uint64_t ids_cap);
/* Flattened taxonomy: nodes in breadth-first order (root = node 0, the children of a node
 * are consecutive), childless Sequence nodes (tree.rs:102-107) dropped -- they never
 * influence Lineage::evaluate.  node_begin/end = Node.confidence_range (tree.rs:190). */
typedef struct {
    uint32_t n_nodes;
    const uint32_t *node_begin;
    const uint32_t *node_end;
    const uint32_t *node_first_child;
    const uint32_t *node_n_children;
    const uint32_t *node_parent; /* 0xFFFFFFFF for the root */
    const uint8_t *node_type;    /* 0 Inner, 1 Taxon, 2 Sequence (tree.rs:181-186) */
} rtx_nodes_view;
int rtx_tree_nodes(const rtx_tree *tree, rtx_nodes_view *out);

/* parse_query_fasta_str, src/parser.rs:117-154.  Returns a handle holding labels and
 * encoded sequences; `skip` labels (already-processed queries, parser.rs:150-153) are dropped. */
typedef struct rtx_queries rtx_queries;
int rtx_queries_parse_fasta(const char *text, uint64_t len, const char *const *skip, uint64_t n_skip,
                            rtx_queries **out);
/* The same for one block of a file that is read piecewise (FASTA ingest of very large query files: the reference
 * reads the whole file, parser.rs:112-115).  Blocks must be cut in front of a header line: rtx_fasta_block_end returns
 * the offset of the last '>' that directly follows a newline (0 if there is none) -- parse text[0, end) with
 * RTX_FASTA_MORE_FOLLOWS and carry text[end, len) over to the next block; every block but the first also gets
 * RTX_FASTA_NOT_FIRST.  The records of all blocks together are exactly those of the whole-file parse. */
#define RTX_FASTA_MORE_FOLLOWS 1u /* not the last block: a trailing header without bases is dropped (the next header replaces it) */
#define RTX_FASTA_NOT_FIRST 2u    /* not the first block: it starts with a header by construction */
uint64_t rtx_fasta_block_end(const char *text, uint64_t len);
int rtx_queries_parse_fasta_block(const char *text, uint64_t len, const char *const *skip, uint64_t n_skip, uint32_t flags,
                                  rtx_queries **out);
void rtx_queries_destroy(rtx_queries *q);
uint64_t rtx_queries_len(const rtx_queries *q);
const char *rtx_queries_label(const rtx_queries *q, uint64_t i);
/* all sequences concatenated + n+1 offsets (the layout rtx_classify_batch takes) */
int rtx_queries_data(const rtx_queries *q, const uint8_t **bases, const uint64_t **base_off);

/* ------------------------------------------------------------------------- */
/* Device index (replaces the read-only `&Tree` argument of raxtax())        */
/* ------------------------------------------------------------------------- */
/* Uploads Tree.k_mer_map (CSR, lists sorted-unique) and the flattened taxonomy
 * (as rtx_nodes_view) to GPU `device`; the library re-encodes the postings into
 * its own HBM layout (per-k-mer reference bitmaps).  n_refs = Tree.num_tips. */
int rtx_index_create(int device, uint64_t n_refs, const uint64_t *offsets /*65537*/,
                     const uint32_t *postings, uint32_t n_nodes, const uint32_t *node_begin,
                     const uint32_t *node_end, const uint32_t *node_first_child,
                     const uint32_t *node_n_children, const uint8_t *node_type, rtx_index **out);
/* One shard of a reference-sharded database (BASELINE.json configs[4]): the handle holds the bitmaps of
 * references [ref_lo, ref_hi) only; offsets/postings and the taxonomy are those of the WHOLE database
 * (global ids).  shard_cuts lists the boundaries of all shards (every rank passes the same list) so that
 * the taxonomy boundary numbering is identical on every rank.  Such a handle is driven with
 * rtx_shard_begin / _count / _prob / _walk (below) instead of rtx_batch_run. */
int rtx_index_create_shard(int device, uint64_t n_refs_total, uint64_t ref_lo, uint64_t ref_hi,
                           const uint64_t *shard_cuts, uint32_t n_cuts, const uint64_t *offsets,
                           const uint32_t *postings, uint32_t n_nodes, const uint32_t *node_begin,
                           const uint32_t *node_end, const uint32_t *node_first_child,
                           const uint32_t *node_n_children, const uint8_t *node_type, rtx_index **out);
/* Index build on the GPU (the k-mer part of Tree::new, src/tree.rs:114-123,134-137): the encoded
 * reference sequences in lineage-sorted order (sequence i = reference id i) instead of posting lists. */
int rtx_index_create_from_sequences(int device, uint64_t n_refs, const uint8_t *seq_bytes, const uint64_t *seq_off,
                                    uint32_t n_nodes, const uint32_t *node_begin, const uint32_t *node_end,
                                    const uint32_t *node_first_child, const uint32_t *node_n_children,
                                    const uint8_t *node_type, rtx_index **out);
/* Convenience: the same from a host tree (GPU build if the tree has no k-mer map). */
int rtx_index_create_from_tree(int device, const rtx_tree *tree, rtx_index **out);
void rtx_index_destroy(rtx_index *index);
uint64_t rtx_index_num_refs(const rtx_index *index);
uint64_t rtx_index_device_bytes(const rtx_index *index); /* HBM held by the index itself */
uint64_t rtx_index_workspace_bytes(const rtx_index *index); /* ... and by everything else of the handle as of the last upload: probability tables, the scratch
                                                               sets of a sub-batch, inputs, result arena and final result arrays (ABI 5) */
/* ... in parts (bytes): [0] probability tables, [1] counts, [2] record segments, [3] boundary prefix sums, [4] per-tile masks and sparse-slot lists,
 * [5] the rest of the scratch sets, [6] inputs and processing order, [7] result arena and final arrays; [8] = scratch sets in use */
int rtx_index_workspace_parts(const rtx_index *index, uint64_t out[9]);
/* queries processed per kernel wave (sub-batch); 0 = choose from free HBM */
int rtx_index_set_batch(rtx_index *index, uint32_t sub_batch);
/* Tuning / test knobs.  RTX_OPT_PROB_MODE: 0 = auto (memoised cmf tables when every query has
 * t <= 1023 and they fit in HBM, else the per-query recurrence kernel), 1 = recurrence only,
 * 2 = tables only (error if unavailable).  Both produce the same probabilities to ~1e-13. */
#define RTX_OPT_SUB_BATCH 1
#define RTX_OPT_PROB_MODE 2
#define RTX_OPT_STAGE_TIMING 6 /* 0 (default): HIP events around hit_count only; 1: around every kernel
                                 (rtx_batch_stage_times then reports all stages; adds ~1 ms per 100k queries) */
#define RTX_OPT_DEBUG_TAPS 14 /* 0 (default); 1: prune_kernel keeps its view of every query of a sub-batch for rtx_debug_prune_detail (tests) */
#define RTX_OPT_DEVICE_EXACT 15 /* 1 (default): a batch uploaded WITHOUT exact-match ids (exact_off == NULL) has them looked up on the device
                                 * (a hash table of the distinct reference sequences, every candidate verified byte by byte; rtx_exact.hip)
                                 * as part of rtx_batch_run -- Tree.sequences.get, raxtax.rs:42 -- when the handle was built from the
                                 * sequences (rtx_index_create_from_tree / _from_sequences); rtx_batch_exact_matches returns the ids.
                                 * 0: such a batch has no exact matches (the behaviour of ABI version 2) */
#define RTX_OPT_CLUSTER 7 /* 1 (default): the queries of a batch are processed in an order that puts related
                             queries next to each other (min-hash sketches, rtx_cluster.hip) so that bitmap rows
                             are reused; 0: input order.  Results are identical and always in input order. */
#define RTX_OPT_PACKED_COUNTS 8 /* 1 (default): with t <= 1023 the hit counts travel from hit_count to taxon_prefix
                                 * packed, 10 bits per reference (low byte + 2 high bits); 0: as u16 (A/B measurements) */
#define RTX_OPT_HIT_PAIR 11 /* 1 (default): with t <= 1023 and RTX_OPT_CLUSTER on, hit_count runs two neighbouring queries per wave (two
                             * sets of bit planes in registers) and loads the bitmap rows they share once (rtx_hit_pair.hip);
                             * identical results; 0: one query per wave (the kernel of longer queries; A/B measurements) */
#define RTX_OPT_TILE_SKIP 10 /* 1 (default): hit_count records the largest count of every tile of 8192 references and
                              * taxon_prefix (lineage.rs:61-66) reads only the tiles that hold a reference whose probability
                              * reaches 1e-30 -- the others add less than n_refs * 1e-30 to any prefix sum, far below what a
                              * confidence (rounded to 1e-2) can show; 0: every reference is summed (A/B measurements) */
#define RTX_OPT_LOCATOR 12 /* 1 (default): when the handle was built from the reference sequences (rtx_index_create_from_sequences /
                            * _from_tree) the processing order of RTX_OPT_CLUSTER is led by the query's position in the
                            * lineage-ordered database (a vote of its 12-mers in a table built from the references), the
                            * min-hashes only break ties; 0: min-hash order alone.  Scheduling only: results are identical */
#define RTX_OPT_TILE_PRUNE 13 /* 1 (default): hit_count counts only the tiles of 8192 references that can hold a reference with any
                               * probability -- decided from upper bounds (the queries counted against a union bitmap over blocks
                               * of 64 references) and a per-query threshold u: the references with a count up to u hold less
                               * than 1e-10 of probability together and are treated as references without a hit (rtx_prune.hip;
                               * every probability and confidence sum stays within 1e-9 of the full count -- measured 1e-11 --,
                               * the result of a query does not depend on the rest of the batch).  Takes effect with t <= 1023,
                               * RTX_OPT_HIT_PAIR = 1, RTX_OPT_TILE_SKIP = 1, the whole database on the handle and 4 tiles or more;
                               * the debug taps recount the tapped sub-batch in full.  0: every tile is counted */
#define RTX_OPT_FINE_BOUNDS 17 /* 1 (default): second stage of the bounds of the tile pruning on databases of 16 tiles or more: the pairs of
                                * queries that the bounds over blocks of 64 references leave 4 or more live tiles are counted against a
                                * second union bitmap over blocks of 8 references (1/8 of the index), which takes the tiles without a block
                                * above the query's threshold off their lists before the counting pass.  Results do not change (a bound is a
                                * bound); 0: first stage only (A/B measurements) */
#define RTX_OPT_RECORDS 18 /* 4 (default; 0 .. 16): the RECORDS path of the tile pruning.  A query that the pruning gives a threshold and at most
                            * this many live tiles only ever needs its references with a count ABOVE the threshold (every other one is a
                            * reference without a hit): the epilogue of hit_count then writes (reference, count) records of those alone instead
                            * of the counts of 8192 references per tile, and one wave per query turns them into the prefix sums of
                            * lineage.rs:61-77 at the few taxonomy boundaries they touch and walks the lineage (rtx_records.hip) in the place
                            * of taxon_prefix's sweeps.  Same thresholds, same probabilities; 0: every query takes the dense epilogue.
                            * Shapes the workspace like the options below. */
#define RTX_OPT_OVERLAP 19 /* 1 (default): the back half of sub-batch k (prob_lookup, taxon_prefix + walk, records tail: chains of dependent round
                            * trips) runs on a second HIP stream beside the front half of sub-batch k + 1 (bounds, counting: VALU and L1
                            * rate); two scratch sets alternate -- taken only if the second one fits free HBM without shrinking the sub-batch.
                            * 2: three stages (bounds | threshold + counting | back half; measured: no faster than two).  0: one stream.
                            * Same results.  Whole-database handles whose batch is on the pruned path (where every tile is counted the second
                            * stream costs more than it hides), one handle per device (handles that rtx_raxtax_multi drives side by side on
                            * ONE device keep to one stream and one scratch set: two sets are ~120 GB at 500 000 references); shapes the workspace. */
#define RTX_OPT_MIN_SUB_BATCHES 20 /* 4 (default; 1 .. 64): a batch on the pruned path is cut into at least this many sub-batches (of 16 384 queries or
                                    * more), so that the host finalises the records of one while the next ones run and only the last one is left
                                    * when the device is done.  rtx_raxtax sets 2 for the time of a call: its chunks follow one another through
                                    * rtx_batch_download_then_run, which hides the last sub-batch of a chunk behind the next chunk.  Shapes the workspace. */
#define RTX_OPT_TWO_LEVEL_BOUNDS 21 /* 1 (default): the bounds pass of the tile pruning on a whole-database handle runs in two levels
                                     * (rtx_bounds2.hip): union bounds over blocks of 256 references for every tile -- four rows of that
                                     * bitmap per load instruction --, and over blocks of 64 only for the groups of four tiles whose
                                     * coarse bound comes close to the query's largest (sixteen rows per load instruction).  Every tile
                                     * keeps a valid upper bound either way: results differ from the one-level pass only through which
                                     * tiles are proven dead before counting (a function of the query alone).  0: the one-level pass
                                     * over blocks of 64 (reference shards always use it).  Values above 1 set the rule that picks the
                                     * refined groups (experiments): c_t | c_m << 16 | lo << 32 | hi << 48, in 1/256 of t. */
#define RTX_OPT_PRUNE_SELF_SAMPLE 22 /* 1 (default): the handle honours the verdict of rtx_index_self_sample (taken when the handle is created
                                      * from the reference sequences or from a tree): on a database whose own references keep 85 % or more
                                      * of its tiles live -- every tile holds relatives of every query: real barcodes of one order -- tile
                                      * pruning stays off whatever RTX_OPT_TILE_PRUNE says (bounds and thresholds would cost more than they
                                      * save).  0: RTX_OPT_TILE_PRUNE alone decides.  Shapes the workspace like RTX_OPT_TILE_PRUNE */
#define RTX_OPT_RUN_AHEAD 23         /* 0 (default).  1: rtx_batch_download_then_run enqueues the staged batch BEFORE the last sub-batch of the batch
                                      * being downloaded has finished: the front half of the next batch's first sub-batch runs beside the back
                                      * half of this batch's last one (RTX_OPT_OVERLAP's two streams never drain between the chunks of a file).
                                      * The result state of a batch then exists twice on the device.  The call may return RTX_RETRY_CHUNK (see
                                      * there); a run under this option leaves the handle's stream NOT covering its back halves until the next
                                      * call that needs it (rtx_batch_sync, a download, the next run, switching the option off).  rtx_raxtax sets
                                      * it for the duration of a call with more than one chunk.  Results are those of the plain sequence.  (2: a test aid -- every
                                      * second run-ahead is abandoned as if the batch had overflowed.) */
/* RTX_OPT_SUB_BATCH, _PACKED_COUNTS, _HIT_PAIR, _TILE_PRUNE and _PROB_MODE shape the workspace that rtx_batch_upload sizes:
 * setting one of them drops the uploaded batch (rtx_batch_run then fails with RTX_ERR_STATE until the batch is uploaded again). */
int rtx_index_set_option(rtx_index *index, int option, uint64_t value);

/* Does tile pruning pay on this database?  Classifies every (n_refs / n_sample)-th reference (n_sample = 0: 1024) as a query with its
 * exact copies left out (RTX_SKIP_EXACT_MATCHES) and tile pruning on, and keeps the share of (query, tile) combinations that stayed live:
 * from 0.85 on the handle leaves tile pruning off (see RTX_OPT_PRUNE_SELF_SAMPLE).  A property of the database: the results of a query
 * never depend on the rest of its batch.  rtx_index_create_from_sequences / _from_tree call it themselves; a handle built from postings
 * (rtx_index_create) has no sequences and keeps pruning on until the caller passes them here.  seq_bytes / seq_off as in
 * rtx_index_create_from_sequences (the references in the order of the handle).  *live_fraction (may be NULL): the share, or -1 where there
 * was nothing to decide (a reference shard, fewer than 4 tiles, no union bitmap).  Any uploaded batch is dropped.
 * The reference has no counterpart: it counts every reference for every query (raxtax.rs:58-64). */
int rtx_index_self_sample(rtx_index *index, const uint8_t *seq_bytes, const uint64_t *seq_off, uint64_t n_refs, uint32_t n_sample, double *live_fraction);
/* *pruning: 1 if tile pruning is on for this handle (RTX_OPT_TILE_PRUNE and the verdict above); *live_fraction: the share the sample kept live, -1 without a sample */
int rtx_index_prune_verdict(const rtx_index *index, int *pruning, double *live_fraction);
/* RTX_OPT_RUN_AHEAD: batches this handle enqueued ahead of the end of the batch before them, and run-aheads it abandoned (RTX_RETRY_CHUNK), since its creation */
int rtx_index_run_ahead_stats(const rtx_index *index, uint64_t *enqueued_ahead, uint64_t *abandoned);
/* Process-wide default for handles created afterwards.  RTX_DEFAULT_SEGMENT_CLASSES (default 1): at index creation
 * every (row, tile) segment of the bitmaps is classified; empty segments are never read and segments with at most
 * 16 references are added from 32-byte slots through byte counters instead of 1-KiB row reads (rtx_segments.hip).
 * 0: every segment is read densely.  Results are identical for either value (A/B measurements). */
#define RTX_DEFAULT_SEGMENT_CLASSES 1
/* RTX_DEFAULT_EXACT_HASH_MASK (default: all ones; 0 restores it): the hash of the device exact-match table is ANDed with this mask.
 * Tests pass a mask of a few bits so that most sequences collide in slot and tag and every probe ends in the byte compare. */
#define RTX_DEFAULT_EXACT_HASH_MASK 2
int rtx_set_default_option(int option, uint64_t value);

/* ------------------------------------------------------------------------- */
/* Classification: the body of raxtax(), src/raxtax.rs:39-84, minus string    */
/* formatting (host mirror); the exact-match lookup of raxtax.rs:42 on the     */
/* device, or the caller passes the ids.                                       */
/* ------------------------------------------------------------------------- */
/* One result row = one EvaluationResult (src/lineage.rs:8-14) before the
 * single-exact-match override of raxtax.rs:73-84 (applied by the caller /
 * host mirror, since it needs `raw_confidence` and the lineage strings). */
typedef struct {
    uint32_t n_queries;
    uint64_t n_rows;
    const uint32_t *t;             /* [n_queries] distinct valid 8-mers (k_mers.len(), raxtax.rs:55) */
    const uint8_t *status;         /* [n_queries] RTX_Q_*                                            */
    const double *global_signal;   /* [n_queries] lineage.rs:86-90                                   */
    const uint64_t *row_begin;     /* [n_queries] rows of query q = row_begin[q] .. row_begin[q] + row_count[q]; */
    const uint32_t *row_count;     /* [n_queries] the rows are stored in processing order, not in query order    */
    const uint32_t *row_lineage;   /* [n_rows] index into tree.lineages (lineage.rs:105)             */
    const uint32_t *row_node;      /* [n_rows] node id (rtx_nodes_view numbering)                    */
    const uint32_t *row_depth;     /* [n_rows] number of confidence values                           */
    const double *row_conf;        /* [n_rows][row_conf_stride] confidence_values, rounded to 2 decimals; entries from row_depth on are 0 */
    const double *row_local_signal;/* [n_rows] lineage.rs:95-102                                     */
    /* ABI 5.  The rows leave the device finalised (rtx_finalise.hip: sorted as lineage.rs:91-93 asks, local signal computed, final
     * layout), the host copies them and nothing else.  row_conf_stride = the deepest lineage of the tree (0 in a view a caller fills
     * by hand = RTX_MAX_DEPTH, the layout of ABI <= 4).  The two byte arrays carry what rtx_result_pack ships (NULL in a hand-made
     * view: the pack converts). */
    uint32_t row_conf_stride;
    const uint8_t *row_depth_u8;        /* [n_rows] row_depth as a byte                                   */
    const uint8_t *row_conf_hundredths; /* [n_rows][row_conf_stride] round(confidence * 100), lineage.rs:128-129 */
} rtx_result_view;

/* Whole path for one batch of queries: H2D, kernels, D2H, host finalisation (sort
 * lineage.rs:91-93 + local signal).  The handle alternates between two host result sets: a view stays
 * valid until the second-next download on the handle, so a caller can format batch c while batch c+1
 * is classified (the host mirror rtx_raxtax does).  exact_ids/exact_off: the ids Tree.sequences.get()
 * returned per query (raxtax.rs:42), or NULL / NULL: the library looks them up on the device (handles built from the reference
 * sequences, RTX_OPT_DEVICE_EXACT; rtx_batch_exact_matches returns them) -- on a handle without that table, or with the option off,
 * NULL means that no query has an exact match. */
/* The upload in two halves, so that the NEXT batch can cross PCIe while the current one is classified (rtx_raxtax does this with its
 * chunks; raxtax.rs:35-36: the reference's workers pick up their next chunk without waiting for anybody either):
 *   rtx_batch_prefetch  validates and stages a batch beside the current one: bases packed two per byte (4-bit codes, parser.rs:11-34)
 *                       into page-locked memory, asynchronous H2D on a stream of its own.  Returns when the host buffers may be reused.
 *   rtx_batch_activate  makes the staged batch the current one (the handle's stream waits for the transfer, the host does not): call
 *                       it once the batch before it has been downloaded, then rtx_batch_run.
 * rtx_batch_upload = prefetch + activate. */
int rtx_batch_prefetch(rtx_index *index, uint64_t n_queries, const uint8_t *bases, const uint64_t *base_off,
                       const uint32_t *exact_ids, const uint64_t *exact_off);
int rtx_batch_activate(rtx_index *index);
/* The packing rtx_batch_prefetch applies, on its own (no device involved): packed[i] = bases[2 i] | bases[2 i + 1] << 4 for n_bases
 * codes (an odd last one alone in its byte; packed holds (n_bases + 1) / 2 bytes).  Returns 1, or 0 if a byte above 15 was seen --
 * no code of parser.rs:11-34; rtx_batch_prefetch sends such a batch as it is. */
int rtx_pack_bases(const uint8_t *bases, uint64_t n_bases, uint8_t *packed);
int rtx_classify_batch(rtx_index *index, uint64_t n_queries, const uint8_t *bases,
                       const uint64_t *base_off, const uint32_t *exact_ids,
                       const uint64_t *exact_off, uint32_t flags, rtx_result_view *out);
/* 1 if the handle can look exact matches up itself (built from the reference sequences, RTX_OPT_DEVICE_EXACT on) */
int rtx_index_has_exact_lookup(const rtx_index *index);
/* The exact matches of the last download as the device found them (the batch was uploaded with exact_off == NULL): CSR over the
 * queries, ids ascending as Tree.sequences holds them (tree.rs:109-112).  Valid as long as the view of that download. */
int rtx_batch_exact_matches(rtx_index *index, const uint64_t **exact_off /*n_queries + 1*/, const uint32_t **exact_ids);

/* The same in stages, so that a caller (bench.py) can keep inputs resident in HBM and
 * time the device part alone, or overlap stages of different batches. */
int rtx_batch_upload(rtx_index *index, uint64_t n_queries, const uint8_t *bases,
                     const uint64_t *base_off, const uint32_t *exact_ids,
                     const uint64_t *exact_off);
int rtx_batch_run(rtx_index *index, uint32_t flags); /* enqueue all kernels (async)        */
int rtx_batch_sync(rtx_index *index);                /* wait for the handle's stream        */
int rtx_batch_download(rtx_index *index, rtx_result_view *out);
/* rtx_batch_download of the current batch that, as soon as the last result records have left the device, makes the STAGED batch
 * (rtx_batch_prefetch) the current one and runs it with `flags` (= rtx_batch_activate + rtx_batch_run) -- the host finalises the last
 * sub-batch of this batch while the device already classifies the next (rtx_raxtax's chunks follow one another without the host's
 * finalisation between them).  Nothing staged: plain rtx_batch_download.  Under RTX_OPT_RUN_AHEAD the staged batch is enqueued before the
 * current one has finished, and the call may return RTX_RETRY_CHUNK (> 0). */
int rtx_batch_download_then_run(rtx_index *index, rtx_result_view *out, uint32_t flags);

/* ---- staged execution of a reference-sharded handle (one sub-batch at a time) --------------------
 * Between the stages the caller exchanges two device buffers with the other shards (RCCL):
 *   after rtx_shard_count: all-reduce(sum) RTX_BUF_HIST   ([n][row_stride] uint32, n = queries of the sub-batch)
 *   after rtx_shard_prob : all-gather  RTX_BUF_PREFIX ([n][n_bnd_local] double); the global prefix row is the
 *                          concatenation over shards of prefix_s[1..] + sum of the totals prefix_s'[last] of
 *                          the shards s' < s, preceded by a 0 (n_bnd_global values) -> rtx_shard_walk.
 * Stages are asynchronous on the handle's stream: rtx_batch_sync before touching a buffer. */
#define RTX_BUF_HIST 1
#define RTX_BUF_PREFIX 2
int rtx_shard_begin(rtx_index *index, uint32_t *n_sub_batches, uint32_t *sub_batch);
/* Tile pruning on a reference shard (RTX_OPT_SHARD_PRUNE = 1 before the upload; 4 tiles or more on the shard; queries with t <= 1023):
 * the threshold of a query follows from the best block of 64 references ANYWHERE in the database, so the counting of a sub-batch
 * stops once for an exchange:   rtx_shard_bounds  ->  all-gather RTX_BUF_BEST, keep per query the record with the largest first word
 * (ties: the lowest shard) in every shard's buffer  ->  rtx_shard_count (counts the live tiles only)  ->  as before.
 * rtx_shard_prunes (after rtx_shard_begin): 1 if this run prunes -- all shards must agree (same options, same queries); the shards
 * then process the queries in their min-hash order, which is the same on every shard. */
#define RTX_OPT_SHARD_PRUNE 16
int rtx_shard_prunes(const rtx_index *index);
int rtx_shard_bounds(rtx_index *index, uint32_t sub_batch_idx, uint32_t flags);
int rtx_shard_count(rtx_index *index, uint32_t sub_batch_idx, uint32_t flags);
int rtx_shard_prob(rtx_index *index, uint32_t sub_batch_idx);
int rtx_shard_walk(rtx_index *index, uint32_t sub_batch_idx, const double *prefix_global /* device */);
int rtx_shard_info(const rtx_index *index, uint64_t *ref_lo, uint64_t *ref_hi, uint32_t *n_bnd_global,
                   uint32_t *n_bnd_local, uint32_t *first_bnd);
int rtx_device_buffer(rtx_index *index, int which, void **device_ptr, uint64_t *row_stride_elems);
/* The same for the scratch set of sub-batch `sub_batch_idx`: a staged run alternates between two sets, so that
 * the exchange of sub-batch i (on the caller's stream or RCCL's) can overlap with rtx_shard_count of sub-batch i + 1.
 * RTX_BUF_COUNTS: the u16 hit counts ([n][row_stride], needs RTX_OPT_PACKED_COUNTS = 0) -- what a k-mer-sharded
 * database (SURVEY.md 8e mode A, the literal wording of BASELINE.json configs[4]) all-reduces; rtx_shard_rehist then
 * rebuilds the histogram of prob.rs:13-19 from the summed counts before rtx_shard_prob. */
#define RTX_BUF_COUNTS 3
#define RTX_BUF_BEST 4 /* [n][66] uint32: {largest bound of a block of 64 references, 0, exact counts of that block's references} */
int rtx_shard_buffer(rtx_index *index, uint32_t sub_batch_idx, int which, void **device_ptr, uint64_t *row_stride_elems);
int rtx_shard_rehist(rtx_index *index, uint32_t sub_batch_idx);
/* The HIP stream (hipStream_t) every kernel of this handle is enqueued on: a caller that interleaves its own device
 * work (collectives) with rtx_shard_* orders it against this stream instead of synchronising the host. */
int rtx_index_stream(rtx_index *index, void **hip_stream);

/* Per-kernel device time of the last rtx_batch_run, from HIP events on the library's
 * stream (ms, summed over sub-batches), and launch counts (0 launches = stage not timed, see
 * RTX_OPT_STAGE_TIMING).  Stage order: */
#define RTX_STAGE_KMER_EXTRACT 0
#define RTX_STAGE_HIT_COUNT 1
#define RTX_STAGE_PROB_TABLE 2
#define RTX_STAGE_TAXON_PREFIX 3
#define RTX_STAGE_LINEAGE_WALK 4
#define RTX_STAGE_TILE_BOUNDS 5 /* tile pruning (RTX_OPT_TILE_PRUNE): the queries counted against the union bitmap (the hit_count kernel again); 0 launches if the run did not prune */
#define RTX_STAGE_TILE_PRUNE 6  /* ... prune_kernel (thresholds, live tiles) + the row lists of the live tiles */
#define RTX_STAGE_EXACT_MATCH 7 /* Tree.sequences.get on the device (rtx_exact.hip; RTX_OPT_DEVICE_EXACT), once per run: reported with sub-batch 0 */
#define RTX_STAGE_ORDER 8       /* processing order of the batch (rtx_cluster.hip: sketch, locator, radix sort, inverse), once per run: reported with sub-batch 0 */
#define RTX_STAGE_PAIR_UNION 9  /* union row lists of the pairs of neighbouring queries (pair_union_kernel), once per sub-batch */
#define RTX_NUM_STAGES 10
int rtx_batch_stage_times(rtx_index *index, float ms[RTX_NUM_STAGES], uint32_t launches[RTX_NUM_STAGES]);
/* Algorithmic work of the last rtx_batch_run: sum over queries of
 * H_q = sum_r count_q[r] (postings touched) and of L_q (query bytes), and the bitmap
 * bytes the hit_count kernel actually requested. */
int rtx_batch_work(rtx_index *index, uint64_t *sum_hits, uint64_t *sum_query_bytes,
                   uint64_t *bitmap_bytes_read);
/* bitmap_bytes_read split by the kind of launch of the hit_count kernel: the tiles of the database it counted, and -- with tile pruning --
 * the bounds pass on the union bitmap */
int rtx_batch_work_split(rtx_index *index, uint64_t *live_bytes, uint64_t *bounds_bytes);
/* Algorithmic work of the probability stage of the last rtx_batch_run (prob.rs:43-90):
 * sum over queries of D_q (n_q + 1) -- the points of the pmf/cmf grid the reference evaluates, D_q = number of
 * distinct hit counts, n_q = t_q / 2 -- and of D_q.  ops_prob = 3 x grid points (2 exp + 1 log each). */
int rtx_batch_prob_work(rtx_index *index, uint64_t *sum_grid_points, uint64_t *sum_distinct_counts);

/* ------------------------------------------------------------------------- */
/* Parity / debug taps (full vectors; not used on the fast path)              */
/* ------------------------------------------------------------------------- */
/* Valid after rtx_batch_run + rtx_batch_sync for queries of the LAST sub-batch only when
 * the batch spans several sub-batches; tests keep n_queries <= sub-batch.  */
int rtx_debug_kmers(rtx_index *index, uint64_t query, uint16_t *kmers /*cap 65535*/, uint32_t *t);
int rtx_debug_hit_counts(rtx_index *index, uint64_t query, uint16_t *counts /*n_refs*/);
int rtx_debug_prob_table(rtx_index *index, uint64_t query, double *table_over_z /*t+1*/, double *z);
int rtx_debug_probs(rtx_index *index, uint64_t query, double *probs /*n_refs*/);
/* tile pruning: the largest bound of every tile of 8192 references as the bounds pass left it for a query of the last sub-batch
 * (what prune_kernel's tile-aware threshold and the live masks are derived from); RTX_ERR_STATE if the last run did not prune */
int rtx_debug_tile_bounds(rtx_index *index, uint64_t query, uint16_t *tile_ub /*ceil(n_refs / 8192)*/);
/* processing order of the last run (RTX_OPT_CLUSTER / RTX_OPT_LOCATOR): perm[position] = query */
int rtx_debug_order(rtx_index *index, uint32_t *perm /*n_queries*/);
/* queries per sub-batch of the uploaded batch and the number of sub-batches: positions [(n_sub - 1) * sub_batch, n_queries) of the
 * processing order are the last sub-batch, the one the taps can read */
int rtx_batch_sub_batch(const rtx_index *index, uint32_t *sub_batch, uint32_t *n_sub_batches);
/* A batch is cut into LENGTH CLASSES (t <= 255 / t <= 1023 / t <= 2047 (ABI 5) / longer reads whose probability arrays fit LDS / up to t = 65 535): the class
 * leads the processing order, every class runs through sub-batches of its own shape (bit planes, pair kernel and tile pruning, memoised
 * tables or the recurrence kernel), so that one long read does not move a batch of barcodes off the fast path; results come back in
 * input order.  rtx_batch_sub_batch reports the sub-batch size of the LAST class and the number of sub-batches of the whole batch;
 * rtx_batch_last_sub_batch the positions [first, first + n) of the processing order that form the last sub-batch (the one the taps read);
 * rtx_batch_classes, per class c < *n_classes <= 5 (4 until ABI 4: out[16]): out[4c] queries, [4c + 1] longest query (bases), [4c + 2] sub-batch size,
 * [4c + 3] bit planes | tables << 8 | pair kernel << 9 | tile pruning << 10 | records path << 11 | global-memory forms << 12 (the last
 * four after rtx_batch_run). */
int rtx_batch_last_sub_batch(const rtx_index *index, uint64_t *first, uint32_t *n);
int rtx_batch_classes(const rtx_index *index, uint32_t *n_classes, uint64_t out[20]);
/* tile pruning of the last run (RTX_OPT_TILE_PRUNE): out[0] (pair, tile) blocks that are counted for at least one of their two queries, [1] pairs,
 * [2] sum of the lower bounds of the best hit, [3] sum of the thresholds, [4] sum of the largest tile bounds, [5] queries, [6] bounds below a count
 * they bound (must be 0), [7] (query, tile) combinations that are counted, [8] (query, tile) combinations with a count above the query's
 * threshold -- what exact knowledge would have counted --, [9] queries with a threshold; all 0 if the run did not prune.
 * The fine bounds pass (RTX_OPT_FINE_BOUNDS): [10] (query, tile) combinations it took off the lists, [11] its (pair, fine tile) blocks,
 * [12] (pair, tile) blocks the counting pass was left with (0 if the pass did not run: then [0] is that number).
 * The records path (RTX_OPT_RECORDS): [13] records written (references above their query's threshold), [14] queries on the path,
 * [15] of those, queries whose boundary entries did not fit LDS (prefix row written out, walked from memory) */
int rtx_debug_prune_stats(rtx_index *index, uint64_t *out /*16*/);
/* table / Z of a query of the last sub-batch as the PRUNED run computed it (0 for the counts up to the query's threshold), its Z and the
 * threshold; must be called before any other tap (those recount the sub-batch in full) */
int rtx_debug_pruned_prob_table(rtx_index *index, uint64_t query, double *table_over_z /*t+1*/, double *z, uint32_t *threshold);
/* The last sub-batch exactly as the run left it -- NO recount (the taps above prove the kernel that counts every tile; this one shows
 * what the timed, pruned path computed); must be called before any recounting tap.  counts[n_refs]: what hit_count wrote (0xFFFF for the
 * references of tiles it did not visit for the query's pair); tile_live[ntiles]: 1 if the tile was visited; hist[t + 1]: the histogram
 * of prob.rs:13-19 as prune_kernel (bin 0 = the references never counted) and hit_count (every counted reference) left it; the query's
 * threshold (0: none) and i* + 1 (rtx_prune.hip).  Any output pointer may be NULL. */
int rtx_debug_run_counts(rtx_index *index, uint64_t query, uint16_t *counts /*n_refs*/, uint8_t *tile_live /*ntiles*/,
                         uint32_t *hist /*t+1*/, uint32_t *threshold, uint32_t *i1);
/* A query on the records path (RTX_OPT_RECORDS) wrote no counts but the (reference, count) records of the counts above its threshold:
 * rtx_debug_run_counts then returns, for the visited tiles, the count of every reference that has a record and 0 for every other one
 * (read back from the record segments, which must be in ascending reference order).  n_segments: 0 = the query took the dense epilogues. */
int rtx_debug_run_mode(rtx_index *index, uint64_t query, uint32_t *n_segments);
/* prune_kernel's view of a query of the last sub-batch (needs RTX_OPT_DEBUG_TAPS = 1 before the run): out[0] the block of 64 references
 * with the largest bound, [1] M = the best exact count in it, [2] the threshold, [3] i* + 1, [4] the largest bound, [5] t, [8 .. 72) the exact
 * counts of the block's references (0: behind the end of the database, or zeroed by RTX_SKIP_EXACT_MATCHES) */
int rtx_debug_prune_detail(rtx_index *index, uint64_t query, uint32_t *out /*72*/);
/* Lineage::new(label, tree, probs).evaluate() (src/lineage.rs:61-112) on a caller-supplied
 * probability vector: runs taxon_prefix + lineage_walk + the host finalisation for one
 * pseudo-query.  Lets the reference's lineage KATs pin the device walk.  Small trees only
 * (n_refs up to a few thousand: the workspace is sized as for a query with n_refs k-mers). */
int rtx_debug_evaluate(rtx_index *index, const double *probs /*n_refs*/, rtx_result_view *out);

/* ------------------------------------------------------------------------- */
/* Host mirror of the output formatting (src/lineage.rs:17-48, utils.rs:62-89) */
/* ------------------------------------------------------------------------- */
/* Applies the single-exact-match override (raxtax.rs:73-84, unless RTX_RAW_CONFIDENCE or
 * RTX_SKIP_EXACT_MATCHES is set in flags) and formats the `.out` (and, if tsv_buf != NULL,
 * `.tsv`) lines of query q of a result view.  Returns bytes written (excluding NUL) or a
 * negative RTX_ERR_*; lines of one query are '\n'-joined without a trailing newline. */
int64_t rtx_format_query(const rtx_tree *tree, const rtx_result_view *res, uint64_t q,
                         const char *label, const uint8_t *seq, uint64_t seq_len,
                         const uint32_t *exact_ids, uint64_t n_exact, uint32_t flags, char *out_buf,
                         uint64_t out_cap, char *tsv_buf, uint64_t tsv_cap, int64_t *tsv_len);

/* Compact byte record of a result view: what a rank ships in the multi-GPU result gather (BASELINE.json
 * configs[3]; layout in raxtax_amd/dist_util.py: 32 B header, 25 B per query (begin, global signal, row count, t,
 * status) + 13 + L B per row (lineage u32, depth, L confidences as hundredths, local signal), L = depth of the
 * deepest row, so every level of every row travels).  buf == NULL: returns the size needed; else the bytes written
 * or a negative RTX_ERR_*. */
int64_t rtx_result_pack(const rtx_result_view *res, uint8_t *buf, uint64_t cap);

/* The writer's side of that gather (main.rs:126-136 for the results of every rank): the `.out` lines of all queries of ONE packed record
 * buffer, formatted natively on `threads` threads (0: the library's budget) -- the text rtx_format_query gives for the view the records
 * were packed from.  labels: [n] the labels of the buffer's queries; exact_one: [n] the id of a query's ONLY exact match (the override of
 * raxtax.rs:73-84; 0xFFFFFFFF: none or several) or NULL; flags: RTX_SKIP_EXACT_MATCHES / RTX_RAW_CONFIDENCE switch the override off as in the
 * reference.  out: the texts back to back, each NUL-terminated, lines of one query '\n'-joined; line_off: [n + 1] where each starts (a query
 * without rows -- status != 0 -- has an empty text) or NULL.  Returns the bytes written, with out == NULL the bytes needed, or a negative RTX_ERR_*;
 * a buffer that is too small: -(bytes needed) - RTX_NEED_BASE (every value below -RTX_NEED_BASE; ABI 5) -- the text has been formatted to be
 * measured, the caller allocates and calls once more. */
#define RTX_NEED_BASE 1024
int64_t rtx_records_format(const rtx_tree *tree, const uint8_t *records, uint64_t n_bytes, const char *const *labels,
                           const uint32_t *exact_one, uint32_t flags, char *out, uint64_t cap, uint64_t *line_off, uint32_t threads);

/* ------------------------------------------------------------------------- */
/* Host mirror of raxtax() itself (src/raxtax.rs:14-97)                       */
/* ------------------------------------------------------------------------- */
/* Receives one message per query, in input order: the `(label, out_lines, tsv_lines?)`
 * triple the reference sends on its crossbeam channel (raxtax.rs:85-87).  A non-zero
 * return plays the role of a closed channel: rtx_raxtax stops with RTX_ERR_SENDER. */
typedef int (*rtx_sender_fn)(void *ctx, const char *label, const char *out_lines, const char *tsv_lines);
/* Same arguments, in the same order and with the same meaning as the reference function,
 * plus the device index that replaces the CPU traversal of `tree`.  chunk_size = queries
 * per device batch (0 = all at once; the reference's rayon chunking has no other effect).
 * Exact-match lookups (raxtax.rs:42), the lineage-consistency warning (raxtax.rs:43-53,
 * written to stderr) and the override (raxtax.rs:73-84) run on the host. */
int rtx_raxtax(rtx_index *index, const rtx_tree *tree, uint64_t n_queries, const char *const *labels,
               const uint8_t *bases, const uint64_t *base_off, int skip_exact_matches, int raw_confidence,
               uint64_t chunk_size, rtx_sender_fn sender, void *sender_ctx, int tsv);

/* The same on several device handles at once -- one per GPU of the node (the index replicated, BASELINE.json configs[3]), or several
 * on one GPU: chunks of `chunk_size` queries (0: one chunk per handle) are dealt to the handles in turn, every handle is driven by a
 * thread of its own inside this call, and the messages reach `sender` in input order, as from rtx_raxtax.  The reference's parallel
 * driver is one call inside the process too (par_chunks over the rayon pool, raxtax.rs:35-36, main.rs:40-57). */
int rtx_raxtax_multi(rtx_index *const *indices, uint32_t n_indices, const rtx_tree *tree, uint64_t n_queries,
                     const char *const *labels, const uint8_t *bases, const uint64_t *base_off, int skip_exact_matches,
                     int raw_confidence, uint64_t chunk_size, rtx_sender_fn sender, void *sender_ctx, int tsv);
/* A ready-made sender that discards the messages and only counts them: ctx = NULL or uint64_t[2] {messages, bytes of text} */
int rtx_sender_discard(void *ctx, const char *label, const char *out_lines, const char *tsv_lines);
/* Busy seconds of the stages of the last rtx_raxtax / rtx_raxtax_multi call of this process (which stage bounds an end-to-end run):
 * busy[0] exact-match lookup on the host (0 when the device does it), busy[1] device stage of the busiest handle (upload, kernels,
 * download), busy[2] formatting of the busiest handle, busy[3] sender; *n_chunks = chunks of that call.  The counterpart of the
 * reference's `timer!` log lines (raxtax.rs:13, prob.rs:7, lineage.rs:79). */
int rtx_raxtax_last_timing(double busy[4], uint64_t *n_chunks);

/* ------------------------------------------------------------------------- */
/* Host thread budget (the reference: rayon pool of std::thread::available_parallelism threads, main.rs:40-57, utils.rs:139-158) */
/* ------------------------------------------------------------------------- */
/* Every worker pool of the library (FASTA parsing, result finalisation, formatting, record packing) is sized from the CPUs this
 * process may really use -- the affinity mask capped by the cgroup CPU quota, not the logical CPUs of the machine -- divided by the
 * number of ranks that share the host.  That number is LOCAL_WORLD_SIZE (exported by torch.distributed.run) unless set here; the
 * handles driven by one rtx_raxtax_multi call divide their share among themselves. */
int rtx_set_host_share(uint32_t n_ranks_on_this_host);
uint32_t rtx_host_threads(void); /* threads a pool of this process may use (>= 1) */

#ifdef __cplusplus
}
#endif
#endif /* RAXTAX_HIP_H */
