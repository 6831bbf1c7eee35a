/*
 * oracle.h -- CPU restatement ("oracle") of the raxtax per-query classification
 * hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / reported CPU baseline.  The
 * product (libraxtax_hip.so, raxtax_amd/) never links, imports or calls it.
 *
 * Every function cites the reference file:line (noahares/raxtax v1.5.0) whose
 * behaviour it restates.  The reference is Rust and cannot be built in this
 * image (no cargo/rustc), so the restatement is pinned by the reference's own
 * unit-test vectors (tests/golden/reference_kats.json, transcribed from
 * src/utils.rs:209-273, src/parser.rs:167-299, src/lineage.rs:192-334,
 * src/prob.rs:209-235).  Third-party arithmetic on the path:
 * statrs 0.16.x `ln_binomial` (Cargo.toml:39, version unpinned, no lockfile,
 * source absent from the reference tree) -- restated from its published
 * algorithm (cached factorial table <= 170, Lanczos ln_gamma above).  The exact
 * digits of table[m] are therefore "parity unpinned" beyond the reference's
 * own 1e-7 property tests (prob.rs:209-235).
 */
#ifndef RAXTAX_ORACLE_H
#define RAXTAX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAXD 64          /* max lineage depth the row struct can carry      */
#define ORC_NUM_KMERS 65536  /* 2 << 15, tree.rs:52                              */

enum { ORC_NODE_INNER = 0, ORC_NODE_TAXON = 1, ORC_NODE_SEQUENCE = 2 }; /* tree.rs:181-186 */

typedef struct orc_tree orc_tree;
typedef struct orc_queries orc_queries;

typedef struct {
    uint64_t idx;              /* index into tree.lineages (first ref of the node) */
    uint32_t depth;            /* number of confidence values                       */
    uint32_t pad;
    double conf[ORC_MAXD];     /* confidence_values (rounded to 2 decimals)         */
    double expd[ORC_MAXD];     /* expected confidence prefix (range/num_tips)       */
    double local_signal;
    double global_signal;
} orc_row;

/* ---- utils.rs ---------------------------------------------------------- */
int      orc_map_four_to_two_bit_repr(uint8_t c);                 /* utils.rs:17-25, -1 = None */
uint32_t orc_sequence_to_kmers(const uint8_t *seq, uint64_t len, uint16_t *out); /* utils.rs:27-40 */
void     orc_decompress_sequence(const uint8_t *seq, uint64_t len, char *out);  /* utils.rs:70-81 */
double   orc_euclidean_distance_l1(const double *a, const double *b, uint64_t n); /* utils.rs:91-105 */
double   orc_euclidean_norm(const double *v, uint64_t n);          /* utils.rs:107-116 */
double   orc_cosine_similarity(const double *a, const double *b, uint64_t n); /* utils.rs:118-129 */

/* ---- statrs 0.16 restatement ------------------------------------------- */
double orc_ln_gamma(double x);
double orc_ln_factorial(uint64_t x);
double orc_ln_binomial(uint64_t n, uint64_t k);

/* ---- parser.rs --------------------------------------------------------- */
int orc_map_dna_char(int ch);                                      /* parser.rs:11-34, -1 = panic */
orc_tree *orc_parse_reference_fasta_str(const char *s, int *err);  /* parser.rs:46-105 */
orc_queries *orc_parse_query_fasta_str(const char *s, const char *const *skip, uint64_t n_skip,
                                       int *err);                  /* parser.rs:117-154 */
uint64_t orc_queries_len(const orc_queries *q);
const char *orc_queries_label(const orc_queries *q, uint64_t i);
uint64_t orc_queries_seq(const orc_queries *q, uint64_t i, const uint8_t **seq);
void orc_queries_free(orc_queries *q);

/* ---- tree.rs ----------------------------------------------------------- */
orc_tree *orc_tree_new(uint64_t n, const char *const *lineages, const uint8_t *seq_bytes,
                       const uint64_t *seq_off);                   /* tree.rs:46-140 */
void orc_tree_free(orc_tree *t);
uint64_t orc_tree_num_tips(const orc_tree *t);
const char *orc_tree_lineage(const orc_tree *t, uint64_t i);       /* sorted order */
uint64_t orc_tree_original_index(const orc_tree *t, uint64_t i);   /* sorted idx -> input idx */
uint64_t orc_tree_kmer_list(const orc_tree *t, uint32_t kmer, const uint32_t **ids);
uint64_t orc_tree_total_postings(const orc_tree *t);
void orc_tree_export_csr(const orc_tree *t, uint64_t *offsets /*65537*/, uint32_t *postings);
uint64_t orc_tree_exact_matches(const orc_tree *t, const uint8_t *seq, uint64_t len,
                                const uint32_t **ids);             /* tree.sequences.get, raxtax.rs:42 */
uint64_t orc_tree_num_nodes(const orc_tree *t);                    /* pre-order, root = 0, all types */
void orc_tree_export_nodes(const orc_tree *t, uint64_t *lo, uint64_t *hi, int64_t *parent,
                           uint8_t *type, uint32_t *n_children);
const char *orc_tree_node_label(const orc_tree *t, uint64_t node);

/* ---- raxtax.rs / prob.rs / lineage.rs ----------------------------------- */
/* raxtax.rs:41,55-68: fill(0), k-mer set, posting traversal, optional exact zeroing */
uint32_t orc_hit_counts(const orc_tree *t, const uint8_t *seq, uint64_t len, int skip_exact,
                        uint16_t *counts);
/* prob.rs:8-103.  Returns 0, or <0 where the reference would panic
 * (-1: t==0 underflow prob.rs:21; -2: zip_eq mismatch n==0 prob.rs:162; -3: probs_sum<=0 prob.rs:98). */
int orc_highest_hit_prob_per_reference(uint16_t total_num_k_mers, uint64_t num_trials,
                                       const uint16_t *intersection_sizes, uint64_t n_refs,
                                       double *out);
/* Same computation, but exposes the unnormalised per-count table (0 for absent m) and Z. */
int orc_prob_table(uint16_t total_num_k_mers, uint64_t num_trials, const uint16_t *sizes,
                   uint64_t n_refs, double *table /* t+1 */, double *z);
/* prob.rs:121-170 for one (m): writes n+1 ln-pmf values */
void orc_iterative_pmf_ln(uint64_t t, uint64_t n, uint64_t m, double ln_total, double *out);
/* lineage.rs:61-179: rows sorted as lineage.rs:91-93; returns row count (or -cap-1 on overflow) */
int orc_lineage_evaluate(const orc_tree *t, const double *probs, orc_row *rows, int cap);
/* raxtax.rs:39-88 for one query; returns row count, <0 = reference would panic */
int orc_classify(const orc_tree *t, const uint8_t *seq, uint64_t len, int skip_exact,
                 int raw_confidence, orc_row *rows, int cap);
/* lineage.rs:17-29 / 31-48 + utils.rs:62-68,83-89; returns bytes written (excl. NUL) or -1 */
int64_t orc_format_out(const orc_tree *t, const char *label, const orc_row *rows, int n,
                       char *buf, uint64_t cap);
int64_t orc_format_tsv(const orc_tree *t, const char *label, const orc_row *rows, int n,
                       const uint8_t *seq, uint64_t len, char *buf, uint64_t cap);
/* raxtax.rs:35-88 over a batch with rayon-like chunking (main.rs:119-124).  rows_out is
 * n_q*cap rows (may be NULL), n_rows_out n_q ints (may be NULL).  Returns #queries that
 * would have panicked in the reference.  Used as bench.py's cpu_baseline ("port"). */
int64_t orc_classify_batch(const orc_tree *t, uint64_t n_q, const uint8_t *bases,
                           const uint64_t *base_off, int skip_exact, int raw_confidence,
                           int threads, orc_row *rows_out, int cap, int32_t *n_rows_out,
                           int format_strings);

/* The same with the pool pinned to one logical CPU per physical core (utils.rs:139-158, `--pin`):
 * pin_cpus from orc_physical_core_ids (utils.rs:160-197). */
int orc_physical_core_ids(int *ids, int cap);
int64_t orc_classify_batch_ex(const orc_tree *t, uint64_t n_q, const uint8_t *bases,
                              const uint64_t *base_off, int skip_exact, int raw_confidence,
                              int threads, orc_row *rows_out, int cap, int32_t *n_rows_out,
                              int format_strings, const int *pin_cpus, int n_pin);
/* Parity taps over many queries at once (full-size tests): hit counts (raxtax.rs:41,55-68) and the
 * normalised probability table table[m]/Z (prob.rs:8-103) of every query, on `threads` threads. */
void orc_hit_counts_batch(const orc_tree *t, uint64_t n_q, const uint8_t *bases, const uint64_t *base_off,
                          int skip_exact, int threads, uint16_t *counts_out, uint32_t *t_out);
void orc_prob_tables_batch(uint64_t n_q, const uint32_t *t_arr, const uint16_t *counts, uint64_t n_refs,
                           int threads, double *tables, uint64_t tstride, double *z, int32_t *rc_out);

#ifdef __cplusplus
}
#endif
#endif
