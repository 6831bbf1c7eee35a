/*
 * oracle.c -- CPU restatement of the raxtax hot path.  TEST INFRASTRUCTURE ONLY
 * (see oracle.h).  Plain C11, no dependencies beyond libc/libm (+OpenMP for the
 * batch driver).  Faithful to the reference's data structures and passes:
 * Vec<Vec<u32>> postings, one u16[N] buffer per chunk, log-space pmf/cmf grid,
 * sequential prefix sums + recursive tree walk.  The only deliberate deviation
 * is iteration order over the hit-count histogram: the reference iterates an
 * ahash HashMap (random order, prob.rs:13-19,64-71); the oracle iterates the
 * distinct counts in ascending order, which is one of the orders the reference
 * itself may take.
 */
#define _GNU_SOURCE
#include "oracle.h"

#include <ctype.h>
#include <math.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* small helpers                                                             */
/* ------------------------------------------------------------------------ */
static void *xmalloc(size_t n) {
    void *p = malloc(n ? n : 1);
    if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
    return p;
}
static void *xcalloc(size_t n, size_t s) {
    void *p = calloc(n ? n : 1, s ? s : 1);
    if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
    return p;
}
static void *xrealloc(void *q, size_t n) {
    void *p = realloc(q, n ? n : 1);
    if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
    return p;
}
static char *xstrndup(const char *s, size_t n) {
    char *p = xmalloc(n + 1);
    memcpy(p, s, n);
    p[n] = 0;
    return p;
}

/* ------------------------------------------------------------------------ */
/* utils.rs                                                                  */
/* ------------------------------------------------------------------------ */

/* utils.rs:17-25 */
int orc_map_four_to_two_bit_repr(uint8_t c) {
    switch (c) {
    case 1: return 0;
    case 2: return 1;
    case 4: return 2;
    case 8: return 3;
    default: return -1;
    }
}

/* utils.rs:27-40: distinct valid 8-mers, first base in bits 15:14, ascending. */
uint32_t orc_sequence_to_kmers(const uint8_t *seq, uint64_t len, uint16_t *out) {
    if (len < 8) return 0;
    uint8_t *seen = xcalloc(ORC_NUM_KMERS / 8, 1); /* stands in for the HashSet */
    for (uint64_t w = 0; w + 8 <= len; w++) {
        uint32_t k = 0;
        int ok = 1;
        for (int j = 0; j < 8; j++) {
            int c = orc_map_four_to_two_bit_repr(seq[w + j]);
            if (c < 0) { ok = 0; break; }
            k |= (uint32_t)c << (14 - 2 * j);
        }
        if (ok) seen[k >> 3] |= (uint8_t)(1u << (k & 7));
    }
    uint32_t n = 0;
    for (uint32_t k = 0; k < ORC_NUM_KMERS; k++) /* .sorted() */
        if (seen[k >> 3] & (1u << (k & 7))) out[n++] = (uint16_t)k;
    free(seen);
    return n;
}

/* utils.rs:70-81 */
void orc_decompress_sequence(const uint8_t *seq, uint64_t len, char *out) {
    for (uint64_t i = 0; i < len; i++) {
        switch (seq[i]) {
        case 1: out[i] = 'A'; break;
        case 2: out[i] = 'C'; break;
        case 4: out[i] = 'G'; break;
        case 8: out[i] = 'T'; break;
        default: out[i] = '-';
        }
    }
    out[len] = 0;
}

/* utils.rs:91-105.  Returns NaN where the reference asserts. */
double orc_euclidean_distance_l1(const double *a, const double *b, uint64_t n) {
    if (n == 0) return 0.0;
    double a_sum = 0.0, b_sum = 0.0;
    for (uint64_t i = 0; i < n; i++) a_sum += a[i];
    for (uint64_t i = 0; i < n; i++) b_sum += b[i];
    if (!(a_sum > 0.0) || !(b_sum > 0.0)) return NAN;
    double s = 0.0;
    for (uint64_t i = 0; i < n; i++) {
        double d = a[i] / a_sum - b[i] / b_sum;
        s += d * d; /* powi(2) */
    }
    return sqrt(s);
}

/* utils.rs:107-116 */
double orc_euclidean_norm(const double *v, uint64_t n) {
    double s = 0.0;
    for (uint64_t i = 0; i < n; i++) s += v[i] * v[i];
    return sqrt(s);
}

/* utils.rs:118-129 */
double orc_cosine_similarity(const double *a, const double *b, uint64_t n) {
    double na = orc_euclidean_norm(a, n), nb = orc_euclidean_norm(b, n);
    double s = 0.0;
    for (uint64_t i = 0; i < n; i++) s += a[i] * b[i];
    return s / (na * nb);
}

/* ------------------------------------------------------------------------ */
/* statrs 0.16.x: function::gamma::ln_gamma, function::factorial::{ln_factorial,
 * ln_binomial} -- restated from the published crate source (not under
 * /root/reference): Lanczos approximation g = 10.900511, 11 coefficients
 * (Godfrey), factorial cache of 171 f64 entries.                            */
/* ------------------------------------------------------------------------ */
static const double GAMMA_R = 10.900511;
static const double GAMMA_DK[11] = {
    2.48574089138753565546e-5, 1.05142378581721974210,  -3.45687097222016235469,
    4.51227709466894823700,    -2.98285225323576655721, 1.05639711577126713077,
    -1.95428773191645869583e-1, 1.70970543404441224307e-2, -5.71926117404305781283e-4,
    4.63399473359905636708e-6, -2.71994908488607703910e-9};
static const double LN_2_SQRT_E_OVER_PI = 0.6207822376352452223455184457816472122518527279025978;
static const double LN_PI = 1.1447298858494001741434273513530587116472948129153;

double orc_ln_gamma(double x) {
    if (x < 0.5) {
        double s = GAMMA_DK[0];
        for (int i = 1; i < 11; i++) s += GAMMA_DK[i] / ((double)i - x);
        return LN_PI - log(sin(M_PI * x)) - log(s) - LN_2_SQRT_E_OVER_PI -
               (0.5 - x) * log((0.5 - x + GAMMA_R) / M_E);
    }
    double s = GAMMA_DK[0];
    for (int i = 1; i < 11; i++) s += GAMMA_DK[i] / (x + (double)i - 1.0);
    return log(s) + LN_2_SQRT_E_OVER_PI + (x - 0.5) * log((x - 0.5 + GAMMA_R) / M_E);
}

static double g_fcache[171];
static int g_fcache_ready = 0;
static void fcache_init(void) {
    if (g_fcache_ready) return;
    g_fcache[0] = 1.0;
    for (int i = 1; i <= 170; i++) g_fcache[i] = g_fcache[i - 1] * (double)i;
    __atomic_store_n(&g_fcache_ready, 1, __ATOMIC_RELEASE);
}

double orc_ln_factorial(uint64_t x) {
    fcache_init();
    if (x <= 170) return log(g_fcache[x]);
    return orc_ln_gamma((double)x + 1.0);
}

double orc_ln_binomial(uint64_t n, uint64_t k) {
    if (k > n) return -INFINITY;
    return orc_ln_factorial(n) - orc_ln_factorial(k) - orc_ln_factorial(n - k);
}

/* ------------------------------------------------------------------------ */
/* tree.rs                                                                   */
/* ------------------------------------------------------------------------ */
typedef struct onode {
    char *label;
    uint64_t lo, hi; /* confidence_range */
    int type;
    struct onode **children;
    uint32_t n_children, cap_children;
} onode;

typedef struct {
    uint32_t *ids;
    uint64_t n, cap;
} olist;

typedef struct {
    uint64_t hash;
    uint64_t first; /* a sorted index whose sequence is the key */
    uint32_t *ids;
    uint32_t n, cap;
    int used;
} oseq_entry;

struct orc_tree {
    onode *root;
    uint64_t n; /* number of input sequences */
    uint64_t num_tips;
    char **lineages;     /* sorted */
    uint64_t *orig_idx;  /* sorted idx -> input idx */
    uint8_t *seq_bytes;  /* sorted order, concatenated */
    uint64_t *seq_off;   /* n+1 */
    olist *k_mer_map;    /* 65536 */
    oseq_entry *seq_tab; /* open addressing */
    uint64_t seq_tab_size;
    /* flattened pre-order view (built lazily) */
    onode **flat;
    int64_t *flat_parent;
    uint64_t n_nodes;
};

static onode *node_new(const char *label, size_t label_len, uint64_t idx, int type) {
    onode *n = xcalloc(1, sizeof(onode));
    n->label = xstrndup(label, label_len);
    n->lo = idx;      /* tree.rs:197-204 */
    n->hi = idx + 1;
    n->type = type;
    return n;
}
static void node_add_child(onode *p, onode *c) {
    if (p->n_children == p->cap_children) {
        p->cap_children = p->cap_children ? p->cap_children * 2 : 2;
        p->children = xrealloc(p->children, p->cap_children * sizeof(onode *));
    }
    p->children[p->n_children++] = c;
}
static void node_free(onode *n) {
    for (uint32_t i = 0; i < n->n_children; i++) node_free(n->children[i]);
    free(n->children);
    free(n->label);
    free(n);
}

static uint64_t hash_bytes(const uint8_t *p, uint64_t len) {
    uint64_t h = 0xcbf29ce484222325ull;
    for (uint64_t i = 0; i < len; i++) { h ^= p[i]; h *= 0x100000001b3ull; }
    h ^= len;
    h *= 0x9E3779B97F4A7C15ull;
    return h ^ (h >> 29);
}

static const uint8_t *tree_seq(const orc_tree *t, uint64_t i, uint64_t *len) {
    *len = t->seq_off[i + 1] - t->seq_off[i];
    return t->seq_bytes + t->seq_off[i];
}

static oseq_entry *seq_tab_find(const orc_tree *t, const uint8_t *seq, uint64_t len, int insert,
                                uint64_t self_idx) {
    uint64_t h = hash_bytes(seq, len);
    uint64_t mask = t->seq_tab_size - 1;
    for (uint64_t p = h & mask;; p = (p + 1) & mask) {
        oseq_entry *e = &t->seq_tab[p];
        if (!e->used) {
            if (!insert) return NULL;
            e->used = 1;
            e->hash = h;
            e->first = self_idx;
            return e;
        }
        if (e->hash == h) {
            uint64_t l2;
            const uint8_t *s2 = tree_seq(t, e->first, &l2);
            if (l2 == len && memcmp(s2, seq, len) == 0) return e;
        }
    }
}

/* stable merge sort of indices by lineage string, bytewise (tree.rs:53-54) */
static int lineage_cmp(const char *a, const char *b) { return strcmp(a, b); }
static void merge_sort_idx(uint64_t *idx, uint64_t *tmp, uint64_t n, const char *const *lin) {
    if (n < 2) return;
    uint64_t h = n / 2;
    merge_sort_idx(idx, tmp, h, lin);
    merge_sort_idx(idx + h, tmp, n - h, lin);
    uint64_t i = 0, j = h, k = 0;
    while (i < h && j < n) {
        if (lineage_cmp(lin[idx[j]], lin[idx[i]]) < 0) tmp[k++] = idx[j++];
        else tmp[k++] = idx[i++];
    }
    while (i < h) tmp[k++] = idx[i++];
    while (j < n) tmp[k++] = idx[j++];
    memcpy(idx, tmp, n * sizeof(uint64_t));
}

/* tree.rs:46-140 */
orc_tree *orc_tree_new(uint64_t n, const char *const *lineages, const uint8_t *seq_bytes,
                       const uint64_t *seq_off) {
    if (n > 0xFFFFFFFFull) return NULL; /* check_lineage_size, tree.rs:24-31 */
    orc_tree *t = xcalloc(1, sizeof(orc_tree));
    t->n = n;
    t->root = node_new("root", 4, 0, ORC_NODE_INNER); /* tree.rs:49 */
    t->k_mer_map = xcalloc(ORC_NUM_KMERS, sizeof(olist)); /* tree.rs:52 */

    /* tree.rs:53-54: zip + stable sort by lineage */
    uint64_t *order = xmalloc(n * sizeof(uint64_t));
    uint64_t *tmp = xmalloc(n * sizeof(uint64_t));
    for (uint64_t i = 0; i < n; i++) order[i] = i;
    merge_sort_idx(order, tmp, n, lineages);
    free(tmp);
    t->orig_idx = order;
    t->lineages = xmalloc(n * sizeof(char *));
    t->seq_off = xmalloc((n + 1) * sizeof(uint64_t));
    t->seq_off[0] = 0;
    for (uint64_t i = 0; i < n; i++) {
        t->lineages[i] = xstrndup(lineages[order[i]], strlen(lineages[order[i]]));
        t->seq_off[i + 1] = t->seq_off[i] + (seq_off[order[i] + 1] - seq_off[order[i]]);
    }
    t->seq_bytes = xmalloc(t->seq_off[n]);
    for (uint64_t i = 0; i < n; i++)
        memcpy(t->seq_bytes + t->seq_off[i], seq_bytes + seq_off[order[i]],
               seq_off[order[i] + 1] - seq_off[order[i]]);

    /* tree.rs:50-51: sequence map (key -> ids in sorted order) */
    t->seq_tab_size = 16;
    while (t->seq_tab_size < 2 * n + 1) t->seq_tab_size <<= 1;
    t->seq_tab = xcalloc(t->seq_tab_size, sizeof(oseq_entry));

    uint64_t confidence_idx = 0; /* tree.rs:55 */
    for (uint64_t idx = 0; idx < n; idx++) {
        const char *lineage = t->lineages[idx];
        uint64_t slen;
        const uint8_t *sequence = tree_seq(t, idx, &slen);
        /* tree.rs:69-70: levels = lineage.split(',') */
        size_t n_levels = 1;
        for (const char *p = lineage; *p; p++) n_levels += (*p == ',');
        size_t last_level_idx = n_levels - 1;
        onode *current = t->root;
        const char *p = lineage;
        for (size_t level = 0; level < n_levels; level++) {
            const char *e = strchr(p, ',');
            size_t llen = e ? (size_t)(e - p) : strlen(p);
            int node_type = (level == last_level_idx) ? ORC_NODE_TAXON : ORC_NODE_INNER;
            /* tree.rs:78-98 */
            if (current->n_children > 0) {
                const char *name = current->children[current->n_children - 1]->label;
                if (!(strlen(name) == llen && memcmp(name, p, llen) == 0))
                    node_add_child(current, node_new(p, llen, confidence_idx, node_type));
                current->hi = confidence_idx + 1;
            } else {
                node_add_child(current, node_new(p, llen, confidence_idx, node_type));
                current->hi = confidence_idx + 1;
            }
            if (level == last_level_idx) confidence_idx += 1; /* tree.rs:97-99 */
            current = current->children[current->n_children - 1];
            p = e ? e + 1 : p + llen;
        }
        /* tree.rs:102-107 */
        node_add_child(current, node_new(current->label, strlen(current->label),
                                         confidence_idx - 1, ORC_NODE_SEQUENCE));
        current->hi = confidence_idx;

        /* tree.rs:109-112 */
        oseq_entry *e = seq_tab_find(t, sequence, slen, 1, idx);
        if (e->n == e->cap) {
            e->cap = e->cap ? e->cap * 2 : 1;
            e->ids = xrealloc(e->ids, e->cap * sizeof(uint32_t));
        }
        e->ids[e->n++] = (uint32_t)idx;

        /* tree.rs:114-123 (+ unique/sorted of :134-137 applied on the fly: ids arrive ascending) */
        for (uint64_t w = 0; w + 8 <= slen; w++) {
            uint32_t k = 0;
            int ok = 1;
            for (int j = 0; j < 8; j++) {
                int c = orc_map_four_to_two_bit_repr(sequence[w + j]);
                if (c < 0) { ok = 0; break; }
                k |= (uint32_t)c << (14 - 2 * j);
            }
            if (!ok) continue;
            olist *l = &t->k_mer_map[k];
            if (l->n && l->ids[l->n - 1] == (uint32_t)idx) continue;
            if (l->n == l->cap) {
                l->cap = l->cap ? l->cap * 2 : 4;
                l->ids = xrealloc(l->ids, l->cap * sizeof(uint32_t));
            }
            l->ids[l->n++] = (uint32_t)idx;
        }
    }
    t->root->hi = confidence_idx; /* tree.rs:127 */
    t->num_tips = confidence_idx; /* tree.rs:138 */
    return t;
}

void orc_tree_free(orc_tree *t) {
    if (!t) return;
    node_free(t->root);
    for (uint64_t i = 0; i < t->n; i++) free(t->lineages[i]);
    free(t->lineages);
    free(t->orig_idx);
    free(t->seq_bytes);
    free(t->seq_off);
    for (uint32_t k = 0; k < ORC_NUM_KMERS; k++) free(t->k_mer_map[k].ids);
    free(t->k_mer_map);
    for (uint64_t i = 0; i < t->seq_tab_size; i++) free(t->seq_tab[i].ids);
    free(t->seq_tab);
    free(t->flat);
    free(t->flat_parent);
    free(t);
}

uint64_t orc_tree_num_tips(const orc_tree *t) { return t->num_tips; }
const char *orc_tree_lineage(const orc_tree *t, uint64_t i) { return t->lineages[i]; }
uint64_t orc_tree_original_index(const orc_tree *t, uint64_t i) { return t->orig_idx[i]; }
uint64_t orc_tree_kmer_list(const orc_tree *t, uint32_t kmer, const uint32_t **ids) {
    *ids = t->k_mer_map[kmer].ids;
    return t->k_mer_map[kmer].n;
}
uint64_t orc_tree_total_postings(const orc_tree *t) {
    uint64_t s = 0;
    for (uint32_t k = 0; k < ORC_NUM_KMERS; k++) s += t->k_mer_map[k].n;
    return s;
}
void orc_tree_export_csr(const orc_tree *t, uint64_t *offsets, uint32_t *postings) {
    uint64_t s = 0;
    for (uint32_t k = 0; k < ORC_NUM_KMERS; k++) {
        offsets[k] = s;
        if (postings && t->k_mer_map[k].n)
            memcpy(postings + s, t->k_mer_map[k].ids, t->k_mer_map[k].n * sizeof(uint32_t));
        s += t->k_mer_map[k].n;
    }
    offsets[ORC_NUM_KMERS] = s;
}
uint64_t orc_tree_exact_matches(const orc_tree *t, const uint8_t *seq, uint64_t len,
                                const uint32_t **ids) {
    oseq_entry *e = seq_tab_find(t, seq, len, 0, 0);
    if (!e) { *ids = NULL; return 0; }
    *ids = e->ids;
    return e->n;
}

static void flatten_rec(orc_tree *t, onode *n, int64_t parent, uint64_t *pos) {
    uint64_t me = (*pos)++;
    if (t->flat) { t->flat[me] = n; t->flat_parent[me] = parent; }
    for (uint32_t i = 0; i < n->n_children; i++) flatten_rec(t, n->children[i], (int64_t)me, pos);
}
static void ensure_flat(orc_tree *t) {
    if (t->flat) return;
    uint64_t cnt = 0;
    flatten_rec(t, t->root, -1, &cnt);
    t->n_nodes = cnt;
    t->flat = xmalloc(cnt * sizeof(onode *));
    t->flat_parent = xmalloc(cnt * sizeof(int64_t));
    cnt = 0;
    flatten_rec(t, t->root, -1, &cnt);
}
uint64_t orc_tree_num_nodes(const orc_tree *t) {
    ensure_flat((orc_tree *)t);
    return t->n_nodes;
}
void orc_tree_export_nodes(const orc_tree *t, uint64_t *lo, uint64_t *hi, int64_t *parent,
                           uint8_t *type, uint32_t *n_children) {
    ensure_flat((orc_tree *)t);
    for (uint64_t i = 0; i < t->n_nodes; i++) {
        lo[i] = t->flat[i]->lo;
        hi[i] = t->flat[i]->hi;
        parent[i] = t->flat_parent[i];
        type[i] = (uint8_t)t->flat[i]->type;
        n_children[i] = t->flat[i]->n_children;
    }
}
const char *orc_tree_node_label(const orc_tree *t, uint64_t node) {
    ensure_flat((orc_tree *)t);
    return t->flat[node]->label;
}

/* ------------------------------------------------------------------------ */
/* parser.rs                                                                 */
/* ------------------------------------------------------------------------ */

/* parser.rs:11-34 */
int orc_map_dna_char(int ch) {
    const int a = 1, c = 2, g = 4, t = 8;
    switch (toupper(ch)) {
    case 'A': return a;
    case 'C': return c;
    case 'G': return g;
    case 'T': return t;
    case 'W': return a | t;
    case 'S': return c | g;
    case 'M': return a | c;
    case 'K': return g | t;
    case 'R': return a | g;
    case 'Y': return c | t;
    case 'B': return c | g | t;
    case 'D': return a | g | t;
    case 'H': return a | c | t;
    case 'V': return a | c | g;
    case 'N': return a | c | g | t;
    default: return -1; /* panic!("Unexpected character") */
    }
}

typedef struct {
    const char **line; /* pointers into a private copy */
    size_t *len;
    size_t n;
    char *copy;
} olines;

/* parser.rs:53-57 / 124-128: lines(), trim(), drop empty and ';' lines.
 * (ASCII whitespace only; the reference's str::trim also strips Unicode spaces.) */
static olines split_lines(const char *s) {
    olines L = {0};
    size_t slen = strlen(s);
    L.copy = xstrndup(s, slen);
    size_t cap = 64;
    L.line = xmalloc(cap * sizeof(char *));
    L.len = xmalloc(cap * sizeof(size_t));
    size_t i = 0;
    while (i < slen) {
        size_t j = i;
        while (j < slen && L.copy[j] != '\n') j++;
        size_t b = i, e = j;
        while (b < e && isspace((unsigned char)L.copy[b])) b++;
        while (e > b && isspace((unsigned char)L.copy[e - 1])) e--;
        if (e > b && L.copy[b] != ';') {
            if (L.n == cap) {
                cap *= 2;
                L.line = xrealloc(L.line, cap * sizeof(char *));
                L.len = xrealloc(L.len, cap * sizeof(size_t));
            }
            L.line[L.n] = L.copy + b;
            L.len[L.n] = e - b;
            L.n++;
        }
        i = j + 1;
    }
    return L;
}
static void free_lines(olines *L) {
    free(L->line);
    free(L->len);
    free(L->copy);
}

typedef struct {
    uint8_t *p;
    uint64_t n, cap;
} obuf;
static void obuf_push(obuf *b, uint8_t v) {
    if (b->n == b->cap) {
        b->cap = b->cap ? b->cap * 2 : 256;
        b->p = xrealloc(b->p, b->cap);
    }
    b->p[b->n++] = v;
}

/* `tax=([^;]+);` first match (parser.rs:50,70-78) */
static int find_tax(const char *s, size_t n, size_t *b, size_t *e) {
    for (size_t i = 0; i + 4 <= n; i++) {
        if (memcmp(s + i, "tax=", 4) != 0) continue;
        size_t j = i + 4;
        while (j < n && s[j] != ';') j++;
        if (j < n && j > i + 4) { *b = i + 4; *e = j; return 1; }
    }
    return 0;
}

/* parser.rs:46-105.  err: 1 empty, 2 not FASTA, 3 bad annotation, 4 count mismatch, 5 bad char */
orc_tree *orc_parse_reference_fasta_str(const char *s, int *err) {
    *err = 0;
    if (!s || !*s) { *err = 1; return NULL; }
    olines L = split_lines(s);
    if (L.n == 0 || L.line[0][0] != '>') { *err = 2; free_lines(&L); return NULL; }
    char **labels = NULL;
    size_t n_labels = 0, cap_labels = 0;
    obuf bytes = {0}, cur = {0};
    uint64_t *off = xmalloc(sizeof(uint64_t));
    size_t n_seqs = 0, cap_off = 1;
    off[0] = 0;
#define PUSH_SEQ()                                                       \
    do {                                                                 \
        for (uint64_t q_ = 0; q_ < cur.n; q_++) obuf_push(&bytes, cur.p[q_]); \
        if (n_seqs + 2 > cap_off) {                                      \
            cap_off = cap_off * 2 + 2;                                   \
            off = xrealloc(off, cap_off * sizeof(uint64_t));             \
        }                                                                \
        off[++n_seqs] = bytes.n;                                         \
        cur.n = 0;                                                       \
    } while (0)
    for (size_t li = 0; li < L.n && !*err; li++) {
        const char *line = L.line[li];
        size_t len = L.len[li];
        if (line[0] == '>') {
            size_t b, e;
            if (!find_tax(line + 1, len - 1, &b, &e)) { *err = 3; break; }
            if (n_labels == cap_labels) {
                cap_labels = cap_labels ? cap_labels * 2 : 64;
                labels = xrealloc(labels, cap_labels * sizeof(char *));
            }
            labels[n_labels++] = xstrndup(line + 1 + b, e - b);
            if (cur.n) PUSH_SEQ(); /* parser.rs:80-83 */
        } else {
            for (size_t i = 0; i < len; i++) {
                int c = orc_map_dna_char((unsigned char)line[i]);
                if (c < 0) { *err = 5; break; }
                obuf_push(&cur, (uint8_t)c);
            }
        }
    }
    orc_tree *t = NULL;
    if (!*err) {
        PUSH_SEQ(); /* parser.rs:98 */
        if (n_labels != n_seqs) *err = 4;
        else t = orc_tree_new(n_labels, (const char *const *)labels, bytes.p, off);
    }
#undef PUSH_SEQ
    for (size_t i = 0; i < n_labels; i++) free(labels[i]);
    free(labels);
    free(bytes.p);
    free(cur.p);
    free(off);
    free_lines(&L);
    return t;
}

struct orc_queries {
    uint64_t n;
    char **labels;
    uint8_t **seqs;
    uint64_t *lens;
};

/* parser.rs:117-154 */
orc_queries *orc_parse_query_fasta_str(const char *s, const char *const *skip, uint64_t n_skip,
                                       int *err) {
    *err = 0;
    if (!s || !*s) { *err = 1; return NULL; }
    olines L = split_lines(s);
    if (L.n == 0 || L.line[0][0] != '>') { *err = 2; free_lines(&L); return NULL; }
    orc_queries *q = xcalloc(1, sizeof(orc_queries));
    uint64_t cap = 0;
    char *cur_label = xstrndup("", 0);
    obuf cur = {0};
#define PUSH_Q()                                                        \
    do {                                                                \
        if (q->n == cap) {                                              \
            cap = cap ? cap * 2 : 64;                                   \
            q->labels = xrealloc(q->labels, cap * sizeof(char *));      \
            q->seqs = xrealloc(q->seqs, cap * sizeof(uint8_t *));       \
            q->lens = xrealloc(q->lens, cap * sizeof(uint64_t));        \
        }                                                               \
        q->labels[q->n] = xstrndup(cur_label, strlen(cur_label));       \
        q->seqs[q->n] = xmalloc(cur.n);                                 \
        memcpy(q->seqs[q->n], cur.p, cur.n);                            \
        q->lens[q->n] = cur.n;                                          \
        q->n++;                                                         \
    } while (0)
    for (size_t li = 0; li < L.n && !*err; li++) {
        const char *line = L.line[li];
        size_t len = L.len[li];
        if (line[0] == '>') {
            if (cur.n) { PUSH_Q(); cur.n = 0; } /* parser.rs:138-141 */
            free(cur_label);
            cur_label = xstrndup(line + 1, len - 1);
        } else {
            for (size_t i = 0; i < len; i++) {
                int c = orc_map_dna_char((unsigned char)line[i]);
                if (c < 0) { *err = 5; break; }
                obuf_push(&cur, (uint8_t)c);
            }
        }
    }
    if (!*err) PUSH_Q(); /* parser.rs:149 */
#undef PUSH_Q
    free(cur_label);
    free(cur.p);
    free_lines(&L);
    if (*err) { orc_queries_free(q); return NULL; }
    /* parser.rs:150-153: drop already-processed labels */
    if (n_skip) {
        uint64_t w = 0;
        for (uint64_t i = 0; i < q->n; i++) {
            int drop = 0;
            for (uint64_t k = 0; k < n_skip && !drop; k++) drop = strcmp(q->labels[i], skip[k]) == 0;
            if (drop) { free(q->labels[i]); free(q->seqs[i]); continue; }
            q->labels[w] = q->labels[i];
            q->seqs[w] = q->seqs[i];
            q->lens[w] = q->lens[i];
            w++;
        }
        q->n = w;
    }
    return q;
}
uint64_t orc_queries_len(const orc_queries *q) { return q->n; }
const char *orc_queries_label(const orc_queries *q, uint64_t i) { return q->labels[i]; }
uint64_t orc_queries_seq(const orc_queries *q, uint64_t i, const uint8_t **seq) {
    *seq = q->seqs[i];
    return q->lens[i];
}
void orc_queries_free(orc_queries *q) {
    if (!q) return;
    for (uint64_t i = 0; i < q->n; i++) { free(q->labels[i]); free(q->seqs[i]); }
    free(q->labels);
    free(q->seqs);
    free(q->lens);
    free(q);
}

/* ------------------------------------------------------------------------ */
/* raxtax.rs:41,55-68                                                        */
/* ------------------------------------------------------------------------ */
static uint32_t hit_counts_buf(const orc_tree *t, const uint8_t *seq, uint64_t len, int skip_exact,
                               uint16_t *counts, uint16_t *kbuf) {
    memset(counts, 0, t->num_tips * sizeof(uint16_t)); /* :41 */
    uint32_t nk = orc_sequence_to_kmers(seq, len, kbuf); /* :55 */
    for (uint32_t i = 0; i < nk; i++) {                  /* :58-64 */
        const olist *l = &t->k_mer_map[kbuf[i]];
        for (uint64_t j = 0; j < l->n; j++) counts[l->ids[j]] += 1;
    }
    if (skip_exact) { /* :65-68 */
        const uint32_t *ids;
        uint64_t ne = orc_tree_exact_matches(t, seq, len, &ids);
        for (uint64_t j = 0; j < ne; j++) counts[ids[j]] = 0;
    }
    return nk;
}

uint32_t orc_hit_counts(const orc_tree *t, const uint8_t *seq, uint64_t len, int skip_exact,
                        uint16_t *counts) {
    uint16_t *kbuf = xmalloc((len + 1) * sizeof(uint16_t));
    uint32_t nk = hit_counts_buf(t, seq, len, skip_exact, counts, kbuf);
    free(kbuf);
    return nk;
}

/* ------------------------------------------------------------------------ */
/* prob.rs                                                                   */
/* ------------------------------------------------------------------------ */

/* prob.rs:105-119 */
static double only_last_pmf(uint64_t t, uint64_t n, uint64_t m, double ln_total) {
    if (m == t) return 1.0;
    if (m == 0) return 0.0;
    double num_possible_matches = orc_ln_binomial(m + n - 1, n);
    return exp(num_possible_matches - ln_total);
}

/* prob.rs:121-170 for one intersection size m; out has n+1 entries. Requires n >= 1
 * in the general arm (the reference's zip_eq panics at n == 0). */
void orc_iterative_pmf_ln(uint64_t t, uint64_t n, uint64_t m, double ln_total, double *out) {
    if (m == t) { /* :130-133 */
        for (uint64_t i = 0; i <= n; i++) out[i] = -INFINITY;
        out[n] = 0.0;
        return;
    }
    if (m == 0) { /* :134-137 */
        for (uint64_t i = 0; i <= n; i++) out[i] = -INFINITY;
        out[0] = 0.0;
        return;
    }
    double impossible_init = orc_ln_binomial(t - m + n - 1, n); /* :143-146 */
    out[0] = impossible_init - ln_total;                        /* :158 */
    double possible = 0.0, impossible = impossible_init;
    for (uint64_t i = 1; i <= n; i++) {
        possible += log((double)(m + i - 1) / (double)i); /* :139-142 */
        double imp;
        if (i < n) { /* :147-155 */
            impossible -= log((double)(t - m + n - i) / (double)(n - i + 1));
            imp = impossible;
        } else {
            imp = 0.0; /* .chain([0.0]) */
        }
        out[i] = possible + imp - ln_total; /* :160-163 */
    }
}

/* Scratch reused from query to query by the batch driver.  The reference allocates per query as well
 * (prob.rs:13-19 HashMap, :43-61 one Vec per distinct count, lineage.rs:62-66 the prefix Vec); a port that
 * mmap()s and page-faults 1-2 MB per query would understate the CPU baseline, so the buffers are kept --
 * the arithmetic and its order are untouched. */
typedef struct {
    uint64_t *hist;  /* 65536 entries, all zero between queries */
    uint32_t *ms;    /* 65536 */
    double *pmf, *cmf, *prod, *table, *prefix;
    size_t pmf_cap, cmf_cap, prod_cap, table_cap, prefix_cap;
} orc_ws;

static void ws_init(orc_ws *w) {
    memset(w, 0, sizeof *w);
    w->hist = xcalloc(65536, sizeof(uint64_t));
    w->ms = xmalloc(65536 * sizeof(uint32_t));
}
static void ws_free(orc_ws *w) {
    free(w->hist); free(w->ms); free(w->pmf); free(w->cmf); free(w->prod); free(w->table); free(w->prefix);
    memset(w, 0, sizeof *w);
}
static double *ws_grow(double **p, size_t *cap, size_t want) {
    if (want > *cap) {
        free(*p);
        *cap = want + want / 4 + 16;
        *p = xmalloc(*cap * sizeof(double));
    }
    return *p;
}

static int prob_table_ws(orc_ws *ws, uint16_t total_num_k_mers, uint64_t num_trials, const uint16_t *sizes,
                         uint64_t n_refs, double *table, double *z) {
    const uint64_t t = total_num_k_mers, n = num_trials;
    if (t == 0) return -1; /* t + n - 1 underflows u64, prob.rs:21 */
    /* prob.rs:13-19 histogram (dense array instead of HashMap) */
    uint64_t *hist = ws->hist;
    uint32_t max_size = 0;
    for (uint64_t r = 0; r < n_refs; r++) {
        hist[sizes[r]] += 1;
        if (sizes[r] > max_size) max_size = sizes[r];
    }
    uint32_t *ms = ws->ms;
    uint32_t D = 0;
    int any_full = 0;
    for (uint32_t m = 0; m <= max_size; m++)
        if (hist[m]) { ms[D++] = m; if (m == t) any_full = 1; }
    for (uint64_t m = 0; m <= t; m++) table[m] = 0.0;
    double ln_total = orc_ln_binomial(t + n - 1, n); /* :20-23 */
    int rc = 0;
    if (any_full) { /* :24-41 */
        for (uint32_t d = 0; d < D; d++)
            if (ms[d] <= t) table[ms[d]] = only_last_pmf(t, n, ms[d], ln_total);
    } else {
        if (n == 0) { rc = -2; goto done; } /* zip_eq length mismatch, :162 */
        const uint64_t W = n + 1;
        double *pmf = ws_grow(&ws->pmf, &ws->pmf_cap, (size_t)D * W);
        double *cmf = ws_grow(&ws->cmf, &ws->cmf_cap, (size_t)D * W);
        double *prod = ws_grow(&ws->prod, &ws->prod_cap, W);
        for (uint32_t d = 0; d < D; d++) { /* :43-48 */
            /* counts > t cannot occur (count <= |K(q)| = t) */
            orc_iterative_pmf_ln(t, n, ms[d], ln_total, pmf + (size_t)d * W);
            double sum = 0.0; /* :49-61 */
            for (uint64_t i = 0; i < W; i++) {
                double p = pmf[(size_t)d * W + i];
                if (p != -INFINITY) sum += exp(p);
                cmf[(size_t)d * W + i] = log(sum);
            }
        }
        for (uint64_t i = 0; i < W; i++) { /* :62-73 */
            double s = 0.0;
            for (uint32_t d = 0; d < D; d++) s += (double)hist[ms[d]] * cmf[(size_t)d * W + i];
            prod[i] = s;
        }
        for (uint32_t d = 0; d < D; d++) { /* :74-90 */
            double s = 0.0;
            for (uint64_t i = 0; i < W; i++) {
                double p = pmf[(size_t)d * W + i], c = cmf[(size_t)d * W + i];
                if (c == -INFINITY || prod[i] == -INFINITY) s += 0.0;
                else s += exp(p + prod[i] - c);
            }
            table[ms[d]] = s;
        }
    }
    if (z) { /* :97: probs_sum over references, in reference order */
        double s = 0.0;
        for (uint64_t r = 0; r < n_refs; r++) s += table[sizes[r]];
        *z = s;
    }
done:
    for (uint32_t d = 0; d < D; d++) hist[ms[d]] = 0; /* the workspace invariant: all zero between queries */
    return rc;
}

int orc_prob_table(uint16_t total_num_k_mers, uint64_t num_trials, const uint16_t *sizes,
                   uint64_t n_refs, double *table, double *z) {
    orc_ws ws;
    ws_init(&ws);
    int rc = prob_table_ws(&ws, total_num_k_mers, num_trials, sizes, n_refs, table, z);
    ws_free(&ws);
    return rc;
}

/* prob.rs:8-103 */
static int highest_hit_prob_ws(orc_ws *ws, uint16_t total_num_k_mers, uint64_t num_trials,
                               const uint16_t *intersection_sizes, uint64_t n_refs, double *out) {
    double *table = ws_grow(&ws->table, &ws->table_cap, (size_t)total_num_k_mers + 1);
    double z = 0.0;
    int rc = prob_table_ws(ws, total_num_k_mers, num_trials, intersection_sizes, n_refs, table, &z);
    if (rc == 0) {
        if (!(z > 0.0)) rc = -3; /* assert!(probs_sum > 0.0), :98 */
        else
            for (uint64_t r = 0; r < n_refs; r++) out[r] = table[intersection_sizes[r]] / z; /* :92-102 */
    }
    return rc;
}

int orc_highest_hit_prob_per_reference(uint16_t total_num_k_mers, uint64_t num_trials,
                                       const uint16_t *intersection_sizes, uint64_t n_refs,
                                       double *out) {
    orc_ws ws;
    ws_init(&ws);
    int rc = highest_hit_prob_ws(&ws, total_num_k_mers, num_trials, intersection_sizes, n_refs, out);
    ws_free(&ws);
    return rc;
}

/* ------------------------------------------------------------------------ */
/* lineage.rs                                                                */
/* ------------------------------------------------------------------------ */
typedef struct {
    const orc_tree *tree;
    const double *prefix; /* N+1 */
    double rounding_factor;
    orc_row *rows;
    int n_rows, cap;
    int overflow;
} olin;

static double get_confidence(const olin *L, const onode *n) { /* :114-117 */
    return L->prefix[n->hi] - L->prefix[n->lo];
}

static void push_row(olin *L, uint64_t idx, const double *cp, const double *ep, uint32_t depth) {
    if (L->n_rows >= L->cap || depth > ORC_MAXD) { L->overflow = 1; L->n_rows++; return; }
    orc_row *r = &L->rows[L->n_rows++];
    r->idx = idx;
    r->depth = depth;
    memcpy(r->conf, cp, depth * sizeof(double));
    memcpy(r->expd, ep, depth * sizeof(double));
}

/* lineage.rs:119-179.  conf/expd prefixes live in caller-owned arrays of ORC_MAXD+1. */
static int eval_recurse(olin *L, const onode *node, double *cp, double *ep, uint32_t depth) {
    int no_child_significant = 1, pushed_result = 0;
    const double N = (double)L->tree->num_tips;
    for (uint32_t ci = 0; ci < node->n_children; ci++) {
        const onode *c = node->children[ci];
        double child_conf = round(get_confidence(L, c) * L->rounding_factor) / L->rounding_factor;
        if (child_conf == 0.0) continue;
        no_child_significant = 0;
        if (depth >= ORC_MAXD) { L->overflow = 1; continue; }
        cp[depth] = child_conf;
        ep[depth] = (double)(c->hi - c->lo) / N;
        int child_pushed = eval_recurse(L, c, cp, ep, depth + 1);
        if (!child_pushed && c->type == ORC_NODE_TAXON) {
            push_row(L, c->lo, cp, ep, depth + 1);
            pushed_result = 1;
        }
        pushed_result |= child_pushed;
    }
    if (no_child_significant && node->type == ORC_NODE_INNER) { /* :151-177 */
        const onode *cur = node;
        uint32_t d = depth;
        while (cur->type == ORC_NODE_INNER) {
            /* Iterator::max_by keeps the LAST maximum */
            const onode *best = cur->children[0];
            double best_v = get_confidence(L, best);
            for (uint32_t ci = 1; ci < cur->n_children; ci++) {
                double v = get_confidence(L, cur->children[ci]);
                if (!(v < best_v)) { best = cur->children[ci]; best_v = v; }
            }
            cur = best;
            if (d >= ORC_MAXD) { L->overflow = 1; break; }
            cp[d] = 1.0 / L->rounding_factor;
            ep[d] = (double)(cur->hi - cur->lo) / N;
            d++;
        }
        push_row(L, cur->lo, cp, ep, d);
        pushed_result = 1;
    }
    return pushed_result;
}

/* iterator partial_cmp on f64 slices: lexicographic, shorter-prefix = Less */
static int conf_vec_cmp(const orc_row *a, const orc_row *b) {
    uint32_t n = a->depth < b->depth ? a->depth : b->depth;
    for (uint32_t i = 0; i < n; i++) {
        if (a->conf[i] < b->conf[i]) return -1;
        if (a->conf[i] > b->conf[i]) return 1;
    }
    return (a->depth > b->depth) - (a->depth < b->depth);
}

/* lineage.rs:61-112 */
static int lineage_evaluate_buf(const orc_tree *t, const double *probs, orc_row *rows, int cap, double *prefix);
int orc_lineage_evaluate(const orc_tree *t, const double *probs, orc_row *rows, int cap) {
    double *prefix = xmalloc((t->num_tips + 1) * sizeof(double));
    int n = lineage_evaluate_buf(t, probs, rows, cap, prefix);
    free(prefix);
    return n;
}
static int lineage_evaluate_buf(const orc_tree *t, const double *probs, orc_row *rows, int cap, double *prefix) {
    const uint64_t N = t->num_tips; /* :62-66 */
    prefix[0] = 0.0;
    {
        double s = 0.0;
        for (uint64_t i = 0; i < N; i++) { s += probs[i]; prefix[i + 1] = s; }
    }
    olin L = {t, prefix, 100.0 /* 10^F64_OUTPUT_ACCURACY */, rows, 0, cap, 0};
    double cp[ORC_MAXD + 1], ep[ORC_MAXD + 1];
    eval_recurse(&L, t->root, cp, ep, 0); /* :81 */
    if (L.overflow) return -(L.n_rows) - 1;
    /* :86-90 global signal */
    double gs = 0.0;
    {
        const double inv = 1.0 / (double)N;
        for (uint64_t i = 0; i < N; i++) { double d = probs[i] - inv; gs += d * d; }
        gs = sqrt(gs);
    }
    /* :91-93 stable sort, descending by confidence vector (insertion sort = stable) */
    for (int i = 1; i < L.n_rows; i++) {
        orc_row key = rows[i];
        int j = i - 1;
        while (j >= 0 && conf_vec_cmp(&rows[j], &key) < 0) { rows[j + 1] = rows[j]; j--; }
        rows[j + 1] = key;
    }
    /* :94-110 local signal */
    for (int i = 0; i < L.n_rows; i++) {
        orc_row *r = &rows[i];
        uint32_t s = r->depth - 1;
        for (uint32_t k = 0; k < r->depth; k++)
            if (1.0 > r->expd[k]) { s = k; break; }
        r->local_signal = orc_euclidean_distance_l1(r->conf + s, r->expd + s, r->depth - s);
        r->global_signal = gs;
    }
    return L.n_rows;
}

/* ------------------------------------------------------------------------ */
/* raxtax.rs:39-88 for one query                                             */
/* ------------------------------------------------------------------------ */
static int classify_buf(const orc_tree *t, const uint8_t *seq, uint64_t len, int skip_exact,
                        int raw_confidence, orc_row *rows, int cap, uint16_t *counts,
                        uint16_t *kbuf, double *probs, orc_ws *ws) {
    uint32_t nk = hit_counts_buf(t, seq, len, skip_exact, counts, kbuf);
    if (nk > 65535) return -10; /* assert, :56 */
    uint64_t num_trials = nk / 2; /* :57 */
    int rc = highest_hit_prob_ws(ws, (uint16_t)nk, num_trials, counts, t->num_tips, probs);
    if (rc < 0) return rc;
    int n = lineage_evaluate_buf(t, probs, rows, cap, ws_grow(&ws->prefix, &ws->prefix_cap, t->num_tips + 1)); /* :71 */
    if (n < 0) return -20;
    if (n == 0) return -11; /* assert!(!eval_res.is_empty()), :72 */
    if (!raw_confidence && !skip_exact) { /* :73-84 */
        const uint32_t *ids;
        uint64_t ne = orc_tree_exact_matches(t, seq, len, &ids);
        if (ne == 1) {
            const char *lin = t->lineages[ids[0]];
            uint32_t depth = 1;
            for (const char *p = lin; *p; p++) depth += (*p == ',');
            if (depth > ORC_MAXD) return -20;
            rows[0].idx = ids[0];
            rows[0].depth = depth;
            for (uint32_t k = 0; k < depth; k++) { rows[0].conf[k] = 1.0; rows[0].expd[k] = 0.0; }
            /* local/global signal stay those of the former row 0 */
            n = 1;
        }
    }
    return n;
}

int orc_classify(const orc_tree *t, const uint8_t *seq, uint64_t len, int skip_exact,
                 int raw_confidence, orc_row *rows, int cap) {
    uint16_t *counts = xmalloc(t->num_tips * sizeof(uint16_t));
    uint16_t *kbuf = xmalloc((len + 1) * sizeof(uint16_t));
    double *probs = xmalloc(t->num_tips * sizeof(double));
    orc_ws ws;
    ws_init(&ws);
    int n = classify_buf(t, seq, len, skip_exact, raw_confidence, rows, cap, counts, kbuf, probs, &ws);
    ws_free(&ws);
    free(counts);
    free(kbuf);
    free(probs);
    return n;
}

/* lineage.rs:17-29, utils.rs:62-68 */
int64_t orc_format_out(const orc_tree *t, const char *label, const orc_row *rows, int n, char *buf,
                       uint64_t cap) {
    uint64_t w = 0;
    for (int i = 0; i < n; i++) {
        int k = snprintf(buf + w, cap - w, "%s%s\t%s\t", i ? "\n" : "", label, t->lineages[rows[i].idx]);
        if (k < 0 || (uint64_t)k >= cap - w) return -1;
        w += (uint64_t)k;
        for (uint32_t d = 0; d < rows[i].depth; d++) {
            k = snprintf(buf + w, cap - w, "%s%.2f", d ? "," : "", rows[i].conf[d]);
            if (k < 0 || (uint64_t)k >= cap - w) return -1;
            w += (uint64_t)k;
        }
        k = snprintf(buf + w, cap - w, "\t%.5f\t%.5f", rows[i].local_signal, rows[i].global_signal);
        if (k < 0 || (uint64_t)k >= cap - w) return -1;
        w += (uint64_t)k;
    }
    return (int64_t)w;
}

/* lineage.rs:31-48, utils.rs:83-89: lineage levels interleaved with confidences.
 * itertools::interleave alternates and then drains the longer side. */
int64_t orc_format_tsv(const orc_tree *t, const char *label, const orc_row *rows, int n,
                       const uint8_t *seq, uint64_t len, char *buf, uint64_t cap) {
    char *dec = xmalloc(len + 1);
    orc_decompress_sequence(seq, len, dec);
    uint64_t w = 0;
    int64_t ret = -1;
    for (int i = 0; i < n; i++) {
        int k = snprintf(buf + w, cap - w, "%s%s\t", i ? "\n" : "", label);
        if (k < 0 || (uint64_t)k >= cap - w) goto out;
        w += (uint64_t)k;
        const char *lin = t->lineages[rows[i].idx];
        const char *p = lin;
        uint32_t d = 0;
        int first = 1, lin_done = 0;
        while (!lin_done || d < rows[i].depth) {
            if (!lin_done) {
                const char *e = strchr(p, ',');
                size_t l = e ? (size_t)(e - p) : strlen(p);
                k = snprintf(buf + w, cap - w, "%s%.*s", first ? "" : "\t", (int)l, p);
                if (k < 0 || (uint64_t)k >= cap - w) goto out;
                w += (uint64_t)k;
                first = 0;
                if (e) p = e + 1; else lin_done = 1;
            }
            if (d < rows[i].depth) {
                k = snprintf(buf + w, cap - w, "%s%.2f", first ? "" : "\t", rows[i].conf[d]);
                if (k < 0 || (uint64_t)k >= cap - w) goto out;
                w += (uint64_t)k;
                first = 0;
                d++;
            }
        }
        k = snprintf(buf + w, cap - w, "\t%.5f\t%.5f\t%s", rows[i].local_signal,
                     rows[i].global_signal, dec);
        if (k < 0 || (uint64_t)k >= cap - w) goto out;
        w += (uint64_t)k;
    }
    ret = (int64_t)w;
out:
    free(dec);
    return ret;
}

/* utils.rs:160-197 get_thread_ids: one logical CPU per (core_id, physical_package_id) pair, taken in sorted
 * order of (core, socket, cpu), over the CPUs this process may run on.  Returns the number of ids written. */
int orc_physical_core_ids(int *ids, int cap) {
    cpu_set_t mask;
    CPU_ZERO(&mask);
    if (sched_getaffinity(0, sizeof mask, &mask) != 0) return 0;
    typedef struct { long core, socket; int cpu; } ent;
    ent *all = xmalloc(CPU_SETSIZE * sizeof(ent));
    int n = 0;
    for (int cpu = 0; cpu < CPU_SETSIZE; cpu++) {
        if (!CPU_ISSET(cpu, &mask)) continue;
        char path[128];
        long core = -1, socket = -1;
        snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/core_id", cpu);
        FILE *f = fopen(path, "r");
        if (f) { if (fscanf(f, "%ld", &core) != 1) core = -1; fclose(f); }
        snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/physical_package_id", cpu);
        f = fopen(path, "r");
        if (f) { if (fscanf(f, "%ld", &socket) != 1) socket = -1; fclose(f); }
        if (core < 0 || socket < 0) { core = cpu; socket = 0; } /* no topology files: every CPU counts as a core */
        all[n].core = core; all[n].socket = socket; all[n].cpu = cpu;
        n++;
    }
    for (int i = 1; i < n; i++) { /* sorted() on (core, socket, cpu) */
        ent k = all[i];
        int j = i - 1;
        while (j >= 0 && (all[j].core > k.core || (all[j].core == k.core && (all[j].socket > k.socket ||
               (all[j].socket == k.socket && all[j].cpu > k.cpu))))) { all[j + 1] = all[j]; j--; }
        all[j + 1] = k;
    }
    int out = 0;
    for (int i = 0; i < n; i++) {
        if (i && all[i].core == all[i - 1].core && all[i].socket == all[i - 1].socket) continue; /* used_physical */
        if (out < cap) ids[out] = all[i].cpu;
        out++;
    }
    free(all);
    return out < cap ? out : cap;
}

/* raxtax.rs:35-88 over a batch: par_chunks(chunk_size) with one u16[N] buffer per chunk.
 * pin_cpus != NULL: thread i of the pool runs on pin_cpus[i] (utils.rs:139-158, --pin); threads is then
 * clamped to n_pin. */
int64_t orc_classify_batch_ex(const orc_tree *t, uint64_t n_q, const uint8_t *bases,
                              const uint64_t *base_off, int skip_exact, int raw_confidence,
                              int threads, orc_row *rows_out, int cap, int32_t *n_rows_out,
                              int format_strings, const int *pin_cpus, int n_pin) {
    if (threads < 1) threads = 1;
    if (pin_cpus && n_pin > 0 && threads > n_pin) threads = n_pin; /* utils.rs:150 */
    /* main.rs:119-124 */
    uint64_t chunk = threads == 1 ? n_q : (n_q / ((uint64_t)threads * 10) + 1);
    if (threads != 1 && chunk < 100) chunk = 100;
    if (chunk == 0) chunk = 1;
    uint64_t n_chunks = (n_q + chunk - 1) / chunk;
    uint64_t max_len = 0;
    for (uint64_t q = 0; q < n_q; q++)
        if (base_off[q + 1] - base_off[q] > max_len) max_len = base_off[q + 1] - base_off[q];
    int64_t bad = 0;
#pragma omp parallel num_threads(threads) reduction(+ : bad)
    {
        cpu_set_t old_mask;
        int pinned = 0;
        if (pin_cpus && n_pin > 0) {
            extern int omp_get_thread_num(void);
            const int me = omp_get_thread_num();
            if (me < n_pin && sched_getaffinity(0, sizeof old_mask, &old_mask) == 0) {
                cpu_set_t m;
                CPU_ZERO(&m);
                CPU_SET(pin_cpus[me], &m);
                pinned = sched_setaffinity(0, sizeof m, &m) == 0;
            }
        }
        orc_ws ws;
        ws_init(&ws);
#pragma omp for schedule(dynamic, 1)
        for (uint64_t c = 0; c < n_chunks; c++) {
            uint16_t *counts = xmalloc(t->num_tips * sizeof(uint16_t)); /* :38 */
            uint16_t *kbuf = xmalloc((max_len + 1) * sizeof(uint16_t));
            double *probs = xmalloc(t->num_tips * sizeof(double));
            orc_row *local = rows_out ? NULL : xmalloc((size_t)cap * sizeof(orc_row));
            char *sbuf = format_strings ? xmalloc(1 << 20) : NULL;
            uint64_t q1 = (c + 1) * chunk < n_q ? (c + 1) * chunk : n_q;
            for (uint64_t q = c * chunk; q < q1; q++) {
                orc_row *rows = rows_out ? rows_out + q * (uint64_t)cap : local;
                int n = classify_buf(t, bases + base_off[q], base_off[q + 1] - base_off[q], skip_exact,
                                     raw_confidence, rows, cap, counts, kbuf, probs, &ws);
                if (n < 0) bad++;
                if (n_rows_out) n_rows_out[q] = n;
                if (format_strings && n > 0) {
                    char label[32];
                    snprintf(label, sizeof label, "q%llu", (unsigned long long)q);
                    (void)orc_format_out(t, label, rows, n, sbuf, 1 << 20); /* :85 */
                }
            }
            free(counts);
            free(kbuf);
            free(probs);
            free(local);
            free(sbuf);
        }
        ws_free(&ws);
        if (pinned) (void)sched_setaffinity(0, sizeof old_mask, &old_mask); /* pool threads outlive the region */
    }
    return bad;
}

int64_t orc_classify_batch(const orc_tree *t, uint64_t n_q, const uint8_t *bases,
                           const uint64_t *base_off, int skip_exact, int raw_confidence,
                           int threads, orc_row *rows_out, int cap, int32_t *n_rows_out,
                           int format_strings) {
    return orc_classify_batch_ex(t, n_q, bases, base_off, skip_exact, raw_confidence, threads, rows_out, cap,
                                 n_rows_out, format_strings, NULL, 0);
}

/* Parity taps over many queries at once (tests at full database size): raxtax.rs:41,55-68 for every query,
 * counts_out = n_q rows of num_tips u16; t_out[q] = number of distinct k-mers. */
void orc_hit_counts_batch(const orc_tree *t, uint64_t n_q, const uint8_t *bases, const uint64_t *base_off,
                          int skip_exact, int threads, uint16_t *counts_out, uint32_t *t_out) {
    if (threads < 1) threads = 1;
    uint64_t max_len = 0;
    for (uint64_t q = 0; q < n_q; q++)
        if (base_off[q + 1] - base_off[q] > max_len) max_len = base_off[q + 1] - base_off[q];
#pragma omp parallel num_threads(threads)
    {
        uint16_t *kbuf = xmalloc((max_len + 1) * sizeof(uint16_t));
#pragma omp for schedule(dynamic, 1)
        for (uint64_t q = 0; q < n_q; q++) {
            uint32_t nk = hit_counts_buf(t, bases + base_off[q], base_off[q + 1] - base_off[q], skip_exact,
                                         counts_out + q * t->num_tips, kbuf);
            if (t_out) t_out[q] = nk;
        }
        free(kbuf);
    }
}

/* prob.rs:8-103 for many count vectors: tables[q * tstride + m] = table[m] / Z (the probability of a reference
 * with m hits; 0 for absent m), z[q] = Z, rc[q] = the return code of orc_prob_table. */
void orc_prob_tables_batch(uint64_t n_q, const uint32_t *t_arr, const uint16_t *counts, uint64_t n_refs, int threads,
                           double *tables, uint64_t tstride, double *z, int32_t *rc_out) {
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
        orc_ws ws;
        ws_init(&ws);
#pragma omp for schedule(dynamic, 1)
        for (uint64_t q = 0; q < n_q; q++) {
            double *tb = tables + q * tstride;
            double zz = 0.0;
            int rc = t_arr[q] + 1 <= tstride ? prob_table_ws(&ws, (uint16_t)t_arr[q], t_arr[q] / 2, counts + q * n_refs, n_refs, tb, &zz) : -9;
            if (rc == 0 && !(zz > 0.0)) rc = -3;
            if (rc == 0)
                for (uint64_t m = 0; m <= t_arr[q]; m++) tb[m] /= zz;
            if (z) z[q] = zz;
            if (rc_out) rc_out[q] = rc;
        }
        ws_free(&ws);
    }
}
