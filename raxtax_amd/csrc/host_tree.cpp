// Host mirror of the reference's index build and FASTA parsing:
//   Tree::new                 src/tree.rs:46-140
//   parse_reference_fasta_str src/parser.rs:46-105
//   parse_query_fasta_str     src/parser.rs:117-154
//   map_dna_char              src/parser.rs:11-34
// This is plumbing around the device hot path (SURVEY.md 8f "next" rows 1-2): it produces
// exactly the read-only inputs raxtax() takes (k_mer_map, taxonomy, exact-sequence map).
// Written for this library (arena tree, counting-sort CSR build); not a translation.
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>

#include "rtx_internal.hpp"

namespace rtx {

static thread_local std::string g_err;

void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}

size_t BytesHash::operator()(std::string_view s) const noexcept {
    // 64-bit multiply-rotate over 8-byte words (the reference uses ahash; any hash will do)
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (s.size() * 0xff51afd7ed558ccdull);
    const char *p = s.data();
    size_t n = s.size();
    while (n >= 8) {
        uint64_t w;
        memcpy(&w, p, 8);
        h = (h ^ w) * 0xc4ceb9fe1a85ec53ull;
        h = (h << 29) | (h >> 35);
        p += 8;
        n -= 8;
    }
    uint64_t w = 0;
    memcpy(&w, p, n);
    h = (h ^ w) * 0xff51afd7ed558ccdull;
    return (size_t)(h ^ (h >> 32));
}

// one-hot nibble -> 2-bit code, 0xFF for anything else (src/utils.rs:17-25)
static inline uint8_t two_bit(uint8_t c) {
    switch (c) {
        case 1: return 0;
        case 2: return 1;
        case 4: return 2;
        case 8: return 3;
        default: return 0xFF;
    }
}

// Calls f(kmer) for every valid 8-base window of seq (src/tree.rs:114-123 / utils.rs:27-40):
// rolling 16-bit code, first base of the window in bits 15:14.
template <class F>
static inline void for_each_window_kmer(const uint8_t *seq, uint64_t len, F &&f) {
    uint32_t code = 0;
    uint32_t valid_run = 0;  // consecutive valid bases ending here
    for (uint64_t i = 0; i < len; i++) {
        uint8_t c = two_bit(seq[i]);
        if (c == 0xFF) {
            valid_run = 0;
            code = 0;
            continue;
        }
        code = ((code << 2) | c) & 0xFFFFu;
        if (++valid_run >= 8) f(code);
    }
}

bool derive_flat_nodes(uint64_t n_refs, uint32_t n_nodes, const uint32_t *begin, const uint32_t *end,
                       const uint32_t *first_child, const uint32_t *n_children, const uint8_t *type,
                       FlatNodes &out) {
    if (n_nodes == 0) { set_error("taxonomy has no nodes"); return false; }
    out.begin.assign(begin, begin + n_nodes);
    out.end.assign(end, end + n_nodes);
    out.first_child.assign(first_child, first_child + n_nodes);
    out.n_children.assign(n_children, n_children + n_nodes);
    out.type.assign(type, type + n_nodes);
    out.parent.assign(n_nodes, kNoNode);
    out.depth.assign(n_nodes, 0);
    out.max_depth = 0;
    uint64_t next = 1;  // BFS order: children blocks appear in node order
    for (uint32_t v = 0; v < n_nodes; v++) {
        if (out.type[v] > kSequence) { set_error("node %u: bad type %u", v, out.type[v]); return false; }
        if (out.begin[v] > out.end[v] || out.end[v] > n_refs) {
            set_error("node %u: range [%u,%u) outside [0,%llu)", v, out.begin[v], out.end[v],
                      (unsigned long long)n_refs);
            return false;
        }
        if (out.n_children[v] == 0) continue;
        if (out.first_child[v] != next || next + out.n_children[v] > n_nodes) {
            set_error("node %u: children [%u,+%u) are not in breadth-first order", v, out.first_child[v],
                      out.n_children[v]);
            return false;
        }
        for (uint32_t c = 0; c < out.n_children[v]; c++) {
            out.parent[next + c] = v;
            out.depth[next + c] = out.depth[v] + 1;
            out.max_depth = std::max(out.max_depth, out.depth[v] + 1);
        }
        next += out.n_children[v];
    }
    if (next != n_nodes) { set_error("taxonomy has %u nodes but %llu are reachable", n_nodes, (unsigned long long)next); return false; }
    return true;
}

// parser.rs:11-34; returns 0xFF where the reference panics
static inline uint8_t map_dna_char(unsigned char ch) {
    constexpr uint8_t a = 1, c = 2, g = 4, t = 8;
    switch (ch >= 'a' && ch <= 'z' ? ch - 32 : ch) {
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
        default: return 0xFF;
    }
}

// lines(): trimmed, empty and ';'-prefixed lines dropped (parser.rs:53-57,124-128).
// ASCII whitespace only (the reference's str::trim also strips Unicode spaces).
static std::vector<std::string_view> fasta_lines(const char *text, uint64_t len) {
    std::vector<std::string_view> out;
    auto is_ws = [](char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f'; };
    uint64_t i = 0;
    while (i < len) {
        uint64_t j = i;
        while (j < len && text[j] != '\n') j++;
        uint64_t b = i, e = j;
        while (b < e && is_ws(text[b])) b++;
        while (e > b && is_ws(text[e - 1])) e--;
        if (e > b && text[b] != ';') out.emplace_back(text + b, e - b);
        i = j + 1;
    }
    return out;
}

// ---- parallel FASTA parsing: the text is cut at lines that begin with '>' -------------------------------
struct FastaPiece {
    std::vector<std::string> labels;
    std::vector<uint8_t> bytes;
    std::vector<uint64_t> off;
    bool last_empty = false;  // queries: the piece ends in a header without sequence
    std::string err;
};

static unsigned parse_threads(uint64_t len) {
    return (unsigned)std::min<uint64_t>(rtx::host_threads(16u), std::max<uint64_t>(1, len >> 20));  // >= 1 MiB per piece
}

// cut points [0, ..., len]: every inner one is the position of a '>' that directly follows a newline
static std::vector<uint64_t> fasta_cuts(const char *text, uint64_t len, unsigned parts) {
    std::vector<uint64_t> cuts{0};
    for (unsigned i = 1; i < parts; i++) {
        uint64_t pos = len * i / parts;
        if (pos <= cuts.back()) continue;
        const void *nl;
        while (pos < len && (nl = memchr(text + pos, '\n', len - pos)) != nullptr) {
            pos = (uint64_t)((const char *)nl - text) + 1;
            if (pos < len && text[pos] == '>') break;
        }
        if (pos >= len || text[pos] != '>') break;
        if (pos > cuts.back()) cuts.push_back(pos);
    }
    cuts.push_back(len);
    return cuts;
}

template <class F>
static void run_pieces(size_t np, F fn) {
    if (np <= 1) { for (size_t i = 0; i < np; i++) fn(i); return; }
    std::vector<std::thread> th;
    for (size_t i = 0; i < np; i++) th.emplace_back(fn, i);
    for (auto &t : th) t.join();
}

// first match of `tax=([^;]+);` (parser.rs:50,70-78)
static bool find_tax(std::string_view s, std::string_view &out) {
    size_t pos = 0;
    while ((pos = s.find("tax=", pos)) != std::string_view::npos) {
        size_t b = pos + 4, e = s.find(';', b);
        if (e != std::string_view::npos && e > b) {
            out = s.substr(b, e - b);
            return true;
        }
        pos += 1;
    }
    return false;
}

// The device walk carries RTX_MAX_DEPTH levels per result row (the reference has no limit, lineage.rs:119-179): a deeper taxonomy is
// refused where the tree is built, with the lineage named -- not later, at index creation, with a number only.
int check_tree_depth(const rtx_tree &t) {
    if (t.flat.max_depth <= RTX_MAX_DEPTH) return RTX_OK;
    size_t worst = 0, worst_levels = 0;
    for (size_t i = 0; i < t.lineages.size(); i++) {
        size_t levels = 1;
        for (char c : t.lineages[i]) levels += c == ',';
        if (levels > worst_levels) { worst_levels = levels; worst = i; }
    }
    const std::string &l = t.lineages.empty() ? std::string() : t.lineages[worst];
    set_error("lineage of %zu levels, RTX_MAX_DEPTH = %u are carried per result row: \"%.160s%s\"", worst_levels, RTX_MAX_DEPTH, l.c_str(), l.size() > 160 ? "..." : "");
    return RTX_ERR_DEPTH;
}

void flatten_tree(rtx_tree &t);
static void flatten(rtx_tree &t) { flatten_tree(t); }
void flatten_tree(rtx_tree &t) {
    // BFS from the root; childless Sequence nodes are dropped (they cannot influence
    // Lineage::evaluate: they are neither Inner nor Taxon and have nothing to recurse into).
    FlatNodes &f = t.flat;
    std::vector<uint32_t> order{0};  // arena ids in BFS order
    f.begin.clear(); f.end.clear(); f.first_child.clear(); f.n_children.clear();
    f.parent.clear(); f.depth.clear(); f.type.clear();
    f.parent.push_back(kNoNode);
    f.depth.push_back(0);
    f.max_depth = 0;
    for (size_t qi = 0; qi < order.size(); qi++) {
        const Node &nd = t.nodes[order[qi]];
        uint32_t first = (uint32_t)order.size(), cnt = 0;
        for (uint32_t c : nd.children) {
            const Node &ch = t.nodes[c];
            if (ch.type == kSequence && ch.children.empty()) continue;
            order.push_back(c);
            f.parent.push_back((uint32_t)qi);
            f.depth.push_back(f.depth[qi] + 1);
            f.max_depth = std::max(f.max_depth, f.depth[qi] + 1);
            cnt++;
        }
        f.begin.push_back((uint32_t)nd.lo);
        f.end.push_back((uint32_t)nd.hi);
        f.first_child.push_back(cnt ? first : 0);
        f.n_children.push_back(cnt);
        f.type.push_back((uint8_t)nd.type);
    }
}

// Tree.k_mer_map (tree.rs:114-123,134-137) as CSR by counting sort: pass 1 counts the distinct k-mers of every
// reference, pass 2 scatters reference ids in ascending order, which leaves every list sorted and unique.
static void build_kmer_map(rtx_tree &tr) {
    rtx_tree *t = &tr;
    const uint64_t n = t->n;
    t->csr_off.assign(RTX_NUM_KMERS + 1, 0);
    std::vector<uint32_t> stamp(RTX_NUM_KMERS, 0);  // last reference (idx+1) that touched the k-mer
    for (uint64_t idx = 0; idx < n; idx++) {
        const uint32_t tag = (uint32_t)idx + 1;
        for_each_window_kmer(t->seq_bytes.data() + t->seq_off[idx], t->seq_off[idx + 1] - t->seq_off[idx],
                             [&](uint32_t k) {
                                 if (stamp[k] != tag) { stamp[k] = tag; t->csr_off[k + 1]++; }
                             });
    }
    for (uint32_t k = 0; k < RTX_NUM_KMERS; k++) t->csr_off[k + 1] += t->csr_off[k];
    t->postings.resize(t->csr_off[RTX_NUM_KMERS]);
    std::vector<uint64_t> cursor(t->csr_off.begin(), t->csr_off.end() - 1);
    std::fill(stamp.begin(), stamp.end(), 0);
    for (uint64_t idx = 0; idx < n; idx++) {
        const uint32_t tag = (uint32_t)idx + 1;
        for_each_window_kmer(t->seq_bytes.data() + t->seq_off[idx], t->seq_off[idx + 1] - t->seq_off[idx],
                             [&](uint32_t k) {
                                 if (stamp[k] != tag) { stamp[k] = tag; t->postings[cursor[k]++] = (uint32_t)idx; }
                             });
    }
}

static int build_tree(std::vector<std::string> &&lineages_in, const uint8_t *seq_bytes, const uint64_t *seq_off,
                      rtx_tree **out, bool with_kmer_map = true) {
    const uint64_t n = lineages_in.size();
    if (n > 0xFFFFFFFFull) {  // check_lineage_size, tree.rs:24-31 (IndexType = u32)
        set_error("too many database sequences for 32-bit indices");
        return RTX_ERR_INVALID;
    }
    auto t = new rtx_tree();
    t->n = n;
    // tree.rs:53-54: stable sort of (lineage, sequence) pairs by lineage, bytewise
    t->orig_idx.resize(n);
    std::iota(t->orig_idx.begin(), t->orig_idx.end(), 0);
    std::stable_sort(t->orig_idx.begin(), t->orig_idx.end(),
                     [&](uint64_t a, uint64_t b) { return lineages_in[a] < lineages_in[b]; });
    t->lineages.resize(n);
    t->seq_off.assign(n + 1, 0);
    for (uint64_t i = 0; i < n; i++) {
        uint64_t o = t->orig_idx[i];
        t->lineages[i] = std::move(lineages_in[o]);
        t->seq_off[i + 1] = t->seq_off[i] + (seq_off[o + 1] - seq_off[o]);
    }
    t->seq_bytes.resize(t->seq_off[n]);
    for (uint64_t i = 0; i < n; i++) {
        uint64_t o = t->orig_idx[i];
        memcpy(t->seq_bytes.data() + t->seq_off[i], seq_bytes + seq_off[o], seq_off[o + 1] - seq_off[o]);
    }

    // taxonomy, tree.rs:56-107 (arena instead of owned child vectors)
    t->nodes.reserve(2 * n + 16);
    t->nodes.push_back(Node{"root", 0, 1, kInner, {}});
    uint64_t confidence_idx = 0;
    for (uint64_t idx = 0; idx < n; idx++) {
        std::string_view lin = t->lineages[idx];
        uint32_t cur = 0;
        size_t pos = 0;
        for (;;) {
            size_t comma = lin.find(',', pos);
            bool last = comma == std::string_view::npos;
            std::string_view label = lin.substr(pos, last ? std::string_view::npos : comma - pos);
            // tree.rs:78-98: compare with the label of the current node's LAST child only
            bool need_new = t->nodes[cur].children.empty() ||
                            std::string_view(t->nodes[t->nodes[cur].children.back()].label) != label;
            if (need_new) {
                t->nodes.push_back(Node{std::string(label), confidence_idx, confidence_idx + 1,
                                        last ? kTaxon : kInner, {}});
                t->nodes[cur].children.push_back((uint32_t)t->nodes.size() - 1);
            }
            t->nodes[cur].hi = confidence_idx + 1;
            if (last) confidence_idx += 1;  // tree.rs:97-99
            cur = t->nodes[cur].children.back();
            if (last) break;
            pos = comma + 1;
        }
        // tree.rs:102-107: per-reference Sequence node
        t->nodes.push_back(Node{t->nodes[cur].label, confidence_idx - 1, confidence_idx, kSequence, {}});
        t->nodes[cur].children.push_back((uint32_t)t->nodes.size() - 1);
        t->nodes[cur].hi = confidence_idx;
    }
    t->nodes[0].hi = confidence_idx;  // tree.rs:127
    t->num_tips = confidence_idx;     // tree.rs:138

    // Tree.sequences (tree.rs:50-51,109-112): ids pushed in sorted order
    t->sequences.reserve(n * 2);
    for (uint64_t idx = 0; idx < n; idx++) {
        std::string_view key((const char *)t->seq_bytes.data() + t->seq_off[idx], t->seq_off[idx + 1] - t->seq_off[idx]);
        t->sequences[key].push_back((uint32_t)idx);
    }

    if (with_kmer_map) build_kmer_map(*t);  // else: bitmaps are built on the GPU from the sequences, or rtx_tree_build_kmer_map later
    flatten(*t);
    if (int rc = check_tree_depth(*t)) { delete t; return rc; }
    *out = t;
    return RTX_OK;
}

}  // namespace rtx

using namespace rtx;

extern "C" {

int rtx_abi_version(void) { return RTX_ABI_VERSION; }
const char *rtx_last_error(void) { return rtx::g_err.c_str(); }

int rtx_tree_build(uint64_t n, const char *lineage_bytes, const uint64_t *lineage_off,
                   const uint8_t *seq_bytes, const uint64_t *seq_off, rtx_tree **out) {
    return rtx_tree_build_ex(n, lineage_bytes, lineage_off, seq_bytes, seq_off, 0, out);
}

int rtx_tree_build_ex(uint64_t n, const char *lineage_bytes, const uint64_t *lineage_off, const uint8_t *seq_bytes,
                      const uint64_t *seq_off, uint32_t flags, rtx_tree **out) {
    if (!out || (n && (!lineage_bytes || !lineage_off || !seq_off))) {
        set_error("rtx_tree_build: null argument");
        return RTX_ERR_INVALID;
    }
    try {
        std::vector<std::string> lin(n);
        for (uint64_t i = 0; i < n; i++) lin[i].assign(lineage_bytes + lineage_off[i], lineage_off[i + 1] - lineage_off[i]);
        return build_tree(std::move(lin), seq_bytes, seq_off, out, !(flags & RTX_TREE_SKIP_KMER_MAP));
    } catch (const std::bad_alloc &) {
        set_error("rtx_tree_build: out of host memory");
        return RTX_ERR_OOM;
    }
}

int rtx_tree_parse_reference_fasta(const char *text, uint64_t len, rtx_tree **out) {
    return rtx_tree_parse_reference_fasta_ex(text, len, 0, out);
}

int rtx_tree_build_kmer_map(rtx_tree *tree) {
    if (!tree) { set_error("null argument"); return RTX_ERR_INVALID; }
    try {
        if (tree->csr_off.empty()) build_kmer_map(*tree);
        return RTX_OK;
    } catch (const std::bad_alloc &) {
        set_error("out of host memory");
        return RTX_ERR_OOM;
    }
}

int rtx_tree_parse_reference_fasta_ex(const char *text, uint64_t len, uint32_t flags, rtx_tree **out) {
    if (!out) { set_error("null argument"); return RTX_ERR_INVALID; }
    if (!text || len == 0) { set_error("File is empty"); return RTX_ERR_PARSE; }  // parser.rs:47-49
    try {
        // the file is cut at header lines into pieces that are parsed in parallel with the reference's
        // line rules (parser.rs:53-98); a sequence is attached when the next header (or the end) is reached
        const std::vector<uint64_t> cuts = fasta_cuts(text, len, parse_threads(len));
        const size_t np = cuts.size() - 1;
        std::vector<FastaPiece> pieces(np);
        auto parse_piece = [&](size_t pi) {
            FastaPiece &P = pieces[pi];
            const bool last = pi + 1 == np;
            auto lines = fasta_lines(text + cuts[pi], cuts[pi + 1] - cuts[pi]);
            if (pi == 0 && (lines.empty() || lines[0][0] != '>')) { P.err = "Not a valid FASTA file"; return; }
            std::vector<uint8_t> cur;
            P.off.push_back(0);
            auto push = [&]() {
                P.bytes.insert(P.bytes.end(), cur.begin(), cur.end());
                P.off.push_back(P.bytes.size());
                cur.clear();
            };
            for (std::string_view line : lines) {
                if (line[0] == '>') {
                    std::string_view tax;
                    if (!find_tax(line.substr(1), tax)) {
                        P.err = "Unexpected taxonomical annotation detected in label " + std::string(line.substr(1));
                        return;
                    }
                    P.labels.emplace_back(tax);
                    if (!cur.empty()) push();  // parser.rs:80-83
                } else {
                    for (char ch : line) {
                        uint8_t c = map_dna_char((unsigned char)ch);
                        if (c == 0xFF) { P.err = std::string("Unexpected character: ") + ch; return; }
                        cur.push_back(c);
                    }
                }
            }
            if (last || !cur.empty()) push();  // parser.rs:98 (a later piece starts with a header)
        };
        run_pieces(np, parse_piece);
        std::vector<std::string> labels;
        std::vector<uint8_t> bytes;
        std::vector<uint64_t> off{0};
        for (FastaPiece &P : pieces) {
            if (!P.err.empty()) { set_error("%s", P.err.c_str()); return RTX_ERR_PARSE; }
            for (std::string &l : P.labels) labels.push_back(std::move(l));
            const uint64_t base = bytes.size();
            bytes.insert(bytes.end(), P.bytes.begin(), P.bytes.end());
            for (size_t i = 1; i < P.off.size(); i++) off.push_back(base + P.off[i]);
        }
        if (labels.size() != off.size() - 1) {
            set_error("Number of sequences does not match number of labels");
            return RTX_ERR_PARSE;
        }
        return build_tree(std::move(labels), bytes.data(), off.data(), out, !(flags & RTX_TREE_SKIP_KMER_MAP));
    } catch (const std::bad_alloc &) {
        set_error("out of host memory");
        return RTX_ERR_OOM;
    }
}

void rtx_tree_destroy(rtx_tree *tree) { delete tree; }
uint64_t rtx_tree_num_tips(const rtx_tree *tree) { return tree ? tree->num_tips : 0; }
const char *rtx_tree_lineage(const rtx_tree *tree, uint64_t i) {
    return (tree && i < tree->lineages.size()) ? tree->lineages[i].c_str() : nullptr;
}
uint64_t rtx_tree_original_index(const rtx_tree *tree, uint64_t i) {
    return (tree && i < tree->orig_idx.size()) ? tree->orig_idx[i] : ~0ull;
}
int rtx_tree_kmer_csr(const rtx_tree *tree, const uint64_t **offsets, const uint32_t **postings) {
    if (!tree || !offsets || !postings) { set_error("null argument"); return RTX_ERR_INVALID; }
    if (tree->csr_off.empty()) { set_error("tree was built with RTX_TREE_SKIP_KMER_MAP"); return RTX_ERR_STATE; }
    *offsets = tree->csr_off.data();
    *postings = tree->postings.data();
    return RTX_OK;
}
uint64_t rtx_tree_exact_matches(const rtx_tree *tree, const uint8_t *seq, uint64_t len, const uint32_t **ids) {
    if (ids) *ids = nullptr;
    if (!tree) return 0;
    auto it = tree->sequences.find(std::string_view((const char *)seq, len));
    if (it == tree->sequences.end()) return 0;
    if (ids) *ids = it->second.data();
    return it->second.size();
}
uint64_t rtx_tree_exact_matches_batch(const rtx_tree *tree, uint64_t n_queries, const uint8_t *bases,
                                      const uint64_t *base_off, uint64_t *exact_off, uint32_t *exact_ids,
                                      uint64_t ids_cap) {
    if (!tree || !base_off || !exact_off) return 0;
    uint64_t total = 0;
    exact_off[0] = 0;
    for (uint64_t q = 0; q < n_queries; q++) {
        auto it = tree->sequences.find(std::string_view((const char *)bases + base_off[q], base_off[q + 1] - base_off[q]));
        if (it != tree->sequences.end()) {
            for (uint32_t id : it->second) {
                if (exact_ids && total < ids_cap) exact_ids[total] = id;
                total++;
            }
        }
        exact_off[q + 1] = total;
    }
    return total;
}
int rtx_tree_nodes(const rtx_tree *tree, rtx_nodes_view *out) {
    if (!tree || !out) { set_error("null argument"); return RTX_ERR_INVALID; }
    const FlatNodes &f = tree->flat;
    out->n_nodes = f.size();
    out->node_begin = f.begin.data();
    out->node_end = f.end.data();
    out->node_first_child = f.first_child.data();
    out->node_n_children = f.n_children.data();
    out->node_parent = f.parent.data();
    out->node_type = f.type.data();
    return RTX_OK;
}

int rtx_queries_parse_fasta(const char *text, uint64_t len, const char *const *skip, uint64_t n_skip,
                            rtx_queries **out) {
    return rtx_queries_parse_fasta_block(text, len, skip, n_skip, 0, out);
}

uint64_t rtx_fasta_block_end(const char *text, uint64_t len) {
    if (!text) return 0;
    for (uint64_t i = len; i > 1; i--)
        if (text[i - 1] == '>' && text[i - 2] == '\n') return i - 1;
    return 0;
}

int rtx_queries_parse_fasta_block(const char *text, uint64_t len, const char *const *skip, uint64_t n_skip, uint32_t flags,
                                  rtx_queries **out) {
    if (!out) { set_error("null argument"); return RTX_ERR_INVALID; }
    if (!text || len == 0) { set_error("File is empty"); return RTX_ERR_PARSE; }
    const bool more_follows = (flags & RTX_FASTA_MORE_FOLLOWS) != 0;
    try {
        std::unordered_map<std::string_view, int> skipset;
        for (uint64_t i = 0; i < n_skip; i++) skipset.emplace(skip[i], 1);
        // pieces cut at header lines, parsed in parallel with the reference's rules (parser.rs:124-153): a record
        // is pushed when the next header arrives, but only if it has bases; the very last record always is
        const std::vector<uint64_t> cuts = fasta_cuts(text, len, parse_threads(len));
        const size_t np = cuts.size() - 1;
        std::vector<FastaPiece> pieces(np);
        auto parse_piece = [&](size_t pi) {
            FastaPiece &P = pieces[pi];
            const bool last = pi + 1 == np && !more_follows;  // a later block starts with a header as well
            auto lines = fasta_lines(text + cuts[pi], cuts[pi + 1] - cuts[pi]);
            if (pi == 0 && !(flags & RTX_FASTA_NOT_FIRST) && (lines.empty() || lines[0][0] != '>')) { P.err = "Not a valid FASTA file"; return; }
            std::string cur_label;
            std::vector<uint8_t> cur;
            P.off.push_back(0);
            auto push = [&]() {  // parser.rs:139,149-153
                if (skipset.count(cur_label)) return;
                P.labels.push_back(cur_label);
                P.bytes.insert(P.bytes.end(), cur.begin(), cur.end());
                P.off.push_back(P.bytes.size());
            };
            for (std::string_view line : lines) {
                if (line[0] == '>') {
                    if (!cur.empty()) { push(); cur.clear(); }
                    cur_label.assign(line.substr(1));
                } else {
                    for (char ch : line) {
                        uint8_t c = map_dna_char((unsigned char)ch);
                        if (c == 0xFF) { P.err = std::string("Unexpected character: ") + ch; return; }
                        cur.push_back(c);
                    }
                }
            }
            if (last || !cur.empty()) push();  // a header without bases is overwritten by the next header
        };
        run_pieces(np, parse_piece);
        auto q = new rtx_queries();
        for (FastaPiece &P : pieces) {
            if (!P.err.empty()) { set_error("%s", P.err.c_str()); delete q; return RTX_ERR_PARSE; }
            for (std::string &l : P.labels) q->labels.push_back(std::move(l));
            const uint64_t base = q->bases.size();
            q->bases.insert(q->bases.end(), P.bytes.begin(), P.bytes.end());
            for (size_t i = 1; i < P.off.size(); i++) q->base_off.push_back(base + P.off[i]);
        }
        *out = q;
        return RTX_OK;
    } catch (const std::bad_alloc &) {
        set_error("out of host memory");
        return RTX_ERR_OOM;
    }
}
void rtx_queries_destroy(rtx_queries *q) { delete q; }
uint64_t rtx_queries_len(const rtx_queries *q) { return q ? q->labels.size() : 0; }
const char *rtx_queries_label(const rtx_queries *q, uint64_t i) {
    return (q && i < q->labels.size()) ? q->labels[i].c_str() : nullptr;
}
int rtx_queries_data(const rtx_queries *q, const uint8_t **bases, const uint64_t **base_off) {
    if (!q || !bases || !base_off) { set_error("null argument"); return RTX_ERR_INVALID; }
    *bases = q->bases.data();
    *base_off = q->base_off.data();
    return RTX_OK;
}

}  // extern "C"
