// Host mirror of the reference's binary database format:
//   Tree::save_to_file   src/tree.rs:147-153   bincode::serialize_into(&mut output, &self)
//   Tree::load_from_file src/tree.rs:155-164   bincode::deserialize(&buffer)
// bincode 1.3.3 with default options (Cargo.toml:23): little-endian, fixed-width integers, usize and every
// length as u64, enum variants as u32 indices, no framing.  Field order = declaration order:
//   Tree { root: Node, lineages: Vec<String>, sequences: HashMap<Vec<u8>, Vec<u32>>,
//          k_mer_map: Vec<Vec<u32>>, num_tips: usize }                       (tree.rs:36-43)
//   Node { label: String, confidence_range: (usize, usize), children: Vec<Node>, node_type: NodeType }
//   NodeType { Inner = 0, Taxon = 1, Sequence = 2 }                           (tree.rs:181-194)
// HashMap entries are written in iteration order, which is arbitrary in the reference (ahash); readers
// must not depend on it.  No reference test pins this layout ("parity unpinned"; tests/test_bin_format.py
// cross-checks it with an independent struct-level parser).
#include <cstdio>
#include <cstring>
#include <fstream>

#include "rtx_internal.hpp"

namespace {

using rtx::Node;
using rtx::NodeType;

static_assert(__BYTE_ORDER__ == __ORDER_LITTLE_ENDIAN__, "bincode integers are little-endian: arrays are copied as they are");

struct Writer {
    std::vector<uint8_t> buf;
    void bytes(const void *p, size_t n) {
        const size_t at = buf.size();
        buf.resize(at + n);
        memcpy(buf.data() + at, p, n);
    }
    void u32(uint32_t v) { bytes(&v, 4); }
    void u64(uint64_t v) { bytes(&v, 8); }
    void str(const std::string &s) { u64(s.size()); bytes(s.data(), s.size()); }
};

void write_node(Writer &w, const rtx_tree &t, uint32_t id) {
    const Node &n = t.nodes[id];
    w.str(n.label);
    w.u64(n.lo);
    w.u64(n.hi);
    w.u64(n.children.size());
    for (uint32_t c : n.children) write_node(w, t, c);
    w.u32((uint32_t)n.type);
}

struct Reader {
    const uint8_t *p, *end;
    bool ok = true;
    bool need(size_t n) { if ((size_t)(end - p) < n) { ok = false; return false; } return true; }
    uint32_t u32() { if (!need(4)) return 0; uint32_t v; memcpy(&v, p, 4); p += 4; return v; }
    uint64_t u64() { if (!need(8)) return 0; uint64_t v; memcpy(&v, p, 8); p += 8; return v; }
    bool str(std::string &s) {
        const uint64_t n = u64();
        if (!ok || !need(n)) return false;
        s.assign((const char *)p, n);
        p += n;
        return true;
    }
};

bool read_node(Reader &r, rtx_tree &t, uint32_t &id_out, int depth) {
    if (depth > 4096) { r.ok = false; return false; }
    const uint32_t id = (uint32_t)t.nodes.size();
    t.nodes.emplace_back();
    {
        std::string label;
        if (!r.str(label)) return false;
        t.nodes[id].label = std::move(label);
    }
    t.nodes[id].lo = r.u64();
    t.nodes[id].hi = r.u64();
    const uint64_t nch = r.u64();
    if (!r.ok || nch > (uint64_t)(r.end - r.p)) { r.ok = false; return false; }
    std::vector<uint32_t> kids;
    kids.reserve(nch);
    for (uint64_t i = 0; i < nch; i++) {
        uint32_t c;
        if (!read_node(r, t, c, depth + 1)) return false;
        kids.push_back(c);
    }
    t.nodes[id].children = std::move(kids);
    const uint32_t ty = r.u32();
    if (!r.ok || ty > 2) { r.ok = false; return false; }
    t.nodes[id].type = (NodeType)ty;
    id_out = id;
    return true;
}

}  // namespace

namespace rtx {
void flatten_tree(rtx_tree &t);  // host_tree.cpp
int check_tree_depth(const rtx_tree &t);
}

extern "C" {

int rtx_tree_save_bin(const rtx_tree *tree, const char *path) {
    if (!tree || !path) { rtx::set_error("null argument"); return RTX_ERR_INVALID; }
    if (tree->csr_off.empty()) {  // built with RTX_TREE_SKIP_KMER_MAP: the file format needs Tree.k_mer_map
        const int rc = rtx_tree_build_kmer_map(const_cast<rtx_tree *>(tree));
        if (rc) return rc;
    }
    Writer w;
    w.buf.reserve(tree->postings.size() * 4 + tree->seq_bytes.size() * 2 + tree->lineages.size() * 128 + (1u << 20));
    write_node(w, *tree, 0);                                   // root
    w.u64(tree->lineages.size());                              // lineages
    for (const std::string &l : tree->lineages) w.str(l);
    w.u64(tree->sequences.size());                             // sequences (map order is arbitrary)
    for (const auto &kv : tree->sequences) {
        w.u64(kv.first.size());
        w.bytes(kv.first.data(), kv.first.size());
        w.u64(kv.second.size());
        w.bytes(kv.second.data(), kv.second.size() * 4);
    }
    w.u64(RTX_NUM_KMERS);                                      // k_mer_map
    for (uint32_t k = 0; k < RTX_NUM_KMERS; k++) {
        const uint64_t b = tree->csr_off[k], e = tree->csr_off[k + 1];
        w.u64(e - b);
        w.bytes(tree->postings.data() + b, (e - b) * 4);
    }
    w.u64(tree->num_tips);                                     // num_tips
    std::ofstream f(path, std::ios::binary);
    if (!f) { rtx::set_error("cannot create %s", path); return RTX_ERR_INVALID; }
    f.write((const char *)w.buf.data(), (std::streamsize)w.buf.size());
    return f.good() ? RTX_OK : RTX_ERR_INVALID;
}

int rtx_tree_load_bin(const char *path, rtx_tree **out) {
    if (!path || !out) { rtx::set_error("null argument"); return RTX_ERR_INVALID; }
    std::vector<uint8_t> buf;
    {
        FILE *f = fopen(path, "rb");
        if (!f) { rtx::set_error("cannot open %s", path); return RTX_ERR_PARSE; }
        fseek(f, 0, SEEK_END);
        const long sz = ftell(f);
        fseek(f, 0, SEEK_SET);
        if (sz < 0) { fclose(f); rtx::set_error("cannot read %s", path); return RTX_ERR_PARSE; }
        buf.resize((size_t)sz);
        const size_t got = sz ? fread(buf.data(), 1, (size_t)sz, f) : 0;
        fclose(f);
        if (got != (size_t)sz) { rtx::set_error("cannot read %s", path); return RTX_ERR_PARSE; }
    }
    Reader r{buf.data(), buf.data() + buf.size()};
    auto t = new rtx_tree();
    auto fail = [&](const char *what) { rtx::set_error("%s: not a raxtax database (%s)", path, what); delete t; return RTX_ERR_PARSE; };
    uint32_t root;
    if (!read_node(r, *t, root, 0)) return fail("root");
    const uint64_t nl = r.u64();
    if (!r.ok || nl > buf.size()) return fail("lineages");
    t->lineages.resize(nl);
    for (uint64_t i = 0; i < nl; i++)
        if (!r.str(t->lineages[i])) return fail("lineage string");
    t->n = nl;
    // sequences: key bytes + ids; rebuild the per-reference sequences from them
    const uint64_t ns = r.u64();
    if (!r.ok || ns > buf.size()) return fail("sequences");
    std::vector<std::pair<std::pair<const uint8_t *, uint64_t>, std::vector<uint32_t>>> entries(ns);
    std::vector<uint64_t> len_of(nl, ~0ull);
    std::vector<const uint8_t *> ptr_of(nl, nullptr);
    for (uint64_t s = 0; s < ns; s++) {
        const uint64_t klen = r.u64();
        if (!r.ok || !r.need(klen)) return fail("sequence key");
        const uint8_t *kp = r.p;
        r.p += klen;
        const uint64_t nid = r.u64();
        if (!r.ok || nid > nl || nid > (uint64_t)(r.end - r.p) / 4) return fail("sequence ids");  // nid * 4 cannot wrap below
        entries[s].first = {kp, klen};
        if (!r.need(nid * 4)) return fail("sequence ids");
        entries[s].second.resize(nid);
        memcpy(entries[s].second.data(), r.p, nid * 4);
        r.p += nid * 4;
        for (const uint32_t id : entries[s].second) {
            if (id >= nl) return fail("sequence id");
            len_of[id] = klen;
            ptr_of[id] = kp;
        }
    }
    t->seq_off.assign(nl + 1, 0);
    for (uint64_t i = 0; i < nl; i++) {
        if (len_of[i] == ~0ull) return fail("reference without sequence");
        t->seq_off[i + 1] = t->seq_off[i] + len_of[i];
    }
    t->seq_bytes.resize(t->seq_off[nl]);
    for (uint64_t i = 0; i < nl; i++) memcpy(t->seq_bytes.data() + t->seq_off[i], ptr_of[i], len_of[i]);
    t->sequences.reserve(ns * 2);
    for (auto &e : entries) {
        const uint32_t first = e.second.empty() ? 0 : e.second[0];
        std::string_view key((const char *)t->seq_bytes.data() + t->seq_off[first], e.first.second);
        if (!e.second.empty()) t->sequences[key] = std::move(e.second);
    }
    // k_mer_map
    const uint64_t nk = r.u64();
    if (!r.ok || nk != RTX_NUM_KMERS) return fail("k_mer_map length");
    t->csr_off.assign(RTX_NUM_KMERS + 1, 0);
    t->postings.resize((size_t)(r.end - r.p) / 4);  // upper bound, trimmed below
    uint64_t np = 0;
    for (uint32_t k = 0; k < RTX_NUM_KMERS; k++) {
        const uint64_t n = r.u64();
        // count first, bytes second: n * 4 wraps for n >= 2^62 and would pass need()
        if (!r.ok || n > (uint64_t)(r.end - r.p) / 4 || !r.need(n * 4)) return fail("posting list");
        memcpy(t->postings.data() + np, r.p, n * 4);
        r.p += n * 4;
        for (uint64_t i = 0; i < n; i++) {
            if (t->postings[np + i] >= nl) return fail("posting id");
            // sorted unique (tree.rs:134-137): the reference-sharded index cuts a list with lower_bound
            if (i && t->postings[np + i] <= t->postings[np + i - 1]) return fail("posting list not strictly ascending");
        }
        np += n;
        t->csr_off[k + 1] = np;
    }
    t->postings.resize(np);
    t->num_tips = r.u64();
    if (!r.ok || r.p != r.end) return fail("trailing bytes");
    if (t->num_tips != nl) return fail("num_tips");
    t->orig_idx.resize(nl);
    for (uint64_t i = 0; i < nl; i++) t->orig_idx[i] = i;  // the input order is not stored in the file
    rtx::flatten_tree(*t);
    if (int rc = rtx::check_tree_depth(*t)) { delete t; return rc; }
    *out = t;
    return RTX_OK;
}

}  // extern "C"
