// Host mirror of raxtax() (src/raxtax.rs:14-97): exact-match lookup, device classification in
// batches, single-exact-match override, formatting, one message per query to the sender.
#include "host_raxtax.hpp"

#include <cstdio>
#include <cstring>

#include "rtx_internal.hpp"

namespace {

int run(rtx_index *index, const rtx_tree *tree, uint64_t n_queries, const char *const *labels, const uint8_t *bases,
        const uint64_t *base_off, bool skip_exact_matches, bool raw_confidence, uint64_t chunk_size,
        const raxtax::Sender &sender, bool tsv) {
    if (!index || !tree || !base_off || !labels) { rtx::set_error("rtx_raxtax: null argument"); return RTX_ERR_INVALID; }
    if (rtx_index_num_refs(index) != tree->num_tips) { rtx::set_error("index and tree disagree on num_tips"); return RTX_ERR_INVALID; }
    if (chunk_size == 0 || chunk_size > n_queries) chunk_size = n_queries ? n_queries : 1;
    const uint32_t flags = (skip_exact_matches ? RTX_SKIP_EXACT_MATCHES : 0u) | (raw_confidence ? RTX_RAW_CONFIDENCE : 0u);
    bool warnings = false;
    std::vector<uint32_t> exact_ids;
    std::vector<uint64_t> exact_off;
    std::vector<char> out_buf(1 << 16), tsv_buf(1 << 16);
    for (uint64_t q0 = 0; q0 < n_queries; q0 += chunk_size) {
        const uint64_t nq = std::min<uint64_t>(chunk_size, n_queries - q0);
        exact_ids.clear();
        exact_off.assign(1, 0);
        for (uint64_t q = q0; q < q0 + nq; q++) {
            // let exact_matches = tree.sequences.get(query_sequence), raxtax.rs:42
            const uint32_t *ids = nullptr;
            const uint64_t ne = rtx_tree_exact_matches(tree, bases + base_off[q], base_off[q + 1] - base_off[q], &ids);
            exact_ids.insert(exact_ids.end(), ids, ids + ne);
            exact_off.push_back(exact_ids.size());
            if (!skip_exact_matches && ne > 1) {  // raxtax.rs:43-53 (the info! lines go to the log in the CLI)
                auto parent = [&](uint32_t id) {
                    const std::string &l = tree->lineages[id];
                    const size_t c = l.rfind(',');
                    return c == std::string::npos ? std::string_view() : std::string_view(l).substr(0, c);
                };
                for (uint64_t i = 1; i < ne; i++)
                    if (parent(ids[i]) != parent(ids[0])) {
                        fprintf(stderr, "[WARN ] Exact matches for %s differ above the leafs of the lineage tree!\n", labels[q]);
                        warnings = true;
                        break;
                    }
            }
        }
        rtx_result_view res;
        int rc = rtx_classify_batch(index, nq, bases, base_off + q0, exact_ids.empty() ? nullptr : exact_ids.data(),
                                    exact_off.data(), flags, &res);
        if (rc) return rc;
        for (uint64_t i = 0; i < nq; i++) {
            const uint64_t q = q0 + i;
            if (res.status[i] != RTX_Q_OK) {
                // the reference aborts here (prob.rs:21/162); report and skip the query instead
                fprintf(stderr, "[ERROR] query %s has too few valid 8-mers (%u) to be classified\n", labels[q], res.t[i]);
                continue;
            }
            const uint64_t len = base_off[q + 1] - base_off[q];
            const uint64_t rows = res.row_off[i + 1] - res.row_off[i];
            const size_t need = (rows + 1) * (strlen(labels[q]) + 4096 + 8 * RTX_MAX_DEPTH) + len + 64;
            if (out_buf.size() < need) out_buf.resize(need);
            if (tsv && tsv_buf.size() < need + rows * len) tsv_buf.resize(need + rows * len);
            int64_t tsv_len = 0;
            const int64_t n = rtx_format_query(tree, &res, i, labels[q], bases + base_off[q], len,
                                               exact_ids.data() + exact_off[i], exact_off[i + 1] - exact_off[i], flags,
                                               out_buf.data(), out_buf.size(), tsv ? tsv_buf.data() : nullptr,
                                               tsv_buf.size(), &tsv_len);
            if (n < 0) return (int)n;
            std::optional<std::string> tsv_msg;
            if (tsv) tsv_msg.emplace(tsv_buf.data(), (size_t)tsv_len);
            if (!sender(labels[q], std::string(out_buf.data(), (size_t)n), std::move(tsv_msg))) {
                rtx::set_error("result sink closed");  // sender.send(..)?, raxtax.rs:87
                return RTX_ERR_SENDER;
            }
        }
    }
    if (warnings)  // raxtax.rs:93-95
        fprintf(stderr, "\x1b[33m[WARN ]\x1b[0m Exact matches for some queries differ above the species level! Check the log file for more information!\n");
    return RTX_OK;
}

}  // namespace

namespace raxtax {

int raxtax(const std::vector<std::pair<std::string, std::vector<uint8_t>>> &queries, const rtx_tree *tree,
           rtx_index *index, bool skip_exact_matches, bool raw_confidence, size_t chunk_size, const Sender &sender,
           bool tsv) {
    std::vector<const char *> labels(queries.size());
    std::vector<uint8_t> bases;
    std::vector<uint64_t> off(queries.size() + 1, 0);
    for (size_t i = 0; i < queries.size(); i++) {
        labels[i] = queries[i].first.c_str();
        bases.insert(bases.end(), queries[i].second.begin(), queries[i].second.end());
        off[i + 1] = bases.size();
    }
    return run(index, tree, queries.size(), labels.data(), bases.data(), off.data(), skip_exact_matches, raw_confidence,
               chunk_size, sender, tsv);
}

}  // namespace raxtax

extern "C" int rtx_raxtax(rtx_index *index, const rtx_tree *tree, uint64_t n_queries, const char *const *labels,
                          const uint8_t *bases, const uint64_t *base_off, int skip_exact_matches, int raw_confidence,
                          uint64_t chunk_size, rtx_sender_fn sender, void *sender_ctx, int tsv) {
    if (!sender) { rtx::set_error("rtx_raxtax: null sender"); return RTX_ERR_INVALID; }
    raxtax::Sender s = [&](const std::string &label, std::string &&out, std::optional<std::string> &&t) {
        return sender(sender_ctx, label.c_str(), out.c_str(), t ? t->c_str() : nullptr) == 0;
    };
    return run(index, tree, n_queries, labels, bases, base_off, skip_exact_matches != 0, raw_confidence != 0, chunk_size, s,
               tsv != 0);
}
