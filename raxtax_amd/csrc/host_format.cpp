// Host mirror of the result formatting and of the single-exact-match override:
//   EvaluationResult::get_output_string  src/lineage.rs:17-29
//   EvaluationResult::get_tsv_string     src/lineage.rs:31-48
//   utils::get_results / get_results_tsv src/utils.rs:62-68,83-89
//   utils::decompress_sequence           src/utils.rs:70-81
//   exact-match override                 src/raxtax.rs:73-84
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>

#include "rtx_internal.hpp"

namespace {

struct Out {
    char *buf;
    uint64_t cap, len = 0;
    bool ok = true;
    // room for n more bytes (+ the terminator)?  The formatters ask once per piece that can be long (a label, a lineage) and once per row
    // for everything of bounded size -- the small pieces are then written without a check and without a call (a libc memcpy per tab,
    // comma and number was most of the 0.4 us a query cost)
    bool room(size_t n) {
        if (ok && len + n + 1 > cap) ok = false;
        return ok;
    }
    void put(const char *s, size_t n) {  // (the text is terminated once, by finish())
        if (!room(n)) return;
        memcpy(buf + len, s, n);
        len += n;
    }
    void finish() { if (ok && buf && len < cap) buf[len] = 0; }
    void put(const std::string &s) { put(s.data(), s.size()); }
    void putc(char c) { if (room(1)) buf[len++] = c; }
    void putc_nc(char c) { buf[len++] = c; }  // (no check: inside a reservation)
    void put4_nc(const char *t) { memcpy(buf + len, t, 4); len += 4; }
    void putf(const char *fmt, double v) {
        char tmp[64];
        int n = snprintf(tmp, sizeof tmp, fmt, v);
        put(tmp, (size_t)n);
    }
    // "{:.D}" of a non-negative value below 1e6 (Rust's and printf's fixed notation agree: the exact binary value rounded half to even
    // at D decimals) without snprintf -- eight of them per result row were the whole cost of the format stage (1.2 us per query).
    // v * 10^D is formed in double (relative error 2^-53: below 1e-5 absolute here); unless it lands within 1e-4 of a rounding
    // boundary the nearest integer IS the correctly rounded result, else snprintf decides.  At most 24 bytes: callers reserve them.
    template <int D>
    void put_fixed(double v) {
        static_assert(D >= 1 && D <= 5, "decimals");
        constexpr double kPow[6] = {1.0, 10.0, 100.0, 1e3, 1e4, 1e5};
        if (!(v >= 0.0 && v < 1e6) || std::signbit(v)) { putf(D == 2 ? "%.2f" : "%.5f", v); return; }
        const double s = v * kPow[D];
        const double fl = std::floor(s), frac = s - fl;
        if (frac > 0.4999 && frac < 0.5001) { putf(D == 2 ? "%.2f" : "%.5f", v); return; }  // too close to call
        unsigned long long r = (unsigned long long)fl + (frac > 0.5 ? 1ull : 0ull);
        if (!room(24)) return;
        if (D == 2 && r <= 100ull) { put4_nc(kHundredths.t[r]); return; }  // a confidence: "0.00" .. "1.00" from a table (six of them per row)
        char tmp[24];
        int n = 0;
        for (int d = 0; d < D; d++) { tmp[n++] = (char)('0' + r % 10ull); r /= 10ull; }
        tmp[n++] = '.';
        do { tmp[n++] = (char)('0' + r % 10ull); r /= 10ull; } while (r);
        for (int i = 0; i < n; i++) buf[len + i] = tmp[n - 1 - i];
        len += (uint64_t)n;
    }
    struct Hundredths {
        char t[101][4];
        Hundredths() { for (int k = 0; k <= 100; k++) { t[k][0] = (char)('0' + k / 100); t[k][1] = '.'; t[k][2] = (char)('0' + k / 10 % 10); t[k][3] = (char)('0' + k % 10); } }
    };
    static const Hundredths kHundredths;
};
const Out::Hundredths Out::kHundredths;

}  // namespace

extern "C" int64_t rtx_format_query(const rtx_tree *tree, const rtx_result_view *res, uint64_t q, const char *label,
                                    const uint8_t *seq, uint64_t seq_len, const uint32_t *exact_ids, uint64_t n_exact,
                                    uint32_t flags, char *out_buf, uint64_t out_cap, char *tsv_buf, uint64_t tsv_cap,
                                    int64_t *tsv_len) {
    if (!tree || !res || !label || !out_buf || q >= res->n_queries) {
        rtx::set_error("rtx_format_query: invalid argument");
        return RTX_ERR_INVALID;
    }
    uint64_t r0 = res->row_begin[q], r1 = r0 + res->row_count[q];
    if (r1 == r0) {  // assert!(!eval_res.is_empty()), raxtax.rs:72
        rtx::set_error("query %llu has no result rows (status %u)", (unsigned long long)q, res->status[q]);
        return RTX_ERR_INVALID;
    }
    // raxtax.rs:73-84: exactly one exact match -> one row, confidence 1.0 on every level,
    // signals of the former first row
    const bool override_one =
        !(flags & RTX_RAW_CONFIDENCE) && !(flags & RTX_SKIP_EXACT_MATCHES) && exact_ids && n_exact == 1;
    std::string dec;
    if (tsv_buf) {
        dec.resize(seq_len);
        for (uint64_t i = 0; i < seq_len; i++) {
            switch (seq[i]) {
                case 1: dec[i] = 'A'; break;
                case 2: dec[i] = 'C'; break;
                case 4: dec[i] = 'G'; break;
                case 8: dec[i] = 'T'; break;
                default: dec[i] = '-';
            }
        }
    }
    Out o{out_buf, out_cap}, tv{tsv_buf, tsv_buf ? tsv_cap : 0};
    const size_t label_len = strlen(label);
    const uint64_t n_out = override_one ? 1 : r1 - r0;
    for (uint64_t i = 0; i < n_out; i++) {
        const uint64_t r = r0 + i;
        uint32_t lin_idx = res->row_lineage[r];
        uint32_t depth = res->row_depth[r];
        const double *conf = res->row_conf + r * (res->row_conf_stride ? res->row_conf_stride : RTX_MAX_DEPTH);
        double ones[RTX_MAX_DEPTH];
        if (override_one) {
            lin_idx = exact_ids[0];
            const std::string &l = tree->lineages[lin_idx];
            depth = 1;
            for (char c : l) depth += c == ',';
            if (depth > RTX_MAX_DEPTH) { rtx::set_error("lineage deeper than RTX_MAX_DEPTH"); return RTX_ERR_DEPTH; }
            for (uint32_t d = 0; d < depth; d++) ones[d] = 1.0;
            conf = ones;
        }
        const std::string &lineage = tree->lineages[lin_idx];
        const double local = res->row_local_signal[r0 + (override_one ? 0 : i)];
        const double global = res->global_signal[q];
        if (i) o.putc('\n');
        o.put(label, label_len);
        o.putc('\t');
        o.put(lineage);
        o.putc('\t');
        for (uint32_t d = 0; d < depth; d++) {
            if (d) o.putc(',');
            o.put_fixed<2>(conf[d]);
        }
        o.putc('\t');
        o.put_fixed<5>(local);
        o.putc('\t');
        o.put_fixed<5>(global);
        if (tsv_buf) {
            if (i) tv.putc('\n');
            tv.put(label, label_len);
            tv.putc('\t');
            // levels interleaved with confidences; interleave() drains the longer side
            size_t pos = 0;
            uint32_t d = 0;
            bool lin_done = false, first = true;
            while (!lin_done || d < depth) {
                if (!lin_done) {
                    size_t c = lineage.find(',', pos);
                    if (!first) tv.putc('\t');
                    tv.put(lineage.data() + pos, (c == std::string::npos ? lineage.size() : c) - pos);
                    first = false;
                    if (c == std::string::npos) lin_done = true;
                    else pos = c + 1;
                }
                if (d < depth) {
                    if (!first) tv.putc('\t');
                    tv.put_fixed<2>(conf[d]);
                    first = false;
                    d++;
                }
            }
            tv.putc('\t');
            tv.put_fixed<5>(local);
            tv.putc('\t');
            tv.put_fixed<5>(global);
            tv.putc('\t');
            tv.put(dec);
        }
    }
    o.finish();
    if (tsv_buf) tv.finish();
    if (!o.ok || (tsv_buf && !tv.ok)) { rtx::set_error("rtx_format_query: output buffer too small"); return RTX_ERR_INVALID; }
    if (tsv_len) *tsv_len = tsv_buf ? (int64_t)tv.len : 0;
    return (int64_t)o.len;
}

// ---- compact byte record of a result view (the unit the multi-GPU gather ships, raxtax_amd/dist_util.py) ----
//   int64[4] n_queries, n_rows, L (confidence levels per row = deepest row of the view), version (2)
//   | int64 begin[nq] | f64 global[nq] | u32 count[nq] | u32 t[nq] | u8 status[nq]
//   | u32 lineage[n_rows] | u8 depth[n_rows] | u8 conf[n_rows][L] (hundredths) | f64 local[n_rows]
#include <cmath>
#include <thread>
#include <vector>

extern "C" int64_t rtx_result_pack(const rtx_result_view *res, uint8_t *buf, uint64_t cap) {
    if (!res) { rtx::set_error("rtx_result_pack: null view"); return RTX_ERR_INVALID; }
    const uint64_t nq = res->n_queries, nr = res->n_rows;
    // A view of rtx_batch_download carries the row fields as the record wants them (depths and hundredths as bytes, written by
    // finalise_kernel): the record is then a handful of block copies, L = the stride of the view (the deepest lineage of the tree).
    // A hand-made view (no byte arrays): L = the deepest row, the confidences are converted.
    const bool ready = res->row_depth_u8 && res->row_conf_hundredths && res->row_conf_stride;
    const uint32_t stride = res->row_conf_stride ? res->row_conf_stride : RTX_MAX_DEPTH;
    uint32_t L = 1;  // every level of every row travels: the block is as wide as the deepest row
    if (ready) L = stride;
    else for (uint64_t r = 0; r < nr; r++) L = std::max(L, res->row_depth[r]);
    if (L > RTX_MAX_DEPTH || L > stride) { rtx::set_error("rtx_result_pack: row of depth %u (RTX_MAX_DEPTH %u, stride %u)", L, RTX_MAX_DEPTH, stride); return RTX_ERR_DEPTH; }
    const uint64_t need = 32 + 25 * nq + (13 + (uint64_t)L) * nr;
    if (!buf) return (int64_t)need;  // size query
    if (cap < need) { rtx::set_error("rtx_result_pack: buffer of %llu bytes, need %llu", (unsigned long long)cap, (unsigned long long)need); return RTX_ERR_INVALID; }
    int64_t *hdr = reinterpret_cast<int64_t *>(buf);
    hdr[0] = (int64_t)nq;
    hdr[1] = (int64_t)nr;
    hdr[2] = (int64_t)L;
    hdr[3] = 2;
    uint8_t *p_begin = buf + 32, *p_gs = p_begin + 8 * nq, *p_count = p_gs + 8 * nq, *p_t = p_count + 4 * nq, *p_status = p_t + 4 * nq;
    uint8_t *p_lin = p_status + nq, *p_depth = p_lin + 4 * nr, *p_conf = p_depth + nr, *p_local = p_conf + (uint64_t)L * nr;
    auto work = [&](uint64_t qa, uint64_t qb, uint64_t ra, uint64_t rb) {
        // (row_begin is u64 in the view and i64 in the record: the same bytes)
        memcpy(p_begin + 8 * qa, res->row_begin + qa, 8 * (qb - qa));
        memcpy(p_gs + 8 * qa, res->global_signal + qa, 8 * (qb - qa));
        memcpy(p_count + 4 * qa, res->row_count + qa, 4 * (qb - qa));
        memcpy(p_t + 4 * qa, res->t + qa, 4 * (qb - qa));
        memcpy(p_status + qa, res->status + qa, qb - qa);
        memcpy(p_lin + 4 * ra, res->row_lineage + ra, 4 * (rb - ra));
        memcpy(p_local + 8 * ra, res->row_local_signal + ra, 8 * (rb - ra));
        if (ready) {
            memcpy(p_depth + ra, res->row_depth_u8 + ra, rb - ra);
            memcpy(p_conf + (uint64_t)L * ra, res->row_conf_hundredths + (uint64_t)L * ra, (uint64_t)L * (rb - ra));
        } else {
            for (uint64_t r = ra; r < rb; r++) {
                p_depth[r] = (uint8_t)res->row_depth[r];
                const double *cf = res->row_conf + r * stride;
                for (uint32_t d = 0; d < L; d++) p_conf[(uint64_t)L * r + d] = (uint8_t)std::lrint(cf[d] * 100.0);
            }
        }
    };
    const unsigned nt = nr + nq < 65536 ? 1u : rtx::host_threads(8u);
    if (nt == 1) {
        work(0, nq, 0, nr);
    } else {
        std::vector<std::thread> th;
        for (unsigned i = 0; i < nt; i++) th.emplace_back(work, nq * i / nt, nq * (i + 1) / nt, nr * i / nt, nr * (i + 1) / nt);
        for (auto &t : th) t.join();
    }
    return (int64_t)need;
}

// ---- the writer's side of the multi-GPU gather: `.out` lines straight from packed records ----
// What rank 0 does with the buffers of every rank (BASELINE configs[3]): one text per query, in the buffer's query order, formatted by
// `threads` threads into arenas of their own and laid out back to back in `out`.  Same text as rtx_format_query gives for the view the
// records were packed from (confidences travel as hundredths: "0.57" comes from a table), the single-exact-match override of
// raxtax.rs:73-84 from `exact_one` (the id of the query's only exact match, 0xFFFFFFFF: none or several).  The rows of a buffer lie in the
// processing order of the rank that classified them: a query's row is a cache miss -- the loop prefetches rows, lineage objects and their
// characters three stages ahead (as the format stage of rtx_raxtax does).
extern "C" int64_t rtx_records_format(const rtx_tree *tree, const uint8_t *records, uint64_t n_bytes, const char *const *labels,
                                      const uint32_t *exact_one, uint32_t flags, char *out, uint64_t cap, uint64_t *line_off, uint32_t threads) {
    if (!tree || !records || !labels || n_bytes < 32) { rtx::set_error("rtx_records_format: invalid argument"); return RTX_ERR_INVALID; }
    int64_t hdr[4];
    memcpy(hdr, records, 32);
    const uint64_t nq = (uint64_t)hdr[0], nr = (uint64_t)hdr[1], L = (uint64_t)hdr[2];
    if (hdr[3] != 2 || L == 0 || L > RTX_MAX_DEPTH || n_bytes < 32 + 25 * nq + (13 + L) * nr) { rtx::set_error("rtx_records_format: not a record buffer of version 2"); return RTX_ERR_INVALID; }
    const uint8_t *p_begin = records + 32, *p_gs = p_begin + 8 * nq, *p_count = p_gs + 8 * nq, *p_status = p_count + 8 * nq;
    const uint8_t *p_lin = p_status + nq, *p_depth = p_lin + 4 * nr, *p_conf = p_depth + nr, *p_local = p_conf + L * nr;
    auto rd64 = [](const uint8_t *p) { int64_t v; memcpy(&v, p, 8); return v; };
    auto rd32 = [](const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; };
    auto rdf = [](const uint8_t *p) { double v; memcpy(&v, p, 8); return v; };
    const bool may_override = exact_one && !(flags & RTX_RAW_CONFIDENCE) && !(flags & RTX_SKIP_EXACT_MATCHES);
    const unsigned nt = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(threads ? threads : rtx::host_threads(16u), (nq + 1023) / 1024));
    struct Part { std::vector<char> text; std::vector<uint64_t> len; int rc = 0; };
    std::vector<Part> parts(nt);
    auto work = [&](unsigned w) {
        const uint64_t a = nq * w / nt, b = nq * (w + 1) / nt;
        Part &pt = parts[w];
        pt.len.assign(b - a, 0);
        size_t capw = (b - a) * 128 + 4096, used = 0;
        pt.text.resize(capw);
        auto row0 = [&](uint64_t q) { return (uint64_t)rd64(p_begin + 8 * q); };
        auto live = [&](uint64_t q) { return q < b && p_status[q] == RTX_Q_OK && rd32(p_count + 4 * q) != 0; };
        auto st1 = [&](uint64_t q) { if (live(q)) { const uint64_t r = row0(q); __builtin_prefetch(p_lin + 4 * r); __builtin_prefetch(p_conf + L * r); __builtin_prefetch(p_local + 8 * r); __builtin_prefetch(p_depth + r); } if (q < b) __builtin_prefetch(labels[q]); };
        auto st2 = [&](uint64_t q) { if (live(q)) __builtin_prefetch(&tree->lineages[rd32(p_lin + 4 * row0(q)) < tree->lineages.size() ? rd32(p_lin + 4 * row0(q)) : 0]); };
        auto st3 = [&](uint64_t q) { if (live(q)) { const uint32_t li = rd32(p_lin + 4 * row0(q)); if (li < tree->lineages.size()) { const std::string &l = tree->lineages[li]; __builtin_prefetch(l.data()); __builtin_prefetch(l.data() + 64); } } };
        for (uint64_t q = a; q < std::min(b, a + 24); q++) st1(q);
        for (uint64_t q = a; q < std::min(b, a + 16); q++) st2(q);
        for (uint64_t q = a; q < std::min(b, a + 8); q++) st3(q);
        for (uint64_t q = a; q < b; q++) {
            st1(q + 24); st2(q + 16); st3(q + 8);
            if (p_status[q] != RTX_Q_OK) continue;
            const uint64_t r0 = row0(q), cnt = rd32(p_count + 4 * q);
            if (cnt == 0 || r0 + cnt > nr) { pt.rc = RTX_ERR_INVALID; return; }
            const bool one = may_override && exact_one[q] != 0xFFFFFFFFu;
            const uint64_t n_out = one ? 1 : cnt;
            const size_t llen = strlen(labels[q]);
            size_t need = 64;
            for (uint64_t i = 0; i < n_out; i++) {
                const uint32_t li = one ? exact_one[q] : rd32(p_lin + 4 * (r0 + i));
                if (li >= tree->lineages.size()) { pt.rc = RTX_ERR_INVALID; return; }
                need += llen + tree->lineages[li].size() + 5 * RTX_MAX_DEPTH + 64;
            }
            if (used + need > capw) { capw = (used + need) * 3 / 2; pt.text.resize(capw); }
            Out o{pt.text.data() + used, capw - used};
            const double global = rdf(p_gs + 8 * q);
            for (uint64_t i = 0; i < n_out; i++) {
                const uint64_t r = r0 + i;
                const uint32_t li = one ? exact_one[q] : rd32(p_lin + 4 * r);
                const std::string &lineage = tree->lineages[li];
                uint32_t depth = p_depth[r];
                if (one) { depth = 1; for (char c : lineage) depth += c == ','; if (depth > RTX_MAX_DEPTH) { pt.rc = RTX_ERR_DEPTH; return; } }
                if (!o.room(llen + lineage.size() + 5 * (size_t)depth + 64)) break;  // everything of this row
                if (i) o.putc_nc('\n');
                memcpy(o.buf + o.len, labels[q], llen);
                o.len += llen;
                o.putc_nc('\t');
                memcpy(o.buf + o.len, lineage.data(), lineage.size());
                o.len += lineage.size();
                o.putc_nc('\t');
                for (uint32_t d = 0; d < depth; d++) {
                    if (d) o.putc_nc(',');
                    const uint32_t h = one ? 100u : (d < L ? p_conf[L * r + d] : 0u);
                    o.put4_nc(Out::kHundredths.t[h <= 100u ? h : 100u]);
                }
                o.putc_nc('\t');
                o.put_fixed<5>(rdf(p_local + 8 * (r0 + (one ? 0 : i))));
                o.putc('\t');
                o.put_fixed<5>(global);
            }
            o.finish();
            if (!o.ok) { pt.rc = RTX_ERR_INVALID; return; }
            pt.len[q - a] = o.len + 1;  // with its NUL
            used += o.len + 1;
        }
        pt.text.resize(used);
    };
    if (nt == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (unsigned w = 0; w < nt; w++) th.emplace_back(work, w);
        for (auto &t : th) t.join();
    }
    uint64_t total = 0;
    for (auto &pt : parts) { if (pt.rc) { rtx::set_error("rtx_records_format: malformed records"); return pt.rc; } total += pt.text.size(); }
    if (!out) return (int64_t)total;  // size query (the text is formatted to be measured: call once with a generous buffer instead where that matters)
    if (cap < total) {  // the caller learns what it needs from this call: -(bytes needed) - RTX_NEED_BASE (no second pass to measure: ADVICE r5)
        rtx::set_error("rtx_records_format: buffer of %llu bytes, need %llu", (unsigned long long)cap, (unsigned long long)total);
        return -(int64_t)total - (int64_t)RTX_NEED_BASE;
    }
    uint64_t at = 0;
    std::vector<uint64_t> base(nt);
    for (unsigned w = 0; w < nt; w++) { base[w] = at; at += parts[w].text.size(); }
    auto place = [&](unsigned w) {
        memcpy(out + base[w], parts[w].text.data(), parts[w].text.size());
        if (line_off) {
            const uint64_t a = nq * w / nt;
            uint64_t o = base[w];
            for (size_t k = 0; k < parts[w].len.size(); k++) { line_off[a + k] = o; o += parts[w].len[k]; }
        }
    };
    if (nt == 1) place(0);
    else {
        std::vector<std::thread> th;
        for (unsigned w = 0; w < nt; w++) th.emplace_back(place, w);
        for (auto &t : th) t.join();
    }
    if (line_off) line_off[nq] = total;
    return (int64_t)total;
}
