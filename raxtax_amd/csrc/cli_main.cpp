// raxtax-hip: minimal command line around the host mirror of raxtax() -- just enough to run the
// reference's plumbing configuration (FASTA database + FASTA queries in, `.out`/`.tsv` lines out) on
// one GPU.  Flag names follow src/io.rs:112-154; checkpointing, logging, `.bin` caching, gzip input
// and thread options are out of scope.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "host_raxtax.hpp"
#include "raxtax_hip.h"

static bool slurp(const std::string &path, std::string &out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::ostringstream ss;
    ss << f.rdbuf();
    out = ss.str();
    return true;
}

int main(int argc, char **argv) {
    std::string db, qf, prefix = "raxtax";
    bool skip_exact = false, raw = false, tsv = false;
    int device = 0;
    size_t chunk = 0;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto val = [&]() -> const char * { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "-d" || a == "--database-path") db = val();
        else if (a == "-i" || a == "--query-file") qf = val();
        else if (a == "-o" || a == "--prefix") prefix = val();
        else if (a == "--skip-exact-matches") skip_exact = true;
        else if (a == "--raw-confidence") raw = true;
        else if (a == "--tsv") tsv = true;
        else if (a == "--device") device = atoi(val());
        else if (a == "--batch") chunk = (size_t)atoll(val());
        else { fprintf(stderr, "usage: raxtax-hip -d DB.fasta -i QUERIES.fasta [-o PREFIX] [--skip-exact-matches] [--raw-confidence] [--tsv] [--device N] [--batch N]\n"); return 64; }
    }
    if (db.empty() || qf.empty()) { fprintf(stderr, "raxtax-hip: -d and -i are required\n"); return 64; }
    std::string db_text, q_text;
    if (!slurp(db, db_text)) { fprintf(stderr, "[ERROR] Failed to parse %s\n", db.c_str()); return 66; }   // exitcode::NOINPUT
    if (!slurp(qf, q_text)) { fprintf(stderr, "[ERROR] Failed to parse %s\n", qf.c_str()); return 66; }
    rtx_tree *tree = nullptr;
    if (rtx_tree_parse_reference_fasta(db_text.data(), db_text.size(), &tree) != RTX_OK) {
        fprintf(stderr, "[ERROR] Failed to parse %s: %s\n", db.c_str(), rtx_last_error());
        return 66;
    }
    rtx_queries *qs = nullptr;
    if (rtx_queries_parse_fasta(q_text.data(), q_text.size(), nullptr, 0, &qs) != RTX_OK) {
        fprintf(stderr, "[ERROR] Failed to parse %s: %s\n", qf.c_str(), rtx_last_error());
        return 66;
    }
    rtx_index *index = nullptr;
    if (rtx_index_create_from_tree(device, tree, &index) != RTX_OK) {
        fprintf(stderr, "[ERROR] %s\n", rtx_last_error());
        return 71;  // exitcode::OSERR
    }
    std::ofstream out(prefix + ".out"), tsv_out;
    if (tsv) tsv_out.open(prefix + ".tsv");
    struct Ctx { std::ofstream *out, *tsv; } ctx{&out, tsv ? &tsv_out : nullptr};
    const uint64_t n = rtx_queries_len(qs);
    std::vector<const char *> labels(n);
    for (uint64_t i = 0; i < n; i++) labels[i] = rtx_queries_label(qs, i);
    const uint8_t *bases;
    const uint64_t *off;
    rtx_queries_data(qs, &bases, &off);
    auto sender = [](void *c, const char *, const char *lines, const char *tsv_lines) -> int {
        Ctx *x = static_cast<Ctx *>(c);
        (*x->out) << lines << '\n';                       // writeln!(output, ...), main.rs:132
        if (x->tsv && tsv_lines) (*x->tsv) << tsv_lines << '\n';
        return x->out->good() ? 0 : 1;
    };
    const int rc = rtx_raxtax(index, tree, n, labels.data(), bases, off, skip_exact, raw, chunk, sender, &ctx, tsv);
    if (rc != RTX_OK) {
        fprintf(stderr, "[ERROR] %s\n", rtx_last_error());
        return rc == RTX_ERR_SENDER ? 75 : 70;  // exitcode::TEMPFAIL / SOFTWARE
    }
    rtx_index_destroy(index);
    rtx_queries_destroy(qs);
    rtx_tree_destroy(tree);
    return 0;
}
