// raxtax-hip: command line around the host mirror of raxtax() -- FASTA/.bin database + FASTA queries in,
// the reference's output files out, on one GPU or, with --gpus N / --devices a,b,..., on several at once (one process, one
// index handle and one driving thread per device: rtx_raxtax_multi; the reference's `-t` of rayon threads, main.rs:40-57).  Flag names and file semantics follow the reference
// (src/io.rs:112-154 Args, :202-263 get_output, :47-90 Checkpoint, :156-187 check_incomplete_output;
// src/main.rs:72-99 database caching, :126-136 writer):
//   PREFIX/raxtax.out   one line per result row            PREFIX/raxtax.tsv  (--tsv)
//   PREFIX/raxtax.ckp   one finished query label per line  PREFIX/raxtax.json checkpoint (flags + DB fingerprint)
//   PREFIX/<db>.bin     bincode database cache (unless --skip-db)
// A rerun with the same flags and database resumes: labels listed in raxtax.ckp are skipped
// (parser.rs:150-153) and half-written result lines of unlisted queries are purged first.
// Inputs ending in .gz / .gzip are decompressed on the fly (utils.rs:42-60 get_reader: the extension decides).
// Out of scope: raxtax.log, progress bars, thread options.
#include <sys/stat.h>
#include <zlib.h>

#include <cctype>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <set>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "host_raxtax.hpp"
#include "raxtax_hip.h"

namespace {

// utils::get_reader (utils.rs:42-60): the file's last extension, lower-cased, "gz" or "gzip" = a GzDecoder in front of the reader
bool is_gz(const std::string &path) {
    const size_t slash = path.find_last_of('/'), dot = path.find_last_of('.');
    if (dot == std::string::npos || (slash != std::string::npos && dot < slash)) return false;
    std::string ext = path.substr(dot + 1);
    for (char &c : ext) c = (char)tolower((unsigned char)c);
    return ext == "gz" || ext == "gzip";
}

// A file read in pieces, plain or gzip-compressed.
struct Input {
    FILE *f = nullptr;
    gzFile g = nullptr;
    bool open(const std::string &path) {
        if (is_gz(path)) {
            g = gzopen(path.c_str(), "rb");
            if (g) gzbuffer(g, 1 << 20);
            return g != nullptr;
        }
        f = fopen(path.c_str(), "rb");
        return f != nullptr;
    }
    // up to n bytes; fewer only at the end of the data; (size_t)-1 on a read / decompression error
    size_t read(char *buf, size_t n) {
        if (f) return fread(buf, 1, n, f);
        size_t got = 0;
        while (got < n) {
            const int k = gzread(g, buf + got, (unsigned)std::min<size_t>(n - got, 1u << 30));
            if (k < 0) return (size_t)-1;
            if (k == 0) break;
            got += (size_t)k;
        }
        if (got < n) {  // the end of the data -- or of a truncated / corrupt stream: zlib hands out what it decoded and keeps the error
            int err = Z_OK;
            (void)gzerror(g, &err);
            if (err != Z_OK && err != Z_STREAM_END) return (size_t)-1;
        }
        return got;
    }
    void close() {
        if (f) fclose(f);
        if (g) gzclose(g);
        f = nullptr;
        g = nullptr;
    }
};

bool slurp(const std::string &path, std::string &out) {
    Input in;
    if (!in.open(path)) return false;
    out.clear();
    const size_t piece = (size_t)16 << 20;
    for (;;) {
        const size_t have = out.size();
        out.resize(have + piece);
        const size_t got = in.read(&out[have], piece);
        if (got == (size_t)-1) { in.close(); return false; }
        out.resize(have + got);
        if (got < piece) break;
    }
    in.close();
    return true;
}

bool is_file(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode); }
bool is_dir(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode); }

// FileFingerprint (io.rs:24-45): path, size, mtime in seconds
std::string fingerprint(const std::string &path) {
    struct stat st;
    if (stat(path.c_str(), &st) != 0) return "";
    char *abs = realpath(path.c_str(), nullptr);
    std::ostringstream ss;
    ss << (abs ? abs : path.c_str()) << "|" << (long long)st.st_size << "|" << (long long)st.st_mtime;
    free(abs);
    return ss.str();
}

std::string checkpoint_json(const std::string &fp, bool raw, bool skip, bool tsv) {
    std::ostringstream ss;
    ss << "{\n  \"db_fingerprint\": \"" << fp << "\",\n  \"raw_confidence\": " << (raw ? "true" : "false")
       << ",\n  \"skip_exact_matches\": " << (skip ? "true" : "false") << ",\n  \"tsv\": " << (tsv ? "true" : "false") << "\n}\n";
    return ss.str();
}

// check_incomplete_output (io.rs:156-187): keep only the lines whose first field is a finished query
void purge_incomplete(const std::string &path, const std::set<std::string> &done) {
    std::ifstream in(path);
    if (!in) return;
    std::vector<std::string> keep;
    bool rewrite = false;
    std::string line;
    while (std::getline(in, line)) {
        const size_t tab = line.find('\t');
        if (tab != std::string::npos && done.count(line.substr(0, tab))) keep.push_back(line);
        else rewrite = true;
    }
    in.close();
    if (!rewrite) return;
    const std::string tmp = path + ".tmp";
    {
        std::ofstream out(tmp, std::ios::trunc);
        for (const std::string &l : keep) out << l << '\n';
    }
    rename(tmp.c_str(), path.c_str());
}

struct Sink {
    std::ofstream out, tsv, ckp;
    bool want_tsv = false;
};

}  // namespace

int main(int argc, char **argv) {
    std::string db, qf, prefix = "raxtax";
    bool skip_exact = false, raw = false, tsv = false, only_db = false, skip_db = false, clean = false, redo = false;
    bool timing = false;
    using clk = std::chrono::steady_clock;
    auto t_prev = clk::now();
    std::ostringstream t_log;
    auto lap = [&](const char *what) {  // --timing: seconds per stage on stderr
        const auto now = clk::now();
        t_log << (t_log.tellp() > 0 ? ", " : "") << '"' << what << "\": " << std::chrono::duration<double>(now - t_prev).count();
        t_prev = now;
    };
    std::vector<int> devices{0};
    size_t chunk = 0;  // --batch: queries per chunk of rtx_raxtax; 0 = chosen per block of the query file (below)
    size_t block_bytes = (size_t)256 << 20;  // query file read and parsed in blocks of this size
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto val = [&]() -> const char * { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "-d" || a == "--database-path") db = val();
        else if (a == "-i" || a == "--query-file") qf = val();
        else if (a == "-o" || a == "--prefix") prefix = val();
        else if (a == "--skip-exact-matches") skip_exact = true;
        else if (a == "--raw-confidence") raw = true;
        else if (a == "--tsv") tsv = true;
        else if (a == "--only-db") only_db = true;
        else if (a == "--skip-db") skip_db = true;
        else if (a == "-c" || a == "--clean") clean = true;
        else if (a == "--redo") redo = true;
        else if (a == "--timing") timing = true;
        // CPU tuning flags of the reference (io.rs:139-153): accepted so that existing command lines keep working,
        // without effect on the device path.  -t takes a value; clap also accepts -tN, --threads=N, -vv, -qq.
        else if (a == "-t" || a == "--threads") (void)val();
        else if (a.rfind("--threads=", 0) == 0 || (a.size() > 2 && a[0] == '-' && a[1] == 't' && isdigit((unsigned char)a[2]))) {}
        else if (a == "--pin") {}
        else if (a == "--verbose" || a == "--quiet" || (a.size() >= 2 && a[0] == '-' && a[1] != '-' &&
                                                        a.find_first_not_of(a[1] == 'v' ? "v" : "q", 1) == std::string::npos &&
                                                        (a[1] == 'v' || a[1] == 'q'))) {}
        else if (a == "--device") devices.assign(1, atoi(val()));
        else if (a == "--gpus") {  // devices 0 .. N-1
            const int n = atoi(val());
            devices.clear();
            for (int d = 0; d < n; d++) devices.push_back(d);
        } else if (a == "--devices") {  // an explicit list; a device may be named twice (two handles on one GPU)
            devices.clear();
            std::stringstream ss(val());
            std::string tok;
            while (std::getline(ss, tok, ',')) if (!tok.empty()) devices.push_back(atoi(tok.c_str()));
        }
        else if (a == "--batch") chunk = (size_t)atoll(val());
        else if (a == "--block-bytes") block_bytes = std::max<size_t>(1, (size_t)atoll(val()));
        else {
            fprintf(stderr, "usage: raxtax-hip -d DB.(fasta|bin) [-i QUERIES.fasta] [-o PREFIX] [--skip-exact-matches] [--raw-confidence] "
                            "[--tsv] [--only-db] [--skip-db] [-c] [--redo] [--device N | --gpus N | --devices a,b,..] [--batch N] [--block-bytes N]\n"
                            "       (-t/--threads N, --pin, -v, -q of the reference are accepted and ignored)\n");
            return 64;
        }
    }
    if (devices.empty()) { fprintf(stderr, "raxtax-hip: --gpus / --devices need at least one device\n"); return 64; }
    if (db.empty() || (qf.empty() && !only_db) || (only_db && skip_db)) {
        fprintf(stderr, "raxtax-hip: -d is required, -i unless --only-db; --only-db conflicts with --skip-db\n");
        return 64;
    }
    const std::string ckp_json = prefix + "/raxtax.json", ckp_path = prefix + "/raxtax.ckp";
    const std::string out_path = prefix + "/raxtax.out", tsv_path = prefix + "/raxtax.tsv";
    // ---- checkpoint (io.rs:202-263)
    std::set<std::string> done;
    const std::string want_ckp = checkpoint_json(fingerprint(db), raw, skip_exact, tsv);
    bool resume = false;
    if (!redo && is_file(ckp_json)) {
        std::string have;
        slurp(ckp_json, have);
        if (have == want_ckp) {  // checkpoint_valid (io.rs:288-302)
            std::ifstream p(ckp_path);
            std::string l;
            while (std::getline(p, l)) done.insert(l);
            purge_incomplete(out_path, done);
            if (tsv) purge_incomplete(tsv_path, done);
            resume = true;
            fprintf(stderr, "[INFO ] Restarting from checkpoint %s\n", ckp_json.c_str());
        }
    }
    if (is_dir(prefix) && !is_file(ckp_json) && !redo) {
        fprintf(stderr, "[ERROR] Output folder %s already exists! Please specify another folder with -o <PATH> or run with --redo "
                        "to force overriding existing files!\n", prefix.c_str());
        return 73;  // exitcode::CANTCREAT
    }
    mkdir(prefix.c_str(), 0777);

    // ---- database: try the binary format first, then FASTA (parser.rs:37-44)
    rtx_tree *tree = nullptr;
    bool store_db = false;
    if (rtx_tree_load_bin(db.c_str(), &tree) != RTX_OK) {
        std::string db_text;
        // Tree.k_mer_map is only needed for the .bin cache: the device index is built from the sequences
        if (!slurp(db, db_text) || rtx_tree_parse_reference_fasta_ex(db_text.data(), db_text.size(), RTX_TREE_SKIP_KMER_MAP, &tree) != RTX_OK) {
            fprintf(stderr, "[ERROR] Failed to parse %s: %s\n", db.c_str(), rtx_last_error());
            return 66;  // exitcode::NOINPUT
        }
        store_db = true;
    }
    lap("database");
    std::string db_bin;
    if (store_db && !skip_db) {  // main.rs:72-86, io.rs:269-286
        std::string base = db.substr(db.find_last_of('/') == std::string::npos ? 0 : db.find_last_of('/') + 1);
        const size_t dot = base.find_last_of('.');
        db_bin = prefix + "/" + (dot == std::string::npos ? base : base.substr(0, dot)) + ".bin";
        if (is_file(db_bin) && !redo && !resume) {
            fprintf(stderr, "[ERROR] Output database file %s already exists! Delete it or run with --redo\n", db_bin.c_str());
            return 73;
        }
    }
    // the database cache is written on a thread of its own while the queries are parsed and classified
    std::thread bin_writer;
    int bin_rc = RTX_OK;
    std::string bin_err;
    auto start_bin_writer = [&]() {
        if (db_bin.empty()) return;
        bin_writer = std::thread([&]() {
            bin_rc = rtx_tree_save_bin(tree, db_bin.c_str());
            if (bin_rc != RTX_OK) bin_err = rtx_last_error();
        });
    };
    auto join_bin_writer = [&]() -> bool {
        if (bin_writer.joinable()) bin_writer.join();
        if (bin_rc != RTX_OK) { fprintf(stderr, "[ERROR] Failed to write database: %s\n", bin_err.c_str()); return false; }
        return true;
    };
    {
        const std::string tmp = ckp_json + ".tmp";  // Checkpoint::save: tmp + rename (io.rs:72-78)
        std::ofstream f(tmp, std::ios::trunc);
        f << (resume ? want_ckp : checkpoint_json(fingerprint(db), raw, skip_exact, tsv));
        f.close();
        rename(tmp.c_str(), ckp_json.c_str());
    }
    if (only_db) {
        start_bin_writer();
        const bool ok = join_bin_writer();
        lap("database_cache");
        if (timing) fprintf(stderr, "{%s}\n", t_log.str().c_str());
        return ok ? 0 : 74;
    }

    // ---- queries: the file is read and parsed block by block on a thread of its own (cut in front of header lines,
    // rtx_fasta_block_end), so that ingest overlaps with classification and memory stays bounded for very large
    // files (the reference reads the whole file, parser.rs:112-115).  Already finished labels are dropped
    // (parser.rs:150-153).
    std::vector<const char *> skip;
    for (const std::string &l : done) skip.push_back(l.c_str());
    struct Parsed { rtx_queries *qs = nullptr; int rc = RTX_OK; std::string err; bool end = false; };
    std::mutex qmu;
    std::condition_variable qcv;
    std::deque<Parsed> ready;  // at most two blocks ahead
    bool stop_reader = false;
    std::thread reader([&]() {
        auto push = [&](Parsed &&pz) {
            std::unique_lock<std::mutex> g(qmu);
            qcv.wait(g, [&] { return ready.size() < 2 || stop_reader; });
            ready.push_back(std::move(pz));
            qcv.notify_all();
        };
        Input f;
        if (!f.open(qf)) { Parsed e; e.rc = RTX_ERR_PARSE; e.err = "cannot open file"; e.end = true; push(std::move(e)); return; }
        std::string buf;
        bool first = true, eof = false;
        while (!eof) {
            const size_t have = buf.size();
            buf.resize(have + block_bytes);
            const size_t got = f.read(&buf[have], block_bytes);
            if (got == (size_t)-1) { Parsed e; e.rc = RTX_ERR_PARSE; e.err = "read error (corrupt gzip stream?)"; e.end = true; push(std::move(e)); break; }
            buf.resize(have + got);
            eof = got < block_bytes;
            uint64_t end = buf.size();
            uint32_t flags = first ? 0u : RTX_FASTA_NOT_FIRST;
            if (!eof) {
                end = rtx_fasta_block_end(buf.data(), buf.size());
                if (end == 0) continue;  // no header inside the block yet: read on
                flags |= RTX_FASTA_MORE_FOLLOWS;
            }
            Parsed pz;
            pz.rc = rtx_queries_parse_fasta_block(buf.data(), end, skip.empty() ? nullptr : skip.data(), skip.size(), flags, &pz.qs);
            if (pz.rc != RTX_OK) pz.err = rtx_last_error();
            pz.end = eof || pz.rc != RTX_OK;
            const bool failed = pz.rc != RTX_OK;
            push(std::move(pz));
            if (failed) break;
            buf.erase(0, end);
            first = false;
            {
                std::lock_guard<std::mutex> g(qmu);
                if (stop_reader) break;
            }
        }
        f.close();
    });
    auto stop_and_join_reader = [&]() {
        {
            std::lock_guard<std::mutex> g(qmu);
            stop_reader = true;
            qcv.notify_all();
        }
        reader.join();
        for (Parsed &pz : ready) rtx_queries_destroy(pz.qs);
    };
    // one index handle per device, created side by side (each on a thread of its own: the builds run on their GPUs)
    std::vector<rtx_index *> indices(devices.size(), nullptr);
    {
        std::vector<int> rcs(devices.size(), RTX_OK);
        std::vector<std::string> errs(devices.size());
        std::vector<std::thread> th;
        for (size_t k = 0; k < devices.size(); k++)
            th.emplace_back([&, k] {
                rcs[k] = rtx_index_create_from_tree(devices[k], tree, &indices[k]);
                if (rcs[k] != RTX_OK) errs[k] = rtx_last_error();
            });
        for (auto &t : th) t.join();
        for (size_t k = 0; k < devices.size(); k++)
            if (rcs[k] != RTX_OK) {
                fprintf(stderr, "[ERROR] device %d: %s\n", devices[k], errs[k].c_str());
                stop_and_join_reader();
                return 71;  // exitcode::OSERR
            }
    }
    lap("index");
    if (timing) {  // the handle's own verdict on tile pruning (rtx_index_self_sample: a property of the database)
        int on = 1;
        double share = -1.0;
        if (rtx_index_prune_verdict(indices[0], &on, &share) == RTX_OK && share >= 0.0)
            fprintf(stderr, "[TIMING] tile pruning %s: a sample of the database's own references keeps %.1f %% of its tiles live\n", on ? "on" : "off", 100.0 * share);
    }
    start_bin_writer();  // after the index: rtx_index_create_from_tree looks at the tree's k-mer map
    Sink sink;
    const auto mode = (redo || !resume) ? std::ios::trunc : std::ios::app;
    sink.out.open(out_path, mode);
    sink.ckp.open(ckp_path, mode);
    sink.want_tsv = tsv;
    if (tsv) sink.tsv.open(tsv_path, mode);
    // the writer of main.rs:127-135: result lines, then the label into the progress file
    auto sender = [](void *c, const char *label, const char *lines, const char *tsv_lines) -> int {
        Sink *s = static_cast<Sink *>(c);
        if (s->want_tsv && tsv_lines) s->tsv << tsv_lines << '\n';
        s->out << lines << '\n';
        s->ckp << label << '\n';
        return s->out.good() && s->ckp.good() ? 0 : 1;
    };
    int rc = RTX_OK;
    uint64_t n = 0;
    bool parse_failed = false;
    for (;;) {
        Parsed pz;
        {
            std::unique_lock<std::mutex> g(qmu);
            qcv.wait(g, [&] { return !ready.empty(); });
            pz = std::move(ready.front());
            ready.pop_front();
            qcv.notify_all();
        }
        if (pz.rc != RTX_OK) {
            fprintf(stderr, "[ERROR] Failed to parse %s: %s\n", qf.c_str(), pz.err.c_str());
            parse_failed = true;
            break;
        }
        const uint64_t nb = rtx_queries_len(pz.qs);
        if (nb) {
            std::vector<const char *> labels(nb);
            for (uint64_t i = 0; i < nb; i++) labels[i] = rtx_queries_label(pz.qs, i);
            const uint8_t *bases;
            const uint64_t *off;
            rtx_queries_data(pz.qs, &bases, &off);
            // Chunks of 131 072 queries are what a device handles best (two sub-batches of 65 536 on its two streams, the next chunk enqueued ahead:
            // 43 ms per 524 288 queries against 101-113 with chunks of 32 768, the default until round 6); smaller ones when the block would
            // otherwise leave a device without two chunks of its own, never below 32 768.
            const size_t per_dev = (size_t)((nb + 2 * indices.size() - 1) / (2 * indices.size()));
            const size_t chunk_now = chunk ? chunk : std::min<size_t>(131072, std::max<size_t>(32768, per_dev));
            rc = rtx_raxtax_multi(indices.data(), (uint32_t)indices.size(), tree, nb, labels.data(), bases, off, skip_exact, raw, chunk_now, sender, &sink, tsv);
            n += nb;
            if (timing) {  // busy seconds of the pipeline stages of this block (which stage bounds the run)
                double busy[4];
                uint64_t nch = 0;
                if (rtx_raxtax_last_timing(busy, &nch) == RTX_OK)
                    fprintf(stderr, "[TIMING] pipeline busy seconds over %llu chunk(s) on %zu handle(s): lookup %.3f, device %.3f (busiest handle), format %.3f, sender %.3f\n",
                            (unsigned long long)nch, indices.size(), busy[0], busy[1], busy[2], busy[3]);
                uint64_t ahead = 0, abandoned = 0;  // RTX_OPT_RUN_AHEAD (the first handle, since its creation)
                if (rtx_index_run_ahead_stats(indices[0], &ahead, &abandoned) == RTX_OK)
                    fprintf(stderr, "[TIMING] chunks enqueued ahead of the end of the chunk before them: %llu, abandoned: %llu (queries of this block: %llu)\n", (unsigned long long)ahead,
                            (unsigned long long)abandoned, (unsigned long long)nb);
            }
        }
        rtx_queries_destroy(pz.qs);
        if (rc != RTX_OK || pz.end) break;
    }
    stop_and_join_reader();
    sink.out.flush();
    sink.ckp.flush();
    if (tsv) sink.tsv.flush();
    if (parse_failed) { join_bin_writer(); return 66; }
    lap("classify_and_write");
    if (!join_bin_writer()) return 74;
    lap("database_cache_wait");
    if (timing) fprintf(stderr, "{\"n_queries\": %llu, %s}\n", (unsigned long long)n, t_log.str().c_str());
    if (rc != RTX_OK) {
        fprintf(stderr, "[ERROR] %s\nRerun raxtax-hip to continue from the last checkpoint.\n", rtx_last_error());
        return rc == RTX_ERR_SENDER ? 75 : 70;  // exitcode::TEMPFAIL / SOFTWARE
    }
    if (clean) {  // Checkpoint::cleanup (io.rs:80-89)
        remove(ckp_json.c_str());
        remove(ckp_path.c_str());
        if (!db_bin.empty()) remove(db_bin.c_str());
    }
    for (rtx_index *ix : indices) rtx_index_destroy(ix);
    rtx_tree_destroy(tree);
    return 0;
}
