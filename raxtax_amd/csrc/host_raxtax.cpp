// Host mirror of raxtax() (src/raxtax.rs:14-97): exact-match lookup, device classification in
// batches, single-exact-match override, formatting, one message per query to the sender.
#include "host_raxtax.hpp"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>

#include "rtx_internal.hpp"

namespace {

// Runs fn(i) for i in [0, n) on up to `nt` threads (contiguous ranges).
template <class F>
void parallel_ranges(uint64_t n, unsigned nt, F fn) {
    nt = (unsigned)std::min<uint64_t>(nt, (n + 255) / 256);
    if (nt <= 1) { fn(0, n); return; }
    std::vector<std::thread> th;
    for (unsigned i = 0; i < nt; i++) th.emplace_back(fn, n * i / nt, n * (i + 1) / nt);
    for (auto &t : th) t.join();
}

// A few worker threads that live as long as a run of the pipeline: run(n, fn) hands fn(0) .. fn(n - 1) to them and returns when all are
// done (sixteen std::thread per chunk were half a millisecond of the format stage per chunk: created one after the other).
class WorkerPool {
public:
    explicit WorkerPool(unsigned n) {
        for (unsigned i = 0; i < n; i++) th_.emplace_back([this] { loop(); });
    }
    ~WorkerPool() {
        { std::lock_guard<std::mutex> g(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    void run(unsigned n, const std::function<void(unsigned)> &fn) {
        if (n == 0) return;
        if (th_.empty() || n == 1) { for (unsigned r = 0; r < n; r++) fn(r); return; }
        std::unique_lock<std::mutex> g(mu_);
        job_ = &fn;
        n_jobs_ = n;
        next_ = done_ = 0;
        cv_.notify_all();
        cv_done_.wait(g, [&] { return done_ == n_jobs_; });
        job_ = nullptr;
        n_jobs_ = 0;
    }
private:
    void loop() {
        std::unique_lock<std::mutex> g(mu_);
        for (;;) {
            cv_.wait(g, [&] { return stop_ || (job_ && next_ < n_jobs_); });
            if (stop_) return;
            const unsigned r = next_++;
            const std::function<void(unsigned)> *job = job_;
            g.unlock();
            (*job)(r);
            g.lock();
            if (++done_ == n_jobs_) cv_done_.notify_all();
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_, cv_done_;
    const std::function<void(unsigned)> *job_ = nullptr;
    unsigned n_jobs_ = 0, next_ = 0, done_ = 0;
    bool stop_ = false;
};

struct LastTiming { double busy[4]; uint64_t n_chunks; };
std::mutex g_timing_mu;
LastTiming g_timing{{0, 0, 0, 0}, 0};

// (label, out_lines, tsv_lines or null) as C strings: what the C ABI's callback takes; false = the sink is closed
using RawSender = std::function<bool(const char *, const char *, const char *)>;

// One chunk of queries travelling through the stages of run().
struct Chunk {
    uint64_t q0 = 0, nq = 0;
    std::vector<uint32_t> exact_ids;   // tree.sequences.get(query) per query, raxtax.rs:42
    std::vector<uint64_t> exact_off;
    std::vector<uint8_t> differ;       // raxtax.rs:43-53: exact matches with different parents
    std::vector<uint8_t> status;       // res.status / res.t copied by the format stage: the sender must not read `res`, whose host set the
    std::vector<uint32_t> t;           // handle reuses once this chunk is formatted
    // The messages of the chunk: every formatting thread appends the NUL-terminated `.out` (and `.tsv`) text of its queries to an arena
    // of its own; msg_off[i] locates query i's text in the arena of the thread that took it (a std::string per message was a malloc in
    // the format thread and a free in the sender's, per query: half of the format stage)
    struct Arena {
        char *p = nullptr;
        size_t cap = 0, len = 0;
        char *room(size_t need) {  // a pointer behind the text so far with `need` bytes of room
            if (len + need > cap) {
                size_t c = cap ? cap + cap / 2 : (size_t)1 << 20;
                while (c < len + need) c += c / 2;
                char *np = static_cast<char *>(realloc(p, c));
                if (!np) return nullptr;
                p = np;
                cap = c;
            }
            return p + len;
        }
        void release() { free(p); p = nullptr; cap = len = 0; }
        ~Arena() { free(p); }
        Arena() = default;
        Arena(const Arena &) = delete;
        Arena(Arena &&o) noexcept : p(o.p), cap(o.cap), len(o.len) { o.p = nullptr; o.cap = o.len = 0; }
        Arena &operator=(Arena &&o) noexcept {
            if (this != &o) { free(p); p = o.p; cap = o.cap; len = o.len; o.p = nullptr; o.cap = o.len = 0; }
            return *this;
        }
    };
    std::vector<Arena> out_arena, tsv_arena;   // one per formatting thread
    std::vector<uint64_t> msg_off, tsv_off;    // [nq] offset of the query's text in its thread's arena
    std::vector<uint32_t> msg_arena;           // [nq] which arena
    rtx_result_view res{};
    int stage = 0;                     // 1: exact matches looked up (or left to the device), 2: classified, 3: formatted, 4: sent
};

// raxtax() (src/raxtax.rs:14-97) as a pipeline over chunks of `chunk_size` queries on one or several device handles -- the
// reference's `par_chunks(chunk_size)` (raxtax.rs:35-36) with GPUs in the place of rayon workers.  Chunk c belongs to handle
// c mod n_dev; per handle
//   device thread : classification of its chunks, one after the other  (rtx_classify_batch, raxtax.rs:42,55-71)
//   format thread : lineage check of the exact matches (raxtax.rs:43-53), override + formatting (raxtax.rs:73-84), chunk by chunk
// and for all of them
//   lookup thread : exact-match ids (host hash map, raxtax.rs:42) -- only for handles without the device lookup
//                   (rtx_index_has_exact_lookup): otherwise the ids come back with the results of the device stage
//   calling thread: the sender, one message per query in INPUT order (raxtax.rs:85-87): chunk 0, 1, 2 ... as they become ready.
// A handle keeps two result sets, so the view of its k-th chunk stays valid until its (k + 2)-th is classified.
int run(rtx_index *const *indices, uint32_t n_dev, const rtx_tree *tree, uint64_t n_queries, const char *const *labels,
        const uint8_t *bases, const uint64_t *base_off, bool skip_exact_matches, bool raw_confidence, uint64_t chunk_size,
        const RawSender &sender, bool tsv) {
    if (!indices || n_dev == 0 || !tree || !base_off || !labels) { rtx::set_error("rtx_raxtax: null argument"); return RTX_ERR_INVALID; }
    for (uint32_t d = 0; d < n_dev; d++) {
        if (!indices[d]) { rtx::set_error("rtx_raxtax: null index handle"); return RTX_ERR_INVALID; }
        if (rtx_index_num_refs(indices[d]) != tree->num_tips) { rtx::set_error("index and tree disagree on num_tips"); return RTX_ERR_INVALID; }
        for (uint32_t e = 0; e < d; e++)
            if (indices[e] == indices[d]) { rtx::set_error("rtx_raxtax_multi: the same handle twice (calls on one handle must be serialised)"); return RTX_ERR_INVALID; }
    }
    if (chunk_size == 0 || chunk_size > n_queries) chunk_size = std::max<uint64_t>(1, (n_queries + n_dev - 1) / n_dev);
    const uint32_t flags = (skip_exact_matches ? RTX_SKIP_EXACT_MATCHES : 0u) | (raw_confidence ? RTX_RAW_CONFIDENCE : 0u);
    const uint64_t n_chunks = (n_queries + chunk_size - 1) / chunk_size;
    // thread budget: this process's share of the host's CPUs (rtx::host_threads), split over the handles driven here
    const unsigned nt_lookup = rtx::host_threads(4u), nt_format = rtx::host_threads(16u, n_dev);
    std::vector<uint8_t> dev_lookup(n_dev);
    bool any_host_lookup = false;
    for (uint32_t d = 0; d < n_dev; d++) {
        dev_lookup[d] = rtx_index_has_exact_lookup(indices[d]) != 0;
        any_host_lookup = any_host_lookup || !dev_lookup[d];
    }

    // chunks follow one another through rtx_batch_download_then_run (the last sub-batch of a chunk is finalised beside the next chunk): two
    // large sub-batches per chunk instead of four small ones (RTX_OPT_MIN_SUB_BATCHES); the handles get their setting back below
    struct MinSubs {  // (the caller's own RTX_OPT_MIN_SUB_BATCHES comes back, and the handle's last batch stays what it was: ADVICE r5)
        rtx_index *const *ix; uint32_t n;
        std::vector<uint32_t> was;
        MinSubs(rtx_index *const *i, uint32_t k, uint64_t n_chunks) : ix(i), n(n_chunks > 1 ? k : 0), was(n) { for (uint32_t d = 0; d < n; d++) was[d] = rtx::index_swap_min_subs(ix[d], 2); }
        ~MinSubs() { for (uint32_t d = 0; d < n; d++) (void)rtx::index_swap_min_subs(ix[d], was[d]); }
    } min_subs_guard(indices, n_dev, n_chunks);
    // ... and chunk c + 1 is enqueued before the last sub-batch of chunk c has finished (RTX_OPT_RUN_AHEAD: the two streams of a handle do not
    // drain between chunks); not for handles that share a device (one stream each, below), and only with hardware queues to spare (host_threads.cpp)
    struct RunAhead {
        rtx_index *const *ix; uint32_t n;
        std::vector<uint32_t> was;
        RunAhead(rtx_index *const *i, uint32_t k, uint64_t n_chunks) : ix(i), n(n_chunks > 1 && rtx::hw_queues_for_run_ahead() ? k : 0), was(n) { for (uint32_t d = 0; d < n; d++) { was[d] = rtx::index_swap_run_ahead(ix[d], 1); if (was[d] > 1u) (void)rtx::index_swap_run_ahead(ix[d], was[d]); } }  // (2: the test aid stays)
        ~RunAhead() { for (uint32_t d = 0; d < n; d++) (void)rtx::index_swap_run_ahead(ix[d], was[d]); }
    } run_ahead_guard(indices, n_dev, n_chunks);
    // handles that share a device (rehearsals of the multi-GPU path on one GPU) run on one stream each for the duration of the call
    struct SharedDevice {
        rtx_index *const *ix; uint32_t n;
        SharedDevice(rtx_index *const *i, uint32_t k) : ix(i), n(k) {
            for (uint32_t d = 0; d < n; d++)
                for (uint32_t e = 0; e < n; e++)
                    if (e != d && rtx::index_device(ix[e]) == rtx::index_device(ix[d])) rtx::index_set_shared_device(ix[d], true);
        }
        ~SharedDevice() { for (uint32_t d = 0; d < n; d++) rtx::index_set_shared_device(ix[d], false); }
    } shared_guard(indices, n_dev);
    std::vector<Chunk> chunks(n_chunks);
    for (uint64_t c = 0; c < n_chunks; c++) {
        chunks[c].q0 = c * chunk_size;
        chunks[c].nq = std::min<uint64_t>(chunk_size, n_queries - chunks[c].q0);
    }
    // busy seconds of the stages (rtx_raxtax_last_timing: which stage bounds an end-to-end run)
    double busy_lookup = 0, busy_send = 0;
    std::vector<double> busy_device(n_dev, 0.0), busy_format(n_dev, 0.0);
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    std::mutex mu;
    std::condition_variable cv;
    int failed = RTX_OK;          // first error of any stage (under mu)
    std::string failed_msg;
    bool warnings = false;
    auto fail = [&](int rc, const std::string &msg) {
        std::lock_guard<std::mutex> g(mu);
        if (failed == RTX_OK) { failed = rc; failed_msg = msg; }
        cv.notify_all();
    };
    // blocks until chunk c has reached `stage`; false if the run has failed meanwhile
    auto wait_stage = [&](uint64_t c, int stage) {
        std::unique_lock<std::mutex> g(mu);
        cv.wait(g, [&] { return failed != RTX_OK || chunks[c].stage >= stage; });
        return failed == RTX_OK;
    };
    auto set_stage = [&](uint64_t c, int stage) {
        std::lock_guard<std::mutex> g(mu);
        chunks[c].stage = stage;
        cv.notify_all();
    };
    const uint64_t ahead = 2ull * n_dev;  // chunks a stage may run ahead of the next one
    std::mutex pool_mu;
    std::vector<Chunk::Arena> arena_pool;  // message arenas between the sender (done with a chunk) and the format threads (next chunk)

    std::thread lookup([&] {
        for (uint64_t c = 0; c < n_chunks; c++) {
            Chunk &ch = chunks[c];
            if (dev_lookup[c % n_dev]) { set_stage(c, 1); continue; }  // Tree.sequences.get runs on the device, inside rtx_classify_batch
            if (c >= ahead && !wait_stage(c - ahead, 2)) return;
            const double t_l0 = now();
            std::vector<const uint32_t *> ptr(ch.nq);
            std::vector<uint32_t> cnt(ch.nq);
            parallel_ranges(ch.nq, nt_lookup, [&](uint64_t a, uint64_t b) {
                for (uint64_t i = a; i < b; i++) {
                    const uint64_t q = ch.q0 + i;
                    const uint32_t *ids = nullptr;
                    cnt[i] = (uint32_t)rtx_tree_exact_matches(tree, bases + base_off[q], base_off[q + 1] - base_off[q], &ids);
                    ptr[i] = ids;
                }
            });
            ch.exact_off.assign(ch.nq + 1, 0);
            for (uint64_t i = 0; i < ch.nq; i++) ch.exact_off[i + 1] = ch.exact_off[i] + cnt[i];
            ch.exact_ids.resize(ch.exact_off[ch.nq]);
            for (uint64_t i = 0; i < ch.nq; i++) std::copy(ptr[i], ptr[i] + cnt[i], ch.exact_ids.begin() + ch.exact_off[i]);
            busy_lookup += now() - t_l0;
            set_stage(c, 1);
        }
    });

    // The device stage of a handle: activate -> run -> STAGE THE NEXT CHUNK -> download.  The queries of chunk c + n_dev are packed (two
    // bases per byte, pinned memory) and cross PCIe on the handle's transfer stream while the kernels of chunk c run
    // (rtx_batch_prefetch / rtx_batch_activate): the reference's workers pick up their next chunk without a pause either (raxtax.rs:35-36).
    auto stage_chunk = [&](uint32_t d, uint64_t c) -> int {
        Chunk &ch = chunks[c];
        if (dev_lookup[d]) return rtx_batch_prefetch(indices[d], ch.nq, bases, base_off + ch.q0, nullptr, nullptr);
        return rtx_batch_prefetch(indices[d], ch.nq, bases, base_off + ch.q0, ch.exact_ids.empty() ? nullptr : ch.exact_ids.data(), ch.exact_off.data());
    };
    auto device_loop = [&](uint32_t d) {
        bool running = false;  // the kernels of chunk c have been enqueued already (behind the download of the chunk before it)
        for (uint64_t c = d; c < n_chunks; c += n_dev) {
            if (!wait_stage(c, 1)) return;
            if (c >= ahead && !wait_stage(c - ahead, 3)) return;  // the result set of this handle's second-last chunk is reused now
            Chunk &ch = chunks[c];
            const double t_d0 = now();
            int rc = RTX_OK;
            if (!running) {  // the handle's first chunk
                rc = stage_chunk(d, c);
                if (!rc) rc = rtx_batch_activate(indices[d]);
                if (!rc) rc = rtx_batch_run(indices[d], flags);
            }
            running = false;
            const uint64_t nxt = c + n_dev;
            bool staged = false;
            if (!rc && nxt < n_chunks) {
                if (!wait_stage(nxt, 1)) return;   // its exact-match ids (host lookup), ready long ago as a rule
                rc = stage_chunk(d, nxt);
                staged = !rc;
            }
            // the device goes on with the next chunk as soon as the last records of this one have left it: the host's finalisation of
            // the last sub-batch (and this thread's bookkeeping) no longer stand between two chunks (rtx_batch_download_then_run)
            if (!rc) rc = staged ? rtx_batch_download_then_run(indices[d], &ch.res, flags) : rtx_batch_download(indices[d], &ch.res);
            if (rc == RTX_RETRY_CHUNK) {  // the run-ahead was abandoned (this chunk outgrew a buffer with the next one enqueued already): this chunk on its own
                rc = dev_lookup[d] ? rtx_batch_upload(indices[d], ch.nq, bases, base_off + ch.q0, nullptr, nullptr)
                                   : rtx_batch_upload(indices[d], ch.nq, bases, base_off + ch.q0, ch.exact_ids.empty() ? nullptr : ch.exact_ids.data(), ch.exact_off.data());
                if (!rc) rc = rtx_batch_run(indices[d], flags);
                if (!rc) rc = rtx_batch_download(indices[d], &ch.res);
                staged = false;  // (the next chunk is staged, activated and run at the head of the loop)
            }
            if (!rc && staged) running = true;
            if (!rc && dev_lookup[d]) {
                const uint64_t *xo = nullptr;
                const uint32_t *xi = nullptr;
                rc = rtx_batch_exact_matches(indices[d], &xo, &xi);
                if (!rc) {  // copied: the format thread reads them while the next chunk is classified
                    ch.exact_off.assign(xo, xo + ch.nq + 1);
                    ch.exact_ids.assign(xi, xi + xo[ch.nq]);
                }
            }
            if (rc) { fail(rc, rtx_last_error()); return; }
            busy_device[d] += now() - t_d0;
            set_stage(c, 2);
        }
    };

    auto format_loop = [&](uint32_t d) {
        WorkerPool pool(nt_format > 1 ? nt_format : 0);
        for (uint64_t c = d; c < n_chunks; c += n_dev) {
            if (!wait_stage(c, 2)) return;
            if (c >= 2 * ahead && !wait_stage(c - 2 * ahead, 4)) return;  // bounded memory: the sender is at most 4 chunks per handle behind
            Chunk &ch = chunks[c];
            const double t_f0 = now();
            ch.differ.assign(ch.nq, 0);
            ch.status.assign(ch.res.status, ch.res.status + ch.nq);
            ch.t.assign(ch.res.t, ch.res.t + ch.nq);
            ch.msg_off.assign(ch.nq, 0);
            ch.msg_arena.assign(ch.nq, 0);
            if (tsv) ch.tsv_off.assign(ch.nq, 0);
            const unsigned nt_f = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(nt_format, (ch.nq + 255) / 256));
            ch.out_arena.clear();
            ch.tsv_arena.clear();
            {   // arenas the sender is done with come back through the pool: a fresh megabyte per thread and chunk was a page fault per 4 KB of text
                std::lock_guard<std::mutex> g(pool_mu);
                for (unsigned r = 0; r < nt_f * (tsv ? 2u : 1u); r++) {
                    std::vector<Chunk::Arena> &dst = r < nt_f ? ch.out_arena : ch.tsv_arena;
                    if (!arena_pool.empty()) { dst.push_back(std::move(arena_pool.back())); arena_pool.pop_back(); }
                    else dst.emplace_back();
                    dst.back().len = 0;
                }
            }
            std::atomic<int> rc_fmt{0};
            auto format_range = [&](unsigned r) {
                const uint64_t a = ch.nq * r / nt_f, b = ch.nq * (r + 1) / nt_f;
                // (worked on as locals and handed back: the arenas of neighbouring threads share cache lines in the vector, and `len` moves per query)
                Chunk::Arena oa = std::move(ch.out_arena[r]), tsv_local;
                if (tsv) tsv_local = std::move(ch.tsv_arena[r]);
                Chunk::Arena *ta = tsv ? &tsv_local : nullptr;
                struct Back {
                    Chunk &c; unsigned r; Chunk::Arena &o, *t;
                    ~Back() { c.out_arena[r] = std::move(o); if (t) c.tsv_arena[r] = std::move(*t); }
                } back{ch, r, oa, ta};
                // The rows of a view lie in the PROCESSING order of the batch and the queries are formatted in input order: every row array
                // (confidences: 256 bytes per row, lineage, depth, local signal) and the lineage string are cache misses per query -- what
                // the stage waited for (0.6 us per query on sixteen threads where the arithmetic needs 0.2).  Three prefetch stages ahead of
                // the formatting: the row arrays of query i + 24, the string object of i + 16 (its index has arrived), the characters of i + 8.
                auto row_of = [&](uint64_t j) { return ch.res.row_begin[j]; };
                auto stage1 = [&](uint64_t j) {
                    if (j >= b || ch.res.status[j] != RTX_Q_OK || ch.res.row_count[j] == 0) return;
                    const uint64_t r = row_of(j);
                    __builtin_prefetch(ch.res.row_conf + r * (ch.res.row_conf_stride ? ch.res.row_conf_stride : RTX_MAX_DEPTH));
                    __builtin_prefetch(ch.res.row_lineage + r);
                    __builtin_prefetch(ch.res.row_depth + r);
                    __builtin_prefetch(ch.res.row_local_signal + r);
                };
                auto stage2 = [&](uint64_t j) {
                    if (j >= b) return;
                    __builtin_prefetch(labels[ch.q0 + j]);  // (the labels are the caller's strings, one allocation each)
                    if (ch.res.status[j] != RTX_Q_OK || ch.res.row_count[j] == 0) return;
                    __builtin_prefetch(&tree->lineages[ch.res.row_lineage[row_of(j)]]);
                };
                auto stage3 = [&](uint64_t j) {
                    if (j >= b || ch.res.status[j] != RTX_Q_OK || ch.res.row_count[j] == 0) return;
                    const std::string &l = tree->lineages[ch.res.row_lineage[row_of(j)]];
                    __builtin_prefetch(l.data());
                    __builtin_prefetch(l.data() + 64);
                };
                for (uint64_t j = a; j < std::min<uint64_t>(b, a + 24); j++) stage1(j);
                for (uint64_t j = a; j < std::min<uint64_t>(b, a + 16); j++) stage2(j);
                for (uint64_t j = a; j < std::min<uint64_t>(b, a + 8); j++) stage3(j);
                for (uint64_t i = a; i < b; i++) {
                    stage1(i + 24);
                    stage2(i + 16);
                    stage3(i + 8);
                    const uint64_t q = ch.q0 + i;
                    const uint64_t ne = ch.exact_off[i + 1] - ch.exact_off[i];
                    if (!skip_exact_matches && ne > 1) {  // raxtax.rs:43-53 (the info! lines go to the log in the CLI)
                        const uint32_t *ids = ch.exact_ids.data() + ch.exact_off[i];
                        auto parent = [&](uint32_t id) {
                            const std::string &l = tree->lineages[id];
                            const size_t k = l.rfind(',');
                            return k == std::string::npos ? std::string_view() : std::string_view(l).substr(0, k);
                        };
                        for (uint64_t j = 1; j < ne; j++)
                            if (parent(ids[j]) != parent(ids[0])) { ch.differ[i] = 1; break; }
                    }
                    if (ch.res.status[i] != RTX_Q_OK) continue;
                    const uint64_t len = base_off[q + 1] - base_off[q];
                    const uint64_t rows = ch.res.row_count[i];
                    const size_t need = (rows + 1) * (strlen(labels[q]) + 4096 + 8 * RTX_MAX_DEPTH) + len + 64;
                    char *ob = oa.room(need);
                    char *tb = ta ? ta->room(need + rows * len) : nullptr;
                    if (!ob || (ta && !tb)) { rc_fmt = RTX_ERR_OOM; return; }
                    int64_t tsv_len = 0;
                    const int64_t n = rtx_format_query(tree, &ch.res, i, labels[q], bases + base_off[q], len,
                                                       ch.exact_ids.data() + ch.exact_off[i], ne, flags, ob, need,
                                                       tb, tb ? need + rows * len : 0, &tsv_len);
                    if (n < 0) { rc_fmt = (int)n; return; }
                    ch.msg_arena[i] = r;
                    ch.msg_off[i] = oa.len;
                    oa.len += (size_t)n + 1;  // (rtx_format_query terminates its text)
                    if (ta) { ch.tsv_off[i] = ta->len; ta->len += (size_t)tsv_len + 1; }
                }
            };
            pool.run(nt_f, format_range);
            if (rc_fmt) { fail(rc_fmt, "formatting a result failed"); return; }
            busy_format[d] += now() - t_f0;
            set_stage(c, 3);
        }
    };

    std::vector<std::thread> workers;
    for (uint32_t d = 0; d < n_dev; d++) {
        workers.emplace_back(device_loop, d);
        workers.emplace_back(format_loop, d);
    }
    // the sender: messages in input order
    for (uint64_t c = 0; c < n_chunks; c++) {
        if (!wait_stage(c, 3)) break;
        Chunk &ch = chunks[c];
        const double t_s0 = now();
        bool closed = false;
        for (uint64_t i = 0; i < ch.nq && !closed; i++) {
            const uint64_t q = ch.q0 + i;
            if (ch.differ[i]) {
                fprintf(stderr, "[WARN ] Exact matches for %s differ above the leafs of the lineage tree!\n", labels[q]);
                warnings = true;
            }
            if (ch.status[i] != RTX_Q_OK) {
                // the reference aborts here (prob.rs:21/162); report and skip the query instead
                if (ch.status[i] == RTX_Q_ALL_KMERS) fprintf(stderr, "[ERROR] query %s holds every 8-mer (t = 65536): the reference asserts t < 65536 (raxtax.rs:56)\n", labels[q]);
                else fprintf(stderr, "[ERROR] query %s has too few valid 8-mers (%u) to be classified\n", labels[q], ch.t[i]);
                continue;
            }
            const uint32_t r = ch.msg_arena[i];
            if (!sender(labels[q], ch.out_arena[r].p + ch.msg_off[i], tsv ? ch.tsv_arena[r].p + ch.tsv_off[i] : nullptr)) closed = true;
        }
        if (closed) { fail(RTX_ERR_SENDER, "result sink closed"); break; }  // sender.send(..)?, raxtax.rs:87
        std::vector<uint32_t>().swap(ch.exact_ids);
        std::vector<uint64_t>().swap(ch.exact_off);
        std::vector<uint8_t>().swap(ch.differ);
        std::vector<uint8_t>().swap(ch.status);
        std::vector<uint32_t>().swap(ch.t);
        {
            std::lock_guard<std::mutex> g(pool_mu);
            for (auto &a : ch.out_arena) arena_pool.push_back(std::move(a));
            for (auto &a : ch.tsv_arena) arena_pool.push_back(std::move(a));
        }
        std::vector<Chunk::Arena>().swap(ch.out_arena);
        std::vector<Chunk::Arena>().swap(ch.tsv_arena);
        std::vector<uint64_t>().swap(ch.msg_off);
        std::vector<uint64_t>().swap(ch.tsv_off);
        std::vector<uint32_t>().swap(ch.msg_arena);
        busy_send += now() - t_s0;
        set_stage(c, 4);
    }
    lookup.join();
    for (auto &t : workers) t.join();
    {
        double bd = 0, bf = 0;
        for (uint32_t d = 0; d < n_dev; d++) { bd = std::max(bd, busy_device[d]); bf = std::max(bf, busy_format[d]); }
        std::lock_guard<std::mutex> g(g_timing_mu);
        g_timing = {{any_host_lookup ? busy_lookup : 0.0, bd, bf, busy_send}, n_chunks};
    }
    if (failed != RTX_OK) { rtx::set_error("%s", failed_msg.c_str()); return failed; }
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
    // (the C++ face hands out owned strings, as the reference's channel does)
    RawSender raw = [&](const char *label, const char *out, const char *t) {
        return sender(std::string(label), std::string(out), t ? std::optional<std::string>(std::string(t)) : std::nullopt);
    };
    return run(&index, 1, tree, queries.size(), labels.data(), bases.data(), off.data(), skip_exact_matches, raw_confidence,
               chunk_size, raw, tsv);
}

}  // namespace raxtax

extern "C" int rtx_raxtax(rtx_index *index, const rtx_tree *tree, uint64_t n_queries, const char *const *labels,
                          const uint8_t *bases, const uint64_t *base_off, int skip_exact_matches, int raw_confidence,
                          uint64_t chunk_size, rtx_sender_fn sender, void *sender_ctx, int tsv) {
    if (!sender) { rtx::set_error("rtx_raxtax: null sender"); return RTX_ERR_INVALID; }
    RawSender s = [&](const char *label, const char *out, const char *t) { return sender(sender_ctx, label, out, t) == 0; };
    return run(&index, 1, tree, n_queries, labels, bases, base_off, skip_exact_matches != 0, raw_confidence != 0, chunk_size, s,
               tsv != 0);
}

// The same over several device handles (one per GPU, or several on one): the reference's parallel driver is one call inside the
// process as well (par_chunks over a rayon pool, raxtax.rs:35-36, main.rs:40-57).  Chunks of `chunk_size` queries are dealt to the
// handles in turn, one driving thread per handle; the messages reach the sender in input order.
extern "C" int rtx_raxtax_multi(rtx_index *const *indices, uint32_t n_indices, const rtx_tree *tree, uint64_t n_queries,
                                const char *const *labels, const uint8_t *bases, const uint64_t *base_off, int skip_exact_matches,
                                int raw_confidence, uint64_t chunk_size, rtx_sender_fn sender, void *sender_ctx, int tsv) {
    if (!sender) { rtx::set_error("rtx_raxtax_multi: null sender"); return RTX_ERR_INVALID; }
    RawSender s = [&](const char *label, const char *out, const char *t) { return sender(sender_ctx, label, out, t) == 0; };
    return run(indices, n_indices, tree, n_queries, labels, bases, base_off, skip_exact_matches != 0, raw_confidence != 0, chunk_size, s,
               tsv != 0);
}

extern "C" int rtx_raxtax_last_timing(double busy[4], uint64_t *n_chunks) {
    if (!busy) { rtx::set_error("rtx_raxtax_last_timing: null argument"); return RTX_ERR_INVALID; }
    std::lock_guard<std::mutex> g(g_timing_mu);
    for (int i = 0; i < 4; i++) busy[i] = g_timing.busy[i];
    if (n_chunks) *n_chunks = g_timing.n_chunks;
    return RTX_OK;
}

// A sender that keeps nothing: counts the messages and their bytes into ctx (uint64_t[2]) if given.  For callers that time the
// path through rtx_raxtax without a disk behind it (bench.py: value_end_to_end).
extern "C" int rtx_sender_discard(void *ctx, const char *label, const char *out_lines, const char *tsv_lines) {
    (void)label;
    if (ctx) {
        uint64_t *c = static_cast<uint64_t *>(ctx);
        c[0] += 1;
        c[1] += (out_lines ? strlen(out_lines) : 0) + (tsv_lines ? strlen(tsv_lines) : 0);
    }
    return 0;
}
