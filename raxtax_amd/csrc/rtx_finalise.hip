// Finalisation of the result rows on the device (lineage.rs:91-110, utils.rs:91-105).  The walks of a sub-batch leave every query's
// rows in the arena as they found them ({node, hundredths per level}); until round 5 the host sorted them, computed expected vectors and
// local signals and laid them out -- 170 thread-milliseconds per 1 M single-row queries, 10 ms of a 78-ms step on sixteen threads and
// 43 ms on the two threads a rank of an eight-GPU job on a 16-CPU host is granted (real barcodes, ten rows per query: 7 / 33 ms of a 27-ms
// step).  Here a launch behind the walks of a sub-batch does all of it: per query the order of lineage.rs:91-93 (a stable sort, done as a
// rank count: rows are few), confidence values and local signal (rtx_math.hpp: the arithmetic the host loop did, operation by operation),
// the rows of the launch back to back in the arrays of the host's view, the per-query fields scattered to the input order.  The host
// copies ranges and nothing else (rtx_api_download.hip).
//
// A workgroup (512 threads) takes 128 consecutive positions of the processing order.  First a thread per QUERY: the row counts are scanned, the rows of
// the workgroup reserved with ONE atomic on the cursor of the final arrays, the per-query fields written.  Then a thread per ROW (a search
// in the scanned counts says whose row it is): the rows of as many queries as fit are staged in LDS (hundredths, depth, node), every thread
// counts how many rows of its query come before its own -- the stable descending order of lineage.rs:91-93 as a rank count, rows are
// few -- and writes its row at that rank.  What bounds it is the chain of dependent loads per row, so everything a row is compared with
// comes from LDS and the node's side of the local signal from one table row.  (The first version sorted a query's rows by one wave, the
// queries of a workgroup one after the other: 1.8 ms per launch of 32 768 real barcodes with eleven rows each; the second compared rows
// in global memory -- 208 rows of one query are 208 x 4 dependent round trips per thread: 2.7 ms.)
#include <hip/hip_runtime.h>

#include "rtx_hit_common.hpp"
#include "rtx_kernels.hpp"
#include "rtx_math.hpp"
#include "rtx_wave.hpp"

namespace rtx {

constexpr uint32_t kFinQueries = 128;  // positions of the processing order a workgroup takes
constexpr uint32_t kFinThreads = 512;  // ... with four threads per query: real barcodes bring eleven rows per query, and a row is a chain of dependent loads
constexpr uint32_t kFinWaves = kFinQueries / 64;
constexpr uint32_t kFinKeyWords = 3072;  // LDS words for the hundredths of the staged rows: 1536 rows of a tree of up to 8 levels, 384 of 32 levels
constexpr uint32_t kFinMaxStage = 1536;
static_assert(kFinKeyWords / (RTX_MAX_DEPTH / 4) >= kWalkMaxRows, "the rows of one query must fit a pass");

__global__ __launch_bounds__(kFinThreads) void finalise_kernel(FinaliseParams p) {
    __shared__ uint32_t s_wtot[kFinWaves];
    __shared__ unsigned long long s_base;
    __shared__ uint32_t s_excl[kFinQueries + 1];            // rows of the workgroup in front of query i's
    __shared__ unsigned long long s_start[kFinQueries];     // where the walk left them
    __shared__ uint32_t s_key[kFinKeyWords];                // staged rows: hundredths (kw words each),
    __shared__ uint32_t s_node[kFinMaxStage];               // node,
    __shared__ uint8_t s_dep[kFinMaxStage];                 // depth
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t i = blockIdx.x * kFinQueries + tid;
    const bool valid = tid < kFinQueries && i < p.nq;  // the first kFinQueries threads take a query each
    const uint64_t pos = p.q0 + (valid ? i : 0u);
    uint32_t n = valid ? p.n_rows[pos] : 0u;
    const unsigned long long start = valid ? p.row_start[pos] : 0ull;
    if (n > kWalkMaxRows || start + n > p.arena_cap) n = 0;  // the arena overflowed (flagged by the walk): the run is repeated
    const uint32_t incl = wave_incl_scan_u32(n);
    if (lane == 63u && wave < kFinWaves) s_wtot[wave] = incl;
    __syncthreads();
    uint32_t before = incl - n, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < kFinWaves; w++) {
        const uint32_t wt = s_wtot[w];
        if (w < wave) before += wt;
        total += wt;
    }
    if (tid == 0) {
        unsigned long long base = atomicAdd(p.fin_cursor, (unsigned long long)total);
        if (base + total > p.row_cap) {
            atomicOr(p.flags_out, 1u);
            base = ~0ull;
        }
        s_base = base;
        s_excl[kFinQueries] = total;
    }
    if (tid < kFinQueries) {
        s_excl[tid] = before;
        s_start[tid] = start;
    }
    __syncthreads();
    const unsigned long long base = s_base;
    const bool room = base != ~0ull;  // (else: the queries keep consistent fields; the host repeats the run with larger arrays)
    if (valid) {
        const uint32_t q = p.perm[pos];
        const uint32_t t = p.t_all[pos];
        p.o_t[q] = t;
        p.o_status[q] = t > 65535u ? (uint8_t)RTX_Q_ALL_KMERS : p.status[pos];  // (kmer_extract: a read with every 8-mer was left uncounted)
        p.o_gs[q] = p.gs[pos];
        p.o_row_begin[q] = room ? base + before : 0ull;
        p.o_row_count[q] = room ? n : 0u;
    }
    if (!room) return;
    const uint32_t D = p.D, kw = (D + 3u) >> 2;  // words of hundredths a row carries (DevRow: nine words, k from the second on)
    const uint32_t cap = kFinKeyWords / kw < kFinMaxStage ? kFinKeyWords / kw : kFinMaxStage;
    for (uint32_t qa = 0; qa < kFinQueries;) {  // passes over whole queries whose rows fit the staging area (block-uniform)
        const uint32_t ra = s_excl[qa];
        uint32_t qb = qa + 1u;
        while (qb < kFinQueries && s_excl[qb + 1u] - ra <= cap) qb++;
        const uint32_t rb = s_excl[qb];
        for (uint32_t j = ra + tid; j < rb; j += kFinThreads) {  // a thread per row: stage it
            uint32_t lo = qa, hi = qb;  // the query whose rows hold row j: the last one with s_excl <= j (queries without rows share an offset)
            while (hi - lo > 1u) {
                const uint32_t mid = (lo + hi) >> 1;
                if (s_excl[mid] <= j) lo = mid; else hi = mid;
            }
            const DevRow *row = p.arena + s_start[lo] + (j - s_excl[lo]);
            const uint32_t node = row->node;
            const uint32_t *src = reinterpret_cast<const uint32_t *>(row->k);
            for (uint32_t w = 0; w < kw; w++) s_key[(j - ra) * kw + w] = fin_be32(src[w]);  // (level 0 in the top byte: rows compare as numbers)
            s_node[j - ra] = node;
            s_dep[j - ra] = p.node_depth[node];
        }
        __syncthreads();
        for (uint32_t j = ra + tid; j < rb; j += kFinThreads) {  // ... rank it among the rows of its query, finish it
            uint32_t lo = qa, hi = qb;
            while (hi - lo > 1u) {
                const uint32_t mid = (lo + hi) >> 1;
                if (s_excl[mid] <= j) lo = mid; else hi = mid;
            }
            const uint32_t e0 = s_excl[lo], nr = s_excl[lo + 1u] - e0, r = j - e0, l0 = e0 - ra;  // l0: the query's first staged row
            const uint32_t dr = s_dep[l0 + r], node = s_node[l0 + r];
            uint32_t kr[RTX_MAX_DEPTH / 4];
#pragma unroll
            for (uint32_t w = 0; w < RTX_MAX_DEPTH / 4; w++) kr[w] = w < kw ? s_key[(l0 + r) * kw + w] : 0u;
            uint32_t rank = 0;
            if (nr > 1u) {
#pragma unroll 4
                for (uint32_t x = 0; x < nr; x++) {  // fin_row_before_words (rtx_math.hpp), this row's words in registers
                    const uint32_t *kx = &s_key[(l0 + x) * kw];
                    const uint32_t dx = s_dep[l0 + x];
                    int c = 0;
#pragma unroll
                    for (uint32_t w = 0; w < RTX_MAX_DEPTH / 4; w++)
                        if (w < kw && c == 0) { const uint32_t a = kx[w]; c = a > kr[w] ? 1 : (a < kr[w] ? -1 : 0); }
                    rank += (c != 0 ? c > 0 : (dx != dr ? dx > dr : x < r)) ? 1u : 0u;
                }
            }
            const unsigned long long out = base + e0 + rank;
            p.r_lineage[out] = p.node_begin[node];  // lineage.rs:105: the index of the first lineage below the node
            p.r_node[out] = node;
            p.r_depth[out] = dr;
            p.r_depth8[out] = (uint8_t)dr;
            double *c = p.r_conf + out * D;
            uint8_t *h = p.r_hund + out * D;
            const uint32_t *kwords = &s_key[(l0 + r) * kw];
            auto kbyte = [&](uint32_t d) { return (kwords[d >> 2] >> (24u - 8u * (d & 3u))) & 255u; };  // the row's hundredths at level d
            const uint32_t sig0 = p.node_sig0[node];
            for (uint32_t d = 0; d < D; d++) {
                const uint32_t kk = d < dr ? kbyte(d) : 0u;
                c[d] = (double)kk / 100.0;  // == round(x * 100) / 100, lineage.rs:128-129
                h[d] = (uint8_t)kk;
            }
            p.r_local[out] = fin_local_signal(kbyte, p.node_eb + (size_t)node * D, sig0, dr);
        }
        __syncthreads();
        qa = qb;
    }
}

void launch_finalise(hipStream_t s, const FinaliseParams &p) {
    if (p.nq) hipLaunchKernelGGL(finalise_kernel, dim3((p.nq + kFinQueries - 1u) / kFinQueries), dim3(kFinThreads), 0, s, p);
}

}  // namespace rtx
