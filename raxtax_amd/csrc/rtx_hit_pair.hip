// hit_count for two neighbouring queries per wave (src/raxtax.rs:41,58-68, prob.rs:13-19).
//
// hit_count_kernel (rtx_kernels.hip) runs at the rate at which an XCD's L2 hands 1-KiB row segments to the CUs: more
// rows in flight per CU than it has do not make it faster, fewer bytes do.  The queries of a
// batch are processed in an order that puts related ones next to each other (rtx_cluster.hip); two neighbours share
// more than half of their rows.  Here ONE wave takes two consecutive queries of the processing order and one tile,
// keeps two sets of bit planes in registers (256 VGPRs, two waves per SIMD, 32 rows in flight per wave) and loads
// every row the two share once:
//
//   pair_union_kernel   once per sub-batch and pair: the union of the two row lists in ascending row order, every
//                       entry with its position in the list of A and / or B (the dense masks of kmer_extract are
//                       indexed by those positions); ranks by binary search in LDS
//   prologue            per tile: the union entries whose segment is dense in this tile, split into three lists in
//                       LDS: rows of both queries, of A only, of B only
//   row loop            groups of 32 rows, four buffers of eight in flight; the shared rows are folded first, into ONE
//                       plane set that is then copied (one load AND one fold per row of the union), then A's rows into
//                       A's planes, then B's; the three segments run as one pipeline (fold_seg); no barrier, no
//                       data-dependent control flow inside a segment
//   epilogue            per query, unchanged (rtx_hit_common.hpp), A then B
//
// The counts are those of hit_count_kernel bit for bit.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "rtx_hit_common.hpp"

namespace rtx {

constexpr uint32_t kPairCap = kHitListCap;            // entries per list and round
constexpr uint32_t kPairListDw = kPairCap + 192u;     // + padding to a multiple of 8 + the look-ahead of the row loop
constexpr uint32_t kPairMaskWords = 64u;              // dense-mask words (u64) per query and tile kept in LDS: rstride <= 4096
constexpr uint32_t kPairSidDw = 2u * kSparseIt * 64u;     // slot ids of the sparse segments of both queries (read at the start, used in the epilogue)
constexpr uint32_t kPairLdsBytes = 3u * kPairListDw * 4u + 2u * kPairMaskWords * 8u + 32u * 4u + kPairSidDw * 4u;  // lists | dense masks | 32 x the zero row | sparse slot ids
#ifndef RTX_PAIR_NB
#define RTX_PAIR_NB 3  // (four until round 6: 24 rows in flight and 13 spilled registers beat 32 rows and 46 -- hit_count 24.3 -> 22.8 ms per step at configs[2], every tile counted: 60.5 -> 57.4 ms per 65 536 queries)
#endif
#ifndef RTX_PAIR_WAVES
#define RTX_PAIR_WAVES 2
#endif
constexpr int kPairNB = RTX_PAIR_NB;                   // buffers of eight rows per wave (ten planes and fewer; eleven planes: three)

// ---------------------------------------------------------------------------
// Union of the row lists of the queries 2 * pair and 2 * pair + 1 of a sub-batch (ascending row ids, as kmer_extract
// leaves them).  Entry = {row | inA << 30 | inB << 31, posA | posB << 16} (0xFFFF: not in that list).
// One wave per pair.  Both lists are sorted and free of repeats, so the place of an element in the union follows from ranks alone:
//   a = A[i]:           i + rank_B(a) - (common elements below a)      rank_B(a) = elements of B below a (binary search in LDS)
//   b = B[j] not in A:  j + rank_A(b) - (common elements below b)
// and the common elements below are a running count over the list's own order (ballots).  Eleven dependent LDS reads per 64 elements --
// until round 4 the kernel scattered both lists into two 64 K-bit sets and read them out word by word (82 k cycles per pair, 4.15 ms
// per 1 M queries).  At most 2048 rows per list (the pair kernel runs with t <= 2047; t <= 1023 until round 6).
// ---------------------------------------------------------------------------
constexpr uint32_t kUnionMaxRows = 2048u;
__device__ __forceinline__ uint32_t lower_bound_u16(const uint16_t *l, uint32_t n, uint32_t key) {  // elements of l[0 .. n) below key; n <= kUnionMaxRows
    uint32_t lo = 0;
#pragma unroll
    for (uint32_t step = kUnionMaxRows; step >= 1u; step >>= 1) {  // lo + step - 1 < n and l[lo + step - 1] < key: the first lo + step are below
        const uint32_t probe = lo + step;
        const uint32_t v = probe <= n ? (uint32_t)l[probe - 1u] : 0xFFFFFFFFu;
        lo = v < key ? probe : lo;
    }
    return lo;
}

__global__ __launch_bounds__(64) void pair_union_kernel(const uint32_t *__restrict__ rows, const uint32_t *__restrict__ nrows,
                                                        uint32_t rstride, uint32_t nq, uint2 *__restrict__ urec,
                                                        uint32_t *__restrict__ nu, uint32_t ustride) {
    __shared__ uint16_t la[kUnionMaxRows], lb[kUnionMaxRows];
    const uint32_t pair = blockIdx.x, lane = threadIdx.x;
    const uint32_t qa = pair * 2u, qb = qa + 1u;
    uint32_t na = nrows[qa], nb = qb < nq ? nrows[qb] : 0u;
    na = na < kUnionMaxRows ? na : kUnionMaxRows;  // (never more: t <= 2047)
    nb = nb < kUnionMaxRows ? nb : kUnionMaxRows;
    const uint32_t *ra = rows + (size_t)qa * rstride, *rb = rows + (size_t)(qb < nq ? qb : qa) * rstride;
    uint2 *out = urec + (size_t)pair * ustride;
    const uint32_t nmax = na > nb ? na : nb;
    for (uint32_t i0 = 0; i0 < nmax; i0 += 256) {  // four chunks of each list per turn: the eight loads leave together
        uint32_t va[4], vb[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t i = i0 + (uint32_t)k * 64u + lane;
            va[k] = i < na ? ra[i] : 0u;
            vb[k] = i < nb ? rb[i] : 0u;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t i = i0 + (uint32_t)k * 64u + lane;
            if (i < na) la[i] = (uint16_t)va[k];
            if (i < nb) lb[i] = (uint16_t)vb[k];
        }
    }
    wave_lds_sync();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t ca = 0, cb = 0;  // wave-uniform: common elements in front of the chunk, in the order of A / of B
    for (uint32_t i0 = 0; i0 < nmax; i0 += 64) {  // a chunk of A and a chunk of B per turn: two independent chains of LDS reads
        const uint32_t i = i0 + lane;
        const bool in_a = i < na, in_b = i < nb;
        const uint32_t a = in_a ? (uint32_t)la[i] : 0xFFFFu, bv = in_b ? (uint32_t)lb[i] : 0xFFFFu;
        const uint32_t rka = lower_bound_u16(lb, nb, a), rkb = lower_bound_u16(la, na, bv);
        const bool eqa = in_a && rka < nb && (uint32_t)lb[rka < nb ? rka : 0u] == a;
        const bool eqb = in_b && rkb < na && (uint32_t)la[rkb < na ? rkb : 0u] == bv;
        const unsigned long long bea = __ballot(eqa), beb = __ballot(eqb);
        if (in_a) out[i + rka - (ca + (uint32_t)__popcll(bea & lt_mask))] = make_uint2(a | (1u << 30) | ((uint32_t)eqa << 31), i | ((eqa ? rka : 0xFFFFu) << 16));
        if (in_b && !eqb) out[i + rkb - (cb + (uint32_t)__popcll(beb & lt_mask))] = make_uint2(bv | (1u << 31), 0xFFFFu | (i << 16));
        ca += (uint32_t)__popcll(bea);
        cb += (uint32_t)__popcll(beb);
    }
    const uint32_t n_common = ca;
    if (lane == 0) nu[pair] = na + nb - n_common;
}

// One segment of the row loop: `ng` groups of 32 rows of `list`, folded into both plane sets (MODE 0), A's (1) or B's (2).
// On entry the four buffers hold (have requested) the first group; every buffer is requested again as soon as it has
// been folded -- in the last group with the first rows of the NEXT segment (`next`), so that the three segments of a
// wave run as one pipeline: the load latency is exposed once per wave, not once per list.
template <int NP, int MODE, int NB>
__device__ __forceinline__ void fold_seg(uint32_t (&pa)[4][NP], uint32_t (&pb)[4][NP], uint4 (&buf)[NB][8], const uint32_t *list,
                                         uint32_t ng, const uint32_t *next, uint32_t lane, __amdgpu_buffer_rsrc_t rsrc,
                                         uint32_t voff) {
    static_assert(NB >= 2 && NB <= 4, "groups of 16, 24 or 32 rows");
    constexpr uint32_t GR = 8u * (uint32_t)NB;
    for (uint32_t g = 0; g < ng; g++) {
        const uint32_t *src = g + 1 < ng ? list + (g + 1) * GR : next;  // wave-uniform
        const uint32_t idn = src[lane < GR ? lane : 0u];
        uint4 ca[NB], cb[NB];
#pragma unroll
        for (int b = 0; b < NB; b++) {
            if (MODE != 2) ca[b] = tree8<NP>(pa, buf[b]);
            if (MODE != 1) cb[b] = tree8<NP>(pb, buf[b]);
            load8v_at(buf[b], rsrc, voff, idn, b * 8);
        }
        if (MODE != 2) {
            const uint4 c4a = csa_plane<NP, 3>(pa, ca[0], ca[1]);
            if constexpr (NB == 2) ripple4<NP, 4>(pa, c4a);
            else {
                const uint4 c4b = NB == 3 ? half_plane<NP, 3>(pa, ca[NB - 1]) : csa_plane<NP, 3>(pa, ca[NB - 2], ca[NB - 1]);
                ripple4<NP, 5>(pa, csa_plane<NP, 4>(pa, c4a, c4b));
            }
        }
        if (MODE != 1) {
            const uint4 c4a = csa_plane<NP, 3>(pb, cb[0], cb[1]);
            if constexpr (NB == 2) ripple4<NP, 4>(pb, c4a);
            else {
                const uint4 c4b = NB == 3 ? half_plane<NP, 3>(pb, cb[NB - 1]) : csa_plane<NP, 3>(pb, cb[NB - 2], cb[NB - 1]);
                ripple4<NP, 5>(pb, csa_plane<NP, 4>(pb, c4a, c4b));
            }
        }
    }
}

// One (pair, tile) block: the rows of the pair in this tile, then the epilogue of each query that is live there.
template <int NP, bool kPacked, int kBounds>  // kBounds: 0 the database, 1 the union bitmap over blocks of 64 (bounds), 2 over blocks of 8 (fine bounds)
__device__ __forceinline__ void pair_tile_block(const HitParams &p, uint32_t *lds_dw, const uint32_t pair, const uint32_t tile, const uint32_t lane) {
    const uint32_t qa = pair * 2u, qb = qa + 1u;
    bool has_a = true, has_b = qb < p.nq;
    uint32_t fine_a = 0, fine_b = 0;  // kBounds == 2: the live tiles of either query among the eight database tiles this fine tile covers
    if (kBounds == 2) {
        const uint32_t T0 = tile * 8u;  // (a multiple of 8: the eight bits lie in one word)
        const uint32_t *lw = p.live + (size_t)qa * p.live_words + (T0 >> 5);
        fine_a = (lw[0] >> (T0 & 31u)) & 0xFFu;
        fine_b = has_b ? (lw[p.live_words] >> (T0 & 31u)) & 0xFFu : 0u;
        has_a = fine_a != 0u;
        has_b = fine_b != 0u;
        if (!has_a && !has_b) return;
    }
    if (kBounds == 1 && p.bounds_heavy) {  // behind the two-level bounds pass: only the queries it left to this one
        has_a = p.bounds_heavy[qa] != 0u;
        has_b = has_b && p.bounds_heavy[qb] != 0u;
        if (!has_a && !has_b) return;
    }
    if (!kBounds && p.live) {  // tile pruning (rtx_prune.hip): a mask per query -- the rows of a query are folded only where its tile is live
        const uint32_t *lw = p.live + (size_t)qa * p.live_words + (tile >> 5);
        has_a = (lw[0] >> (tile & 31u)) & 1u;
        has_b = has_b && ((lw[p.live_words] >> (tile & 31u)) & 1u);
        if (!has_a && !has_b) return;  // nothing in this tile can matter to either query
    }
    uint32_t *l_both = lds_dw, *l_a = lds_dw + kPairListDw, *l_b = lds_dw + 2u * kPairListDw;
    unsigned long long *m_a = reinterpret_cast<unsigned long long *>(lds_dw + 3u * kPairListDw), *m_b = m_a + kPairMaskWords;
    uint32_t *l_zero = reinterpret_cast<uint32_t *>(m_b + kPairMaskWords);
    if (lane < 32u) l_zero[lane] = p.zero_row << 10;
    uint32_t *l_sid = l_zero + 32;  // [2][kSparseIt * 64]
    const uint32_t col = tile * 1024u + lane * 16u;
    const bool active = col < p.stride_bytes;
    uint32_t pa[4][NP], pb[4][NP];
#pragma unroll
    for (int w = 0; w < 4; w++)
#pragma unroll
        for (int b = 0; b < NP; b++) { pa[w][b] = 0; pb[w][b] = 0; }

    const uint32_t mwords = p.rstride >> 6;
    const uint2 *urec = p.pair_urec + (size_t)pair * p.pair_ustride;
    // Everything the prologue needs comes in with ONE round trip: the first 512 union entries (clamped index: what lies
    // behind the union is masked below), the dense masks of the tile, the slot ids of the sparse segments, the counts.
    // (Requesting the second 512 entries and the numbers the epilogues start with -- t, the threshold -- here as well changed nothing
    // or cost registers: tools/NOTES.md.)
    constexpr int kRecChunks = 8;
    auto load_recs = [&](uint2 (&rec)[kRecChunks], uint32_t from) {
#pragma unroll
        for (int c = 0; c < kRecChunks; c++) {
            const uint32_t i = from + (uint32_t)c * 64u + lane;
            rec[c] = urec[i < p.pair_ustride ? i : p.pair_ustride - 1u];
        }
    };
    uint2 rec[kRecChunks];
    load_recs(rec, 0);
    const uint32_t ns_a = kBounds || !has_a ? 0u : p.nsparse[(size_t)qa * p.ntiles + tile], ns_b = !kBounds && has_b ? p.nsparse[(size_t)qb * p.ntiles + tile] : 0u;
    const uint32_t *srows_a = p.srows + ((size_t)qa * p.ntiles + tile) * (kSegMaxSparseRows + 1);
    const uint32_t *srows_b = p.srows + ((size_t)(has_b ? qb : qa) * p.ntiles + tile) * (kSegMaxSparseRows + 1);
    {
        const unsigned long long *dm_a = p.dmask + ((size_t)qa * p.ntiles + tile) * mwords;
        const unsigned long long *dm_b = p.dmask + ((size_t)(has_b ? qb : qa) * p.ntiles + tile) * mwords;
        uint32_t sa[kSparseIt], sb[kSparseIt];
#pragma unroll
        for (int it = 0; it < kSparseIt; it++) {  // unconditional: the lists have kSegMaxSparseRows + 1 entries
            sa[it] = kBounds ? 0u : srows_a[(uint32_t)it * 64u + lane];
            sb[it] = kBounds ? 0u : srows_b[(uint32_t)it * 64u + lane];
        }
        for (uint32_t i = lane; i < mwords; i += 64) {  // (a query whose tile is dead has no dense row here: its lists were never built)
            m_a[i] = has_a ? (kBounds ? ~0ull : dm_a[i]) : 0ull;  // a union bitmap is read densely: every row of the query counts
            m_b[i] = has_b ? (kBounds ? ~0ull : dm_b[i]) : 0ull;
        }
#pragma unroll
        for (int it = 0; it < kSparseIt; it++) {
            l_sid[(uint32_t)it * 64u + lane] = sa[it];
            l_sid[(kSparseIt + (uint32_t)it) * 64u + lane] = sb[it];
        }
    }
    wave_lds_sync();
    const uint32_t n_u = p.pair_nu[pair];
    const __amdgpu_buffer_rsrc_t rsrc = tile_rsrc(p.bitmap, p.n_rows1, tile);
    const uint32_t voff = lane * 16u;
    const uint32_t zero_off = p.zero_row << 10;  // the lists hold row offsets inside the tile's region (row << 10)
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t rows_loaded = 0;
    uint32_t u0 = 0;
    bool first_round = true;
    while (u0 < n_u) {  // one round unless a list would overflow (t > kPairCap - 64)
        uint32_t n_both = 0, n_a = 0, n_b = 0;
        bool room = true;
        while (room && u0 < n_u) {
            // eight chunks of 64 union entries per round trip (the first has been requested above)
            if (u0) load_recs(rec, u0);
#pragma unroll
            for (int c = 0; c < kRecChunks; c++) {
                if (u0 >= n_u) break;
                if (n_both + 64u > kPairCap || n_a + 64u > kPairCap || n_b + 64u > kPairCap) { room = false; break; }
                const bool in = u0 + lane < n_u;
                const uint32_t pos_a = rec[c].y & 0xFFFFu, pos_b = rec[c].y >> 16;
                const bool da = in && (rec[c].x & (1u << 30)) && ((m_a[pos_a >> 6] >> (pos_a & 63u)) & 1ull);
                const bool db = in && (rec[c].x & (1u << 31)) && ((m_b[pos_b >> 6] >> (pos_b & 63u)) & 1ull);
                const uint32_t row = (rec[c].x & 0x3FFFFFFFu) << 10;
                // shared rows go to the shared list in the first round (B's planes are still empty there: fold once, copy);
                // in a later round to both of the other lists
                const bool sh = da && db && first_round, oa = da && !sh, ob = db && !sh;
                const unsigned long long bb = __ballot(sh), ba = __ballot(oa), bo = __ballot(ob);
                if (sh) l_both[n_both + (uint32_t)__popcll(bb & lt_mask)] = row;
                if (oa) l_a[n_a + (uint32_t)__popcll(ba & lt_mask)] = row;
                if (ob) l_b[n_b + (uint32_t)__popcll(bo & lt_mask)] = row;
                n_both += (uint32_t)__popcll(bb);
                n_a += (uint32_t)__popcll(ba);
                n_b += (uint32_t)__popcll(bo);
                u0 += 64;
            }
        }
        rows_loaded += n_both + n_a + n_b;
        // groups of 32 rows (eleven planes: 24 -- 88 plane registers leave room for three buffers of eight rows, with four the kernel
        // spilled 128 registers); the rows a list lacks for its last group are the zero row
        constexpr int NB = NP > 10 ? 3 : kPairNB;
        constexpr uint32_t GR = 8u * (uint32_t)NB;
        const uint32_t g_both = (n_both + GR - 1u) / GR, g_a = (n_a + GR - 1u) / GR, g_b = (n_b + GR - 1u) / GR;
        for (uint32_t i = n_both + lane; i < g_both * GR; i += 64) l_both[i] = zero_off;
        for (uint32_t i = n_a + lane; i < g_a * GR; i += 64) l_a[i] = zero_off;
        for (uint32_t i = n_b + lane; i < g_b * GR; i += 64) l_b[i] = zero_off;
        wave_lds_sync();
        const uint32_t *first = g_both ? l_both : (g_a ? l_a : (g_b ? l_b : nullptr));
        if (first) {
            uint4 buf[NB][8];
            const uint32_t idv = first[lane < GR ? lane : 0u];
#pragma unroll
            for (int b = 0; b < NB; b++) load8v_at(buf[b], rsrc, voff, idv, b * 8);
            const uint32_t *after_a = g_b ? l_b : l_zero, *after_both = g_a ? l_a : after_a;
            // the shared rows are folded ONCE, into A's planes while B's are still empty, and copied: every row of the
            // union costs one fold (first round only: later rounds -- t > kPairCap - 64 -- have no shared list)
            if (g_both) {
                fold_seg<NP, 1, NB>(pa, pb, buf, l_both, g_both, after_both, lane, rsrc, voff);
#pragma unroll
                for (int w = 0; w < 4; w++)
#pragma unroll
                    for (int b = 0; b < NP; b++) pb[w][b] = pa[w][b];
            }
            if (g_a) fold_seg<NP, 1, NB>(pa, pb, buf, l_a, g_a, after_a, lane, rsrc, voff);
            if (g_b) fold_seg<NP, 2, NB>(pa, pb, buf, l_b, g_b, l_zero, lane, rsrc, voff);
        }
        wave_lds_sync();  // the lists are rewritten (next round) or become the histogram and the byte counters
        first_round = false;
    }
    if (lane == 0 && p.group_rows) atomicAdd(&p.group_rows[p.group_base + pair], rows_loaded);

    if (kBounds == 1) {  // the largest bound per tile of the database and the best block of each query; nothing else leaves the wave
        if (has_a) bounds_epilogue<NP>(p, pa, qa, tile, lane);
        if (has_b) bounds_epilogue<NP>(p, pb, qb, tile, lane);
        return;
    }
    if (kBounds == 2) {  // the tiles among the eight whose every block of 8 stays at or below the query's threshold are taken off its list
        if (has_a) fine_epilogue<NP>(p, pa, qa, tile, lane, fine_a);
        if (has_b) fine_epilogue<NP>(p, pb, qb, tile, lane, fine_b);
        if (lane == 0u && p.fine_stats) atomicAdd(&p.fine_stats[(size_t)(pair & (kPruneStatCopies - 1u)) * 8u + 1u], 1ull);
        return;
    }
    // epilogues: histogram (4 KiB; 8 KiB with eleven planes: t <= 2047) / byte counters of the whole tile (8 KiB) over the lists (dead now)
    // -- with eleven planes also over the masks and the slot ids: the slots of the sparse segments of BOTH queries are requested first
    // (their ids wait in LDS since the prologue), nothing else of the prologue is read again
    uint32_t *hist_lds = lds_dw;
    uint32_t *cnt8 = lds_dw + (NP > 10 ? 2048u : 1024u);
    uint4 pre_a[kSparseIt][kSparseV], pre_b[kSparseIt][kSparseV];
    auto thr_of = [&](uint32_t q) -> uint32_t { return p.prune_thr ? (uint32_t)__builtin_amdgcn_readfirstlane((int)p.prune_thr[q]) : 0u; };
    if (ns_a) sparse_prefetch(p, lane, ns_a, l_sid, pre_a);
    if (ns_b) sparse_prefetch(p, lane, ns_b, l_sid + kSparseIt * 64u, pre_b);
    // a query on the records path (RecordRef: pruned, few live tiles) leaves (reference, count) records of the counts above its threshold
    // instead of 8192 counts
    if (has_a) {
        const uint32_t ka = rec_slot_of(p, qa, tile, lane);
        if (ka != 0xFFFFFFFFu) rec_epilogue<NP>(p, pa, qa, tile, lane, p.t[qa], active, hist_lds, cnt8, ns_a, pre_a, thr_of(qa), ka, p.rec.stride);
        else hit_epilogue_x<NP, kPacked, true, true, true>(p, pa, qa, tile, lane, p.t[qa], active, hist_lds, cnt8, ns_a, srows_a, pre_a, thr_of(qa));
    }
    if (has_b) {
        wave_lds_sync();
        const uint32_t kb = rec_slot_of(p, qb, tile, lane);
        if (kb != 0xFFFFFFFFu) rec_epilogue<NP>(p, pb, qb, tile, lane, p.t[qb], active, hist_lds, cnt8, ns_b, pre_b, thr_of(qb), kb, p.rec.stride);
        else hit_epilogue_x<NP, kPacked, true, true, true>(p, pb, qb, tile, lane, p.t[qb], active, hist_lds, cnt8, ns_b, srows_b, pre_b, thr_of(qb));
    }
}

// kBounds: the bounds pass of the tile pruning -- the bitmap is the union bitmap (every row dense: the caller passes constant masks and
// no sparse lists), the epilogue keeps the largest bound per tile of the database and the best block instead of counts (bounds_epilogue).
//
// Grid: pairs x tiles -- or, behind tile pruning, one-dimensional over the list of the (pair, tile) blocks with a live query
// (live_items_kernel).  A pruned sub-batch of the bench workload keeps 1.5 of 62 tiles per pair: with the two-dimensional grid 97 % of
// its million workgroups came up only to read their mask and leave, which took a third of the launch.
template <int NP, bool kPacked, int kBounds, bool kItems>
__global__ __launch_bounds__(64, RTX_PAIR_WAVES) void hit_count_pair_kernel(HitParams p) {
    extern __shared__ uint32_t lds_dw[];
    const uint32_t lane = threadIdx.x;
    if (kItems) {
        // XCD x (= workgroup id mod 8) takes the x-th eighth of the list.  Workgroup y = id / 8 of the XCD starts with entry y of
        // the eighth; if there are more entries than workgroups (queries far from their best hit keep tens of tiles each) it goes on
        // with whatever entry is next in the XCD's queue (p.n_items[1 + x], counted from the number of workgroups on).  The workgroups
        // that are resident on an XCD thus always work on ONE stretch of the list -- neighbouring pairs of few tiles, which share rows
        // and with them the XCD's L2 -- however long the blocks of the others take.  (With a fixed share of the list per workgroup
        // -- a stride, or a run of consecutive entries -- the resident ones spread over many tiles: 2.2 times slower at 36 live tiles
        // per pair.)  The queue ends for every wave: an index at or behind the end of the eighth.
        const uint32_t n_items = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.n_items[0]), g8 = gridDim.x >> 3, x = blockIdx.x & 7u;
        const uint32_t e8 = (n_items + 7u) >> 3, first = x * e8;
        const uint32_t end = first + e8 < n_items ? first + e8 : n_items;  // (first >= n_items: nothing for this XCD)
        uint32_t j = first + (blockIdx.x >> 3);
        while (j < end) {
            const uint32_t item = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.items[j]);  // wave-uniform: pair and tile stay scalar
            uint32_t lane_v = lane;
            asm volatile("" : "+v"(lane_v));  // nothing that depends on the lane is kept across blocks: the block needs every register (256)
            pair_tile_block<NP, kPacked, kBounds>(p, lds_dw, item / p.ntiles, item % p.ntiles, lane_v);
            wave_lds_sync();  // the histogram of this block's epilogue becomes the next block's lists
            if (e8 <= g8) break;  // every entry had a workgroup of its own
            uint32_t nxt = 0;
            if (lane_v == 0u) nxt = atomicAdd(&p.n_items[1u + x], 1u);
            j = first + g8 + (uint32_t)__builtin_amdgcn_readfirstlane((int)nxt);
        }
        return;
    }
    // pairs are dealt to the XCDs like the queries of hit_count_kernel: XCD x takes a contiguous slice of the sub-batch
    const uint32_t np8 = gridDim.x >> 3;
    const uint32_t pair = blockIdx.x < np8 * 8u ? (blockIdx.x & 7u) * np8 + (blockIdx.x >> 3) : blockIdx.x;
    pair_tile_block<NP, kPacked, kBounds>(p, lds_dw, pair, blockIdx.y, lane);
}

// The (pair, tile) blocks in which a query of the pair is live: entry = pair * ntiles + tile.  prune_kernel left the number of
// live tiles of every pair; live_offsets_kernel (one workgroup) scans them, live_items_kernel (a thread per pair) writes the entries.
__global__ __launch_bounds__(1024) void live_offsets_kernel(const uint32_t *__restrict__ pair_live, uint32_t np, uint32_t *__restrict__ off,
                                                            uint32_t *__restrict__ n_items) {
    // pieces of 8192 pairs, a thread takes eight neighbours (a wave reads and writes 2 KB in one stretch; with a thread per run of
    // np / 1024 pairs every load and store of the one workgroup was a line of its own: 48 us per 32 768 pairs)
    __shared__ uint32_t wsum[2][16];  // double-buffered: one barrier per piece
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    constexpr uint32_t kPer = 8;
    uint32_t carry = 0, buf = 0;
    for (uint32_t base = 0; base < np; base += 1024u * kPer, buf ^= 1u) {
        const uint32_t i0 = base + tid * kPer;
        uint32_t c[kPer], cnt = 0;
#pragma unroll
        for (uint32_t j = 0; j < kPer; j++) c[j] = i0 + j < np ? pair_live[i0 + j] : 0u;
#pragma unroll
        for (uint32_t j = 0; j < kPer; j++) cnt += c[j];
        const uint32_t incl = wave_incl_scan_u32(cnt);
        if (lane == 63u) wsum[buf][wave] = incl;
        __syncthreads();
        uint32_t o = carry + incl - cnt, tot = 0;
#pragma unroll
        for (uint32_t w = 0; w < 16u; w++) {
            const uint32_t sw = wsum[buf][w];
            o += w < wave ? sw : 0u;
            tot += sw;
        }
#pragma unroll
        for (uint32_t j = 0; j < kPer; j++) {
            if (i0 + j < np) off[i0 + j] = o;
            o += c[j];
        }
        carry += tot;
    }
    if (tid == 0u) n_items[0] = carry;
    if (tid < 8u) n_items[1u + tid] = 0;  // the queues of the XCDs (hit_count_pair_kernel)
}

// Groups of 1024 pairs, and inside a group the entries TILE by tile: the workgroups of the counting pass that run at the same time then
// work on few tiles and many neighbouring pairs, whatever the number of live tiles per pair -- rows are shared in L2 between pairs of
// the same tile.  (Pair by pair, queries far from their best hit -- tens of live tiles each -- spread the resident workgroups over
// all tiles of 57 pairs: 0.64 M queries/s at 10 % divergence where the two-dimensional grid, tile-major by construction, gave 1.36 M.)
// The order of the pairs inside a (group, tile) run is whatever the LDS atomics make it: it decides nothing but scheduling.
__global__ __launch_bounds__(1024) void live_items_kernel(const uint32_t *__restrict__ live, uint32_t live_words, uint32_t nq, uint32_t ntiles,
                                                          const uint32_t *__restrict__ off, uint32_t *__restrict__ items) {
    extern __shared__ uint32_t tl[];  // [ntiles] entries of the group in the tile -> where they start -> cursor
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t np = (nq + 1u) / 2u, g0 = blockIdx.x * 1024u, pair = g0 + tid;
    for (uint32_t T = tid; T < ntiles; T += 1024) tl[T] = 0;
    __syncthreads();
    const uint32_t nw = (ntiles + 31u) >> 5;
    const bool valid = pair < np;
    const uint32_t *wa = live + (size_t)((valid ? pair : 0u) * 2u) * live_words;
    const bool hb = valid && pair * 2u + 1u < nq;  // the mask of a missing second query was never written
    if (valid)
        for (uint32_t w = 0; w < nw; w++) {
            uint32_t bits = wa[w] | (hb ? wa[live_words + w] : 0u);
            while (bits) {
                atomicAdd(&tl[w * 32u + (uint32_t)__builtin_ctz(bits)], 1u);
                bits &= bits - 1u;
            }
        }
    __syncthreads();
    if (tid < 64u) {  // exclusive scan over the tiles by one wave
        uint32_t running = 0;
        for (uint32_t T0 = 0; T0 < ntiles; T0 += 64) {
            const uint32_t v = T0 + lane < ntiles ? tl[T0 + lane] : 0u;
            const uint32_t incl = wave_incl_scan_u32(v);
            if (T0 + lane < ntiles) tl[T0 + lane] = running + incl - v;
            running += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
    }
    __syncthreads();
    if (!valid) return;
    const uint32_t base = off[g0];  // entries in front of the group
    for (uint32_t w = 0; w < nw; w++) {
        uint32_t bits = wa[w] | (hb ? wa[live_words + w] : 0u);
        while (bits) {
            const uint32_t tile = w * 32u + (uint32_t)__builtin_ctz(bits);
            bits &= bits - 1u;
            items[base + atomicAdd(&tl[tile], 1u)] = pair * ntiles + tile;
        }
    }
}

void launch_pair_union(hipStream_t s, const uint32_t *rows, const uint32_t *nrows, uint32_t rstride, uint32_t nq, uint2 *urec,
                       uint32_t *nu, uint32_t ustride) {
    hipLaunchKernelGGL(pair_union_kernel, dim3((nq + 1u) / 2u), dim3(64), 0, s, rows, nrows, rstride, nq, urec, nu, ustride);
}

#ifndef RTX_ITEM_GRID_HALVES
#define RTX_ITEM_GRID_HALVES 4  // workgroups per pass of the list of live blocks, in halves of the number of pairs
#endif
// NP bit planes: 10 (t <= 1023), 11 (t <= 2047: u16 counts), or 8 when every query of the batch has t <= 255 (amplicons of ~200 bp, the reference's example data:
// no ripple into planes 8 and 9, no high-bit words in the epilogue -- where short reads spend most of a block: few rows, the same
// 8192 counts to unpack -- and sixteen registers less)
template <int NP>
static void launch_hit_count_pair_np(hipStream_t s, const HitParams &p, uint32_t nq, uint32_t ntiles) {
    static_assert(3u * kPairListDw >= 1024u + 2048u + 64u, "histogram (t <= 1023) and byte counters (+ pad words) alias the lists");
    static_assert(kPairLdsBytes >= (2048u + 2048u + 64u) * 4u, "eleven planes: histogram (t <= 2047) and byte counters alias the whole prologue");
    static_assert(kSegMaxSparseRows + 1 >= kSparseIt * 64, "the slot id lists are read without a bound");
    const uint32_t np = (nq + 1u) / 2u;
    // with the list of live blocks: two blocks' worth of workgroups per pair and pass (the bench workload keeps 1.5), never more than the blocks there are
    const dim3 grid = p.items ? dim3((uint32_t)((std::min<uint64_t>((uint64_t)np * ntiles, std::max<uint64_t>((uint64_t)RTX_ITEM_GRID_HALVES * np / 2u, 2048ull)) + 7u) & ~7ull)) : dim3(np, ntiles);  // (a multiple of 8: the kernel deals a pass to the XCDs)
    if (NP <= 10 && p.counts_lo) {  // (the packed format ends at ten planes)
        if (p.items) hipLaunchKernelGGL((hit_count_pair_kernel<(NP <= 10 ? NP : 10), true, false, true>), grid, dim3(64), kPairLdsBytes, s, p);
        else hipLaunchKernelGGL((hit_count_pair_kernel<(NP <= 10 ? NP : 10), true, false, false>), grid, dim3(64), kPairLdsBytes, s, p);
    } else {
        if (p.items) hipLaunchKernelGGL((hit_count_pair_kernel<NP, false, false, true>), grid, dim3(64), kPairLdsBytes, s, p);
        else hipLaunchKernelGGL((hit_count_pair_kernel<NP, false, false, false>), grid, dim3(64), kPairLdsBytes, s, p);
    }
}
void launch_hit_count_pair(hipStream_t s, const HitParams &p, uint32_t nq, uint32_t ntiles, int planes) {
    if (planes <= 8) launch_hit_count_pair_np<8>(s, p, nq, ntiles);
    else if (planes <= 10) launch_hit_count_pair_np<10>(s, p, nq, ntiles);
    else launch_hit_count_pair_np<11>(s, p, nq, ntiles);  // reads of 1 031 .. 2 054 bases: u16 counts
}

void launch_live_items(hipStream_t s, const uint32_t *live, uint32_t live_words, const uint32_t *pair_live, uint32_t nq, uint32_t ntiles, uint32_t *off,
                       uint32_t *items, uint32_t *n_items) {
    const uint32_t np = (nq + 1u) / 2u;
    hipLaunchKernelGGL(live_offsets_kernel, dim3(1), dim3(1024), 0, s, pair_live, np, off, n_items);
    hipLaunchKernelGGL(live_items_kernel, dim3((np + 1023u) / 1024u), dim3(1024), (size_t)ntiles * 4, s, live, live_words, nq, ntiles, off, items);
}

void launch_hit_count_pair_bounds(hipStream_t s, const HitParams &p, uint32_t nq, uint32_t u_ntiles, int planes) {
    if (planes <= 8) hipLaunchKernelGGL((hit_count_pair_kernel<8, true, 1, false>), dim3((nq + 1u) / 2u, u_ntiles), dim3(64), kPairLdsBytes, s, p);
    else if (planes <= 10) hipLaunchKernelGGL((hit_count_pair_kernel<10, true, 1, false>), dim3((nq + 1u) / 2u, u_ntiles), dim3(64), kPairLdsBytes, s, p);
    else hipLaunchKernelGGL((hit_count_pair_kernel<11, false, 1, false>), dim3((nq + 1u) / 2u, u_ntiles), dim3(64), kPairLdsBytes, s, p);
}

void launch_hit_count_pair_bounds_items(hipStream_t s, const HitParams &p, uint32_t nq, uint32_t u_ntiles, int planes) {
    const uint32_t np = (nq + 1u) / 2u;
    // (a grid of one residency: on the bench workload the list is all but empty, and the queues of the XCDs walk a long one)
    const dim3 grid((uint32_t)((std::min<uint64_t>((uint64_t)np * u_ntiles, 4096ull) + 7u) & ~7ull));
    if (planes <= 8) hipLaunchKernelGGL((hit_count_pair_kernel<8, true, 1, true>), grid, dim3(64), kPairLdsBytes, s, p);
    else if (planes <= 10) hipLaunchKernelGGL((hit_count_pair_kernel<10, true, 1, true>), grid, dim3(64), kPairLdsBytes, s, p);
    else hipLaunchKernelGGL((hit_count_pair_kernel<11, false, 1, true>), grid, dim3(64), kPairLdsBytes, s, p);
}

// ---------------------------------------------------------------------------
// The fine bounds pass: which pairs, which fine tiles.  A pair qualifies with kFineMinLive live tiles or more (prune_kernel's
// number); it gets an item for every fine tile (eight tiles of the database) in which either query has a live tile.  The items are
// grouped by fine tile -- the workgroups that run together then read one region of the fine bitmap: fine_count (entries per fine
// tile), fine_scan (one thread: offsets, the header of the list), fine_scatter.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t fine_pair_mask(const uint32_t *live, uint32_t live_words, uint32_t pair, uint32_t nq, uint32_t U) {
    const uint32_t T0 = U * 8u;
    const uint32_t *wa = live + (size_t)(pair * 2u) * live_words + (T0 >> 5);
    uint32_t m = (wa[0] >> (T0 & 31u)) & 0xFFu;
    if (pair * 2u + 1u < nq) m |= (wa[live_words] >> (T0 & 31u)) & 0xFFu;
    return m;
}
__global__ __launch_bounds__(256) void fine_count_kernel(const uint32_t *__restrict__ live, uint32_t live_words, const uint32_t *__restrict__ pair_live,
                                                         uint32_t nq, uint32_t f_ntiles, uint32_t *__restrict__ cnt) {
    // one atomic per wave and fine tile (the lanes that hold an item are counted with a ballot): at 10 % divergence every pair qualifies, and
    // 32 768 atomics of a launch on each of eight addresses took 0.4 ms
    const uint32_t pair = blockIdx.x * 256u + threadIdx.x, np = (nq + 1u) / 2u, lane = threadIdx.x & 63u;
    const bool on = pair < np && pair_live[pair < np ? pair : 0u] >= kFineMinLive;
    for (uint32_t U = 0; U < f_ntiles; U++) {
        const unsigned long long b = __ballot(on && fine_pair_mask(live, live_words, pair, nq, U) != 0u);
        if (b && lane == (uint32_t)__builtin_ctzll(b)) atomicAdd(&cnt[U], (uint32_t)__popcll(b));
    }
}
__global__ void fine_scan_kernel(uint32_t *__restrict__ cnt, uint32_t f_ntiles, uint32_t *__restrict__ n_items) {  // one thread
    uint32_t run = 0;
    for (uint32_t U = 0; U < f_ntiles; U++) {
        const uint32_t c = cnt[U];
        cnt[U] = run;  // becomes the cursor of fine tile U
        run += c;
    }
    n_items[0] = run;
    for (uint32_t x = 1; x <= 8u; x++) n_items[x] = 0;  // the queues of the XCDs
}
__global__ __launch_bounds__(256) void fine_scatter_kernel(const uint32_t *__restrict__ live, uint32_t live_words, const uint32_t *__restrict__ pair_live,
                                                           uint32_t nq, uint32_t f_ntiles, uint32_t *__restrict__ cursor, uint32_t *__restrict__ items) {
    const uint32_t pair = blockIdx.x * 256u + threadIdx.x, np = (nq + 1u) / 2u, lane = threadIdx.x & 63u;
    const bool on = pair < np && pair_live[pair < np ? pair : 0u] >= kFineMinLive;
    for (uint32_t U = 0; U < f_ntiles; U++) {  // (as fine_count_kernel: a wave takes its entries of a fine tile with one atomic)
        const bool has = on && fine_pair_mask(live, live_words, pair, nq, U) != 0u;
        const unsigned long long b = __ballot(has);
        if (b == 0ull) continue;  // wave-uniform
        const int leader = __builtin_ctzll(b);
        uint32_t base = 0;
        if ((int)lane == leader) base = atomicAdd(&cursor[U], (uint32_t)__popcll(b));
        base = (uint32_t)__shfl((int)base, leader, 64);
        if (has) items[base + (uint32_t)__popcll(b & ((1ull << lane) - 1ull))] = pair * f_ntiles + U;
    }
}
// live tiles per pair again, from the masks as the fine pass left them (live_offsets_kernel sizes the list of the counting pass with them)
__global__ __launch_bounds__(256) void pair_live_recount_kernel(const uint32_t *__restrict__ live, uint32_t live_words, uint32_t nq, uint32_t ntiles,
                                                                uint32_t *__restrict__ pair_live, unsigned long long *__restrict__ stats) {
    const uint32_t pair = blockIdx.x * 256u + threadIdx.x, np = (nq + 1u) / 2u;
    uint32_t n = 0;
    if (pair < np) {
        const uint32_t *wa = live + (size_t)(pair * 2u) * live_words;
        const bool hb = pair * 2u + 1u < nq;
        for (uint32_t w = 0; w < (ntiles + 31u) >> 5; w++) n += (uint32_t)__popc(wa[w] | (hb ? wa[live_words + w] : 0u));
        pair_live[pair] = n;
    }
    if (stats) {  // reporting: the (pair, tile) blocks the counting pass is left with
        n = wave_incl_scan_u32(n);
        if ((threadIdx.x & 63u) == 63u && n) atomicAdd(&stats[(size_t)(blockIdx.x & (kPruneStatCopies - 1u)) * 8u + 2u], (unsigned long long)n);
    }
}

// cnt: [f_ntiles] scratch; items: [pairs * f_ntiles]; n_items: [9]
void launch_fine_bounds(hipStream_t s, const HitParams &p, uint32_t nq, uint32_t ntiles, uint32_t f_ntiles, uint32_t *pair_live, uint32_t *cnt,
                        uint32_t *items, uint32_t *n_items, int planes) {
    const uint32_t np = (nq + 1u) / 2u, nb = (np + 255u) / 256u;
    (void)hipMemsetAsync(cnt, 0, (size_t)f_ntiles * 4, s);
    hipLaunchKernelGGL(fine_count_kernel, dim3(nb), dim3(256), 0, s, p.live, p.live_words, pair_live, nq, f_ntiles, cnt);
    hipLaunchKernelGGL(fine_scan_kernel, dim3(1), dim3(1), 0, s, cnt, f_ntiles, n_items);
    hipLaunchKernelGGL(fine_scatter_kernel, dim3(nb), dim3(256), 0, s, p.live, p.live_words, pair_live, nq, f_ntiles, cnt, items);
    HitParams fp = p;
    fp.items = items;
    fp.n_items = n_items;
    // as many workgroups as a pass of the counting list: the list is walked through the queues of the XCDs
    const dim3 grid((uint32_t)((std::min<uint64_t>((uint64_t)np * f_ntiles, std::max<uint64_t>((uint64_t)RTX_ITEM_GRID_HALVES * np / 2u, 2048ull)) + 7u) & ~7ull));
    if (planes <= 8) hipLaunchKernelGGL((hit_count_pair_kernel<8, true, 2, true>), grid, dim3(64), kPairLdsBytes, s, fp);
    else if (planes <= 10) hipLaunchKernelGGL((hit_count_pair_kernel<10, true, 2, true>), grid, dim3(64), kPairLdsBytes, s, fp);
    else hipLaunchKernelGGL((hit_count_pair_kernel<11, false, 2, true>), grid, dim3(64), kPairLdsBytes, s, fp);
    hipLaunchKernelGGL(pair_live_recount_kernel, dim3(nb), dim3(256), 0, s, p.live, p.live_words, nq, ntiles, pair_live, p.fine_stats);
}

}  // namespace rtx
