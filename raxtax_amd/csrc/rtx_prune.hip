// Tile pruning (RTX_OPT_TILE_PRUNE): which tiles of 8192 references hit_count has to visit for a pair of queries.
//
// Most references of a large database share far too few k-mers with a query to receive any probability (DESIGN.md
// section 8: dropping every reference of the tiles whose largest count stays below 300 changes no prefix sum of the
// reference algorithm by more than rounding noise).  hit_count can skip such a tile if it KNOWS beforehand that every
// count in it is small ("Tile pruning", has the argument in full):
//   * upper bounds: the union bitmap -- one column per block of 2^kPruneShift consecutive references, bit (k, block) set
//     if ANY reference of the block contains k-mer k -- is counted like the database itself (hit_count_pair_kernel on one
//     tile instead of 62 at N = 500k, blocks of 64); a block's count bounds the count of each of its references.
//     ub(T) = max over the blocks of tile T.
//   * the best hits: the exact counts of the references of the block with the largest bound (the query's nearest relatives:
//     M = the largest of them, H = those with at least 0.8 M).
//   * the threshold (prob.rs:49-90 restated): Z = sum_r table[m_r] >= 1, table[m_r] = sum_i pmf_{m_r}(i) prod_{r' != r}
//     cmf_{m_r'}(i), cmf_m(i) falls with m.  G(i) = prod_{h in H} cmf_{m_h}(i).  (1) All of Z's mass at i <= i* is at most
//     G(i*) (|H| + N (i* + 1)) =: delta: a term of a reference outside H carries every factor of G, and the terms of h in H
//     sum to at most cmf_h(i*) G(i*) / cmf_h(i*).  (One reference alone, G = cmf_M, gives u = 230 on the bench workload where
//     the product over the block gives 350 -- 4.1 against 1.1 tiles per query hold a count above it.)
//     (2) For i > i* the references that are dropped (counts <= u) multiply prod by at least 1 - N tail_u(i*), and
//     hold at most N tail_u(i*) of probability themselves; tail_u(i*) <= (n - i*) pmf_u(i* + 1) once pmf_u falls.
//     i* = the largest i with delta <= eps, u = the largest count below min H with N (n - i*) pmf_u(i* + 1) <= eps (below
//     min H: the members of H are never dropped, so (1) holds for the pruned run as well): every probability and every
//     prefix sum of the pruned run is within a few eps of the full one.  eps = 1e-10 (rtx_math.hpp: kPruneEpsHD; 1e-12 until round 3; north_star asks for 1e-6).
//     With a full-overlap reference (M = t) prob.rs:24-41 applies: table[m] = pmf_m(n), table[t] = 1: u = the largest
//     count with N pmf_u(n) <= eps.
//     (3) The tighter version of (2), the one in force.  (2) prices every dropped reference at the tail of the largest dropped
//     count at i*, where G is still ~ eps / N^2 -- but a dropped reference only matters where its excursion beats the product of
//     the cmfs of H.  With f'_r(i) = pmf_r(i) prod_{kept r' != r} cmf_{r'}(i) <= G(i) the density of the pruned Z' over i (every
//     kept reference, a member of H included: pmf_h G / cmf_h <= G), F' = sum_r f'_r <= min(Z', N G):
//       beta  = what the dropped references hold themselves at i > i*  <= sum_{i>i*} G(i) sum_{dropped} pmf_r(i) <= N sum_{i>i*} G(i) pmf_u(i)
//               (E[G(X) ; X > i*] rises with the count, G rises with i: worst case everything at u);
//       gamma = what their removal adds to the kept entries = sum_{i>i*} F'(i) (1 - prod_dropped cmf_r(i))
//               <= Z' [ sum_{i in W} min(1, N G(i)) N tail_u(i) + N tail_u(end of W) ]   for any window W = (i*, i_w].
//     With alpha <= delta from (1): every probability and every sum of probabilities over any set of references moves by at most
//     2 (alpha + beta + gamma) / min(Z, Z') (Z, Z' >= 1 - 2 eps).  The kernel takes W = 63 values of i (lanes), tail_u(i) <= the sum of
//     pmf_u over the rest of W + R, R = (n - i_w) pmf_u(i_w + 1) >= tail_u(i_w) once pmf_u falls there (lane 63), and searches the
//     largest u below min H with N [sum_W min(1, N G) S_u + R] <= eps / 2 and N [sum_W G pmf_u + R] <= eps / 2 by bisection, starting
//     from the u of (2) (a valid threshold on its own).  On the bench workload u rises from ~365 to ~440 (t ~ 640, best hit ~580).
//     (4) The tile-aware version of (3) (round 4; whole-database handles -- a reference shard knows only its own tiles and must arrive at
//     the threshold of every other shard, it stays with (3)).  (3) prices EVERY reference of the database at the candidate u.  But the
//     bounds pass has left ub(T), the largest bound of every tile: a reference of tile T has a count of at most ub(T), and the tiles of
//     unrelated clades lie far below any threshold.  With c_T = min(u, ub(T)) for the dropped references of tile T (E[G(X_m); X > i*] and
//     tail_m rise with the count m) and K = the kept references (count > u: they live in tiles with ub(T) > u, so K <= the references of
//     those tiles <= the references of the tiles with ub(T) > the u of (2)):
//       beta  <= sum_T n_T sum_{i>i*} G(i) pmf_{c_T}(i),     gamma <= Z' [ sum_{i in W} min(1, K G(i)) TAIL(i) + TAIL(end of W) ],
//       TAIL(i) = sum_T n_T tail_{c_T}(i).
//     The weights of (3) with K in the place of N, and in the place of N S(u):  sum_T n_T S(min(u, ub(T))),  S(m) = sum_l pmf_m(i_l) W(l).
//     Lane g takes a group of ceil(ntiles / 64) consecutive tiles (the largest of their bounds, the sum of their references) and runs
//     the window once for its bound, pmf advanced by its exact ratio (prune_window_sums, rtx_math.hpp); a candidate costs the two wave
//     sums of (3) for S(u) and two more over the groups.  Worth ~ +15 counts at t ~ 640 on top of eps = 1e-10 (+12 over round 3's 1e-12):
//     tests/test_prune_threshold_cpu.py fills every dead tile to its bound and every live one to the threshold.
// A tile is dead for a query if u >= 1 and ub(T) <= u (a query without a threshold has every tile counted); a (pair, tile)
// block of hit_count_pair_kernel leaves at once if the tile is dead for both queries.  The references that are never counted
// are booked into histogram bin 0: cmf_0 = 1, so they drop out of every product -- the approximation bounded in (2) -- and
// the probability of bin 0 itself is at most G(0) <= eps / N for a query with a threshold (part of delta).  Their
// tiles keep a largest count of 0, and for such a query taxon_prefix leaves out every tile whose largest count is 0
// (PrefixParams::prune_thr): what it drops there is again at most N G(0) <= eps.
#include <hip/hip_runtime.h>

#include "rtx_kernels.hpp"
#include "rtx_math.hpp"
#include "rtx_wave.hpp"

namespace rtx {

static constexpr double kPruneLnEps = kPruneLnEpsHD;  // ln eps, eps = 1e-10 (rtx_math.hpp: the budget of the pruning)
static constexpr double kPruneHalfEps = kPruneHalfEpsHD;
#ifndef RTX_PRUNE_WAVES
#define RTX_PRUNE_WAVES 4
#endif
#ifndef RTX_PRUNE_TURNS
#define RTX_PRUNE_TURNS 16  // rows of the best block in flight per wave: 8 x this (8 until round 4: 0.45 ms per step slower at configs[2])
#endif
static constexpr uint32_t kPruneWavesPerBlock = 4;

__device__ __forceinline__ uint32_t wave_max_u32p(uint32_t v) { return wave_max_u32(v); }

// Four waves per workgroup, one pair of queries per wave.  The kernel is bound by the instructions it issues (round 4 counters: every wave
// active 24 % of its cycles at four waves per SIMD), and every evaluation of a pmf reads six entries of the ln x! table: the workgroup
// stages the part of the table the batch can reach (t + n - 1 <= 1.5 tmax: 12 KB at t <= 1023) in LDS once, for its eight queries.
__global__ __launch_bounds__(64 * kPruneWavesPerBlock, RTX_PRUNE_WAVES) void prune_kernel(PruneParams p, ProbTables tb) {
    extern __shared__ double prune_lds[];  // [nlf] ln x!; then per wave [2][ntiles] u16: the bounds of the tiles for the pair's queries
    for (uint32_t i = threadIdx.x; i < p.nlf; i += 64u * kPruneWavesPerBlock) prune_lds[i] = p.lnfact[i];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, pair = blockIdx.x * kPruneWavesPerBlock + (threadIdx.x >> 6);
    if (pair * 2u >= p.nq) return;  // (behind the only barrier)
    uint16_t *ub_lds = reinterpret_cast<uint16_t *>(prune_lds + p.nlf) + (size_t)(threadIdx.x >> 6) * 2u * p.ntiles;
    const double *lf = prune_lds;
    const double ln_n = log((double)p.n_total);
    uint32_t thr[2] = {0u, 0u};
    uint32_t ist_prev = 0;  // i* + 1 of the pair's first query (the second one's search starts there)
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // wave-uniform: what this wave adds to p.stats
    const bool has_b = pair * 2u + 1u < p.nq;
    for (uint32_t x = 0; x < 2u; x++) {
        if (x == 1u && !has_b) break;  // wave-uniform
        const uint32_t q = pair * 2u + x;
        // ---- 1. bound of every tile, and the block with the largest bound (the lowest one among equals; block 0 if every bound is 0):
        // left by the epilogue of the bounds pass (bounds_epilogue, rtx_hit_common.hpp)
        for (uint32_t T = lane; T < p.ntiles; T += 64) ub_lds[x * p.ntiles + T] = p.tile_ub[(size_t)q * p.tile_ub_stride + T];
        const uint32_t bkey = p.best_key[q];
        uint32_t ub_best = bkey >> 20;
        const uint32_t bb = 0xFFFFFu - (bkey & 0xFFFFFu);
        // ---- 2. exact counts of its references (not those that --skip-exact-matches zeroes); M = the best of them.  The
        // block's references lie in kChunks chunks of eight = that many bytes of a row segment, in neighbouring lane words
        // (ref_slot, rtx_math.hpp); a lane takes one chunk of every eighth row of the list and sums the hits of its eight references.
        uint32_t M = 0;
        uint32_t hm = 0;  // lane l < 2^kPruneShift: the exact count of reference l of the block (0: none, or zeroed)
        if (p.phase == 2u) {  // the best block of the whole database, as the exchange between the shards left it
            const uint32_t *bq = p.best + (size_t)q * kPruneBestWords;
            ub_best = bq[0];
            hm = bq[2u + lane];
            M = wave_max_u32p(hm);
        } else if (p.cbitmap) {
            // Round 5: the database stored once more block by block -- [block][row] 8 bytes, bit j = reference 64 block + j -- so that the
            // 64 references of the best block are ONE 8-byte load per row and lane = row: ten loads per lane for a query of 640 rows, all in
            // flight together (32 for a read of t = 2047), from a region of 512 KB per block that neighbouring queries (same best block) keep in L2.  The walk through
            // the tile-major bitmap below took a 128-byte line per row in eighty dependent-id loads per lane: 5.5 of this kernel's 11.9 ms
            // per step.  A lane adds its <= 16 rows into bit-sliced counters (5 planes x 2 words), the lanes are summed as bit-sliced numbers
            // (lane ^ 32: each keeps one word; then within the halves), and lane l reads counter l & 31 of word l >> 5: reference l.
            static_assert(kPruneShift == 6, "a block of the block-major bitmap is the block of the bounds");
            const uint32_t nr = p.nrows[q];
            const uint32_t *rows = p.rows + (size_t)q * p.rstride;
            const uint32_t zero_row = p.n_rows1 - 1u;
            const uint32_t nr_pad = (nr + 63u) & ~63u;  // the row list is padded with the all-zero row to whole chunks of 64
            const uint2 *C = p.cbitmap + (size_t)bb * p.n_rows1;
            uint32_t pl[2][11];
#pragma unroll
            for (int w = 0; w < 2; w++)
#pragma unroll
                for (int b = 0; b < 11; b++) pl[w][b] = 0;
            for (uint32_t i0 = 0; i0 < nr_pad; i0 += 256u) {  // four chunks of 64 rows per turn (a query of t <= 2047 rows: eight turns at most)
                uint32_t id[4];
                uint2 v[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t i = i0 + (uint32_t)k * 64u + lane;
                    id[k] = i < nr_pad ? rows[i] : zero_row;
                }
#pragma unroll
                for (int k = 0; k < 4; k++) v[k] = C[id[k]];
#pragma unroll
                for (int k = 0; k < 4; k++) {  // + 1 into the counters of the set bits (at most 32 rows per lane: planes 0 .. 5)
                    uint32_t cx = v[k].x, cy = v[k].y;
#pragma unroll
                    for (int b = 0; b < 6; b++) {
                        const uint32_t nx = pl[0][b] & cx, ny = pl[1][b] & cy;
                        pl[0][b] ^= cx;
                        pl[1][b] ^= cy;
                        cx = nx;
                        cy = ny;
                    }
                }
            }
            // lane ^ 32: the lower half of the wave keeps word 0 (references 0 .. 31), the upper half word 1
            uint32_t r[11];
            {
                const bool up = (lane & 32u) != 0u;
                uint32_t o[11];
#pragma unroll
                for (int b = 0; b < 11; b++) {
                    r[b] = up ? pl[1][b] : pl[0][b];
                    o[b] = (uint32_t)__shfl_xor((int)(up ? pl[0][b] : pl[1][b]), 32, 64);
                }
                planes_add<11>(r, o);
            }
#pragma unroll
            for (int d = 16; d >= 1; d >>= 1) {
                uint32_t o[11];
#pragma unroll
                for (int b = 0; b < 11; b++) o[b] = (uint32_t)__shfl_xor((int)r[b], d, 64);
                planes_add<11>(r, o);
            }
            uint32_t cnt = 0;
#pragma unroll
            for (int b = 0; b < 11; b++) cnt |= ((r[b] >> (lane & 31u)) & 1u) << b;
            const uint64_t rr = ((uint64_t)bb << kPruneShift) + (uint64_t)lane;
            bool ok = rr < p.n_refs;
            if (p.flags & RTX_SKIP_EXACT_MATCHES) {
                const uint64_t qin = p.perm[p.q0 + q];
                uint64_t xe0 = 0, xe1 = 0;
                const uint32_t *xids = nullptr;
                exact_range(p.exact, qin, xe0, xe1, xids);
                for (uint64_t e = xe0; e < xe1; e++) ok = ok && (uint64_t)(xids[e] - p.ref_base) != rr;  // (wave-uniform loop)
            }
            hm = ok ? cnt : 0u;
            M = wave_max_u32p(hm);
        } else {
            constexpr uint32_t kChunks = (1u << kPruneShift) / 8u;  // 4 for blocks of 32
            const uint32_t nr = p.nrows[q];
            const uint32_t *rows = p.rows + (size_t)q * p.rstride;
            const uint64_t qin = p.perm[p.q0 + q];
            uint64_t xe0 = 0, xe1 = 0;
            const uint32_t *xids = nullptr;
            if (p.flags & RTX_SKIP_EXACT_MATCHES) exact_range(p.exact, qin, xe0, xe1, xids);
            const uint32_t zero_row = p.n_rows1 - 1u;
            // lane = (row group, chunk): an instruction reads the block's bytes of 64 / kChunks rows, and the chunks of a block are
            // neighbouring lane words of the row segment -- one 128-byte line per row (with lane = row and a gather per chunk
            // every instruction touched 64 lines: eight times the requests)
            constexpr uint32_t kRowsPerTurn = 64u / kChunks;
            const uint32_t ch = lane & (kChunks - 1u), rg = lane / kChunks;
            uint32_t word, bit;
            {
                const uint64_t r0 = ((uint64_t)bb << kPruneShift) + (uint64_t)ch * 8u;
                ref_slot((uint32_t)(r0 < p.n_refs ? r0 : (uint64_t)bb << kPruneShift), p.stride_bytes, word, bit);  // bit = first bit of the chunk's byte
            }
            // this lane's chunk = eight references = one byte of the row segment: the low and the high nibble are spread over the four
            // bytes of a word with one 24-bit multiply each (x * 0x204081 puts bit k of x at bit 8 k: the four shifted copies do not
            // overlap) and summed in byte counters -- a lane sees every kChunks-th row of the list, at most 1024 / 8 = 128 of them
            static_assert(kChunks == 8u, "the byte counters of the best block's exact counts hold the rows of one lane in eight (blocks of 64 references)");
            uint32_t acc_lo = 0u, acc_hi = 0u;
            uint32_t acc[4] = {0u, 0u, 0u, 0u};  // [pair of references of this lane's chunk]: two 16-bit counters
            const uint32_t bit4 = bit + 4u;
            const uint32_t nr_pad = (nr + 63u) & ~63u;  // the row list is padded with the all-zero row to whole chunks of 64
            // The rows are random lines of the best tile's region (HBM, not L2): kTurns turns = kTurns * kRowsPerTurn rows are in flight
            // together, their row ids come in as whole 256-byte pieces of the list (lane l <- entry l, handed to the lanes of a row
            // group with a shuffle) and the ids of the next piece are requested before this piece's lines are waited for -- one exposed
            // round trip per piece (a loop of 32 rows with its ids loaded per row group paid two per 32).
            constexpr uint32_t kTurns = RTX_PRUNE_TURNS, kPiece = kTurns * kRowsPerTurn;  // 64 or 128 rows
            static_assert(kPiece % 64u == 0u && kPiece <= 128u, "a piece of the row list = one or two loads per lane");
            auto load_ids = [&](uint32_t i0, uint32_t (&id)[2]) {
#pragma unroll
                for (uint32_t k = 0; k < kPiece / 64u; k++) {
                    const uint32_t i = i0 + k * 64u + lane;
                    id[k] = i < nr_pad ? rows[i] : zero_row;  // (the list is padded to whole chunks of 64 with the zero row)
                }
            };
            uint32_t id[2] = {zero_row, zero_row}, idn[2] = {zero_row, zero_row};
            load_ids(0, id);
            for (uint32_t i0 = 0; i0 < nr; i0 += kPiece) {
                if (i0 + kPiece < nr) load_ids(i0 + kPiece, idn);  // wave-uniform
                uint32_t w[kTurns];
#pragma unroll
                for (uint32_t u = 0; u < kTurns; u++) {
                    const uint32_t e = u * kRowsPerTurn + rg;  // entry of the piece this lane's row group takes in turn u
                    const uint32_t row = (uint32_t)__shfl((int)id[e >> 6], (int)(e & 63u), 64);
                    w[u] = p.bitmap[bitmap_word(row, word, p.n_rows1)];
                }
#pragma unroll
                for (uint32_t u = 0; u < kTurns; u++) {
                    acc_lo += __umul24((w[u] >> bit) & 0xFu, 0x204081u) & 0x01010101u;
                    acc_hi += __umul24((w[u] >> bit4) & 0xFu, 0x204081u) & 0x01010101u;
                }
                id[0] = idn[0];
                id[1] = idn[1];
                // the byte counters are widened piece by piece (16 rows of this lane at most: a read of t = 2047 brings a lane 256 rows)
                acc[0] += (acc_lo & 0xFFu) | ((acc_lo & 0xFF00u) << 8);
                acc[1] += ((acc_lo >> 16) & 0xFFu) | ((acc_lo >> 24) << 16);
                acc[2] += (acc_hi & 0xFFu) | ((acc_hi & 0xFF00u) << 8);
                acc[3] += ((acc_hi >> 16) & 0xFFu) | ((acc_hi >> 24) << 16);
                acc_lo = 0u;
                acc_hi = 0u;
            }
            // ... summed over the row groups (lane c ends up with chunk c)
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (uint32_t d = 32; d >= kChunks; d >>= 1) acc[k] += (uint32_t)__shfl_xor((int)acc[k], (int)d, 64);
            // lane l <-> reference l of the block (chunk l / 8, reference l % 8 of it); not those behind the end, not those
            // --skip-exact-matches zeroes
            {
                const uint32_t c = lane >> 3, j = lane & 7u;
                const uint32_t v0 = (uint32_t)__shfl((int)acc[0], (int)c, 64), v1 = (uint32_t)__shfl((int)acc[1], (int)c, 64),
                               v2 = (uint32_t)__shfl((int)acc[2], (int)c, 64), v3 = (uint32_t)__shfl((int)acc[3], (int)c, 64);
                const uint32_t vv = (j & 4u) ? ((j & 2u) ? v3 : v2) : ((j & 2u) ? v1 : v0);
                const uint32_t cnt = (vv >> ((j & 1u) * 16u)) & 0xFFFFu;
                const uint64_t r = ((uint64_t)bb << kPruneShift) + (uint64_t)lane;
                bool ok = lane < (1u << kPruneShift) && r < p.n_refs;
                if (p.flags & RTX_SKIP_EXACT_MATCHES)
                    for (uint64_t e = xe0; e < xe1; e++) ok = ok && (uint64_t)(xids[e] - p.ref_base) != r;  // local id; other shards' ids wrap out of range (wave-uniform loop)
                hm = ok ? cnt : 0u;
                M = wave_max_u32p(hm);
            }
        }
        if (p.detail) p.detail[(size_t)q * kPruneDetailWords + 8u + lane] = hm;  // debug tap: the exact counts of the best block's references
        if (p.phase == 1u) {  // a reference shard, first half: its candidate for the best block of the database
            uint32_t *bq = p.best + (size_t)q * kPruneBestWords;
            bq[2u + lane] = hm;
            if (lane == 0) { bq[0] = ub_best; bq[1] = 0u; }
            continue;  // wave-uniform
        }
        // ---- 3. the largest count a skipped tile may hold
        const uint32_t t = p.t[q], n = t >> 1;
        uint32_t u_max = 0, i1_q = 0;
        if (t >= 16u && t <= tb.tmax && M >= 1u && n >= 2u) {
            const double ln_total = ln_binom_tab(lf, t + n - 1, n);
            if (M >= t) {  // full overlap: table[m] = pmf_m(n) = C(m+n-1, n) / C(t+n-1, n), rising with m
                uint32_t mine = 0;
                for (uint32_t u = 1u + lane; u < t; u += 64)
                    if (ln_binom_tab(lf, u + n - 1, n) - ln_total + ln_n <= kPruneLnEps) mine = u;
                u_max = wave_max_u32p(mine);
            } else {
                // G(i) = prod over the counted references of the best block of cmf_m(i) (its members are the query's nearest
                // relatives: the product falls far faster than cmf_M alone); lane l holds the row of reference l.  i* by
                // bisection (G rises with i): the largest i <= n - 2 (a tail is left) with ln G(i) + ln(|H| + N (i + 1)) <= ln eps
                const double *Ct = tb.cmf + tb.off[t];
                if (hm * 5u < M * 4u) hm = 0u;  // H: the members close to the best one (the others would add next to nothing to G)
                const double n_h = (double)__popcll(__ballot(hm != 0u));
                const uint32_t h_min = 0xFFFFFFFFu - wave_max_u32p(hm ? 0xFFFFFFFFu - hm : 0u);  // <= M: H holds the best one
                // ln(|H| + N (i + 1)) lies between its values at i = 0 and at i = n - 2: a sum outside that band (all but the last probes of
                // the search: ln G moves by far more per step) decides without the logarithm -- the same answers, most of the log() calls gone
                const double ln_f_lo = log(n_h + (double)p.n_total), ln_f_hi = log(n_h + (double)p.n_total * (double)(n - 1u));
                auto passes = [&](uint32_t i) -> bool {
                    const double v = hm ? Ct[(size_t)hm * (n + 1) + i] : 0.0;  // ln cmf_m(i)
                    const double sum = wave_sum_f64_dpp(v);
                    if (sum + ln_f_hi <= kPruneLnEps) return true;   // wave-uniform
                    if (sum + ln_f_lo > kPruneLnEps) return false;
                    return sum + log(n_h + (double)p.n_total * (double)(i + 1u)) <= kPruneLnEps;
                };
                uint32_t ist1 = 0;  // i* + 1 in the end (0: none)
                if (passes(0u)) {
                    // passes() is monotone (G and the factor rise with i): the boundary is found by galloping away from a guess and a
                    // bisection of the bracket -- the same i* as a plain bisection of [0, n - 2], in fewer round trips (every probe is a
                    // gather from the 740 MB of tables; the probes around a good guess share their cache lines).  Guess: the second query
                    // of the pair starts at the first one's i* (neighbours are relatives); the first at mean - 6 sd of the best hit.
                    uint32_t g;
                    if (x == 1u && ist_prev) g = ist_prev - 1u;
                    else {
                        const double pm = (double)M / (double)t;
                        const double mu = (double)n * pm, sd = sqrt((double)n * pm * (1.0 - pm) * (double)(t + n) / (double)(t + 1u));
                        const double gg = mu - 6.0 * sd;
                        g = gg > 0.0 ? (uint32_t)gg : 0u;
                    }
                    g = g > n - 2u ? n - 2u : g;
                    uint32_t lo, hi;  // passes(lo), and !passes(hi) or hi == n - 1 (beyond the range)
                    if (passes(g)) {
                        lo = g;
                        uint32_t step = 1u;
                        while (lo + step <= n - 2u && passes(lo + step)) { lo += step; step <<= 1; }  // wave-uniform
                        hi = lo + step <= n - 2u ? lo + step : n - 1u;
                    } else {
                        hi = g;
                        uint32_t step = 1u;
                        while (hi > step && !passes(hi - step)) { hi -= step; step <<= 1; }
                        lo = hi > step ? hi - step : 0u;
                    }
                    while (hi - lo > 1u) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (passes(mid)) lo = mid; else hi = mid;
                    }
                    ist1 = lo + 1u;
                }
                ist_prev = ist1;
                if (ist1) {
                    const uint32_t i1 = ist1;  // = i* + 1: the first i that stays
                    const double ln_len = log((double)(n - i1 + 1u));
                    // the condition holds for a prefix of the counts (below its mode pmf_u(i1) rises with u); taken as the
                    // counts below the smallest one that fails, whatever rounding does to the largest one that passes
                    uint32_t first_fail = h_min;  // counts from min H on are never dropped
                    for (uint32_t u = 1u + lane; u < h_min; u += 64) {
                        // pmf_u falling from i1 on: (u + i1)(n - i1) < (i1 + 1)(t - u + n - i1 - 1)
                        const double up = (double)(u + i1) * (double)(n - i1), dn = (double)(i1 + 1u) * (double)(t - u + n - i1 - 1u);
                        if (!(up < dn && ln_len + ln_pmf_tab(lf, t, n, u, i1, ln_total) + ln_n <= kPruneLnEps)) first_fail = u < first_fail ? u : first_fail;
                    }
                    first_fail = 0xFFFFFFFFu - wave_max_u32p(0xFFFFFFFFu - first_fail);
                    u_max = first_fail - 1u;
                    i1_q = i1;
                    // ---- the tighter criterion (header, "(3)"): the same two error terms, but every i weighted with what G leaves of
                    // it.  Lanes 0 .. 62 <-> the window i = i1 .. i1 + 62; lane 63 <-> the point behind it, j = i1 + 63, which stands for
                    // the whole tail of pmf_u from there on ((n - j + 1) pmf_u(j) once pmf_u falls; G counts as 1 out there).  With
                    //   A(l) = sum_{l' < l} min(1, N G(i_l'))  (what a hit at i_l costs through every smaller i of the window)
                    // the two sums are sum_l pmf_u(i_l) WA(l) and sum_l pmf_u(i_l) WB(l) with weights that do not depend on u:
                    // one exp and two wave sums per candidate.
                    {
                        // (4): lane g <-> group g of `per` consecutive tiles: the largest bound, the references; K = those of the groups above the u of (2)
                        const bool tile_aware = p.phase == 0u;
                        uint32_t gub = 0;
                        double gnd = 0.0;
                        {
                            const uint32_t per = (p.ntiles + 63u) >> 6;
                            for (uint32_t k = 0; k < per; k++) {
                                const uint32_t T = lane * per + k;
                                if (T < p.ntiles) {
                                    const uint32_t b = ub_lds[x * p.ntiles + T];
                                    gub = b > gub ? b : gub;
                                    const uint64_t lo = (uint64_t)T * 8192u, hi = lo + 8192u < p.n_refs ? lo + 8192u : p.n_refs;
                                    gnd += (double)(hi - lo);
                                }
                            }
                        }
                        const double kept = tile_aware ? wave_sum_f64_dpp(gub > u_max ? gnd : 0.0) : (double)p.n_total;
                        const double ln_kept = log(kept > 1.0 ? kept : 1.0);
                        const uint32_t iw = i1 + lane;
                        const bool tail_pt = lane == 63u;
                        const bool vi = iw <= n;
                        double lnG = 0.0;
                        unsigned long long hb = __ballot(hm != 0u);
                        const uint32_t iwc = vi ? iw : n;
                        while (hb) {  // wave-uniform: the members of H, four rows of the table in flight (added in the order of the lanes, as one by one)
                            double v[4];
#pragma unroll
                            for (int k = 0; k < 4; k++) {
                                v[k] = 0.0;
                                if (hb) {
                                    const int h = __builtin_ctzll(hb);
                                    hb &= hb - 1ull;
                                    const uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)hm, h);
                                    v[k] = Ct[(size_t)m * (n + 1) + iwc];
                                }
                            }
#pragma unroll
                            for (int k = 0; k < 4; k++) lnG += v[k];
                        }
                        const double gw = vi && !tail_pt ? exp(lnG) : 0.0;                         // G(i)
                        const double ww = vi && !tail_pt ? exp(fmin(0.0, ln_kept + lnG)) : 0.0;   // min(1, K G(i)), K = N for a shard
                        const double incl = wave_incl_scan_f64_dpp(ww);
                        const double a_tot = readlane_f64(incl, 63);
                        const double len_tail = vi ? (double)(n - iw + 1u) : 0.0;                 // lane 63: the values of i from j on
                        const double WA = tail_pt ? len_tail * (a_tot + 1.0) : (vi ? incl - ww : 0.0);
                        const double WB = tail_pt ? len_tail : gw;
                        const double nn = (double)p.n_total;
                        const bool has_tail = i1 + 63u <= n;  // wave-uniform
                        auto sums_at = [&](uint32_t m, double &a, double &b) {  // S_A(m), S_B(m): lanes <-> the window, one exp and two wave sums
                            const double P = vi && m != 0u ? exp(ln_pmf_tab(lf, t, n, m, iw, ln_total)) : 0.0;  // pmf_m(i_l)
                            a = wave_sum_f64_dpp(P * WA);
                            b = wave_sum_f64_dpp(P * WB);
                        };
                        // (4): the window sums of the groups at their own bounds (capped below min H: a group above it is never dead).  Only the
                        // groups whose bound lies within kPruneFarGap counts of the threshold of (2) -- the query's own clade and its
                        // neighbours, a handful -- are worth a value of their own: the others ("far": unrelated clades, S falls by orders of
                        // magnitude per ten counts) are priced together at S(u_(2) - kPruneFarGap), which bounds each of theirs (S rises with the
                        // count).  More than kPruneMaxNear near groups (a threshold near the background: everything is "near"): lane g runs the
                        // window once for its own bound, pmf advanced by its ratio (prune_window_sums), whatever their number.
                        // Round 5: the groups between the two -- bound within kPruneFarGap counts BELOW the threshold of (2): dead at every
                        // candidate, the bisection starts at that threshold -- are priced together at S(u_(2)), which bounds each of theirs.
                        // They hold a handful of tiles against the N references (3) priced at its own, larger threshold: nothing is lost, and
                        // the groups that get a window sum of their own are those ABOVE u_(2) only (with the two-level bounds pass the tiles
                        // left with a bound over blocks of 256 crowd the band below it: their sums were a quarter of this kernel's time).
                        double gsa = 0.0, gsb = 0.0, far_a = 0.0, far_b = 0.0, n_far = 0.0, mid_a = 0.0, mid_b = 0.0, n_mid = 0.0;
                        bool near = false;
                        if (tile_aware) {
                            const uint32_t m_far = u_max > kPruneFarGap ? u_max - kPruneFarGap : 0u;
                            near = gnd > 0.0 && gub > u_max;
                            const bool mid = gnd > 0.0 && gub > m_far && gub <= u_max;
                            unsigned long long nb = __ballot(near);
                            if ((uint32_t)__popcll(nb) > kPruneMaxNear) {
                                near = gnd > 0.0;
                                const uint32_t gcap = gub < h_min ? gub : h_min - 1u;
                                prune_window_sums(lf, p.inv, t, n, gcap, i1, ln_total, [&](uint32_t l) { return readlane_f64(WA, (int)l); },
                                                  [&](uint32_t l) { return readlane_f64(WB, (int)l); }, gsa, gsb);
                            } else {
                                n_far = wave_sum_f64_dpp(near || mid ? 0.0 : gnd);
                                n_mid = wave_sum_f64_dpp(mid ? gnd : 0.0);
                                if (m_far) sums_at(m_far, far_a, far_b);
                                if (n_mid > 0.0) sums_at(u_max, mid_a, mid_b);  // wave-uniform
                                while (nb) {  // wave-uniform
                                    const int g = __builtin_ctzll(nb);
                                    nb &= nb - 1ull;
                                    const uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)gub, g);
                                    double a, b;
                                    sums_at(m < h_min ? m : h_min - 1u, a, b);
                                    if ((int)lane == g) { gsa = a; gsb = b; }
                                }
                            }
                        }
                        auto crit = [&](uint32_t u) -> bool {
                            bool falling = true;
                            if (has_tail) {  // pmf_u(j + 1) < pmf_u(j) at j = i1 + 63, and the ratio falls with j
                                const uint32_t j = i1 + 63u;
                                falling = (double)(u + j) * (double)(n - j) < (double)(j + 1u) * (double)(t - u + n - j - 1u);
                            }
                            double a, b;
                            sums_at(u, a, b);
                            if (!tile_aware) return falling && nn * a <= kPruneHalfEps && nn * b <= kPruneHalfEps;
                            const bool dead = gub <= u;
                            return falling && wave_sum_f64_dpp(near ? gnd * (dead ? gsa : a) : 0.0) + n_far * far_a + n_mid * mid_a <= kPruneHalfEps &&
                                   wave_sum_f64_dpp(near ? gnd * (dead ? gsb : b) : 0.0) + n_far * far_b + n_mid * mid_b <= kPruneHalfEps;
                        };
                        uint32_t lo = u_max, hi = h_min - 1u;
                        while (lo < hi) {  // wave-uniform
                            const uint32_t mid = (lo + hi + 1u) >> 1;
                            if (crit(mid)) lo = mid; else hi = mid - 1u;
                        }
                        u_max = lo;
                    }
                }
            }
        }
        thr[x] = u_max;
        if (lane == 0) { p.thr_out[q] = (uint16_t)u_max; p.i1_out[q] = (uint16_t)(u_max ? i1_q : 0u); }
        if (p.detail && lane == 0) {
            uint32_t *d = p.detail + (size_t)q * kPruneDetailWords;
            d[0] = bb; d[1] = M; d[2] = u_max; d[3] = u_max ? i1_q : 0u; d[4] = ub_best; d[5] = t; d[6] = 0u; d[7] = 0u;
        }
        // reporting: sums of the lower bound of the best hit, of the threshold, of the largest tile bound
        st[2] += M; st[3] += u_max; st[4] += ub_best; st[5] += 1ull;
        if (ub_best < M) st[6] += 1ull;  // must never happen: a block's bound below one of its references' counts
    }
    if (p.phase == 1u) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- 4. live tiles of either query (a mask per QUERY: the wave of a (pair, tile) block folds the rows of a query only where its
    // tile is live, and leaves at once where neither is), the references never counted
    unsigned long long dead_refs[2] = {0, 0};
    uint32_t n_live = 0, n_qlive = 0;
    uint32_t n_rec[2] = {0u, 0u};  // wave-uniform: live tiles of either query
    for (uint32_t T0 = 0; T0 < p.ntiles; T0 += 64) {
        const uint32_t T = T0 + lane;
        bool la = false, lb = false;
        if (T < p.ntiles) {
            // a query without a threshold has every tile counted -- also those without any of its k-mers: taxon_prefix may
            // have to read them (every reference with count 0 can carry probability if the best hit is weak)
            la = thr[0] == 0u || (uint32_t)ub_lds[T] > thr[0];
            lb = has_b && (thr[1] == 0u || (uint32_t)ub_lds[p.ntiles + T] > thr[1]);
        }
        const unsigned long long ba = __ballot(la), bb2 = __ballot(lb);
        if (lane == 0) {
            uint32_t *wa = p.live + (size_t)(pair * 2u) * p.live_words, *wb = wa + p.live_words;
            wa[T0 >> 5] = (uint32_t)ba;
            if ((T0 >> 5) + 1u < p.live_words) wa[(T0 >> 5) + 1u] = (uint32_t)(ba >> 32);
            if (has_b) {
                wb[T0 >> 5] = (uint32_t)bb2;
                if ((T0 >> 5) + 1u < p.live_words) wb[(T0 >> 5) + 1u] = (uint32_t)(bb2 >> 32);
            }
        }
        if (T < p.ntiles) {
            const uint64_t lo = (uint64_t)T * 8192u, hi = lo + 8192u < p.n_refs ? lo + 8192u : p.n_refs;
            if (!la) dead_refs[0] += hi - lo;
            if (!lb) dead_refs[1] += hi - lo;
        }
        if (p.rec.nslots) {  // the records path: the live tiles of either query in ascending order (the segments of its records, RecordRef)
            const unsigned long long lt = (1ull << lane) - 1ull;
            const uint32_t ka = n_rec[0] + (uint32_t)__popcll(ba & lt), kb = n_rec[1] + (uint32_t)__popcll(bb2 & lt);
            if (la && ka < kRecMaxSlots) p.rec.slots[(size_t)(pair * 2u) * kRecMaxSlots + ka] = (uint16_t)T;
            if (lb && kb < kRecMaxSlots) p.rec.slots[(size_t)(pair * 2u + 1u) * kRecMaxSlots + kb] = (uint16_t)T;
            n_rec[0] += (uint32_t)__popcll(ba);
            n_rec[1] += (uint32_t)__popcll(bb2);
        }
        n_live += (uint32_t)__popcll(ba | bb2);
        n_qlive += (uint32_t)__popcll(ba) + (uint32_t)__popcll(bb2);
    }
    if (p.rec.nslots) {  // a query with a threshold and few live tiles: hit_count writes records, records_tail_kernel reads them
        const uint32_t cap = p.rec_max_slots < kRecMaxSlots ? p.rec_max_slots : kRecMaxSlots;
        const bool ra = thr[0] != 0u && n_rec[0] != 0u && n_rec[0] <= cap, rb = has_b && thr[1] != 0u && n_rec[1] != 0u && n_rec[1] <= cap;
        if (lane < kRecMaxSlots) {
            p.rec.cnt[(size_t)(pair * 2u) * kRecMaxSlots + lane] = 0u;
            if (has_b) p.rec.cnt[(size_t)(pair * 2u + 1u) * kRecMaxSlots + lane] = 0u;
        }
        if (lane == 0) {
            p.rec.nslots[pair * 2u] = (uint16_t)(ra ? n_rec[0] : 0u);
            if (has_b) p.rec.nslots[pair * 2u + 1u] = (uint16_t)(rb ? n_rec[1] : 0u);
            if (p.cnt_row) {  // the queries that take the dense epilogues get a row of the counts buffer (HitParams::cnt_row)
                auto row_for = [&](bool records) -> uint32_t {
                    if (records) return 0xFFFFFFFFu;
                    const uint32_t r = atomicAdd(p.cnt_cursor, 1u);
                    if (r < p.cnt_cap) return r;
                    atomicOr(p.flags_out, 4u);  // the rows ran out: the host enlarges the buffer and repeats the run
                    return 0xFFFFFFFFu;
                };
                p.cnt_row[pair * 2u] = row_for(ra);
                if (has_b) p.cnt_row[pair * 2u + 1u] = row_for(rb);
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { dead_refs[0] += __shfl_xor(dead_refs[0], d, 64); dead_refs[1] += __shfl_xor(dead_refs[1], d, 64); }
    if (lane == 0) {
        p.hist[(size_t)(pair * 2u) * p.hstride] = (uint32_t)dead_refs[0];  // kmer_extract has zeroed the row; hit_count adds the counted ones
        if (has_b) p.hist[(size_t)(pair * 2u + 1u) * p.hstride] = (uint32_t)dead_refs[1];
    }
    if (lane == 0 && p.pair_live) p.pair_live[pair] = n_live;
    st[7] = n_qlive;  // (query, tile) combinations that are counted
    if (p.stats) {  // one atomic instruction per wave (lane k adds counter k), 64 copies of the counters in lines of their own:
                    // thousands of waves adding to ONE address queue up in L2 for longer than everything else here takes
        st[0] = n_live; st[1] = 1ull;
        unsigned long long mine = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) mine = lane == (uint32_t)k ? st[k] : mine;
        if (lane < 8u && mine) atomicAdd(&p.stats[(size_t)(pair & (kPruneStatCopies - 1u)) * 8u + lane], mine);
    }
}

// The block-major copy of the database bitmap (PruneParams::cbitmap) from the tile-major one: a workgroup per (row, tile), a thread per word.
// Byte k of word (lane l, word wi) of tile T holds the references 8192 T + ((4 wi + k) L + l) 8 + [0, 8) (L = lanes of the tile): one byte of
// the 8-byte entry of their block.  The target is zeroed first; only the non-zero bytes are written.  (A grid of two dimensions: the words of
// a database of millions of references are more than the 2^32 work-items one dimension may hold.)
__global__ __launch_bounds__(256) void block_major_build_kernel(const uint32_t *__restrict__ bitmap, uint32_t n_rows1, uint32_t stride_bytes,
                                                                uint8_t *__restrict__ cbitmap) {
    const uint32_t row = blockIdx.x, tile = blockIdx.y, word = threadIdx.x;
    const uint32_t v = bitmap[((size_t)tile * n_rows1 + row) * 256u + word];
    if (v == 0u) return;
    const uint32_t l = word >> 2, wi = word & 3u, L = tile_lanes(stride_bytes, tile);
    if (l >= L) return;
#pragma unroll
    for (uint32_t k = 0; k < 4u; k++) {
        const uint32_t byte = (v >> (8u * k)) & 0xFFu;
        if (byte == 0u) continue;
        const uint32_t ref0 = tile * 8192u + ((wi * 4u + k) * L + l) * 8u;
        cbitmap[((size_t)(ref0 >> 6) * n_rows1 + row) * 8u + ((ref0 & 63u) >> 3)] = (uint8_t)byte;
    }
}
void launch_block_major_build(hipStream_t s, const uint32_t *bitmap, uint32_t n_rows1, uint32_t ntiles, uint32_t stride_bytes, uint8_t *cbitmap) {
    hipLaunchKernelGGL(block_major_build_kernel, dim3(n_rows1, ntiles), dim3(256), 0, s, bitmap, n_rows1, stride_bytes, cbitmap);
}

void launch_prune(hipStream_t s, const PruneParams &p, const ProbTables &tb, uint32_t nq) {
    const uint32_t pairs = (nq + 1u) / 2u;
    hipLaunchKernelGGL(prune_kernel, dim3((pairs + kPruneWavesPerBlock - 1u) / kPruneWavesPerBlock), dim3(64 * kPruneWavesPerBlock),
                       (size_t)p.nlf * sizeof(double) + (size_t)kPruneWavesPerBlock * 2 * p.ntiles * sizeof(uint16_t), s, p, tb);
}

}  // namespace rtx
