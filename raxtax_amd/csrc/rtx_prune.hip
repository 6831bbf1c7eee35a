// Tile pruning (RTX_OPT_TILE_PRUNE): which tiles of 8192 references hit_count has to visit for a pair of queries.
//
// Most references of a large database share far too few k-mers with a query to receive any probability (DESIGN.md
// section 8: dropping every reference of the tiles whose largest count stays below 300 changes no prefix sum of the
// reference algorithm by more than rounding noise).  hit_count can skip such a tile if it KNOWS beforehand that every
// count in it is small:
//   * upper bounds: the union bitmap -- one column per block of 2^kPruneShift consecutive references, bit (k, block) set
//     if ANY reference of the block contains k-mer k -- is counted like the database itself (hit_count_kernel on two
//     tiles instead of 62); a block's count bounds the count of each of its references.  ub(T) = max over the blocks of
//     tile T.
//   * a lower bound M of the best hit: the exact count of one reference of the block with the largest bound.
//   * the threshold (prob.rs:49-90 restated): Z = sum_r table[m_r] >= 1, table[m_r] = sum_i pmf_{m_r}(i) prod_{r' != r}
//     cmf_{m_r'}(i), cmf_m(i) falls with m.  (1) All of Z's mass at i <= i* is at most cmf_M(i*) (1 + N (i* + 1)) =: delta
//     (the best reference's own term is at most its cmf, every other term carries the best reference's cmf as a factor).
//     (2) For i > i* the references of skipped tiles (counts <= u) multiply prod by at least 1 - N tail_u(i*), and
//     hold at most N tail_u(i*) of probability themselves; tail_u(i*) <= (n - i*) pmf_u(i* + 1) once pmf_u falls.
//     i* = the largest i with delta <= eps, u = the largest count with N (n - i*) pmf_u(i* + 1) <= eps: every probability
//     and every prefix sum of the pruned run is within a few eps of the full one.  eps = 1e-12 (north_star asks for 1e-6).
//     With a full-overlap reference (M = t) prob.rs:24-41 applies: table[m] = pmf_m(n), table[t] = 1: u = the largest
//     count with N pmf_u(n) <= eps.
// A tile is dead for a query if ub(T) <= u; a (pair, tile) block of hit_count_pair_kernel leaves at once if the tile
// is dead for both queries.  The references that are never counted are booked into histogram bin 0 (the bin takes part
// in nothing but the global signal).
#include <hip/hip_runtime.h>

#include "rtx_kernels.hpp"
#include "rtx_math.hpp"
#include "rtx_wave.hpp"

namespace rtx {

static constexpr double kPruneLnEps = -27.631021115928547;  // ln 1e-12

__device__ __forceinline__ uint32_t wave_max_u32p(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)v, d, 64);
        v = o > v ? o : v;
    }
    return v;
}

__global__ __launch_bounds__(64) void prune_kernel(PruneParams p, ProbTables tb) {
    extern __shared__ uint16_t ub_lds[];  // [2][ntiles]
    const uint32_t pair = blockIdx.x, lane = threadIdx.x;
    const uint32_t bpt = 8192u >> p.shift;  // blocks per tile
    const double ln_n = log((double)p.n_refs);
    uint32_t thr[2] = {0u, 0u};
    const bool has_b = pair * 2u + 1u < p.nq;
    for (uint32_t x = 0; x < 2u; x++) {
        if (x == 1u && !has_b) break;  // wave-uniform
        const uint32_t q = pair * 2u + x;
        // ---- 1. bound of every tile, and the block with the largest bound.  The counts of the blocks are read as they
        // lie (a wave takes 512 consecutive blocks per turn, 8 per lane): with 32 blocks per reference... per tile of 8192
        // references 8192 >> shift blocks = bpt / 8 lanes; the lanes of a tile meet through DPP-free shuffles
        const uint16_t *uc = p.ucounts + (size_t)q * p.unpad;
        const uint32_t lpt = bpt / 8u;  // lanes per tile in a turn (32 for blocks of 32 references): a power of two <= 64
        uint32_t lmx = 0, lblk = 0;  // this lane's largest bound and its block
        const uint32_t n_blocks_pad = p.ntiles * bpt;
        for (uint32_t b0 = 0; b0 < n_blocks_pad; b0 += 512u) {
            const uint32_t blk = b0 + lane * 8u;
            uint32_t mx = 0, arg = 0;
            if (blk < n_blocks_pad) {
                const uint4 v = *reinterpret_cast<const uint4 *>(uc + blk);
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const uint32_t c = (w[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu;
                    if (c > mx) { mx = c; arg = (uint32_t)j; }
                }
            }
            if (mx > lmx) { lmx = mx; lblk = blk + arg; }
            // the largest count of the tile: over the lpt lanes that hold its blocks
            uint32_t tm = mx;
            for (uint32_t d = 1; d < lpt; d <<= 1) {
                const uint32_t o = (uint32_t)__shfl_xor((int)tm, (int)d, 64);
                tm = o > tm ? o : tm;
            }
            const uint32_t T = blk / bpt;
            if ((lane & (lpt - 1u)) == 0u && T < p.ntiles) ub_lds[x * p.ntiles + T] = (uint16_t)tm;
        }
        const uint32_t ub_best = wave_max_u32p(lmx);
        // the lowest block among those with the largest bound (0 if every bound is 0)
        const uint32_t bb = ub_best ? 0xFFFFFFFFu - wave_max_u32p(lmx == ub_best ? 0xFFFFFFFFu - lblk : 0u) : 0u;
        // ---- 2. exact counts of its references (not those that --skip-exact-matches zeroes); M = the best of them.  The
        // block's references lie in (1 << shift) / 8 chunks of eight = that many bytes of a row segment (ref_slot,
        // rtx_math.hpp); lane l of a turn takes row i0 + l and gathers those bytes, the hits of the eight references of a byte
        // are summed bit-sliced (eight 8-bit counters in two words, flushed to 16 bits every 255 rows).
        uint32_t M = 0;
        {
            const uint32_t nchunk = (1u << p.shift) / 8u;  // 4 for blocks of 32
            const uint32_t nr = p.nrows[q];
            const uint32_t *rows = p.rows + (size_t)q * p.rstride;
            const uint64_t qin = p.perm[p.q0 + q];
            for (uint32_t c = 0; c < nchunk; c++) {
                const uint64_t r0 = ((uint64_t)bb << p.shift) + (uint64_t)c * 8u;
                if (r0 >= p.n_refs) break;  // wave-uniform
                uint32_t word, bit;
                ref_slot((uint32_t)r0, p.stride_bytes, word, bit);  // bit = first bit of the chunk's byte
                // per lane: hits of the eight references over this lane's rows, one 16-bit counter each in four words
                uint32_t acc[4] = {0u, 0u, 0u, 0u};
                for (uint32_t i0 = 0; i0 < nr; i0 += 64) {  // the row list is padded with the all-zero row to whole chunks of 64
                    const uint32_t row = rows[i0 + lane];
                    const uint32_t byte = (p.bitmap[bitmap_word(row, word, p.n_rows1)] >> bit) & 0xFFu;
#pragma unroll
                    for (int k = 0; k < 4; k++) acc[k] += ((byte >> (2 * k)) & 1u) | (((byte >> (2 * k + 1)) & 1u) << 16);
                }
#pragma unroll
                for (int k = 0; k < 4; k++)
#pragma unroll
                    for (int d = 32; d >= 1; d >>= 1) acc[k] += (uint32_t)__shfl_xor((int)acc[k], d, 64);  // at most 16 rows per lane x 64 lanes: fits 16 bits
                // the chunk's eight references: drop those behind the end and those --skip-exact-matches zeroes
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const uint64_t r = r0 + (uint64_t)j;
                    bool ok = r < p.n_refs;
                    if (ok && (p.flags & RTX_SKIP_EXACT_MATCHES)) {
                        bool hit = false;
                        for (uint64_t e = p.exact_off[qin] + lane; e < p.exact_off[qin + 1]; e += 64) hit = hit || (uint64_t)p.exact_ids[e] == r;
                        ok = __ballot(hit) == 0ull;
                    }
                    const uint32_t cnt = (acc[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu;
                    if (ok && cnt > M) M = cnt;
                }
            }
        }
#ifdef RTX_PRUNE_CHECK  // debug: the count of the best block recomputed from the union bitmap must equal what the counting pass left
        if (p.ubitmap) {
            uint32_t word, bit;
            ref_slot(bb, p.ustride_bytes, word, bit);
            const uint32_t nr = p.nrows[q];
            const uint32_t *rows = p.rows + (size_t)q * p.rstride;
            uint32_t cub = 0;
            for (uint32_t i0 = 0; i0 < nr; i0 += 64) {
                const uint32_t row = rows[i0 + lane];
                cub += (p.ubitmap[bitmap_word(row, word, p.n_rows1)] >> bit) & 1u;
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) cub += (uint32_t)__shfl_xor((int)cub, d, 64);
            if (lane == 0 && p.stats && cub != (uint32_t)uc[bb]) atomicAdd(&p.stats[7], 1ull);
        }
#endif
        // ---- 3. the largest count a skipped tile may hold
        const uint32_t t = p.t[q], n = t >> 1;
        uint32_t u_max = 0;
        if (t >= 16u && t <= tb.tmax && M >= 1u && n >= 2u) {
            const double *lf = p.lnfact;
            const double ln_total = ln_binom_tab(lf, t + n - 1, n);
            if (M >= t) {  // full overlap: table[m] = pmf_m(n) = C(m+n-1, n) / C(t+n-1, n), rising with m
                uint32_t mine = 0;
                for (uint32_t u = 1u + lane; u < t; u += 64)
                    if (ln_binom_tab(lf, u + n - 1, n) - ln_total + ln_n <= kPruneLnEps) mine = u;
                u_max = wave_max_u32p(mine);
            } else {
                const double *lc = tb.cmf + tb.off[t] + (size_t)M * (n + 1);  // ln cmf_M(i)
                uint32_t mine = 0xFFFFFFFFu;  // i* + 1 in the end (0: none)
                uint32_t ist1 = 0;
                for (uint32_t i = lane; i + 2u <= n; i += 64)  // i* <= n - 2: a tail is left
                    if (lc[i] + log(1.0 + (double)p.n_refs * (double)(i + 1u)) <= kPruneLnEps) ist1 = i + 1u;
                (void)mine;
                ist1 = wave_max_u32p(ist1);
                if (ist1) {
                    const uint32_t i1 = ist1;  // = i* + 1: the first i that stays
                    const double ln_len = log((double)(n - i1 + 1u));
                    // the condition holds for a prefix of the counts (below its mode pmf_u(i1) rises with u); taken as the
                    // counts below the smallest one that fails, whatever rounding does to the largest one that passes
                    uint32_t first_fail = M;  // counts from M on are never skipped
                    for (uint32_t u = 1u + lane; u < M; u += 64) {
                        // pmf_u falling from i1 on: (u + i1)(n - i1) < (i1 + 1)(t - u + n - i1 - 1)
                        const double up = (double)(u + i1) * (double)(n - i1), dn = (double)(i1 + 1u) * (double)(t - u + n - i1 - 1u);
                        if (!(up < dn && ln_len + ln_pmf_tab(lf, t, n, u, i1, ln_total) + ln_n <= kPruneLnEps)) first_fail = u < first_fail ? u : first_fail;
                    }
                    first_fail = 0xFFFFFFFFu - wave_max_u32p(0xFFFFFFFFu - first_fail);
                    u_max = first_fail - 1u;
                }
            }
        }
        thr[x] = u_max;
        if (lane == 0 && p.stats) {  // reporting: sums of the lower bound of the best hit, of the threshold, of the largest tile bound
            atomicAdd(&p.stats[2], (unsigned long long)M);
            atomicAdd(&p.stats[3], (unsigned long long)u_max);
            atomicAdd(&p.stats[4], (unsigned long long)ub_best);
            atomicAdd(&p.stats[5], 1ull);
            if (ub_best < M) atomicAdd(&p.stats[6], 1ull);  // must never happen: a block's bound below one of its references' counts
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- 4. live tiles of the pair, the references never counted
    unsigned long long dead_refs = 0;
    uint32_t n_live = 0;
    for (uint32_t T0 = 0; T0 < p.ntiles; T0 += 64) {
        const uint32_t T = T0 + lane;
        bool live = false;
        if (T < p.ntiles) {
            live = (uint32_t)ub_lds[T] > thr[0];
            if (has_b) live = live || (uint32_t)ub_lds[p.ntiles + T] > thr[1];
        }
        const unsigned long long bl = __ballot(live);
        if (lane == 0) {
            p.live[(size_t)pair * p.live_words + (T0 >> 5)] = (uint32_t)bl;
            if ((T0 >> 5) + 1u < p.live_words) p.live[(size_t)pair * p.live_words + (T0 >> 5) + 1u] = (uint32_t)(bl >> 32);
        }
        if (T < p.ntiles && !live) {
            const uint64_t lo = (uint64_t)T * 8192u, hi = lo + 8192u < p.n_refs ? lo + 8192u : p.n_refs;
            dead_refs += hi - lo;
        }
        n_live += (uint32_t)__popcll(bl);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) dead_refs += __shfl_xor(dead_refs, d, 64);
    if (lane == 0) {
        p.hist[(size_t)(pair * 2u) * p.hstride] = (uint32_t)dead_refs;  // kmer_extract has zeroed the row; hit_count adds the counted ones
        if (has_b) p.hist[(size_t)(pair * 2u + 1u) * p.hstride] = (uint32_t)dead_refs;
        if (p.stats) { atomicAdd(&p.stats[0], (unsigned long long)n_live); atomicAdd(&p.stats[1], 1ull); }
    }
}

void launch_prune(hipStream_t s, const PruneParams &p, const ProbTables &tb, uint32_t nq) {
    hipLaunchKernelGGL(prune_kernel, dim3((nq + 1u) / 2u), dim3(64), (size_t)2 * p.ntiles * sizeof(uint16_t), s, p, tb);
}

}  // namespace rtx
