// The bounds pass of the tile pruning in TWO LEVELS (round 5; rtx_prune.hip has the argument the bounds serve).
//
// The one-level pass (hit_count_pair_kernel<.., 1, ..>) counts every pair of queries against the union bitmap over blocks of 64
// references: ~845 rows of 1 KiB per pair and 512 Ki references, one load instruction and ~9 VALU operations per row and lane -- bound
// by the instructions it issues and by the L1 rate (a 1-KiB row per 16 cycles and CU), 18 of the 82 ms of a step at configs[2], 24 of
// 40 at N = 5 M.  Almost all of that work proves that tiles of UNRELATED clades are dead, which far coarser blocks prove as well:
//
//   level A   union bitmap over blocks of 256 references, rows of 256 bytes (an "A-tile": 2048 blocks = 64 tiles of the database):
//             FOUR rows per load instruction (the four DPP rows of the wave take a row each: lane = 16 (row of the instruction) + s,
//             s = which 16 bytes) -- a quarter of the load instructions and of the folds.  Result: a bound for every tile of the database.
//   level B   union bitmap over blocks of 64 (the bounds of the one-level pass), but stored in "B-tiles" of 4 tiles of the database =
//             rows of 64 bytes: SIXTEEN rows per load instruction (lane = 4 (row) + s).  Only for the B-tiles whose level-A bound comes
//             within `delta` of the query's largest: the query's own clade and its neighbours, one or two of the sixteen B-tiles of an
//             A-tile on the bench workload.  Their tiles get the tight bounds, and the best block of 64 is taken from them.
//
// The lanes that worked on different rows of an instruction hold partial counters: they are summed as bit-sliced numbers (full adders
// on the planes), the cross-row stages with ds_bpermute and each lane keeping half of what it held (two stages: 4 -> 2 -> 1 words),
// level B's four lanes of a row with DPP rotations.  Maxima are taken on the planes (bit by bit from the top: 4 operations per plane
// for 32 counters) -- nothing is unpacked.
//
// Which B-tiles are refined is a heuristic and decides nothing but time: EVERY tile ends with a valid upper bound (level A's or level
// B's), and prune_kernel derives threshold and live tiles from whatever bounds it is given.  A tile that would have been dead with the
// bound of level B and is left with level A's stays live and is counted (its epilogue then finds nothing above the threshold).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "rtx_hit_common.hpp"

namespace rtx {

constexpr uint32_t kB2Pad = 256u;                        // the largest group of rows of either level (level B: 16 instructions x 16 rows)
// entries per list (t <= 1023; eleven planes: t <= 2047), padded with the zero row to multiples of kB2Pad
constexpr uint32_t b2_cap(int planes) { return planes > 10 ? 2048u : 1024u; }
// lists | zero rows; behind them per A-tile [2][64] u16: the level-A bounds both queries' lanes found
constexpr uint32_t b2_lds_dw(int planes) { return 3u * b2_cap(planes) + kB2Pad; }

// eight load instructions of R rows each: this lane's rows are list[8 * grp .. + 8) of the unit, its bytes col .. col + 16 of each
template <int SHIFT>
__device__ __forceinline__ void load_unit(uint4 (&buf)[8], __amdgpu_buffer_rsrc_t rsrc, const uint32_t *unit, uint32_t grp, uint32_t col) {
    const uint4 i0 = reinterpret_cast<const uint4 *>(unit)[grp * 2u], i1 = reinterpret_cast<const uint4 *>(unit)[grp * 2u + 1u];
    const uint32_t id[8] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (id[j] << SHIFT) + col, 0, 0);
        buf[j] = make_uint4(v.x, v.y, v.z, v.w);
    }
}

// One list: `ng` groups of NB units (8 instructions of R rows each) folded into A's planes (MODE 1) or B's (MODE 2).  As fold_seg
// (rtx_hit_pair.hip): on entry the buffers hold the first group, every buffer is requested again as soon as it has been folded -- in
// the last group with the first rows of the next list --, so that the lists of a level run as one pipeline.
template <int NP, int MODE, int R, int NB, int SHIFT>
__device__ __forceinline__ void fold_list(uint32_t (&pa)[4][NP], uint32_t (&pb)[4][NP], uint4 (&buf)[4][8], const uint32_t *list, uint32_t ng,
                                          const uint32_t *next, uint32_t grp, uint32_t col, __amdgpu_buffer_rsrc_t rsrc) {
    constexpr uint32_t kUnit = 8u * (uint32_t)R, kGroup = kUnit * (uint32_t)NB;
    for (uint32_t g = 0; g < ng; g++) {
        const uint32_t *src = g + 1 < ng ? list + (g + 1) * kGroup : next;  // wave-uniform
        uint4 c[NB];
#pragma unroll
        for (int b = 0; b < NB; b++) {
            c[b] = MODE == 1 ? tree8<NP>(pa, buf[b]) : tree8<NP>(pb, buf[b]);
            load_unit<SHIFT>(buf[b], rsrc, src + (uint32_t)b * kUnit, grp, col);
        }
        if (MODE == 1) {
            if constexpr (NB == 4) ripple4<NP, 5>(pa, csa_plane<NP, 4>(pa, csa_plane<NP, 3>(pa, c[0], c[1]), csa_plane<NP, 3>(pa, c[2], c[3])));
            else ripple4<NP, 4>(pa, csa_plane<NP, 3>(pa, c[0], c[1]));
        } else {
            if constexpr (NB == 4) ripple4<NP, 5>(pb, csa_plane<NP, 4>(pb, csa_plane<NP, 3>(pb, c[0], c[1]), csa_plane<NP, 3>(pb, c[2], c[3])));
            else ripple4<NP, 4>(pb, csa_plane<NP, 3>(pb, c[0], c[1]));
        }
    }
}

// The four DPP rows of the wave hold partial counters of the same columns: summed over the rows, DPP row r ends with word r of its
// lane's 16 bytes (two halving exchanges: with lane ^ 32 a lane keeps words {0, 1} or {2, 3}, with lane ^ 16 one of the two).
template <int NP>
__device__ __forceinline__ void reduce_rows(const uint32_t (&pl)[4][NP], uint32_t lane, uint32_t (&out)[NP]) {
    const bool hi = (lane & 32u) != 0u, mid = (lane & 16u) != 0u;
    uint32_t k0[NP], k1[NP];
    {
        uint32_t s0[NP], s1[NP];
#pragma unroll
        for (int p = 0; p < NP; p++) {
            k0[p] = hi ? pl[2][p] : pl[0][p];
            k1[p] = hi ? pl[3][p] : pl[1][p];
            s0[p] = (uint32_t)__shfl_xor((int)(hi ? pl[0][p] : pl[2][p]), 32, 64);
            s1[p] = (uint32_t)__shfl_xor((int)(hi ? pl[1][p] : pl[3][p]), 32, 64);
        }
        planes_add<NP>(k0, s0);
        planes_add<NP>(k1, s1);
    }
    uint32_t s[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) {
        out[p] = mid ? k1[p] : k0[p];
        s[p] = (uint32_t)__shfl_xor((int)(mid ? k0[p] : k1[p]), 16, 64);
    }
    planes_add<NP>(out, s);
}

__device__ __forceinline__ uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }

template <int NP>
__global__ __launch_bounds__(64, 2) void bounds2_kernel(Bounds2Params p) {
    extern __shared__ uint32_t b2_lds[];
    const uint32_t lane = threadIdx.x;
    // pairs are dealt to the XCDs as contiguous slices of the sub-batch (neighbours share rows: the XCD's L2)
    const uint32_t np8 = gridDim.x >> 3;
    const uint32_t pair = blockIdx.x < np8 * 8u ? (blockIdx.x & 7u) * np8 + (blockIdx.x >> 3) : blockIdx.x;
    const uint32_t qa = pair * 2u, qb = qa + 1u;
    const bool has_b = qb < p.nq;
    constexpr uint32_t kB2Cap = b2_cap(NP);
    uint32_t *l_both = b2_lds, *l_a = b2_lds + kB2Cap, *l_b = b2_lds + 2u * kB2Cap, *l_zero = b2_lds + 3u * kB2Cap;
    uint16_t *l_keep = reinterpret_cast<uint16_t *>(l_zero + kB2Pad);  // [n_atiles][2][64] level-A bound of every lane's tile, per query
    for (uint32_t i = lane; i < kB2Pad; i += 64) l_zero[i] = p.zero_row;

    // ---- the three lists: rows of both queries, of A only, of B only (every row of a query counts: a union bitmap has no classes)
    const uint2 *urec = p.pair_urec + (size_t)pair * p.pair_ustride;
    const uint32_t n_u = std::min<uint32_t>(p.pair_nu[pair], p.pair_ustride);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t n_both = 0, n_a = 0, n_b = 0;
    for (uint32_t u0 = 0; u0 < n_u; u0 += 512u) {  // eight chunks of 64 union entries per round trip
        uint2 rec[8];
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const uint32_t i = u0 + (uint32_t)c * 64u + lane;
            rec[c] = urec[i < p.pair_ustride ? i : p.pair_ustride - 1u];
        }
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const uint32_t i = u0 + (uint32_t)c * 64u + lane;
            const bool in = i < n_u;
            const bool ia = in && (rec[c].x & (1u << 30)), ib = in && (rec[c].x & (1u << 31)) && has_b;
            const uint32_t row = rec[c].x & 0x3FFFFFFFu;
            const bool sh = ia && ib, oa = ia && !ib, ob = ib && !ia;
            const unsigned long long bs = __ballot(sh), ba = __ballot(oa), bo = __ballot(ob);
            // (a list never overflows: t <= 1023 / 2047 rows per query; the guard keeps a corrupt union from writing beyond it)
            const uint32_t ps = n_both + (uint32_t)__popcll(bs & lt_mask), pa_ = n_a + (uint32_t)__popcll(ba & lt_mask), pb_ = n_b + (uint32_t)__popcll(bo & lt_mask);
            if (sh && ps < kB2Cap) l_both[ps] = row;
            if (oa && pa_ < kB2Cap) l_a[pa_] = row;
            if (ob && pb_ < kB2Cap) l_b[pb_] = row;
            n_both += (uint32_t)__popcll(bs);
            n_a += (uint32_t)__popcll(ba);
            n_b += (uint32_t)__popcll(bo);
        }
    }
    n_both = std::min(n_both, kB2Cap);
    n_a = std::min(n_a, kB2Cap);
    n_b = std::min(n_b, kB2Cap);
    for (uint32_t i = n_both + lane; i < ((n_both + kB2Pad - 1u) & ~(kB2Pad - 1u)); i += 64) l_both[i] = p.zero_row;
    for (uint32_t i = n_a + lane; i < ((n_a + kB2Pad - 1u) & ~(kB2Pad - 1u)); i += 64) l_a[i] = p.zero_row;
    for (uint32_t i = n_b + lane; i < ((n_b + kB2Pad - 1u) & ~(kB2Pad - 1u)); i += 64) l_b[i] = p.zero_row;
    wave_lds_sync();

    const uint32_t t_a = p.t[qa], t_b = has_b ? p.t[qb] : 0u;
    uint16_t *tub_a = p.tile_ub + (size_t)qa * p.tile_ub_stride, *tub_b = p.tile_ub + (size_t)(has_b ? qb : qa) * p.tile_ub_stride;
    uint32_t pa[4][NP], pb[4][NP];
    uint4 buf[4][8];
    uint32_t n_instr = 0;  // load instructions of this wave (1 KiB each): the work accounting of the roofline

    // one level: the three lists against the region `rsrc` of a bitmap, R rows per load instruction
    auto fold_level = [&](auto r_tag, auto nb_tag, auto shift_tag, __amdgpu_buffer_rsrc_t rsrc, uint32_t grp, uint32_t col, bool need_a, bool need_b) {
        constexpr int R = decltype(r_tag)::value, NB = decltype(nb_tag)::value, SHIFT = decltype(shift_tag)::value;
        constexpr uint32_t kGroup = 8u * (uint32_t)R * (uint32_t)NB;
#pragma unroll
        for (int w = 0; w < 4; w++)
#pragma unroll
            for (int b = 0; b < NP; b++) { pa[w][b] = 0; pb[w][b] = 0; }
        // (the rows of one query alone are left out where only the other one asked for the bounds)
        const uint32_t g_both = (n_both + kGroup - 1u) / kGroup, g_a = need_a ? (n_a + kGroup - 1u) / kGroup : 0u, g_b = need_b ? (n_b + kGroup - 1u) / kGroup : 0u;
        const uint32_t *first = g_both ? l_both : (g_a ? l_a : (g_b ? l_b : nullptr));
        if (!first) return;
#pragma unroll
        for (int b = 0; b < NB; b++) load_unit<SHIFT>(buf[b], rsrc, first + (uint32_t)b * 8u * (uint32_t)R, grp, col);
        const uint32_t *after_a = g_b ? l_b : l_zero, *after_both = g_a ? l_a : after_a;
        if (g_both) {  // the rows both queries share are folded once, into A's planes while B's are empty, and copied
            fold_list<NP, 1, R, NB, SHIFT>(pa, pb, buf, l_both, g_both, after_both, grp, col, rsrc);
#pragma unroll
            for (int w = 0; w < 4; w++)
#pragma unroll
                for (int b = 0; b < NP; b++) pb[w][b] = pa[w][b];
        }
        if (g_a) fold_list<NP, 1, R, NB, SHIFT>(pa, pb, buf, l_a, g_a, after_a, grp, col, rsrc);
        if (g_b) fold_list<NP, 2, R, NB, SHIFT>(pa, pb, buf, l_b, g_b, l_zero, grp, col, rsrc);
        n_instr += (g_both + g_a + g_b) * 8u * (uint32_t)NB;
    };

    uint32_t best_a = 0xFFFFFu, best_b = 0xFFFFFu;  // key of the best block of 64: bound << 20 | (0xFFFFF - block); block 0 with bound 0 to start with
    const uint32_t row4 = lane >> 4;  // DPP row
    // ---- level A, every A-tile: bounds of all tiles; the largest of them per query
    uint32_t max_a = 0, max_b = 0;
    for (uint32_t at = 0; at < p.n_atiles; at++) {
        const char *base = reinterpret_cast<const char *>(p.abitmap) + (size_t)at * p.n_rows1 * 256u;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, p.n_rows1 * 256u, 0x00027000);
        fold_level(std::integral_constant<int, 4>{}, std::integral_constant<int, 4>{}, std::integral_constant<int, 8>{}, rsrc, row4, (lane & 15u) * 16u, true, has_b);
        uint32_t ra[NP], rb[NP], cand;
        reduce_rows<NP>(pa, lane, ra);
        reduce_rows<NP>(pb, lane, rb);
        const uint32_t T = at * 64u + (lane & 15u) * 4u + row4;  // word r of the lane's 16 bytes = 32 blocks of 256 = one tile
        const uint32_t ma = planes_max<NP>(ra, cand), mb = planes_max<NP>(rb, cand);
        if (T < p.ntiles) {
            tub_a[T] = (uint16_t)ma;
            if (has_b) tub_b[T] = (uint16_t)mb;
        }
        max_a = umax(max_a, ma);
        max_b = umax(max_b, mb);
        l_keep[at * 128u + lane] = (uint16_t)ma;  // for the second pass
        l_keep[at * 128u + 64u + lane] = (uint16_t)mb;
    }
    max_a = wave_max_u32(max_a);
    max_b = wave_max_u32(max_b);
    wave_lds_sync();
    // B-tiles within `dl` counts of the query's largest level-A bound are refined.  The gap between that bound and the threshold grows as
    // the best hit weakens (tools/study_two_level.py, t ~ 640: 0.29 t at 2 % divergence, 0.36 t at 5 %, 0.42 t at 10 %):
    // dl = c_t t - c_m max, kept within [lo t, hi t] (all in 1/256; defaults in rtx_index.hpp: 1.105 t - 0.8 max within [0.36 t, 0.5 t])
    auto delta_of = [&](uint32_t t, uint32_t mx) -> uint32_t {
        const uint32_t up = p.delta_ct * t, dn = p.delta_cm * mx;
        const uint32_t d = up > dn ? (up - dn) >> 8 : 0u, lo = (p.delta_lo * t) >> 8, hi = (p.delta_hi * t) >> 8;
        return d < lo ? lo : (d > hi ? hi : d);
    };
    // (at least 1: the B-tile of the largest bound is always refined -- the best block, whose exact counts give the threshold, comes from level B)
    const uint32_t dl_a = umax(delta_of(t_a, max_a), 1u), dl_b = umax(delta_of(t_b, max_b), 1u);

    // ---- level B: the B-tiles (4 tiles of the database: the four words of sub-lane s at level A) near the largest bound
    const uint32_t sub4 = lane & 3u, grp16 = lane >> 2;
    // the B-tiles of A-tile `at` the rule of either query asks for (wave-uniform masks: bit s <-> B-tile at * 16 + s)
    auto wanted = [&](uint32_t at, uint32_t &mask_a, uint32_t &mask_b) {
        uint32_t ma = l_keep[at * 128u + lane], mb = l_keep[at * 128u + 64u + lane];
        // the largest bound of the B-tile of sub-lane s = lane & 15: over the four DPP rows
        ma = umax(ma, (uint32_t)__shfl_xor((int)ma, 16, 64));
        ma = umax(ma, (uint32_t)__shfl_xor((int)ma, 32, 64));
        mb = umax(mb, (uint32_t)__shfl_xor((int)mb, 16, 64));
        mb = umax(mb, (uint32_t)__shfl_xor((int)mb, 32, 64));
        mask_a = (uint32_t)(__ballot(ma + dl_a > max_a && ma != 0u && lane < 16u) & 0xFFFFull);
        mask_b = has_b ? (uint32_t)(__ballot(mb + dl_b > max_b && mb != 0u && lane < 16u) & 0xFFFFull) : 0u;
    };
    // A query far from its best hit has EVERY tile near its largest bound: sixteen folds of level B per A-tile would cost twice the one-level
    // pass.  Such a query ("heavy": more than heavy_max B-tiles; its own rule alone decides -- a result stays a function of the query) keeps
    // level A's bounds here and is handed to the one-level pass over blocks of 64 (launch_bounds2), which overwrites them.
    bool heavy_a = false, heavy_b = false;
    if (p.heavy) {
        uint32_t na = 0, nb = 0;
        for (uint32_t at = 0; at < p.n_atiles; at++) {
            uint32_t mask_a, mask_b;
            wanted(at, mask_a, mask_b);
            na += (uint32_t)__popc(mask_a);
            nb += (uint32_t)__popc(mask_b);
        }
        heavy_a = na > p.heavy_max;
        heavy_b = nb > p.heavy_max;
        if (lane == 0u) {
            p.heavy[qa] = heavy_a ? 1u : 0u;
            if (has_b) p.heavy[qb] = heavy_b ? 1u : 0u;
        }
    }
    for (uint32_t at = 0; at < p.n_atiles; at++) {
        // What a query is left with is a function of the query alone (a result never depends on the rest of the batch): the fold of a
        // B-tile serves both queries of the pair, but each takes the bounds of level B only where ITS OWN rule asked for them.
        uint32_t mask_a, mask_b;
        wanted(at, mask_a, mask_b);
        mask_a = heavy_a ? 0u : mask_a;
        mask_b = heavy_b ? 0u : mask_b;
        uint32_t todo = mask_a | mask_b;
        while (todo) {
            const uint32_t s = (uint32_t)__builtin_ctz(todo);
            todo &= todo - 1u;
            const uint32_t bt = at * 16u + s;
            if (bt * 4u >= p.ntiles) break;  // (behind the database: its level-A bounds are 0, never wanted -- a guard)
            const bool for_a = (mask_a >> s) & 1u, for_b = (mask_b >> s) & 1u;
            const char *base = reinterpret_cast<const char *>(p.bbitmap) + (size_t)bt * p.n_rows1 * 64u;
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, p.n_rows1 * 64u, 0x00027000);
            fold_level(std::integral_constant<int, 16>{}, std::integral_constant<int, 2>{}, std::integral_constant<int, 6>{}, rsrc, grp16, sub4 * 16u, for_a, for_b);
            const uint32_t T = bt * 4u + sub4;
            const bool real = T < p.ntiles;
            // the sixteen rows of an instruction: the DPP rows by halving (-> word `row4` of the tile bt * 4 + sub4), then the four
            // lanes of a DPP row that share a sub-lane (lane, lane + 4, + 8, + 12) by rotations; then the largest of the word's 32
            // counters, the tile's over the four words: key = bound << 8 | 127 - block within the tile (the lowest block among equals)
            auto finish = [&](const uint32_t (&pl)[4][NP], uint16_t *tub, uint32_t &best) {
                uint32_t r[NP];
                reduce_rows<NP>(pl, lane, r);
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    uint32_t sh[NP];
#pragma unroll
                    for (int pp = 0; pp < NP; pp++)
                        sh[pp] = k == 0 ? (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r[pp], 0x124, 0xF, 0xF, true)   // row_ror:4
                                        : (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r[pp], 0x128, 0xF, 0xF, true);  // row_ror:8
                    planes_add<NP>(r, sh);
                }
                uint32_t c;
                const uint32_t w = planes_max<NP>(r, c);
                uint32_t key = (w << 8) | (127u - (row4 * 32u + (uint32_t)__builtin_ctz(c)));
                key = umax(key, (uint32_t)__shfl_xor((int)key, 16, 64));
                key = umax(key, (uint32_t)__shfl_xor((int)key, 32, 64));
                if (real && lane < 4u) tub[T] = (uint16_t)(key >> 8);
                const uint32_t g = real ? ((key >> 8) << 20) | (0xFFFFFu - (T * 128u + (127u - (key & 0xFFu)))) : 0u;
                best = umax(best, wave_max_u32(g));
            };
            if (for_a) finish(pa, tub_a, best_a);  // wave-uniform
            if (for_b) finish(pb, tub_b, best_b);
        }
    }
    if (lane == 0u) {
        p.best_key[qa] = heavy_a ? 0u : best_a;  // (a heavy query: the one-level pass meets in an atomicMax on it)
        if (has_b) p.best_key[qb] = heavy_b ? 0u : best_b;
        if (p.group_rows) atomicAdd(&p.group_rows[p.group_base + pair], n_instr);
    }
}

// The (pair, union tile) items of the one-level pass for the pairs with a heavy query, in pair order.  One workgroup; a thread takes 32
// neighbouring pairs (64 flag bytes in four loads), so that a sub-batch of 65 536 queries is ONE pass: a scan per 1024 pairs was a chain
// of 31 dependent round trips, 0.1 ms per sub-batch on a launch that finds nothing on the bench workload.
__global__ __launch_bounds__(1024) void heavy_items_kernel(const uint8_t *__restrict__ heavy, uint32_t nq, uint32_t u_ntiles, uint32_t *__restrict__ items,
                                                           uint32_t *__restrict__ n_items) {
    __shared__ uint32_t wsum[2][16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, np = (nq + 1u) / 2u;
    constexpr uint32_t kPer = 32;
    uint32_t carry = 0, buf = 0;
    for (uint32_t base = 0; base < np; base += 1024u * kPer, buf ^= 1u) {
        const uint32_t p0 = base + tid * kPer;
        uint32_t bits = 0;  // bit j: pair p0 + j has a heavy query
        if (p0 < np) {
            if (p0 + kPer <= np && 2u * (p0 + kPer) <= nq) {  // 64 flags of 32 whole pairs (the scratch is 16-byte aligned, p0 a multiple of 32)
                const uint4 *f = reinterpret_cast<const uint4 *>(heavy + (size_t)p0 * 2u);
                const uint4 v[4] = {f[0], f[1], f[2], f[3]};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t w[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
                    for (int j = 0; j < 4; j++) {  // a word = four flags = two pairs
                        if (w[j] & 0x0000FFFFu) bits |= 1u << (k * 8 + j * 2);
                        if (w[j] & 0xFFFF0000u) bits |= 1u << (k * 8 + j * 2 + 1);
                    }
                }
            } else {
                for (uint32_t j = 0; j < kPer && p0 + j < np; j++) {
                    const uint32_t pr = p0 + j;
                    if (heavy[pr * 2u] || (pr * 2u + 1u < nq && heavy[pr * 2u + 1u])) bits |= 1u << j;
                }
            }
        }
        const uint32_t cnt = (uint32_t)__popc(bits);
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
        while (bits) {
            const uint32_t j = (uint32_t)__builtin_ctz(bits);
            bits &= bits - 1u;
            for (uint32_t ut = 0; ut < u_ntiles; ut++) items[(size_t)o * u_ntiles + ut] = (p0 + j) * u_ntiles + ut;
            o++;
        }
        carry += tot;
    }
    if (tid == 0u) n_items[0] = carry * u_ntiles;
    if (tid >= 1u && tid <= 8u) n_items[tid] = 0;  // the queues of the XCDs (hit_count_pair_kernel)
}

void launch_bounds2(hipStream_t s, const Bounds2Params &p, uint32_t nq, int planes, const HitParams &hp, uint32_t u_ntiles, uint32_t *items) {
    const uint32_t np = (nq + 1u) / 2u;
    const size_t lds = (size_t)b2_lds_dw(planes) * 4u + (size_t)p.n_atiles * 256u;
    if (planes <= 8) hipLaunchKernelGGL((bounds2_kernel<8>), dim3(np), dim3(64), lds, s, p);
    else if (planes <= 10) hipLaunchKernelGGL((bounds2_kernel<10>), dim3(np), dim3(64), lds, s, p);
    else hipLaunchKernelGGL((bounds2_kernel<11>), dim3(np), dim3(64), lds, s, p);  // reads of 1 031 .. 2 054 bases (lists of 2 048 rows: six workgroups per CU by LDS)
    if (!p.heavy || !items) return;
    // the heavy queries: the one-level pass over the union bitmap of blocks of 64 (on the bench workload an all but empty launch)
    uint32_t *n_items = items + (size_t)np * u_ntiles;
    hipLaunchKernelGGL(heavy_items_kernel, dim3(1), dim3(1024), 0, s, p.heavy, nq, u_ntiles, items, n_items);
    HitParams up = hp;
    up.items = items;
    up.n_items = n_items;
    up.bounds_heavy = p.heavy;
    launch_hit_count_pair_bounds_items(s, up, nq, u_ntiles, planes);
}

// ---------------------------------------------------------------------------
// The two bitmaps from the union bitmap over blocks of 64 (tile-major, ref_slot layout; every tile of it is full: 64 lanes): a thread
// per word.  Byte k of word (lane l, word wi) holds the blocks tile * 8192 + ((wi * 4 + k) * 64 + l) * 8 + [0, 8): one byte of a
// B row (plain bit order: blocks 512 bt .. + 512 in 64 bytes), two bits of an A row (blocks of 256 = four of them).  Both zeroed first.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bounds2_build_kernel(const uint32_t *__restrict__ ubitmap, uint32_t n_rows1, uint32_t u_ntiles,
                                                            uint8_t *__restrict__ bbitmap, uint32_t *__restrict__ abitmap) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;  // [tile][row][256 words]
    const uint64_t total = (uint64_t)u_ntiles * n_rows1 * 256u;
    if (i >= total) return;
    const uint32_t v = ubitmap[i];
    if (v == 0u) return;
    const uint32_t word = (uint32_t)(i & 255u), row = (uint32_t)((i >> 8) % n_rows1), tile = (uint32_t)((i >> 8) / n_rows1);
    const uint32_t l = word >> 2, wi = word & 3u;
#pragma unroll
    for (uint32_t k = 0; k < 4u; k++) {
        const uint32_t byte = (v >> (8u * k)) & 0xFFu;
        if (byte == 0u) continue;
        const uint32_t blk0 = tile * 8192u + ((wi * 4u + k) * 64u + l) * 8u;
        const uint32_t bt = blk0 >> 9;
        bbitmap[((size_t)bt * n_rows1 + row) * 64u + ((blk0 & 511u) >> 3)] = (uint8_t)byte;
        const uint32_t ab = blk0 >> 2;  // even: the byte's low nibble is block ab of 256, the high one ab + 1
        const uint32_t bits = ((byte & 0x0Fu) ? 1u : 0u) | ((byte & 0xF0u) ? 2u : 0u);
        const uint32_t at = ab >> 11, bit = ab & 2047u;
        atomicOr(&abitmap[((size_t)at * n_rows1 + row) * 64u + (bit >> 5)], bits << (bit & 31u));
    }
}

void launch_bounds2_build(hipStream_t s, const uint32_t *ubitmap, uint32_t n_rows1, uint32_t u_ntiles, uint8_t *bbitmap, uint32_t *abitmap) {
    const uint64_t total = (uint64_t)u_ntiles * n_rows1 * 256u;
    hipLaunchKernelGGL(bounds2_build_kernel, dim3((uint32_t)((total + 255u) / 256u)), dim3(256), 0, s, ubitmap, n_rows1, u_ntiles, bbitmap, abitmap);
}

}  // namespace rtx
