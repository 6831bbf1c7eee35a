// Folding SEVERAL short bitmap rows per load instruction (round 5): the pieces shared by the two-level bounds pass (rtx_bounds2.hip)
// and the counting pass over sub-tiles (rtx_subcount.hip).
//
// hit_count_pair_kernel reads rows of 1 KiB: one row per load instruction, lane l its bytes 16 l .. 16 l + 15.  A bitmap stored in
// narrower tiles -- rows of 256 or 64 bytes -- lets one instruction fetch R = 4 or 16 rows (lane = (64 / R) (row of the instruction)
// + s, s = which 16 bytes): the same KiB per instruction, the same folds per instruction, a quarter or a sixteenth of the instructions
// per row.  The lanes that took different rows hold partial counters, summed as bit-sliced numbers when the rows are through.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "rtx_hit_common.hpp"

namespace rtx {

constexpr uint32_t kB2Cap = 1024u;                       // entries per list (t <= 1023), padded with the zero row to multiples of kB2Pad
constexpr uint32_t kB2Pad = 256u;                        // the largest group of rows of either level (level B: 16 instructions x 16 rows)
constexpr uint32_t kB2LdsDw = 3u * kB2Cap + kB2Pad;       // lists | zero rows; behind them per A-tile [2][64] u16: the level-A bounds both queries' lanes found

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

// a + b on bit-sliced numbers (the sum fits NP planes: partial counts of disjoint rows of a query with t < 2^NP)
template <int NP>
__device__ __forceinline__ void planes_add(uint32_t (&a)[NP], const uint32_t (&b)[NP]) {
    uint32_t c = a[0] & b[0];
    a[0] ^= b[0];
#pragma unroll
    for (int p = 1; p < NP; p++) {
        const uint32_t cn = __builtin_amdgcn_bitop3_b32(a[p], b[p], c, 0xE8);
        a[p] = __builtin_amdgcn_bitop3_b32(a[p], b[p], c, 0x96);
        c = cn;
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

// the largest of the 32 counters of a bit-sliced word, and the counters that hold it (bit by bit from the top)
template <int NP>
__device__ __forceinline__ uint32_t planes_max(const uint32_t (&r)[NP], uint32_t &cand) {
    uint32_t m = 0;
    cand = 0xFFFFFFFFu;
#pragma unroll
    for (int p = NP - 1; p >= 0; p--) {
        const uint32_t x = cand & r[p];
        const bool nz = x != 0u;
        cand = nz ? x : cand;
        m |= nz ? 1u << p : 0u;
    }
    return m;
}

__device__ __forceinline__ uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }


// slot of `tile` among the record segments of query q (RecordRef::slots: its live tiles at prune time, ascending), or 0xFFFFFFFF
__device__ __forceinline__ uint32_t rec_slot_find(const RecordRef &rr, uint32_t q, uint32_t tile, uint32_t lane) {
    const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)rr.nslots[q]);
    const uint32_t v = lane < n ? (uint32_t)rr.slots[(size_t)q * kRecMaxSlots + lane] : 0xFFFFFFFFu;
    const unsigned long long b = __ballot(v == tile);
    return b ? (uint32_t)__builtin_ctzll(b) : 0xFFFFFFFFu;
}

// The three row lists of a pair in LDS -- rows of both queries, of A only, of B only -- from the union pair_union_kernel left (every row
// of a query counts: these bitmaps have no segment classes), padded with the zero row to multiples of kB2Pad.
struct PairLists {
    uint32_t *l_both, *l_a, *l_b, *l_zero;
    uint32_t n_both, n_a, n_b;
};
__device__ __forceinline__ PairLists build_pair_lists(uint32_t *lds, const uint2 *urec, uint32_t n_u, uint32_t ustride, bool has_b, uint32_t zero_row,
                                                      uint32_t lane) {
    PairLists L;
    L.l_both = lds;
    L.l_a = lds + kB2Cap;
    L.l_b = lds + 2u * kB2Cap;
    L.l_zero = lds + 3u * kB2Cap;
    for (uint32_t i = lane; i < kB2Pad; i += 64) L.l_zero[i] = zero_row;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t n_both = 0, n_a = 0, n_b = 0;
    for (uint32_t u0 = 0; u0 < n_u; u0 += 512u) {  // eight chunks of 64 union entries per round trip
        uint2 rec[8];
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const uint32_t i = u0 + (uint32_t)c * 64u + lane;
            rec[c] = urec[i < ustride ? i : ustride - 1u];
        }
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const uint32_t i = u0 + (uint32_t)c * 64u + lane;
            const bool in = i < n_u;
            const bool ia = in && (rec[c].x & (1u << 30)), ib = in && (rec[c].x & (1u << 31)) && has_b;
            const uint32_t row = rec[c].x & 0x3FFFFFFFu;
            const bool sh = ia && ib, oa = ia && !ib, ob = ib && !ia;
            const unsigned long long bs = __ballot(sh), ba = __ballot(oa), bo = __ballot(ob);
            // (a list never overflows: t <= 1023 rows per query; the guard keeps a corrupt union from writing beyond it)
            const uint32_t ps = n_both + (uint32_t)__popcll(bs & lt_mask), pa_ = n_a + (uint32_t)__popcll(ba & lt_mask), pb_ = n_b + (uint32_t)__popcll(bo & lt_mask);
            if (sh && ps < kB2Cap) L.l_both[ps] = row;
            if (oa && pa_ < kB2Cap) L.l_a[pa_] = row;
            if (ob && pb_ < kB2Cap) L.l_b[pb_] = row;
            n_both += (uint32_t)__popcll(bs);
            n_a += (uint32_t)__popcll(ba);
            n_b += (uint32_t)__popcll(bo);
        }
    }
    n_both = std::min(n_both, kB2Cap);
    n_a = std::min(n_a, kB2Cap);
    n_b = std::min(n_b, kB2Cap);
    for (uint32_t i = n_both + lane; i < ((n_both + kB2Pad - 1u) & ~(kB2Pad - 1u)); i += 64) L.l_both[i] = zero_row;
    for (uint32_t i = n_a + lane; i < ((n_a + kB2Pad - 1u) & ~(kB2Pad - 1u)); i += 64) L.l_a[i] = zero_row;
    for (uint32_t i = n_b + lane; i < ((n_b + kB2Pad - 1u) & ~(kB2Pad - 1u)); i += 64) L.l_b[i] = zero_row;
    wave_lds_sync();
    L.n_both = n_both;
    L.n_a = n_a;
    L.n_b = n_b;
    return L;
}

// One level: the lists against the region `rsrc` of a bitmap, R rows per load instruction, NB buffers of eight instructions in flight
// (rows of 1 << SHIFT bytes; grp: which row of an instruction this lane takes, col: its 16 bytes of the row).  The planes start at zero.
// need_a / need_b: the rows of one query alone are left out where only the other one is asked for.  Returns the load instructions issued.
template <int NP, int R, int NB, int SHIFT>
__device__ __forceinline__ uint32_t fold_lists(uint32_t (&pa)[4][NP], uint32_t (&pb)[4][NP], uint4 (&buf)[4][8], const PairLists &L,
                                               __amdgpu_buffer_rsrc_t rsrc, uint32_t grp, uint32_t col, bool need_a, bool need_b) {
    constexpr uint32_t kGroup = 8u * (uint32_t)R * (uint32_t)NB;
#pragma unroll
    for (int w = 0; w < 4; w++)
#pragma unroll
        for (int b = 0; b < NP; b++) { pa[w][b] = 0; pb[w][b] = 0; }
    const uint32_t g_both = (L.n_both + kGroup - 1u) / kGroup, g_a = need_a ? (L.n_a + kGroup - 1u) / kGroup : 0u, g_b = need_b ? (L.n_b + kGroup - 1u) / kGroup : 0u;
    const uint32_t *first = g_both ? L.l_both : (g_a ? L.l_a : (g_b ? L.l_b : nullptr));
    if (!first) return 0u;
#pragma unroll
    for (int b = 0; b < NB; b++) load_unit<SHIFT>(buf[b], rsrc, first + (uint32_t)b * 8u * (uint32_t)R, grp, col);
    const uint32_t *after_a = g_b ? L.l_b : L.l_zero, *after_both = g_a ? L.l_a : after_a;
    if (g_both) {  // the rows both queries share are folded once, into A's planes while B's are empty, and copied
        fold_list<NP, 1, R, NB, SHIFT>(pa, pb, buf, L.l_both, g_both, after_both, grp, col, rsrc);
#pragma unroll
        for (int w = 0; w < 4; w++)
#pragma unroll
            for (int b = 0; b < NP; b++) pb[w][b] = pa[w][b];
    }
    if (g_a) fold_list<NP, 1, R, NB, SHIFT>(pa, pb, buf, L.l_a, g_a, after_a, grp, col, rsrc);
    if (g_b) fold_list<NP, 2, R, NB, SHIFT>(pa, pb, buf, L.l_b, g_b, L.l_zero, grp, col, rsrc);
    return (g_both + g_a + g_b) * 8u * (uint32_t)NB;
}

// Level B's reduction: the sixteen rows of an instruction -- the DPP rows by halving (reduce_rows: -> word `lane >> 4` of the lane's 16
// bytes), then the four lanes of a DPP row that share a sub-lane (lane, lane + 4, + 8, + 12) by rotations.
template <int NP>
__device__ __forceinline__ void reduce_rows16(const uint32_t (&pl)[4][NP], uint32_t lane, uint32_t (&r)[NP]) {
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
}

}  // namespace rtx
