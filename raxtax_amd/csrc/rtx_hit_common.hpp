// Pieces shared by the two hit_count kernels (rtx_kernels.hip: one wave per (query, tile); rtx_hit_pair.hip: two
// neighbouring queries per wave, the bitmap rows they share loaded once).
#pragma once

#include <hip/hip_runtime.h>

#include "rtx_kernels.hpp"
#include "rtx_math.hpp"
#include "rtx_wave.hpp"

namespace rtx {

// Orders the LDS traffic of ONE wave (some lanes write, others read): the hardware keeps the LDS operations of a
// wave in order, this only stops the compiler from moving them.  No s_barrier: the epilogue below works on LDS that
// belongs to one wave, whatever the other waves of its workgroup are doing.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

// Eight row segments through raw buffer loads.  The bitmap is stored tile by tile (rtx_math.hpp: bitmap_word): ONE
// buffer descriptor per wave covers the tile's region (n_rows1 KiB), a row is the 32-bit offset row << 10 -- the lists
// hold these offsets -- and goes into the load as its SGPR offset (lanes O .. O+7 of a VGPR, taken out with
// v_readlane); the vector operand is the lane's 16-byte column.  `buffer_load_dwordx4 v, v_col, s[desc], s_row offen`:
// two instructions per row, no address arithmetic (a descriptor per row cost five scalar instructions more).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const uint32_t *bitmap, uint32_t n_rows1, uint32_t tile) {
    const char *base = reinterpret_cast<const char *>(bitmap) + (size_t)tile * n_rows1 * 1024u;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, n_rows1 * 1024u, 0x00027000);
}

__device__ __forceinline__ void load8v_at(uint4 (&buf)[8], __amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t idv, int o) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t roff = (uint32_t)__builtin_amdgcn_readlane((int)idv, o + j);
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, roff, 0);
        buf[j] = make_uint4(v.x, v.y, v.z, v.w);
    }
}
template <int O>
__device__ __forceinline__ void load8v(uint4 (&buf)[8], __amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t idv) {
    load8v_at(buf, rsrc, voff, idv, O);
}

// carry of weight 8 of eight 16-byte row segments, per 32-reference word
template <int NP>
__device__ __forceinline__ uint4 tree8(uint32_t (&pl)[4][NP], const uint4 (&a)[8]) {
    uint4 e;
    e.x = planes_tree8<NP>(pl[0], a[0].x, a[1].x, a[2].x, a[3].x, a[4].x, a[5].x, a[6].x, a[7].x);
    e.y = planes_tree8<NP>(pl[1], a[0].y, a[1].y, a[2].y, a[3].y, a[4].y, a[5].y, a[6].y, a[7].y);
    e.z = planes_tree8<NP>(pl[2], a[0].z, a[1].z, a[2].z, a[3].z, a[4].z, a[5].z, a[6].z, a[7].z);
    e.w = planes_tree8<NP>(pl[3], a[0].w, a[1].w, a[2].w, a[3].w, a[4].w, a[5].w, a[6].w, a[7].w);
    return e;
}

template <int NP, int P>
__device__ __forceinline__ uint4 csa_plane(uint32_t (&pl)[4][NP], const uint4 &a, const uint4 &b) {
    uint4 c;
    csa(pl[0][P], a.x, b.x, pl[0][P], c.x);
    csa(pl[1][P], a.y, b.y, pl[1][P], c.y);
    csa(pl[2][P], a.z, b.z, pl[2][P], c.z);
    csa(pl[3][P], a.w, b.w, pl[3][P], c.w);
    return c;
}


template <int NP, int P>
__device__ __forceinline__ uint4 half_plane(uint32_t (&pl)[4][NP], const uint4 &a) {  // a carry-save adder with one input 0
    uint4 c;
    c.x = pl[0][P] & a.x; pl[0][P] ^= a.x;
    c.y = pl[1][P] & a.y; pl[1][P] ^= a.y;
    c.z = pl[2][P] & a.z; pl[2][P] ^= a.z;
    c.w = pl[3][P] & a.w; pl[3][P] ^= a.w;
    return c;
}

template <int NP, int L>
__device__ __forceinline__ void ripple4(uint32_t (&pl)[4][NP], const uint4 &c) {
    planes_ripple<NP, L>(pl[0], c.x);
    planes_ripple<NP, L>(pl[1], c.y);
    planes_ripple<NP, L>(pl[2], c.z);
    planes_ripple<NP, L>(pl[3], c.w);
}

// The row loop with NB buffers of eight rows: every buffer is requested again as soon as it has been folded, so
// 8 * (NB - 1) .. 8 * NB rows are in flight per wave (the loop is bound by rows in flight per CU / load latency).
// list: n8 * 8 row ids, padded with the zero row up to n8 * 8 + NB * 8 + 64 (the loads are unconditional: the rows
// behind the list are the zero row -- predicated loads cost the compiler its register allocation).
template <int NP, int NB>
__device__ __forceinline__ void fold_ring(uint32_t (&pl)[4][NP], const uint32_t *list, uint32_t n8, uint32_t lane,
                                          __amdgpu_buffer_rsrc_t rsrc, uint32_t voff) {
    static_assert(NB >= 3 && NB <= 6, "groups of 24 .. 48 rows");
    constexpr uint32_t GR = NB * 8;
    const uint32_t ng = n8 / NB, nt = n8 - ng * NB;
    uint32_t idv = list[lane];
    uint4 buf[NB][8];
#pragma unroll
    for (int b = 0; b < NB; b++) load8v_at(buf[b], rsrc, voff, idv, b * 8);
    for (uint32_t g = 0; g < ng; g++) {
        const uint32_t idn = list[(g + 1) * GR + lane];
        uint4 c3[NB];
#pragma unroll
        for (int b = 0; b < NB; b++) {
            c3[b] = tree8<NP>(pl, buf[b]);
            load8v_at(buf[b], rsrc, voff, idn, b * 8);
        }
        const uint4 c4a = csa_plane<NP, 3>(pl, c3[0], c3[1]);
        if (NB == 3) {
            const uint4 c4b = half_plane<NP, 3>(pl, c3[2]);
            const uint4 c5 = csa_plane<NP, 4>(pl, c4a, c4b);
            ripple4<NP, 5>(pl, c5);
        } else {
            const uint4 c4b = csa_plane<NP, 3>(pl, c3[2], c3[3]);
            const uint4 c5a = csa_plane<NP, 4>(pl, c4a, c4b);
            if (NB == 4) {
                ripple4<NP, 5>(pl, c5a);
            } else {
                const uint4 c4c = NB == 5 ? half_plane<NP, 3>(pl, c3[NB - 1]) : csa_plane<NP, 3>(pl, c3[4], c3[NB - 1]);
                const uint4 c5b = half_plane<NP, 4>(pl, c4c);
                const uint4 c6 = csa_plane<NP, 5>(pl, c5a, c5b);
                ripple4<NP, 6>(pl, c6);
            }
        }
        idv = idn;
    }
#pragma unroll
    for (int b = 0; b < NB - 1; b++)
        if ((uint32_t)b < nt) {
            const uint4 c3 = tree8<NP>(pl, buf[b]);
            ripple4<NP, 3>(pl, c3);
        }
}

// The same for a row list that TWO queries share (hit_count_pair_kernel): every buffer is folded into both plane sets
// before it is requested again -- one load, two folds.
template <int NP, int NB>
__device__ __forceinline__ void fold_ring2(uint32_t (&pa)[4][NP], uint32_t (&pb)[4][NP], const uint32_t *list, uint32_t n8,
                                           uint32_t lane, __amdgpu_buffer_rsrc_t rsrc, uint32_t voff) {
    static_assert(NB >= 3 && NB <= 4, "groups of 24 or 32 rows");
    constexpr uint32_t GR = NB * 8;
    const uint32_t ng = n8 / NB, nt = n8 - ng * NB;
    uint32_t idv = list[lane];
    uint4 buf[NB][8];
#pragma unroll
    for (int b = 0; b < NB; b++) load8v_at(buf[b], rsrc, voff, idv, b * 8);
    for (uint32_t g = 0; g < ng; g++) {
        const uint32_t idn = list[(g + 1) * GR + lane];
        uint4 ca[NB], cb[NB];
#pragma unroll
        for (int b = 0; b < NB; b++) {
            ca[b] = tree8<NP>(pa, buf[b]);
            cb[b] = tree8<NP>(pb, buf[b]);
            load8v_at(buf[b], rsrc, voff, idn, b * 8);
        }
        {
            const uint4 c4a = csa_plane<NP, 3>(pa, ca[0], ca[1]);
            const uint4 c4b = NB == 3 ? half_plane<NP, 3>(pa, ca[2]) : csa_plane<NP, 3>(pa, ca[2], ca[NB - 1]);
            ripple4<NP, 5>(pa, csa_plane<NP, 4>(pa, c4a, c4b));
        }
        {
            const uint4 c4a = csa_plane<NP, 3>(pb, cb[0], cb[1]);
            const uint4 c4b = NB == 3 ? half_plane<NP, 3>(pb, cb[2]) : csa_plane<NP, 3>(pb, cb[2], cb[NB - 1]);
            ripple4<NP, 5>(pb, csa_plane<NP, 4>(pb, c4a, c4b));
        }
        idv = idn;
    }
#pragma unroll
    for (int b = 0; b < NB - 1; b++)
        if ((uint32_t)b < nt) {
            ripple4<NP, 3>(pa, tree8<NP>(pa, buf[b]));
            ripple4<NP, 3>(pb, tree8<NP>(pb, buf[b]));
        }
}

// ---------------------------------------------------------------------------
// Epilogue of a (query, tile) wave: the bit planes `pl` hold the hits through dense segments.  Zeroes exact matches
// (raxtax.rs:65-68), unpacks the planes, adds the hits through sparse segments (byte counters in LDS, half a tile at
// a time), stores the counts (u16, or packed 10 bits per reference), builds the histogram of prob.rs:13-19 with LDS
// atomics and flushes it with one global atomic per non-empty bin.
//   hist_lds: [t + 1] u32 of this wave; cnt8: [1024] u32 (4096 byte counters) of this wave.
// ---------------------------------------------------------------------------
//   kPrefetch: the slots of ALL sparse segments of the tile are requested at once and kept in registers for both half-tile
//   passes (two round trips instead of two per 64 segments and half) -- for kernels with registers to spare in the epilogue;
//   `pre_in` (optional): they have been requested by the caller already (sparse_prefetch).
//   kFullTile: cnt8 holds 8192 byte counters (+ 64 pad words), the sparse segments are scanned once for the whole tile.
//   A (query, tile) has at most 255 sparse segments (kmer_extract), so a byte counter cannot overflow.
constexpr int kSparseIt = (kSegMaxSparseRows + 63) / 64, kSparseV = kSegSlotEntries / 8;

// the slots of all sparse segments of a (query, tile): lane l takes segment it * 64 + l; sid: their slot ids (global or LDS)
__device__ __forceinline__ void sparse_prefetch(const HitParams &p, uint32_t lane, uint32_t ns, const uint32_t *sid_src,
                                                uint4 (&pre)[kSparseIt][kSparseV]) {
    uint32_t sid[kSparseIt];
#pragma unroll
    for (int it = 0; it < kSparseIt; it++) sid[it] = (uint32_t)it * 64u + lane < ns ? sid_src[(uint32_t)it * 64u + lane] : 0xFFFFFFFFu;
#pragma unroll
    for (int it = 0; it < kSparseIt; it++)
#pragma unroll
        for (int i = 0; i < kSparseV; i++) {
            const uint32_t pw = seg_pad(2u * lane) | (seg_pad(2u * lane + 1u) << 16);  // unused entries: pad words behind the byte counters
            pre[it][i] = make_uint4(pw, pw, pw, pw);
            if (sid[it] != 0xFFFFFFFFu) pre[it][i] = reinterpret_cast<const uint4 *>(p.segslots + (size_t)sid[it] * kSegSlotEntries)[i];
        }
}

__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {  // v_pk_max_u16
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    u16x2 x, y;
    __builtin_memcpy(&x, &a, 4);
    __builtin_memcpy(&y, &b, 4);
    const u16x2 m = __builtin_elementwise_max(x, y);
    uint32_t r;
    __builtin_memcpy(&r, &m, 4);
    return r;
}

//   kGlobalHist: reads of tens of kilobases (t up to 65 535: a histogram row does not fit LDS beside the list): every count goes to the
//   query's histogram row in global memory with an atomic of its own, the tile's largest count is taken from the counts themselves.
template <int NP, bool kPacked, bool kPrefetch, bool kFullTile, bool kPreLoaded, bool kGlobalHist = false>
__device__ __forceinline__ void hit_epilogue_x(const HitParams &p, uint32_t (&pl)[4][NP], uint32_t q, uint32_t tile, uint32_t lane,
                                               uint32_t t, bool active, uint32_t *hist_lds, uint32_t *cnt8, uint32_t ns,
                                               const uint32_t *srows, uint4 (&pre)[kSparseIt][kSparseV], uint32_t h_thr) {
    constexpr int kIt = kSparseIt, kVp = kSparseV;
    const bool lists = ns != 0u;  // wave-uniform: byte counters in use
    if (kPrefetch && !kPreLoaded && ns) sparse_prefetch(p, lane, ns, srows, pre);
    const uint32_t pad_word = seg_pad(2u * lane) | (seg_pad(2u * lane + 1u) << 16);  // entries behind a list
    // hits of the sparse segments on the references [half*4096, half*4096 + 4096) of the tile -> cnt8 (at most 255 each)
    auto sparse_hits = [&](uint32_t half) {
#pragma unroll
        for (int i = 0; i < (kFullTile ? 8 : 4); i++) reinterpret_cast<uint4 *>(cnt8)[i * 64 + lane] = make_uint4(0, 0, 0, 0);
        if (kFullTile) cnt8[2048u + lane] = 0;  // 64 pad words: they take the unused entries of the slots (spread: one word would serialise the atomics)
        wave_lds_sync();
        constexpr int kV = kSegSlotEntries / 8;  // uint4 per slot
        auto add_slot = [&](const uint4 (&e)[kV]) {
#pragma unroll
            for (int i = 0; i < kV; i++) {
                const uint32_t wv[4] = {e[i].x, e[i].y, e[i].z, e[i].w};
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const uint32_t id = (wv[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu;  // >= kSegPad (8192): unused entry
                    // full tile: unused entries land in the pad words behind the counters -- no compare, no exec mask per atomic
                    if (kFullTile) atomicAdd(&cnt8[id >> 2], 1u << ((id & 3u) * 8u));
                    else if ((id >> 12) == half) atomicAdd(&cnt8[(id & 4095u) >> 2], 1u << ((id & 3u) * 8u));
                }
            }
        };
        if (kPrefetch) {
#pragma unroll
            for (int it = 0; it < kIt; it++)
                if ((uint32_t)it * 64u < ns) add_slot(pre[it]);  // wave-uniform
        } else {
            for (uint32_t c0 = 0; c0 < ns; c0 += 64) {
                uint4 e[kV];
#pragma unroll
                for (int i = 0; i < kV; i++) e[i] = make_uint4(pad_word, pad_word, pad_word, pad_word);
                if (c0 + lane < ns) {
                    const uint4 *slot = reinterpret_cast<const uint4 *>(p.segslots + (size_t)srows[c0 + lane] * kSegSlotEntries);
#pragma unroll
                    for (int i = 0; i < kV; i++) e[i] = slot[i];
                }
                add_slot(e);
            }
        }
        wave_lds_sync();
    };

    // ---- a pruned query (threshold u = h_thr) whose tile turns out to hold no count above u: nothing of this tile can reach the
    // result -- prob_lookup writes 0 into the table entries up to u, taxon_prefix leaves out every tile whose largest count is at most
    // u -- so nothing is unpacked, stored or histogrammed: the references go to bin 0 as one number, the tile keeps a largest count
    // of 0 and its live bit is cleared (the taps then see a tile that was never counted, which is what its outputs are).  The test
    // works on the bit planes: count = dense part (planes) + sparse part (byte counters, at most `smax` = the OR of all of them), so
    // no count exceeds u if no dense part exceeds u - smax -- a comparison of bit-sliced numbers with a constant, ~3 operations per
    // plane and word.  Queries far from their best hit keep tens of tiles live of which a tenth hold such a count (the bounds of the
    // union bitmap are loose there): their epilogues were a third of the counting pass.
    if (kFullTile && h_thr) {
        uint32_t smax = 0;
        if (lists) {
            sparse_hits(0u);
            if (p.flags & RTX_SKIP_EXACT_MATCHES) {  // raxtax.rs:65-68: the sparse part
                const uint64_t qin = p.perm[p.q0 + q];
                uint64_t e0, e1;
                const uint32_t *xids;
                exact_range(p.exact, qin, e0, e1, xids);
                for (uint64_t e = e0 + lane; e < e1; e += 64) {
                    const uint32_t id = xids[e] - p.ref_base;
                    if (id < p.n_refs && (id >> 13) == tile) reinterpret_cast<uint8_t *>(cnt8)[id & 8191u] = 0;
                }
                wave_lds_sync();
            }
            uint32_t orw = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const uint4 v = reinterpret_cast<const uint4 *>(cnt8)[i * 64 + lane];
                orw |= (v.x | v.y) | (v.z | v.w);
            }
            orw |= orw >> 16;  // per byte position the OR of this lane's counters: at least each of them
            smax = wave_max_u32((orw | (orw >> 8)) & 0xFFu);
        }
        bool any = smax >= h_thr;  // wave-uniform
        if (!any) {
            const uint32_t c = h_thr - smax;  // is any dense count > c ?
            uint32_t gt_any = 0;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                uint32_t gt = 0, eq = 0xFFFFFFFFu;
#pragma unroll
                for (int b = NP - 1; b >= 0; b--) {
                    const uint32_t cb = (c >> b) & 1u ? 0xFFFFFFFFu : 0u;  // scalar
                    gt |= eq & pl[w][b] & ~cb;
                    eq &= ~(pl[w][b] ^ cb);
                }
                gt_any |= gt;
            }
            any = __ballot(active && gt_any != 0u) != 0ull;
        }
        if (!any) {
            if (lane == 0u) {
                const uint32_t in_tile = (((uint64_t)tile + 1u) << 13) <= p.n_refs ? 8192u : (uint32_t)(p.n_refs - ((uint64_t)tile << 13));
                atomicAdd(&p.hist[(size_t)q * p.hstride], in_tile);
                atomicAnd(const_cast<uint32_t *>(p.live) + (size_t)q * p.live_words + (tile >> 5), ~(1u << (tile & 31u)));
            }
            return;
        }
    }
    uint32_t *const hist_g = p.hist + (size_t)q * p.hstride;
    uint32_t mx_g = 0;  // kGlobalHist: the largest count this lane saw
    if (!kGlobalHist) {
        for (uint32_t i = lane; i <= t; i += 64) hist_lds[i] = 0;
        wave_lds_sync();
    }
    // tile pruning gave the query a threshold u: every count up to u is a reference without a hit to prob_lookup (rtx_prob_tables.hip) --
    // they go to bin 0 as one number, and only the groups of references that hold a count above u touch the histogram (h_thr: u or 0)
    const uint32_t h_lo = h_thr ? h_thr + 1u : 0u;

    const uint32_t L = tile_lanes(p.stride_bytes, tile);  // lanes of this tile (64 except in the last one)
    if (active && (p.flags & RTX_SKIP_EXACT_MATCHES)) {  // raxtax.rs:65-68: the dense part
        const uint64_t qin = p.perm[p.q0 + q];
        uint64_t e0, e1;
        const uint32_t *xids;
        exact_range(p.exact, qin, e0, e1, xids);
        for (uint64_t e = e0; e < e1; e++) {
            const uint32_t id = xids[e] - p.ref_base;  // local id; other shards' ids wrap out of range
            if (id < p.n_refs && (id >> 13) == tile) {
                const uint32_t c = (id & 8191u) >> 3, g = c / L;
                if (c - g * L == lane) {
                    const uint32_t w = g >> 2, msk = ~(1u << ((g & 3u) * 8u + (id & 7u)));
#pragma unroll
                    for (int ww = 0; ww < 4; ww++)
                        if ((uint32_t)ww == w) {
#pragma unroll
                            for (int b = 0; b < NP; b++) pl[ww][b] &= msk;
                        }
                }
            }
        }
    }
    // group g = (w, g2) of this lane: references ref0 + g*L*8 + [0, 8) (ref_slot, rtx_math.hpp)
    const uint64_t ref0 = (uint64_t)tile * 8192u + lane * 8u;
    static_assert(!kPacked || NP <= 10, "the packed format holds counts up to 1023");  // 10 bits per reference leave the kernel instead of 16
    const bool tile_full = (((uint64_t)tile + 1u) << 13) <= p.n_refs;  // wave-uniform
    // references of this lane from ref0 on that exist (only looked at in the last tile: |value| < 2^31 there)
    const int32_t refs_left = tile_full ? 0x7FFFFFFF : (int32_t)((int64_t)p.n_refs - (int64_t)ref0);
    const uint32_t crow = p.cnt_row ? (uint32_t)__builtin_amdgcn_readfirstlane((int)p.cnt_row[q]) : q;  // the query's row of the counts buffer
    const bool has_row = crow != 0xFFFFFFFFu;  // (none: the rows ran out -- the run is repeated with a larger buffer; nothing is stored)
    uint16_t *out = p.counts + (size_t)(has_row ? crow : 0u) * p.npad + ref0;
    uint8_t *out_lo = p.counts_lo + (size_t)(has_row ? crow : 0u) * p.npad + ref0;
    constexpr bool kBytes = kPacked && NP <= 8 && !kGlobalHist;  // every count fits a byte: stored and histogrammed as bytes
    uint32_t hiw[8];  // packed format: the two high bits of the 8 references of group g in bits 16 (g & 1) + [0, 16) of hiw[g / 2]
#pragma unroll
    for (int i = 0; i < 8; i++) hiw[i] = 0;
#pragma unroll
    for (int half = 0; half < 2; half++) {  // groups 0-7 = references 0..4095 of a full tile, groups 8-15 = 4096..8191
        if (lists && (kFullTile ? (half == 0 && !h_thr) : true)) {  // (a pruned query's full-tile epilogue has scanned them above)
            sparse_hits((uint32_t)half);
            if (p.flags & RTX_SKIP_EXACT_MATCHES) {  // raxtax.rs:65-68: the sparse part
                const uint64_t qin = p.perm[p.q0 + q];
                uint64_t e0, e1;
                const uint32_t *xids;
                exact_range(p.exact, qin, e0, e1, xids);
                for (uint64_t e = e0 + lane; e < e1; e += 64) {
                    const uint32_t id = xids[e] - p.ref_base;
                    if (id < p.n_refs && (id >> 13) == tile && (kFullTile || ((id >> 12) & 1u) == (uint32_t)half))
                        reinterpret_cast<uint8_t *>(cnt8)[id & (kFullTile ? 8191u : 4095u)] = 0;
                }
                wave_lds_sync();
            }
        }
        if (active) {
#pragma unroll
            for (int wi = 0; wi < 2; wi++) {
                const int w = half * 2 + wi;
                // the low eight planes of the word, unpacked at once (planes_unpack32: 64 operations where four planes_unpack8 take 112)
                uint32_t lo_w[4][2];
                planes_unpack32<NP>(pl[w], lo_w);
#pragma unroll
                for (int g2 = 0; g2 < 4; g2++) {  // 8 references per store, contiguous across lanes
                    const uint32_t lo0 = lo_w[g2][0], lo1 = lo_w[g2][1];
                    const uint32_t goff = (uint32_t)(w * 4 + g2) * L * 8u;
                    uint2 sb = make_uint2(0u, 0u);  // hits through sparse segments (L = 64 there): bytes of the eight references of this group
                    if (lists) sb = *reinterpret_cast<const uint2 *>(cnt8 + (((uint32_t)((kFullTile ? w : wi) * 4 + g2) * 64u + lane) * 2u));
                    uint4 st;   // the eight counts as u16 ...
                    uint2 lo8;  // ... and their low bytes, in reference order
                    if (kBytes) {
                        // eight planes <=> t <= 255: a count -- dense and sparse part together at most t -- fits its byte, so the bytes are
                        // added as they are (no carry leaves a byte) and nothing is widened to u16 on the way to the store and the histogram
                        lo8 = make_uint2(lo0 + sb.x, lo1 + sb.y);
                        st = make_uint4(0u, 0u, 0u, 0u);
                    } else {
                        uint32_t hi0, hi1;
                        planes_unpack8_hi<NP>(pl[w], g2, hi0, hi1);
                        // bytes (lo.b0, hi.b0, lo.b1, hi.b1) -> two u16 counts
                        st.x = __builtin_amdgcn_perm(hi0, lo0, 0x05010400u);
                        st.y = __builtin_amdgcn_perm(hi0, lo0, 0x07030602u);
                        st.z = __builtin_amdgcn_perm(hi1, lo1, 0x05010400u);
                        st.w = __builtin_amdgcn_perm(hi1, lo1, 0x07030602u);
                        if (lists) {
                            // bytes (b0, b1) -> (b0, 0, b1, 0): one v_perm_b32 each (selector byte 0x0C = constant 0)
                            st.x += __builtin_amdgcn_perm(0u, sb.x, 0x0C010C00u);
                            st.y += __builtin_amdgcn_perm(0u, sb.x, 0x0C030C02u);
                            st.z += __builtin_amdgcn_perm(0u, sb.y, 0x0C010C00u);
                            st.w += __builtin_amdgcn_perm(0u, sb.y, 0x0C030C02u);
                        }
                        lo8.x = __builtin_amdgcn_perm(st.y, st.x, 0x06040200u);
                        lo8.y = __builtin_amdgcn_perm(st.w, st.z, 0x06040200u);
                    }
                    if (kPacked) {
                        {   // non-temporal: the counts are read once, much later (taxon_prefix), and should not displace bitmap rows
                            typedef uint32_t u32x2_nt __attribute__((ext_vector_type(2)));
                            u32x2_nt nv;
                            nv.x = lo8.x;
                            nv.y = lo8.y;
                            if (has_row) __builtin_nontemporal_store(nv, reinterpret_cast<u32x2_nt *>(out_lo + goff));
                        }
                        if (NP > 8) {
                            // high bytes (0..3 each) -> 2 bits per reference: byte j moves to bit 2j
                            const uint32_t hb0 = __builtin_amdgcn_perm(st.y, st.x, 0x07050301u), hb1 = __builtin_amdgcn_perm(st.w, st.z, 0x07050301u);
                            const uint32_t h0 = (hb0 | (hb0 >> 6) | (hb0 >> 12) | (hb0 >> 18)) & 0xFFu;
                            const uint32_t h1 = (hb1 | (hb1 >> 6) | (hb1 >> 12) | (hb1 >> 18)) & 0xFFu;
                            const uint32_t h16 = h0 | (h1 << 8);
                            const int gi = w * 4 + g2;
                            hiw[gi >> 1] |= h16 << ((gi & 1) * 16);
                        }
                    } else if (has_row) {
                        *reinterpret_cast<uint4 *>(out + goff) = st;
                    }
                    if (kBytes) {  // the histogram of the byte form
                        const uint32_t bw[2] = {lo8.x, lo8.y};
                        if (h_lo) {  // wave-uniform: a pruned query -- only the counts above its threshold are looked at one by one
                            const uint32_t e = pk_max_u16(lo8.x & 0x00FF00FFu, lo8.y & 0x00FF00FFu), o = pk_max_u16((lo8.x >> 8) & 0x00FF00FFu, (lo8.y >> 8) & 0x00FF00FFu);
                            const uint32_t m2 = pk_max_u16(e, o);
                            const uint32_t m = (m2 & 0xFFFFu) > (m2 >> 16) ? (m2 & 0xFFFFu) : (m2 >> 16);
                            if (__ballot(m >= h_lo) != 0ull) {  // (references behind n_refs have a count of 0)
#pragma unroll
                                for (int j = 0; j < 8; j++) {
                                    const uint32_t c = (bw[j >> 2] >> ((j & 3) * 8)) & 0xFFu;
                                    if (c >= h_lo) atomicAdd(&hist_lds[c], 1u);
                                }
                            }
                        } else if (tile_full) {  // wave-uniform: every reference of the tile exists -- no compare, no exec mask per atomic
#pragma unroll
                            for (int j = 0; j < 8; j++) atomicAdd(&hist_lds[(bw[j >> 2] >> ((j & 3) * 8)) & 0xFFu], 1u);
                        } else {  // the last tile of the database: the references behind n_refs are not counted
                            const int32_t left = refs_left - (int32_t)goff;
                            const uint32_t nvalid = left <= 0 ? 0u : (left < 8 ? (uint32_t)left : 8u);
#pragma unroll
                            for (int j = 0; j < 8; j++)
                                if ((uint32_t)j < nvalid) atomicAdd(&hist_lds[(bw[j >> 2] >> ((j & 3) * 8)) & 0xFFu], 1u);
                        }
                        continue;
                    }
                    const uint32_t cw[4] = {st.x, st.y, st.z, st.w};
                    if (h_lo) {  // wave-uniform: a pruned query -- only the counts above its threshold are looked at one by one
                        const uint32_t m2 = pk_max_u16(pk_max_u16(st.x, st.y), pk_max_u16(st.z, st.w));
                        const uint32_t m = (m2 & 0xFFFFu) > (m2 >> 16) ? (m2 & 0xFFFFu) : (m2 >> 16);
                        if (__ballot(m >= h_lo) != 0ull) {  // (references behind n_refs have a count of 0)
#pragma unroll
                            for (int j = 0; j < 8; j++) {
                                const uint32_t c = (cw[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu;
                                if (c >= h_lo) atomicAdd(&hist_lds[c], 1u);
                            }
                        }
                    } else if (kGlobalHist) {  // the histogram row lives in global memory (reads of tens of kilobases)
                        const int32_t left = refs_left - (int32_t)goff;
                        const uint32_t nvalid = left <= 0 ? 0u : (left < 8 ? (uint32_t)left : 8u);
#pragma unroll
                        for (int j = 0; j < 8; j++)
                            if ((uint32_t)j < nvalid) {
                                const uint32_t c = (cw[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu;
                                atomicAdd(&hist_g[c <= t ? c : t], 1u);
                                mx_g = c > mx_g ? c : mx_g;
                            }
                    } else if (tile_full) {  // wave-uniform: every reference of the tile exists -- no compare, no exec mask per atomic
#pragma unroll
                        for (int j = 0; j < 8; j++) atomicAdd(&hist_lds[(cw[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu], 1u);
                    } else {  // the last tile of the database: the references behind n_refs are not counted
                        const int32_t left = refs_left - (int32_t)goff;
                        const uint32_t nvalid = left <= 0 ? 0u : (left < 8 ? (uint32_t)left : 8u);
#pragma unroll
                        for (int j = 0; j < 8; j++)
                            if ((uint32_t)j < nvalid) atomicAdd(&hist_lds[(cw[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu], 1u);
                    }
                }
            }
        }
    }
    if (kPacked && NP <= 8) {  // no count above 255: the tile's high-bit words are zero
        if (active && has_row) {
            uint4 *dst = reinterpret_cast<uint4 *>(p.counts_hi + (size_t)crow * (p.npad >> 3) + (size_t)tile * 1024u + lane * 16u);
            __builtin_nontemporal_store(u32x4_t{0u, 0u, 0u, 0u}, reinterpret_cast<u32x4_t *>(dst));
            __builtin_nontemporal_store(u32x4_t{0u, 0u, 0u, 0u}, reinterpret_cast<u32x4_t *>(dst) + 1);
        }
    } else if (kPacked) {
        // The high-bit words leave in chunk order (u16 index g * L + lane within the tile): transposed through the
        // byte-counter region of LDS (free now) so that every lane stores 32 contiguous bytes.
        wave_lds_sync();
        uint16_t *tr = reinterpret_cast<uint16_t *>(cnt8);
        if (active) {
#pragma unroll
            for (int gi = 0; gi < 16; gi++) tr[(uint32_t)gi * L + lane] = (uint16_t)(hiw[gi >> 1] >> ((gi & 1) * 16));
        }
        wave_lds_sync();
        if (active && has_row) {
            const uint4 a = reinterpret_cast<const uint4 *>(tr)[lane * 2u], b = reinterpret_cast<const uint4 *>(tr)[lane * 2u + 1u];
            uint4 *dst = reinterpret_cast<uint4 *>(p.counts_hi + (size_t)crow * (p.npad >> 3) + (size_t)tile * 1024u + lane * 16u);
            __builtin_nontemporal_store(u32x4_t{a.x, a.y, a.z, a.w}, reinterpret_cast<u32x4_t *>(dst));
            __builtin_nontemporal_store(u32x4_t{b.x, b.y, b.z, b.w}, reinterpret_cast<u32x4_t *>(dst) + 1);
        }
    }
    wave_lds_sync();
    // every lane of the wave flushes (also those whose columns lie beyond the row)
    uint32_t *hist = hist_g;
    uint32_t mx = mx_g;  // the largest count of this tile = its highest non-empty bin
    uint32_t n_high = 0;
    if (!kGlobalHist)
        for (uint32_t m = lane; m <= t; m += 64) {
            const uint32_t v = hist_lds[m];
            if (v) { atomicAdd(&hist[m], v); mx = m; n_high += v; }
        }
    if (h_lo) {  // the references of the tile with a count up to the threshold (0 for the tile's largest count if there is no other)
        n_high = wave_incl_scan_u32(n_high);
        const uint32_t in_tile = tile_full ? 8192u : (uint32_t)(p.n_refs - ((uint64_t)tile << 13));
        if (lane == 63u && in_tile != n_high) atomicAdd(&hist[0], in_tile - n_high);
    }
    if (p.tile_max) {
        for (int d = 32; d >= 1; d >>= 1) {
            const uint32_t o = __shfl_xor(mx, d, 64);
            mx = o > mx ? o : mx;
        }
        if (lane == 0) p.tile_max[(size_t)q * p.ntiles + tile] = (uint16_t)mx;
    }
}

// ---------------------------------------------------------------------------
// The RECORDS epilogue (RecordRef, rtx_kernels.hpp): a pruned query (threshold u = h_thr > 0) on the records path leaves this tile
// nothing but the references with a count ABOVE u, as (local reference | count << 13) in reference order in segment `slot` of the
// query, plus what the dense epilogue leaves a pruned query elsewhere: the histogram above u (everything else in bin 0), the tile's
// largest count, a cleared live bit if nothing is above u.  The comparison runs on the bit planes (count > u: 3 operations per plane
// and word); only the groups of eight references that hold a candidate are unpacked -- on the bench workload one or two of the 16
// groups of a lane, in a handful of lanes (the query's own species), where the dense epilogue unpacks, packs and stores 8192 counts.
//   count = dense part (planes) + sparse part (byte counters, at most smax): candidates = dense > u - smax, then the exact test.
// Reference order: group g, lane l, reference j <-> local reference (g * L + l) * 8 + j (ref_slot): ascending in (g, l, j).
// ---------------------------------------------------------------------------
template <int NP>
__device__ __forceinline__ uint32_t planes_gt(const uint32_t (&pl)[NP], uint32_t c) {  // bit r: counter r of the word > c (c < 2^NP)
    uint32_t gt = 0, eq = 0xFFFFFFFFu;
#pragma unroll
    for (int b = NP - 1; b >= 0; b--) {
        const uint32_t cb = (c >> b) & 1u ? 0xFFFFFFFFu : 0u;  // scalar
        gt |= eq & pl[b] & ~cb;
        eq &= ~(pl[b] ^ cb);
    }
    return gt;
}

// raxtax.rs:65-68 on the bit planes: the counters of this query's exact matches in this tile are cleared (the dense part)
template <int NP>
__device__ __forceinline__ void zero_exact_dense(const HitParams &p, uint32_t (&pl)[4][NP], uint32_t q, uint32_t tile, uint32_t lane, uint32_t L) {
    const uint64_t qin = p.perm[p.q0 + q];
    uint64_t e0, e1;
    const uint32_t *xids;
    exact_range(p.exact, qin, e0, e1, xids);
    for (uint64_t e = e0; e < e1; e++) {
        const uint32_t id = xids[e] - p.ref_base;  // local id; other shards' ids wrap out of range
        if (id < p.n_refs && (id >> 13) == tile) {
            const uint32_t c = (id & 8191u) >> 3, g = c / L;
            if (c - g * L == lane) {
                const uint32_t w = g >> 2, msk = ~(1u << ((g & 3u) * 8u + (id & 7u)));
#pragma unroll
                for (int ww = 0; ww < 4; ww++)
                    if ((uint32_t)ww == w) {
#pragma unroll
                        for (int b = 0; b < NP; b++) pl[ww][b] &= msk;
                    }
            }
        }
    }
}

template <int NP>
__device__ __forceinline__ void rec_epilogue(const HitParams &p, uint32_t (&pl)[4][NP], uint32_t q, uint32_t tile, uint32_t lane, uint32_t t,
                                             bool active, uint32_t *hist_lds, uint32_t *cnt8, uint32_t ns, uint4 (&pre)[kSparseIt][kSparseV],
                                             uint32_t h_thr, uint32_t slot, uint32_t rec_stride) {
    const bool lists = ns != 0u;  // wave-uniform
    const uint32_t L = tile_lanes(p.stride_bytes, tile);
    const bool skip_exact = (p.flags & RTX_SKIP_EXACT_MATCHES) != 0u;
    uint32_t smax = 0;
    if (lists) {  // the hits through sparse segments of the whole tile -> 8192 byte counters (as the dense epilogue of a full tile)
#pragma unroll
        for (int i = 0; i < 8; i++) reinterpret_cast<uint4 *>(cnt8)[i * 64 + lane] = make_uint4(0, 0, 0, 0);
        cnt8[2048u + lane] = 0;
        wave_lds_sync();
#pragma unroll
        for (int it = 0; it < kSparseIt; it++)
            if ((uint32_t)it * 64u < ns) {  // wave-uniform
#pragma unroll
                for (int i = 0; i < kSparseV; i++) {
                    const uint32_t wv[4] = {pre[it][i].x, pre[it][i].y, pre[it][i].z, pre[it][i].w};
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const uint32_t id = (wv[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu;  // >= kSegPad: unused entry -> the pad words
                        atomicAdd(&cnt8[id >> 2], 1u << ((id & 3u) * 8u));
                    }
                }
            }
        wave_lds_sync();
        if (skip_exact) {  // raxtax.rs:65-68: the sparse part
            const uint64_t qin = p.perm[p.q0 + q];
            uint64_t e0, e1;
            const uint32_t *xids;
            exact_range(p.exact, qin, e0, e1, xids);
            for (uint64_t e = e0 + lane; e < e1; e += 64) {
                const uint32_t id = xids[e] - p.ref_base;
                if (id < p.n_refs && (id >> 13) == tile) reinterpret_cast<uint8_t *>(cnt8)[id & 8191u] = 0;
            }
            wave_lds_sync();
        }
        uint32_t orw = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint4 v = reinterpret_cast<const uint4 *>(cnt8)[i * 64 + lane];
            orw |= (v.x | v.y) | (v.z | v.w);
        }
        orw |= orw >> 16;
        smax = wave_max_u32((orw | (orw >> 8)) & 0xFFu);  // at least every byte counter of the tile
    }
    if (active && skip_exact) zero_exact_dense<NP>(p, pl, q, tile, lane, L);
    // candidates: a count above u needs a dense part above u - smax; with smax > u (never seen: a reference in more than u sparse
    // segments) every reference is one
    uint32_t gt[4];
    {
        const bool all = smax > h_thr;  // wave-uniform
        const uint32_t c = all ? 0u : h_thr - smax;
#pragma unroll
        for (int w = 0; w < 4; w++) gt[w] = !active ? 0u : (all ? 0xFFFFFFFFu : planes_gt<NP>(pl[w], c));
    }
    const uint32_t in_tile = (((uint64_t)tile + 1u) << 13) <= p.n_refs ? 8192u : (uint32_t)(p.n_refs - ((uint64_t)tile << 13));
    uint32_t *hist = p.hist + (size_t)q * p.hstride;
    uint32_t running = 0;  // wave-uniform: records written so far
    uint32_t mx = 0;
    if (__ballot((gt[0] | gt[1]) | (gt[2] | gt[3])) != 0ull) {
        for (uint32_t i = lane; i <= t; i += 64) hist_lds[i] = 0;
        wave_lds_sync();
        uint32_t *seg = p.rec.rec + ((size_t)q * rec_stride + slot) * p.rec.seg_len;
#pragma unroll
        for (int g = 0; g < 16; g++) {
            const int w = g >> 2, g2 = g & 3;
            const uint32_t cand = (gt[w] >> (8 * g2)) & 0xFFu;
            if (__ballot(cand != 0u) == 0ull) continue;  // wave-uniform: no lane holds a candidate in this group
            uint32_t lo0, hi0, lo1, hi1;
            planes_unpack8<NP>(pl[w], g2, lo0, hi0, lo1, hi1);
            uint32_t cw[4];  // the eight counts as u16 pairs
            cw[0] = __builtin_amdgcn_perm(hi0, lo0, 0x05010400u);
            cw[1] = __builtin_amdgcn_perm(hi0, lo0, 0x07030602u);
            cw[2] = __builtin_amdgcn_perm(hi1, lo1, 0x05010400u);
            cw[3] = __builtin_amdgcn_perm(hi1, lo1, 0x07030602u);
            if (lists) {
                const uint2 sb = *reinterpret_cast<const uint2 *>(cnt8 + (((uint32_t)g * 64u + lane) * 2u));
                cw[0] += __builtin_amdgcn_perm(0u, sb.x, 0x0C010C00u);
                cw[1] += __builtin_amdgcn_perm(0u, sb.x, 0x0C030C02u);
                cw[2] += __builtin_amdgcn_perm(0u, sb.y, 0x0C010C00u);
                cw[3] += __builtin_amdgcn_perm(0u, sb.y, 0x0C030C02u);
            }
            const uint32_t rl0 = ((uint32_t)g * L + lane) * 8u;  // local reference of j = 0
            uint32_t m8 = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t c = (cw[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu;
                if (c > h_thr && rl0 + (uint32_t)j < in_tile) m8 |= 1u << j;
            }
            if (!active) m8 = 0;
            const uint32_t n8 = (uint32_t)__popc(m8);
            const uint32_t incl = wave_incl_scan_u32(n8);
            uint32_t pos = running + incl - n8;
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (m8 & (1u << j)) {
                    const uint32_t c = (cw[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu;
                    if (pos < p.rec.seg_len) seg[pos] = (rl0 + (uint32_t)j) | (c << 13);  // (beyond it: the run is repeated with longer segments, below)
                    pos++;
                    atomicAdd(&hist_lds[c], 1u);
                }
            running += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        wave_lds_sync();
        for (uint32_t m = lane; m <= t; m += 64) {  // only bins above u can be non-empty
            const uint32_t v = hist_lds[m];
            if (v) { atomicAdd(&hist[m], v); mx = m; }
        }
    }
    if (lane == 0u) {
        atomicAdd(&hist[0], in_tile - running);  // the references of the tile with a count up to the threshold
        if (running > p.rec.seg_len) atomicOr(p.rec.flags_out, 8u);  // the segment was too short: the host doubles it and repeats the run
        p.rec.cnt[(size_t)q * kRecMaxSlots + slot] = running < p.rec.seg_len ? running : p.rec.seg_len;
        if (running == 0u)  // nothing of this tile can reach the result: as if it had never been counted
            atomicAnd(const_cast<uint32_t *>(p.live) + (size_t)q * p.live_words + (tile >> 5), ~(1u << (tile & 31u)));
    }
    if (p.tile_max) {
        mx = wave_max_u32(mx);
        if (lane == 0) p.tile_max[(size_t)q * p.ntiles + tile] = (uint16_t)mx;
    }
}

// slot of `tile` among the record segments of query q (RecordRef), or 0xFFFFFFFF: the query takes the dense epilogue
__device__ __forceinline__ uint32_t rec_slot_of(const HitParams &p, uint32_t q, uint32_t tile, uint32_t lane) {
    if (!p.rec.nslots) return 0xFFFFFFFFu;
    const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.rec.nslots[q]);
    if (n == 0u) return 0xFFFFFFFFu;
    const uint32_t v = lane < n ? (uint32_t)p.rec.slots[(size_t)q * kRecMaxSlots + lane] : 0xFFFFFFFFu;
    const unsigned long long b = __ballot(v == tile);
    return b ? (uint32_t)__builtin_ctzll(b) : 0xFFFFFFFFu;
}

// ---------------------------------------------------------------------------
// Epilogue of the BOUNDS pass of the tile pruning (rtx_prune.hip): the "references" of this launch are blocks of 2^kPruneShift
// references of the database (the union bitmap), a count is an upper bound of the counts of the block's members.  Nothing is stored
// per block: what prune_kernel needs is the largest bound of every tile of the DATABASE (8192 >> kPruneShift blocks: with blocks of 64
// the 128 blocks that group gi of the 16 lanes of one DPP row hold) and the block with the largest bound of all (the lowest one among
// equals): key = bound << 20 | (0xFFFFF - block), the maximum over the wave, atomicMax over the waves of the query's union tiles.
// No histogram, no count stores, no lists (every row of the union bitmap is read densely).
// ---------------------------------------------------------------------------
template <int NP>
__device__ __forceinline__ void bounds_epilogue(const HitParams &p, uint32_t (&pl)[4][NP], uint32_t q, uint32_t utile, uint32_t lane) {
    static_assert(kPruneShift == 6, "128 blocks per tile of the database = one group of 8 per lane of a DPP row");
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    auto pkmax = [](uint32_t a, uint32_t b) -> uint32_t {
        u16x2 x, y;
        __builtin_memcpy(&x, &a, 4);
        __builtin_memcpy(&y, &b, 4);
        const u16x2 m = __builtin_elementwise_max(x, y);
        uint32_t r;
        __builtin_memcpy(&r, &m, 4);
        return r;
    };
    uint32_t best = 0;  // key of this lane's best block
    uint16_t *tub = p.bounds_tile_ub + (size_t)q * p.bounds_tile_stride;
#pragma unroll
    for (int w = 0; w < 4; w++) {
#pragma unroll
        for (int g2 = 0; g2 < 4; g2++) {
            const uint32_t gi = (uint32_t)(w * 4 + g2);
            uint32_t lo0, hi0, lo1, hi1;
            planes_unpack8<NP>(pl[w], g2, lo0, hi0, lo1, hi1);
            // (count << 3) | (7 - j) of the eight blocks as packed u16: the maximum is the largest count, the lowest block among equals
            const uint32_t vx = (__builtin_amdgcn_perm(hi0, lo0, 0x05010400u) << 3) | 0x00060007u;
            const uint32_t vy = (__builtin_amdgcn_perm(hi0, lo0, 0x07030602u) << 3) | 0x00040005u;
            const uint32_t vz = (__builtin_amdgcn_perm(hi1, lo1, 0x05010400u) << 3) | 0x00020003u;
            const uint32_t vw = (__builtin_amdgcn_perm(hi1, lo1, 0x07030602u) << 3) | 0x00000001u;
            const uint32_t m2 = pkmax(pkmax(vx, vy), pkmax(vz, vw));
            const uint32_t m = (m2 & 0xFFFFu) > (m2 >> 16) ? (m2 & 0xFFFFu) : (m2 >> 16);
            const uint32_t blk0 = utile * 8192u + gi * 512u + lane * 8u;
            const uint32_t key = ((m >> 3) << 20) | (0xFFFFFu - blk0 - 7u + (m & 7u));  // block blk0 + j, j = 7 - (m & 7)
            best = key > best ? key : best;
            const uint32_t tmax = row16_max_u32(m >> 3);
            const uint32_t T = utile * 64u + gi * 4u + (lane >> 4);
            if ((lane & 15u) == 0u && T < p.bounds_ntiles) tub[T] = (uint16_t)tmax;
        }
    }
    best = wave_max_u32(best);
    if (lane == 0) atomicMax(&p.bounds_best[q], best);
}

// ---------------------------------------------------------------------------
// Epilogue of the FINE bounds pass (two-stage bounds): the "references" of this launch are blocks of 8 references,
// one wave has counted a query against the 8192 blocks of fine tile `utile` = the eight tiles utile * 8 .. + 7 of the database.  With
// ref_slot's layout group gi of every lane (byte gi of its 16) holds blocks of tile utile * 8 + gi / 2: the low half of plane word w is
// tile 2 w, the high half tile 2 w + 1.  A tile whose every block bound is at most the query's threshold cannot hold a count above it:
// its live bit is cleared and its references go to bin 0 of the histogram -- exactly what the counting pass would have found out at
// the price of a block of its own (the epilogue's early exit).  The comparison runs on the bit planes (bound > u: ~3 operations per
// plane and word); `bits`: the live tiles of the query among the eight.
// ---------------------------------------------------------------------------
template <int NP>
__device__ __forceinline__ void fine_epilogue(const HitParams &p, uint32_t (&pl)[4][NP], uint32_t q, uint32_t utile, uint32_t lane, uint32_t bits) {
    const uint32_t u = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.prune_thr[q]);
    if (u == 0u || bits == 0u) return;  // (a query without a threshold keeps every tile)
    uint32_t above = 0;  // bit k: some block of tile utile * 8 + k has a bound above u
#pragma unroll
    for (int w = 0; w < 4; w++) {
        uint32_t gt = 0, eq = 0xFFFFFFFFu;
#pragma unroll
        for (int b = NP - 1; b >= 0; b--) {
            const uint32_t cb = (u >> b) & 1u ? 0xFFFFFFFFu : 0u;  // scalar
            gt |= eq & pl[w][b] & ~cb;
            eq &= ~(pl[w][b] ^ cb);
        }
        if (__ballot((gt & 0xFFFFu) != 0u) != 0ull) above |= 1u << (2 * w);
        if (__ballot((gt >> 16) != 0u) != 0ull) above |= 2u << (2 * w);
    }
    uint32_t clear = bits & ~above;  // wave-uniform
    if (clear == 0u || lane != 0u) return;
    const uint32_t T0 = utile * 8u;
    uint32_t refs = 0, n = 0;
    for (uint32_t k = 0; k < 8u; k++)
        if ((clear >> k) & 1u) {
            const uint64_t lo = (uint64_t)(T0 + k) << 13, hi = lo + 8192u < p.fine_n_refs ? lo + 8192u : p.fine_n_refs;
            refs += (uint32_t)(hi - lo);
            n++;
        }
    atomicAnd(const_cast<uint32_t *>(p.live) + (size_t)q * p.live_words + (T0 >> 5), ~(clear << (T0 & 31u)));
    atomicAdd(&p.hist[(size_t)q * p.hstride], refs);
    if (p.fine_stats) atomicAdd(&p.fine_stats[(size_t)(q & (kPruneStatCopies - 1u)) * 8u], (unsigned long long)n);
}

template <int NP, bool kPacked, bool kPrefetch = false, bool kGlobalHist = false>
__device__ __forceinline__ void hit_epilogue(const HitParams &p, uint32_t (&pl)[4][NP], uint32_t q, uint32_t tile, uint32_t lane,
                                             uint32_t t, bool active, uint32_t *hist_lds, uint32_t *cnt8, uint32_t ns,
                                             const uint32_t *srows) {
    uint4 pre[kSparseIt][kSparseV];  // unused without kPrefetch
    hit_epilogue_x<NP, kPacked, kPrefetch, false, false, kGlobalHist>(p, pl, q, tile, lane, t, active, hist_lds, cnt8, ns, srows, pre,
                                                         p.prune_thr ? (uint32_t)__builtin_amdgcn_readfirstlane((int)p.prune_thr[q]) : 0u);
}

}  // namespace rtx
