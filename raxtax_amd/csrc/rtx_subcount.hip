// The counting pass of the RECORDS path over SUB-TILES of 512 references (round 5; src/raxtax.rs:41,58-68 for the references that can
// matter to a pruned query).
//
// A pruned query on the records path (RecordRef: threshold u, a handful of live tiles) needs nothing but its references with a count
// above u.  hit_count_pair_kernel counts all 8192 references of a live tile for that -- ~600 row loads of 1 KiB per (pair, tile) block --
// although the bounds over blocks of 64 references say where in the tile such a reference can sit at all: on the bench workload in one
// or two blocks, the query's own species.  Here a wave takes ONE such query and a B-TILE (4 tiles of the database) in which it
// has a live tile (the row ids come straight from the query's row list: no lists in LDS, 3 waves per SIMD -- the kernel is a chain of short
// folds, and the waves in flight are what hides their round trips):
//   1. the bounds over blocks of 64 of the B-tile (level B of rtx_bounds2.hip: rows of 64 bytes, sixteen per load instruction) against the
//      threshold, on the bit planes: which blocks can hold a count above u -> which SUB-TILES of 512 references (8 blocks: a byte of the
//      row).  A live tile without such a block is dead: its references go to bin 0 as one number, as the epilogue's early exit does.
//   2. every such sub-tile is counted from the database bitmap stored in sub-tiles ([sub-tile][row][64 bytes], bit j = reference
//      512 sub + j): again sixteen rows per load instruction, ~60 instructions where the tile took ~600;
//   3. the counts above u leave as (reference, count) records in reference order, with the histogram entries, the tile's largest count
//      and the number of records -- exactly what rec_epilogue (rtx_hit_common.hpp) leaves for records_tail_kernel.
// The references of the sub-tiles that are not counted have a count of at most u (a block's bound is an upper bound of its members'
// counts): to everything downstream they are references without a hit, like those of a tile that is not counted.
// Queries that are not on the records path (no threshold, or many live tiles) stay with hit_count_pair_kernel; it leaves out the
// queries this kernel takes (HitParams::sub_skip).
#include <hip/hip_runtime.h>

#include "rtx_fold_r.hpp"

namespace rtx {

// bit b of the result: counter b of the word holds a value > c
template <int NP>
__device__ __forceinline__ uint32_t word_gt(const uint32_t (&pl)[NP], uint32_t c) {
    uint32_t gt = 0, eq = 0xFFFFFFFFu;
#pragma unroll
    for (int b = NP - 1; b >= 0; b--) {
        const uint32_t cb = (c >> b) & 1u ? 0xFFFFFFFFu : 0u;  // scalar
        gt |= eq & pl[b] & ~cb;
        eq &= ~(pl[b] ^ cb);
    }
    return gt;
}

// Eight load instructions of sixteen rows each, the row ids straight from the query's row list in global memory (kmer_extract: ascending,
// padded with the zero row to a multiple of 64): lane group g takes the entries unit * 128 + 8 g .. + 8.
__device__ __forceinline__ void load_unit_q(uint4 (&buf)[8], __amdgpu_buffer_rsrc_t rsrc, const uint32_t *rows, uint32_t nr_pad, uint32_t unit, uint32_t grp,
                                            uint32_t col, uint32_t zero_row) {
    const uint32_t i0 = unit * 128u + grp * 8u;
    uint4 a = make_uint4(zero_row, zero_row, zero_row, zero_row), b = a;
    if (i0 < nr_pad) {  // (nr_pad is a multiple of 64: a group of eight lies inside the list or behind it)
        a = *reinterpret_cast<const uint4 *>(rows + i0);
        b = *reinterpret_cast<const uint4 *>(rows + i0 + 4u);
    }
    const uint32_t id[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (id[j] << 6) + col, 0, 0);
        buf[j] = make_uint4(v.x, v.y, v.z, v.w);
    }
}

// All rows of ONE query against a region of 64-byte rows (a B-tile of the union bitmap, a sub-tile of the database): two buffers of
// eight instructions in flight (16 x 16 rows: three buffers spill at the three waves per SIMD the kernel wants), a query of 640 rows is 5
// units -- three groups.  The planes start at zero.
#ifndef RTX_SUB_NBUF
#define RTX_SUB_NBUF 2  // buffers of eight load instructions in flight per wave
#endif
#ifndef RTX_SUB_WAVES
#define RTX_SUB_WAVES 2  // waves per SIMD the kernel is compiled for
#endif
constexpr int kSubBuf = RTX_SUB_NBUF;
static_assert(kSubBuf == 2 || kSubBuf == 3, "two or three buffers");
template <int NP>
__device__ __forceinline__ uint32_t fold_query(uint32_t (&pl)[4][NP], uint4 (&buf)[kSubBuf][8], __amdgpu_buffer_rsrc_t rsrc, const uint32_t *rows, uint32_t nr_pad,
                                               uint32_t n_units, uint32_t grp, uint32_t col, uint32_t zero_row) {
#pragma unroll
    for (int w = 0; w < 4; w++)
#pragma unroll
        for (int b = 0; b < NP; b++) pl[w][b] = 0;
    const uint32_t ng = (n_units + (uint32_t)kSubBuf - 1u) / (uint32_t)kSubBuf;
#pragma unroll
    for (int b = 0; b < kSubBuf; b++) load_unit_q(buf[b], rsrc, rows, nr_pad, (uint32_t)b, grp, col, zero_row);
    for (uint32_t g = 0; g < ng; g++) {
        uint4 c[kSubBuf];
#pragma unroll
        for (int b = 0; b < kSubBuf; b++) {
            c[b] = tree8<NP>(pl, buf[b]);
            if (g + 1u < ng) load_unit_q(buf[b], rsrc, rows, nr_pad, (g + 1u) * (uint32_t)kSubBuf + (uint32_t)b, grp, col, zero_row);  // wave-uniform
        }
        if constexpr (kSubBuf == 2) ripple4<NP, 4>(pl, csa_plane<NP, 3>(pl, c[0], c[1]));
        else ripple4<NP, 5>(pl, csa_plane<NP, 4>(pl, csa_plane<NP, 3>(pl, c[0], c[1]), half_plane<NP, 3>(pl, c[kSubBuf - 1])));
    }
    return ng * 8u * (uint32_t)kSubBuf;
}

// One item: a records-path query and a B-tile (4 tiles of the database) in which it has a live tile.
template <int NP>
__device__ __forceinline__ void sub_item(const SubCountParams &p, uint32_t q, uint32_t bt, uint32_t lane) {
    const uint32_t T0 = bt * 4u;  // (a multiple of 4: the four live bits lie in one word)
    uint32_t *lw = p.live + (size_t)q * p.live_words + (T0 >> 5);
    const uint32_t live4 = (uint32_t)__builtin_amdgcn_readfirstlane((int)((lw[0] >> (T0 & 31u)) & 0xFu));
    if (live4 == 0u) return;
    const uint32_t u = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.prune_thr[q]);
    const uint32_t nr = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.nrows[q]);
    const uint32_t nr_pad = (nr + 63u) & ~63u, n_units = (nr + 127u) >> 7;
    const uint32_t *rows = p.rows + (size_t)q * p.rstride;
    uint32_t pl[4][NP];
    uint4 buf[kSubBuf][8];
    uint32_t n_instr = 0;
    const uint32_t row4 = lane >> 4, sub4 = lane & 3u, grp16 = lane >> 2;
    const bool first4 = (lane & 12u) == 0u;  // the lanes 16 r + c: one of the four copies of every (word r, sub-lane c)

    // ---- 1. bounds over blocks of 64 of the four tiles against the threshold: the sub-tiles that can hold a count above u
    uint32_t nib;  // this lane's word: bit k <-> byte k of the word = sub-tile 4 r + k of tile T0 + c holds a block above u
    {
        const char *base = reinterpret_cast<const char *>(p.bbitmap) + (size_t)bt * p.n_rows1 * 64u;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, p.n_rows1 * 64u, 0x00027000);
        n_instr += fold_query<NP>(pl, buf, rsrc, rows, nr_pad, n_units, grp16, sub4 * 16u, p.zero_row);
        uint32_t r[NP];
        reduce_rows16<NP>(pl, lane, r);
        const uint32_t gt = word_gt<NP>(r, u);
        nib = ((gt & 0xFFu) ? 1u : 0u) | ((gt & 0xFF00u) ? 2u : 0u) | ((gt & 0xFF0000u) ? 4u : 0u) | ((gt & 0xFF000000u) ? 8u : 0u);
    }
    // ---- 2. tile by tile, sub-tile by sub-tile (ascending: the records of a tile leave in reference order)
    for (uint32_t c = 0; c < 4u; c++) {
        const uint32_t T = T0 + c;
        if (T >= p.ntiles) break;
        if (!((live4 >> c) & 1u)) continue;
        const uint32_t slot = rec_slot_find(p.rec, q, T, lane);
        if (slot >= p.rec.stride) continue;  // (a live tile of a records-path query is one of its slots: anything else would be a corrupt mask)
        uint32_t m = 0;  // sub-tiles of this tile to count (16 bits: sub-tile 4 r + k <- bit k of lane 16 r + c)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            // (the lane index must be a constant per unrolled step: the four candidates c are read, one is taken)
            const uint32_t a0 = (uint32_t)__builtin_amdgcn_readlane((int)nib, 16 * r + 0), a1 = (uint32_t)__builtin_amdgcn_readlane((int)nib, 16 * r + 1),
                           a2 = (uint32_t)__builtin_amdgcn_readlane((int)nib, 16 * r + 2), a3 = (uint32_t)__builtin_amdgcn_readlane((int)nib, 16 * r + 3);
            m |= (c == 0u ? a0 : (c == 1u ? a1 : (c == 2u ? a2 : a3))) << (4 * r);
        }
        const uint32_t in_tile = (((uint64_t)T + 1u) << 13) <= p.n_refs ? 8192u : (uint32_t)(p.n_refs - ((uint64_t)T << 13));
        uint32_t run = 0, mx = 0;  // wave-uniform: records written, largest count among them
        uint32_t *seg = p.rec.rec + ((size_t)q * p.rec.stride + slot) * 8192u;
        uint32_t *hist = p.hist + (size_t)q * p.hstride;
        while (m) {
            const uint32_t k = (uint32_t)__builtin_ctz(m);
            m &= m - 1u;
            const uint32_t sub = T * 16u + k;
            const char *base = reinterpret_cast<const char *>(p.sbitmap) + (size_t)sub * p.n_rows1 * 64u;
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(base), 0, p.n_rows1 * 64u, 0x00027000);
            n_instr += fold_query<NP>(pl, buf, rsrc, rows, nr_pad, n_units, grp16, sub4 * 16u, p.zero_row);
            uint32_t r[NP];
            reduce_rows16<NP>(pl, lane, r);  // word row4 of sub-lane sub4: references 512 sub + 128 sub4 + 32 row4 + bit
            if (p.flags & RTX_SKIP_EXACT_MATCHES) {  // raxtax.rs:65-68: the counters of the query's exact matches are cleared
                const uint64_t qin = p.perm[p.q0 + q];
                uint64_t e0, e1;
                const uint32_t *xids;
                exact_range(p.exact, qin, e0, e1, xids);
                for (uint64_t e = e0; e < e1; e++) {  // wave-uniform
                    const uint32_t id = xids[e];
                    if ((uint64_t)id < p.n_refs && (id >> 9) == sub && ((id >> 7) & 3u) == sub4 && ((id >> 5) & 3u) == row4) {
                        const uint32_t msk = ~(1u << (id & 31u));
#pragma unroll
                        for (int b = 0; b < NP; b++) r[b] &= msk;
                    }
                }
            }
            uint32_t gt = first4 ? word_gt<NP>(r, u) : 0u;
            if (__ballot(gt != 0u) == 0ull) continue;  // wave-uniform: nothing above the threshold in this sub-tile
            // reference order: (sub-lane, word, bit) = key 4 sub4 + row4 of the lanes 16 row4 + sub4
            const uint32_t n_mine = (uint32_t)__popc(gt), key = sub4 * 4u + row4;
            uint32_t before = 0, total = 0;
#pragma unroll
            for (int kk = 0; kk < 16; kk++) {
                const uint32_t n_k = (uint32_t)__builtin_amdgcn_readlane((int)n_mine, 16 * (kk & 3) + (kk >> 2));
                before += (uint32_t)kk < key ? n_k : 0u;
                total += n_k;
            }
            uint32_t pos = run + before, my_max = 0;
            const uint32_t ref0 = k * 512u + sub4 * 128u + row4 * 32u;  // local reference (within the tile) of bit 0
            while (gt) {
                const uint32_t b = (uint32_t)__builtin_ctz(gt);
                gt &= gt - 1u;
                uint32_t cnt = 0;
#pragma unroll
                for (int pp = 0; pp < NP; pp++) cnt |= ((r[pp] >> b) & 1u) << pp;
                if (pos < 8192u) seg[pos] = (ref0 + b) | (cnt << 13);
                pos++;
                atomicAdd(&hist[cnt], 1u);
                my_max = cnt > my_max ? cnt : my_max;
            }
            run += total;
            mx = umax(mx, wave_max_u32(my_max));
        }
        // what rec_epilogue leaves behind a tile: the number of records, the references up to the threshold in bin 0, the largest count,
        // and a cleared live bit if nothing of the tile can reach the result (as if it had never been counted)
        if (lane == 0u) {
            p.rec.cnt[(size_t)q * kRecMaxSlots + slot] = run;
            atomicAdd(&hist[0], in_tile - run);
            if (p.tile_max) p.tile_max[(size_t)q * p.ntiles + T] = (uint16_t)mx;
            if (run == 0u) atomicAnd(p.live + (size_t)q * p.live_words + (T >> 5), ~(1u << (T & 31u)));
        }
    }
    if (lane == 0u) {
        if (p.group_rows) atomicAdd(&p.group_rows[p.group_base + (q >> 1)], n_instr);
        if (p.stats) atomicAdd(&p.stats[(size_t)(q & (kPruneStatCopies - 1u)) * 8u + 4u], 1ull);  // (query, B-tile) items
    }
}

// The grid walks the list of (query, B-tile) items like hit_count_pair_kernel walks its blocks: XCD x takes the x-th eighth of the list,
// workgroups beyond the first pass take the next entry of the XCD's queue (the resident ones stay on one stretch of the list).
template <int NP>
__global__ __launch_bounds__(64, RTX_SUB_WAVES) void subcount_kernel(SubCountParams p) {
    const uint32_t lane = threadIdx.x;
    const uint32_t n_items = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.n_items[0]), g8 = gridDim.x >> 3, x = blockIdx.x & 7u;
    const uint32_t e8 = (n_items + 7u) >> 3, first = x * e8;
    const uint32_t end = first + e8 < n_items ? first + e8 : n_items;
    uint32_t j = first + (blockIdx.x >> 3);
    while (j < end) {
        const uint32_t item = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.items[j]);
        uint32_t lane_v = lane;
        asm volatile("" : "+v"(lane_v));  // nothing that depends on the lane is kept across items
        sub_item<NP>(p, item / p.n_btiles, item % p.n_btiles, lane_v);
        if (e8 <= g8) break;  // every entry had a workgroup of its own
        uint32_t nxt = 0;
        if (lane_v == 0u) nxt = atomicAdd(&p.n_items[1u + x], 1u);
        j = first + g8 + (uint32_t)__builtin_amdgcn_readfirstlane((int)nxt);
    }
}

// ---- the list of items: (query, B-tile) with a live tile of a records-path query, grouped by B-tile (count, scan, scatter)
__device__ __forceinline__ bool sub_query_has(const uint32_t *live, uint32_t live_words, uint32_t q, uint32_t U) {
    const uint32_t T0 = U * 4u;
    return ((live[(size_t)q * live_words + (T0 >> 5)] >> (T0 & 31u)) & 0xFu) != 0u;
}
__global__ __launch_bounds__(256) void sub_count_items_kernel(const uint32_t *__restrict__ live, uint32_t live_words, const uint16_t *__restrict__ nslots,
                                                              uint32_t nq, uint32_t n_btiles, uint32_t *__restrict__ cnt) {
    const uint32_t q = blockIdx.x * 256u + threadIdx.x, lane = threadIdx.x & 63u;
    const bool on = q < nq && nslots[q < nq ? q : 0u] != 0u;
    if (__ballot(on) == 0ull) return;
    for (uint32_t U = 0; U < n_btiles; U++) {
        const unsigned long long b = __ballot(on && sub_query_has(live, live_words, q, U));
        if (b && lane == (uint32_t)__builtin_ctzll(b)) atomicAdd(&cnt[U], (uint32_t)__popcll(b));
    }
}
__global__ void sub_scan_items_kernel(uint32_t *__restrict__ cnt, uint32_t n_btiles, uint32_t *__restrict__ n_items) {  // one wave
    const uint32_t lane = threadIdx.x;
    uint32_t run = 0;
    for (uint32_t U0 = 0; U0 < n_btiles; U0 += 64) {
        const uint32_t c = U0 + lane < n_btiles ? cnt[U0 + lane] : 0u;
        const uint32_t incl = wave_incl_scan_u32(c);
        if (U0 + lane < n_btiles) cnt[U0 + lane] = run + incl - c;  // becomes the cursor of B-tile U
        run += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    if (lane == 0u) n_items[0] = run;
    if (lane >= 1u && lane <= 8u) n_items[lane] = 0;  // the queues of the XCDs
}
__global__ __launch_bounds__(256) void sub_scatter_items_kernel(const uint32_t *__restrict__ live, uint32_t live_words, const uint16_t *__restrict__ nslots,
                                                                uint32_t nq, uint32_t n_btiles, uint32_t *__restrict__ cursor, uint32_t *__restrict__ items) {
    const uint32_t q = blockIdx.x * 256u + threadIdx.x, lane = threadIdx.x & 63u;
    const bool on = q < nq && nslots[q < nq ? q : 0u] != 0u;
    if (__ballot(on) == 0ull) return;
    for (uint32_t U = 0; U < n_btiles; U++) {
        const bool has = on && sub_query_has(live, live_words, q, U);
        const unsigned long long b = __ballot(has);
        if (b == 0ull) continue;  // wave-uniform
        const int leader = __builtin_ctzll(b);
        uint32_t base = 0;
        if ((int)lane == leader) base = atomicAdd(&cursor[U], (uint32_t)__popcll(b));
        base = (uint32_t)__shfl((int)base, leader, 64);
        if (has) items[base + (uint32_t)__popcll(b & ((1ull << lane) - 1ull))] = q * n_btiles + U;
    }
}

// cnt: [n_btiles] scratch; items: [queries * n_btiles]; n_items: [9]
void launch_subcount(hipStream_t s, const SubCountParams &p, uint32_t nq, uint32_t *cnt, uint32_t *items, uint32_t *n_items, int planes) {
    const uint32_t nb = (nq + 255u) / 256u;
    (void)hipMemsetAsync(cnt, 0, (size_t)p.n_btiles * 4, s);
    hipLaunchKernelGGL(sub_count_items_kernel, dim3(nb), dim3(256), 0, s, p.live, p.live_words, p.rec.nslots, nq, p.n_btiles, cnt);
    hipLaunchKernelGGL(sub_scan_items_kernel, dim3(1), dim3(64), 0, s, cnt, p.n_btiles, n_items);
    hipLaunchKernelGGL(sub_scatter_items_kernel, dim3(nb), dim3(256), 0, s, p.live, p.live_words, p.rec.nslots, nq, p.n_btiles, cnt, items);
    SubCountParams q = p;
    q.items = items;
    q.n_items = n_items;
    // two items' worth of workgroups per query and pass, never more than the items there can be (a multiple of 8: the XCDs)
    const dim3 grid((uint32_t)((std::min<uint64_t>((uint64_t)nq * p.n_btiles, std::max<uint64_t>(2ull * nq, 2048ull)) + 7u) & ~7ull));
    if (planes <= 8) hipLaunchKernelGGL((subcount_kernel<8>), grid, dim3(64), 0, s, q);
    else hipLaunchKernelGGL((subcount_kernel<10>), grid, dim3(64), 0, s, q);
}

// ---------------------------------------------------------------------------
// The database bitmap in sub-tiles from the tile-major bitmap (ref_slot layout): a thread per word.  Byte k of word (lane l, word wi)
// of tile T holds the references 8192 T + ((4 wi + k) L + l) 8 + [0, 8) (L = lanes of the tile: 64 but in the last one): one byte of a
// sub-tile row (plain bit order).  The target is zeroed first; only the non-zero bytes are written.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void subtile_build_kernel(const uint32_t *__restrict__ bitmap, uint32_t n_rows1, uint32_t ntiles, uint32_t stride_bytes,
                                                            uint8_t *__restrict__ sbitmap) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;  // [tile][row][256 words]
    const uint64_t total = (uint64_t)ntiles * n_rows1 * 256u;
    if (i >= total) return;
    const uint32_t v = bitmap[i];
    if (v == 0u) return;
    const uint32_t word = (uint32_t)(i & 255u), row = (uint32_t)((i >> 8) % n_rows1), tile = (uint32_t)((i >> 8) / n_rows1);
    const uint32_t l = word >> 2, wi = word & 3u, L = tile_lanes(stride_bytes, tile);
    if (l >= L) return;
#pragma unroll
    for (uint32_t k = 0; k < 4u; k++) {
        const uint32_t byte = (v >> (8u * k)) & 0xFFu;
        if (byte == 0u) continue;
        const uint32_t ref0 = tile * 8192u + ((wi * 4u + k) * L + l) * 8u;
        sbitmap[((size_t)(ref0 >> 9) * n_rows1 + row) * 64u + ((ref0 & 511u) >> 3)] = (uint8_t)byte;
    }
}

void launch_subtile_build(hipStream_t s, const uint32_t *bitmap, uint32_t n_rows1, uint32_t ntiles, uint32_t stride_bytes, uint8_t *sbitmap) {
    const uint64_t total = (uint64_t)ntiles * n_rows1 * 256u;
    hipLaunchKernelGGL(subtile_build_kernel, dim3((uint32_t)((total + 255u) / 256u)), dim3(256), 0, s, bitmap, n_rows1, ntiles, stride_bytes, sbitmap);
}

}  // namespace rtx
