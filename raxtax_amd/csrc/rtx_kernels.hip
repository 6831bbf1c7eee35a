// Hand-written gfx950 (CDNA4, wave64) kernels of the raxtax classification hot path.
//
//   bitmap_build      Tree.k_mer_map (CSR)  -> per-k-mer reference bitmaps   (index build, once)
//   kmer_extract      src/utils.rs:17-40    -> sorted distinct 8-mers, t, bitmap-row list
//   hit_count         src/raxtax.rs:41,58-68 + prob.rs:13-19 -> count[r] (u16), histogram
//   prob_table        src/prob.rs:20-103    -> table[m]/Z, Z, global signal
//   taxon_prefix      src/lineage.rs:61-66  -> prefix sums of p_r at taxonomy boundaries
//   lineage_walk      src/lineage.rs:114-179 -> result rows (node, rounded confidences)
//
// No MFMA: the path is integer bit-slicing + f64 VALU, bounded by HBM/L2 bandwidth
// (DESIGN.md has the roofline of every kernel).  Wave = 64 lanes everywhere.
#include <hip/hip_runtime.h>

#include "rtx_kernels.hpp"
#include "rtx_math.hpp"
#include "rtx_wave.hpp"
#include "rtx_hit_common.hpp"
#include "rtx_walk.hpp"

namespace rtx {

static constexpr uint32_t kEmptyRow = 0xFFFFFFFFu;

// ---------------------------------------------------------------------------
// bitmap_build: one workgroup per k-mer; sets bit r of row row_of[k] for every posting r.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bitmap_build_kernel(const uint64_t *__restrict__ off,
                                                           const uint32_t *__restrict__ post,
                                                           const uint32_t *__restrict__ row_of,
                                                           uint32_t *__restrict__ bitmap, uint32_t stride_words, uint32_t n_rows1,
                                                           uint32_t ref_lo, uint32_t ref_hi, uint32_t shift) {  // bit: ref_slot(); shift > 0: the union bitmap of blocks of 2^shift references (rtx_prune.hip)
    const uint32_t k = blockIdx.x;
    const uint32_t row = row_of[k];
    if (row == kEmptyRow) return;
    const uint64_t b = off[k], e = off[k + 1];
    for (uint64_t i = b + threadIdx.x; i < e; i += blockDim.x) {
        const uint32_t g = post[i];
        if (g < ref_lo || g >= ref_hi) continue;  // reference held by another shard
        uint32_t word, bit;
        ref_slot((g - ref_lo) >> shift, stride_words * 4u, word, bit);
        atomicOr(&bitmap[bitmap_word(row, word, n_rows1)], 1u << bit);
    }
}

// ---------------------------------------------------------------------------
// Index build straight from the encoded reference sequences (the k-mer part of Tree::new,
// src/tree.rs:114-123,134-137).  With one bitmap row per k-mer no sort or
// dedup pass is needed: setting bit `ref` of row `k` is idempotent.
//   ref_kmer_mark : which of the 65536 k-mers occur at all (-> row_of on the host)
//   ref_bitmap_set: bit (row_of[k], ref) for every valid window of every reference
//   row_popcount  : posting-list lengths (= set bits per row), for the work accounting
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool window_kmer(const uint8_t *__restrict__ seq, uint64_t w, uint32_t &k) {
    k = 0;
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t c = seq[w + j];
        ok = ok && (c == 1u || c == 2u || c == 4u || c == 8u);
        k |= ((uint32_t)__ffs((int)c) - 1u) << (14 - 2 * j);
    }
    k &= 0xFFFFu;
    return ok;
}

__global__ __launch_bounds__(64) void ref_kmer_mark_kernel(const uint8_t *__restrict__ bases,
                                                           const uint64_t *__restrict__ off, uint64_t n_refs,
                                                           uint32_t *__restrict__ present /*2048 words*/) {
    __shared__ uint32_t bm[2048];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 2048; i += 64) bm[i] = 0;
    __syncthreads();
    for (uint64_t r = blockIdx.x; r < n_refs; r += gridDim.x) {
        const uint64_t b0 = off[r], len = off[r + 1] - b0;
        for (uint64_t w = lane; w + 8 <= len; w += 64) {
            uint32_t k;
            if (window_kmer(bases + b0, w, k)) atomicOr(&bm[k >> 5], 1u << (k & 31u));
        }
    }
    __syncthreads();
    for (uint32_t i = lane; i < 2048; i += 64)
        if (bm[i]) atomicOr(&present[i], bm[i]);
}

__global__ __launch_bounds__(64) void ref_bitmap_set_kernel(const uint8_t *__restrict__ bases,
                                                            const uint64_t *__restrict__ off, uint64_t n_refs,
                                                            const uint32_t *__restrict__ row_of,
                                                            uint32_t *__restrict__ bitmap, uint32_t stride_words, uint32_t n_rows1,
                                                            uint32_t shift) {  // shift > 0: the union bitmap of blocks of 2^shift references
    const uint64_t r = blockIdx.x;
    if (r >= n_refs) return;
    const uint32_t lane = threadIdx.x;
    const uint64_t b0 = off[r], len = off[r + 1] - b0;
    uint32_t word, bitpos;
    ref_slot((uint32_t)(r >> shift), stride_words * 4u, word, bitpos);
    const uint32_t bit = 1u << bitpos;
    for (uint64_t w = lane; w + 8 <= len; w += 64) {
        uint32_t k;
        if (window_kmer(bases + b0, w, k)) atomicOr(&bitmap[bitmap_word(row_of[k], word, n_rows1)], bit);
    }
}

__global__ __launch_bounds__(256) void row_popcount_kernel(const uint32_t *__restrict__ row_of,
                                                           const uint32_t *__restrict__ bitmap, uint32_t stride_words, uint32_t n_rows1,
                                                           uint32_t *__restrict__ list_len) {
    __shared__ uint32_t part[4];
    const uint32_t k = blockIdx.x;
    const uint32_t row = row_of[k];
    if (row == kEmptyRow) {
        if (threadIdx.x == 0) list_len[k] = 0;
        return;
    }
    uint32_t c = 0;
    for (uint32_t i = threadIdx.x; i < stride_words; i += 256) c += __popc(bitmap[bitmap_word(row, i, n_rows1)]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
    if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) list_len[k] = part[0] + part[1] + part[2] + part[3];
}

// ---------------------------------------------------------------------------
// kmer_extract (src/utils.rs:17-40): one wave per query.
// A 65536-bit set in LDS gives HashSet semantics; reading it out word by word in
// ascending order gives `.sorted()`.  Also emits, for hit_count, the list of bitmap rows
// of the k-mers that occur in the index (padded with the all-zero row to a multiple of
// 32 plus one look-ahead group) and H_q = sum of posting-list lengths.
// ---------------------------------------------------------------------------
// 64 x 64 bit-matrix transpose across the wave: lane l gives row l, receives column l (six butterfly stages).
__device__ __forceinline__ unsigned long long transpose64(unsigned long long x, uint32_t lane) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        // bits b with (b & s) == 0
        const unsigned long long m = s == 32 ? 0x00000000FFFFFFFFull : s == 16 ? 0x0000FFFF0000FFFFull : s == 8 ? 0x00FF00FF00FF00FFull
                                   : s == 4 ? 0x0F0F0F0F0F0F0F0Full : s == 2 ? 0x3333333333333333ull : 0x5555555555555555ull;
        const unsigned long long y = __shfl_xor(x, s, 64);
        x = (lane & (uint32_t)s) ? (((y >> s) & m) | (x & ~m)) : ((x & m) | ((y & m) << s));
    }
    return x;
}

// kTilesOnly: the launch behind tile pruning (mode 2) -- without the 8 KB k-mer set of the extraction in LDS twice as many of its waves fit a CU
// (a chain of gathers per query: the waves in flight are what hides them)
template <bool kTilesOnly>
__global__ __launch_bounds__(64) void kmer_extract_kernel(KmerParams p) {
    const uint32_t mode = kTilesOnly ? 2u : p.mode;
    __shared__ __attribute__((aligned(16))) uint32_t bm[kTilesOnly ? 64 : 2048];
    const uint32_t q = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    const uint64_t gq = p.q0 + q;  // position in the processing order: index of the per-query outputs
    const uint64_t qin = p.perm[gq];
    const uint64_t b0 = p.base_off[qin];
    const uint64_t len = p.base_off[qin + 1] - b0;
    const uint8_t *seq = p.bases + b0;

    // mode: 0 = everything; 1 = the k-mers and the row list only; 2 = the per-tile lists only, for the tiles tile pruning
    // left alive (the launch of mode 1 has left the row list; rtx_prune.hip decides between the two)
    uint32_t t = 0, nrows = 0;
    unsigned long long hq = 0;
    uint32_t *rout = p.rows + (size_t)q * p.rstride;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    if (mode == 2u) {
        nrows = p.nrows[q];
    } else {
    for (uint32_t i = lane; i < 2048; i += 64) bm[i] = 0;
    {  // the histogram row hit_count accumulates into with global atomics
        uint32_t *h = p.hist + (size_t)q * p.hstride;
        for (uint32_t i = lane; i < p.hstride; i += 64) h[i] = 0;
    }
    __syncthreads();

    if (len >= 8) {
        // lane l takes the windows [l*wpl, (l+1)*wpl), sixteen at a time: their 23 bases come in as three unaligned
        // 8-byte words (one round trip instead of 23 dependent byte loads; the batch buffer is padded behind its end)
        // and the 2-bit code rolls over them in registers
        const uint64_t nwin = len - 7;
        const uint64_t wpl = (nwin + 63) / 64;
        const uint64_t w0 = (uint64_t)lane * wpl;
        const uint64_t w1 = w0 + wpl < nwin ? w0 + wpl : nwin;
        for (uint64_t i = w0; i < w1; i += 16) {
            unsigned long long v[3];
            __builtin_memcpy(v, seq + i, 24);
            const uint32_t nw = w1 - i < 16 ? (uint32_t)(w1 - i) : 16u;
            uint32_t k = 0, run = 0;  // run: consecutive valid bases ending here
#pragma unroll
            for (int b = 0; b < 23; b++) {
                const uint32_t c = (uint32_t)(v[b >> 3] >> ((b & 7) * 8)) & 0xFFu;
                // one-hot nibble {1,2,4,8} -> {0,1,2,3}; everything else invalidates the windows that contain it
                const bool ok = c == 1u || c == 2u || c == 4u || c == 8u;
                k = ((k << 2) | (((uint32_t)__ffs((int)c) - 1u) & 3u)) & 0xFFFFu;  // first base of the window in bits 15:14
                run = ok ? run + 1u : 0u;
                if (b >= 7 && (uint32_t)(b - 7) < nw && run >= 8u) atomicOr(&bm[k >> 5], 1u << (k & 31u));
            }
        }
    }
    __syncthreads();

    // ascending read-out: round r covers the 64-bit words r*64 .. r*64+63 of the set (conflict-free LDS reads); sixteen
    // scans instead of thirty-two with 32-bit words
    uint16_t *kout = p.kmers + (size_t)q * p.kstride;
    const unsigned long long *bm64 = reinterpret_cast<const unsigned long long *>(bm);
    uint32_t base = 0;
    for (uint32_t r = 0; r < 16; r++) {
        unsigned long long word = bm64[r * 64 + lane];
        const uint32_t cnt = (uint32_t)__popcll(word);
        const uint32_t incl = wave_incl_scan_u32(cnt);
        uint32_t pos = base + incl - cnt;
        const uint32_t kbase = (r * 64 + lane) * 64;
        while (word) {
            const uint32_t bit = (uint32_t)__builtin_ctzll(word);
            word &= word - 1;
            if (pos < p.kstride) kout[pos] = (uint16_t)(kbase + bit);
            pos++;
        }
        base += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    t = base;
    __syncthreads();  // kout visible to the whole wave
    // A read that holds EVERY 8-mer (t = 65 536; only reads of 65 543 bases or more can): the reference asserts that t fits a u16
    // (raxtax.rs:56) and aborts.  Here the query is reported (RTX_Q_ALL_KMERS, set by finalise_kernel from t_all) and counted as one
    // without k-mers; the other queries of the batch are not affected.
    const bool all_kmers = t > 65535u;

    // rows of the k-mers present in the index, in ascending k-mer order (the query's row list, shared by all tiles)
    const uint32_t tt = all_kmers ? 0u : (t < p.kstride ? t : p.kstride);
    // Four chunks of 64 k-mers per turn, all loads of a level issued together: on gfx9 loads and stores share one
    // in-order counter, so every wait for a load also waits for the stores before it -- a chunk-by-chunk loop
    // (load, gather, store) pays two full round trips per chunk.
    for (uint32_t i0 = 0; i0 < tt; i0 += 256) {
        uint32_t kk[4], row[4], ll[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t i = i0 + u * 64 + lane;
            kk[u] = i < tt ? (uint32_t)kout[i] : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            row[u] = kEmptyRow;
            ll[u] = 0;
            if (kk[u] != 0xFFFFFFFFu) {
                const uint2 rl = p.row_len[kk[u]];  // {row, posting-list length}: one gather
                row[u] = rl.x;
                ll[u] = rl.y;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            hq += ll[u];
            const unsigned long long m = __ballot(row[u] != kEmptyRow);
            if (row[u] != kEmptyRow) rout[nrows + __popcll(m & lt_mask)] = row[u];
            nrows += (uint32_t)__popcll(m);
        }
    }
    for (uint32_t i = nrows + lane; i < ((nrows + 63u) & ~63u); i += 64) rout[i] = p.zero_row;
    __syncthreads();  // rout visible to the whole wave
    }  // mode != 2
    if (mode == 1u) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) hq += __shfl_xor(hq, d, 64);
        if (lane == 0) { p.t[q] = t > 65535u ? 0u : t; p.nrows[q] = nrows; p.hq[gq] = hq; p.t_all[gq] = t; }
        return;
    }
    const uint32_t nchunks = (nrows + 63u) >> 6;
    const uint32_t *live = mode == 2u ? p.live + (size_t)q * p.live_words : nullptr;  // the tiles that are counted for this query (rtx_prune.hip)
    // Per tile: which rows have a dense segment there (a 64-bit mask per 64 rows), and the slots of the sparse
    // segments; empty segments are dropped (rtx_segments.hip).
    const uint32_t nt = p.ntiles;
    const uint32_t mstride = p.rstride >> 6;
    unsigned long long *dm = p.dmask + (size_t)q * nt * mstride;
    uint32_t *sout = p.srows + (size_t)q * nt * (kSegMaxSparseRows + 1);
    uint32_t nseg = 0;
    __shared__ unsigned long long l_sb[64];
    __shared__ uint32_t l_base[64];
    // Tile pruning left this query's pair a handful of tiles: one pass per live tile over the tile's class table (two bits per row)
    // instead of the bit tables and their transposes, which cost the same however few tiles are wanted.
    bool few_done = false;
    if (live && p.seg_blocks) {
        const uint32_t lw = (nt + 31u) >> 5;
        uint32_t nlive = 0;
        for (uint32_t w = 0; w < lw; w++) nlive += (uint32_t)__popc(live[w]);
        if (nlive <= kKmerFewLive) {  // wave-uniform
            for (uint32_t w = 0; w < lw; w++) {
                uint32_t bits = (uint32_t)__builtin_amdgcn_readfirstlane((int)live[w]);
                while (bits) {
                    const uint32_t tile = w * 32u + (uint32_t)__builtin_ctz(bits);
                    bits &= bits - 1u;
                    if (tile >= nt) break;
                    uint32_t cd = 0, cs = 0;  // wave-uniform
                    for (uint32_t c0 = 0; c0 < nchunks; c0 += 4) {
                        uint32_t row[4], cls[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const uint32_t i = (c0 + (uint32_t)u) * 64 + lane;
                            row[u] = i < nrows ? rout[i] : kEmptyRow;
                        }
#pragma unroll
                        for (int u = 0; u < 4; u++) {  // two bits per row from the tile's own table (16 KB: L1 / L2)
                            const uint32_t r = row[u] == kEmptyRow ? 0u : row[u];
                            cls[u] = (p.segcls[(size_t)tile * p.cls_stride + (r >> 4)] >> ((r & 15u) * 2u)) & 3u;
                            if (row[u] == kEmptyRow) cls[u] = 0u;
                        }
                        // as in the pass per tile below: the rows taken from a chunk are a prefix of its sparse candidates.  Their slots
                        // come from seginfo -- a gather for the few lanes that hold one (at most kSegMaxSparseRows per tile), the four
                        // chunks' together
                        uint32_t srank[4], slot[4];
                        bool take[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const uint32_t c = c0 + (uint32_t)u;
                            take[u] = false;
                            srank[u] = 0u;
                            if (c >= nchunks) continue;  // wave-uniform
                            const bool sparse = cls[u] == 2u;
                            const unsigned long long ms = __ballot(sparse);
                            srank[u] = cs + (uint32_t)__popcll(ms & lt_mask);
                            take[u] = sparse && srank[u] < kSegMaxSparseRows;
                            const unsigned long long bt = __ballot(take[u]);
                            const unsigned long long md = __ballot(cls[u] == 1u || (sparse && !take[u]));
                            if (lane == 0) dm[(size_t)tile * mstride + c] = md;
                            cd += (uint32_t)__popcll(md);
                            cs += (uint32_t)__popcll(bt);
                        }
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            slot[u] = 0u;
                            if (take[u]) slot[u] = p.seginfo[(size_t)row[u] * p.seg_stride + tile] - 2u;
                        }
#pragma unroll
                        for (int u = 0; u < 4; u++)
                            if (take[u]) sout[(size_t)tile * (kSegMaxSparseRows + 1) + srank[u]] = slot[u];
                    }
                    if (lane == 0) {
                        p.nsparse[(size_t)q * nt + tile] = cs;
                        nseg += cd;
                    }
                }
            }
            few_done = true;
        }
    }
    for (uint32_t tb = 0; p.seg_blocks && !few_done && tb < nt; tb += 64) {  // many tiles: 64 rows x 64 tiles per step
        const uint32_t blk = tb >> 6;
        const uint32_t tile = tb + lane;  // this lane's tile after the transposes
        // a tile that is not counted for this query's pair needs no lists (the transposes still take every lane)
        const bool tlive = tile < nt && (!live || ((live[tile >> 5] >> (tile & 31u)) & 1u));
        uint32_t cd = 0, cs = 0;
        uint4 pend = make_uint4(0, 0, 0, 0);  // slot ids of this lane's tile waiting for their 16-byte store
        // the class tables of four chunks of 64 rows are gathered together (one round trip per four chunks, not one each)
        unsigned long long dbv[4], sbv[4];
        uint32_t basev[4];
        for (uint32_t c = 0; c < nchunks; c++) {
            if ((c & 3u) == 0) {
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t i = (c + (uint32_t)u) * 64 + lane;
                    const uint32_t row = i < nrows ? rout[i] : kEmptyRow;
                    const size_t at = (size_t)(row == kEmptyRow ? 0u : row) * p.seg_blocks + blk;  // unconditional loads, masked below
                    dbv[u] = p.seg_dbits[at];
                    sbv[u] = p.seg_sbits[at];
                    basev[u] = p.seg_sbase[at];
                    if (row == kEmptyRow) { dbv[u] = 0; sbv[u] = 0; }
                }
            }
            unsigned long long db = 0, sb = 0;
            uint32_t base = 0;
#pragma unroll
            for (int u = 0; u < 4; u++)
                if ((c & 3u) == (uint32_t)u) { db = dbv[u]; sb = sbv[u]; base = basev[u]; }
            l_sb[lane] = sb;
            l_base[lane] = base;
            unsigned long long dT = transpose64(db, lane), sT = transpose64(sb, lane);  // bit r = row c*64 + r
            __syncthreads();
            if (tlive) {
                // the first sparse rows go to the slot list (the byte counters of hit_count hold 255 hits: at most
                // kSegMaxSparseRows sparse segments) ...
                // (four slot ids per 16-byte store: every lane writes to a list of its own, so each store is a memory
                // transaction of its own, and those are what bounds this kernel -- a quarter as many)
                while (sT && cs < kSegMaxSparseRows) {
                    const int r = __builtin_ctzll(sT);
                    sT &= sT - 1;
                    const uint32_t sid = l_base[r] + (uint32_t)__popcll(l_sb[r] & lt_mask);
                    const uint32_t k4 = cs & 3u;
                    pend.x = k4 == 0u ? sid : pend.x;
                    pend.y = k4 == 1u ? sid : pend.y;
                    pend.z = k4 == 2u ? sid : pend.z;
                    pend.w = k4 == 3u ? sid : pend.w;
                    if (k4 == 3u) *reinterpret_cast<uint4 *>(sout + (size_t)tile * (kSegMaxSparseRows + 1) + (cs - 3u)) = pend;
                    cs++;
                }
                dT |= sT;  // ... the rest is read densely
                dm[(size_t)tile * mstride + c] = dT;
                cd += (uint32_t)__popcll(dT);
            }
            __syncthreads();
        }
        if (tlive) {
            // the last, incomplete group of four (the entries behind cs are never used: the lists have 256 entries)
            if (cs & 3u) *reinterpret_cast<uint4 *>(sout + (size_t)tile * (kSegMaxSparseRows + 1) + (cs & ~3u)) = pend;
            p.nsparse[(size_t)q * nt + tile] = cs;
            nseg += cd;
        }
    }
    if (!p.seg_blocks) {  // few tiles (at most 12): one pass per tile; lane l keeps the counters of tile l
        uint32_t cd = 0, cs = 0;
        const uint32_t nv = (nt + 3u) >> 2;  // seginfo rows are padded to whole uint4: four tiles per (gather) load
        for (uint32_t c0 = 0; c0 < nchunks; c0 += 4) {  // four chunks per turn, loads of a level together (see above)
            uint32_t row[4];
            uint4 iv[4][3];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t i = (c0 + u) * 64 + lane;
                row[u] = i < nrows ? rout[i] : kEmptyRow;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint4 *info = reinterpret_cast<const uint4 *>(p.seginfo + (size_t)(row[u] == kEmptyRow ? 0u : row[u]) * p.seg_stride);
#pragma unroll
                for (uint32_t v = 0; v < 3; v++) iv[u][v] = v < nv ? info[v] : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t c = c0 + u;
                if (c >= nchunks) break;
                for (uint32_t tile = 0; tile < nt; tile++) {
                    const uint32_t v = tile >> 2;
                    const uint4 q4 = v == 0 ? iv[u][0] : (v == 1 ? iv[u][1] : iv[u][2]);
                    const uint32_t word = (tile & 3u) == 0 ? q4.x : (tile & 3u) == 1 ? q4.y : (tile & 3u) == 2 ? q4.z : q4.w;
                    const uint32_t code = row[u] != kEmptyRow ? word : 0u;
                    const uint32_t ns = (uint32_t)__builtin_amdgcn_readlane((int)cs, (int)tile);
                    const bool sparse = code >= 2u;  // hit_count's byte counters hold 255 hits: more sparse rows are read densely
                    // The rank counts the sparse candidates in front of a row (taken so far + earlier ones of this chunk): it only
                    // grows along the rows, so the rows taken from a chunk are a prefix of its candidates -- their ranks are
                    // positions in the list -- and a row is only taken while fewer than the cap have been
                    const unsigned long long ms = __ballot(sparse);
                    const uint32_t srank = ns + (uint32_t)__popcll(ms & lt_mask);
                    const bool take = sparse && srank < kSegMaxSparseRows;
                    if (take) sout[(size_t)tile * (kSegMaxSparseRows + 1) + srank] = code - 2u;
                    const unsigned long long bt = __ballot(take);
                    const unsigned long long md = __ballot(code == 1u || (sparse && !take));
                    if (lane == tile) {
                        dm[(size_t)tile * mstride + c] = md;
                        cd += (uint32_t)__popcll(md);
                        cs = ns + (uint32_t)__popcll(bt);
                    }
                }
            }
        }
        if (lane < nt) {
            p.nsparse[(size_t)q * nt + lane] = cs;
            nseg += cd;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { hq += __shfl_xor(hq, d, 64); nseg += __shfl_xor(nseg, d, 64); }
    if (lane == 0) {
        if (mode != 2u) {
            p.t[q] = t > 65535u ? 0u : t;  // (a read with every 8-mer: see above)
            p.nrows[q] = nrows;
            p.hq[gq] = hq;
            p.t_all[gq] = t;
        }
        p.nrows_all[gq] = nseg;
    }
}

// ---------------------------------------------------------------------------
// hit_count (src/raxtax.rs:41,58-68): one wave per (query, 8192-reference tile).
// Lane l owns 16 bytes of every bitmap row = 128 references of the tile (which ones: ref_slot,
// rtx_math.hpp), so a wave reads 1 KiB contiguous per row (one global_load_dwordx4 per lane, row base in
// SGPRs).  Rows are folded 32 at a time into NP bit planes per 32-reference word with a
// Harley-Seal carry-save tree (rtx_math.hpp: 31 CSAs + one ripple per 32 rows).  The epilogue zeroes exact matches (raxtax.rs:65-68), unpacks the planes
// to u16 counts, stores them, and builds the hit-count histogram of prob.rs:13-19 with
// LDS atomics, flushed with one global atomic per non-empty bin.
// blockIdx.x = query (fast) so that concurrently resident waves work on the same
// reference tile and popular rows are served from L2.
// ---------------------------------------------------------------------------
// (load8v, tree8, csa_plane and the epilogue shared with hit_count_pair_kernel: rtx_hit_common.hpp)

// Occupancy matters here: with <= 128 VGPRs four waves per SIMD are resident (16 per CU; their 8.4 KB of LDS each
// just fit) -- a variant with 132 VGPRs (three waves) was 16 % slower.  The bound makes the compiler keep it.
#ifndef RTX_HIT_NB
#define RTX_HIT_NB 2
#endif

template <int NP, bool kPacked, bool kGlobalHist = false>
#if RTX_HIT_NB > 2  // RTX_HIT_NB buffers of eight rows in flight per wave, 256 registers, two waves per SIMD (three with NB = 3)
__global__ __launch_bounds__(64, (RTX_HIT_NB == 3 ? 3 : 2)) void hit_count_kernel(HitParams p) {
#else
__global__ __launch_bounds__(64, (NP <= 10 ? 4 : 2)) void hit_count_kernel(HitParams p) {
#endif
    extern __shared__ uint32_t hist_lds[];
    const uint32_t tile = blockIdx.y, lane = threadIdx.x;
    // Workgroups are dealt round-robin to the 8 XCDs (each with an L2 of its own).  Neighbouring slots hold
    // related queries (rtx_cluster.hip): XCD x takes the contiguous slice [x*nq/8, (x+1)*nq/8) of the sub-batch,
    // so that the rows a cluster shares are fetched into one L2 instead of eight.
    const uint32_t nq8 = gridDim.x >> 3;
    const uint32_t q = blockIdx.x < nq8 * 8u ? (blockIdx.x & 7u) * nq8 + (blockIdx.x >> 3) : blockIdx.x;
    const uint32_t t = p.t[q];
    // LDS: [row-id list | histogram] (the list during the row loop, the histogram in the epilogue), then 4 KiB of byte
    // counters for the hits through sparse segments (one half of the tile at a time, in the epilogue)
    uint32_t *cnt8 = hist_lds + p.lds_cnt8_off;  // [1024] dwords = 4096 byte counters, indexed by local id % 4096
    const uint32_t ns = p.nsparse[(size_t)q * p.ntiles + tile];  // 0 in a partial last tile (kmer_extract)
    const uint32_t *srows = p.srows + ((size_t)q * p.ntiles + tile) * (kSegMaxSparseRows + 1);
    const uint32_t col = tile * 1024u + lane * 16u;
    const bool active = col < p.stride_bytes;
    uint32_t pl[4][NP];
#pragma unroll
    for (int w = 0; w < 4; w++)
#pragma unroll
        for (int b = 0; b < NP; b++) pl[w][b] = 0;

    {
        // The dense rows of this tile are the set bits of the tile's masks over the query's row list (the same 2.6 KB
        // for every tile of the query: it stays in L2).  They are compacted into an LDS list first -- 64 rows per step,
        // a lane whose bit is set writes its row at the rank of the bit -- so that the row loop has no data-dependent
        // control flow: it reads 32 row ids per group with one ds_read and takes them out with v_readlane.
        const uint32_t *rows = p.rows + (size_t)q * p.rstride;
        const unsigned long long *masks = p.dmask + ((size_t)q * p.ntiles + tile) * (p.rstride >> 6);
        const uint32_t nchunks = (p.nrows[q] + 63u) >> 6;
        const __amdgpu_buffer_rsrc_t rsrc = tile_rsrc(p.bitmap, p.n_rows1, tile);
        const uint32_t voff = lane * 16u;
        const uint32_t zero_off = p.zero_row << 10;  // the list holds row offsets inside the tile's region (row << 10)
        uint32_t *list = hist_lds;
        const unsigned long long lt_mask = (1ull << lane) - 1ull;
        uint32_t chunk = 0;
        while (chunk < nchunks) {  // one round unless more than kHitListCap - 63 dense rows (the last group of a round is padded)
            uint32_t count = 0;
            // The masks of up to 64 chunks come in with ONE load (lane c <-> chunk + c, taken out with v_readlane) and the
            // row ids four chunks at a time: three or four round trips in front of the first bitmap row instead of two
            // per chunk (mask, then the rows of its set bits).
            const unsigned long long mv = chunk + lane < nchunks ? masks[chunk + lane] : 0ull;
            uint32_t ci = 0;
            bool room = true;
            while (room && chunk < nchunks && ci < 64u) {
                uint32_t rowv[4];
#pragma unroll
                for (int u = 0; u < 4; u++) rowv[u] = chunk + u < nchunks ? rows[(chunk + u) * 64 + lane] : 0u;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (chunk >= nchunks || ci >= 64u) break;
                    if (count + 64u > kHitListCap) { room = false; break; }
                    const unsigned long long m = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(mv >> 32), (int)ci) << 32) |
                                                 (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mv, (int)ci);
                    if ((m >> lane) & 1ull) list[count + (uint32_t)__popcll(m & lt_mask)] = rowv[u] << 10;
                    count += (uint32_t)__popcll(m);
                    chunk++;
                    ci++;
                }
            }
            // groups of 32 rows in the main loop, then up to three groups of 8 (the list is padded to a multiple of 8 only;
            // the entries behind it, read by the look-ahead loads, are zero rows)
            const uint32_t n8 = (count + 7u) >> 3, n32 = n8 >> 2, ntail = n8 & 3u;
#if RTX_HIT_NB > 2
            for (uint32_t i = count + lane; i < n8 * 8u + RTX_HIT_NB * 8u + 64u; i += 64) list[i] = zero_off;
#else
            for (uint32_t i = count + lane; i < n32 * 32u + 32u; i += 64) list[i] = zero_off;
#endif
            __syncthreads();
#if RTX_HIT_NB > 2
            if (n8) fold_ring<NP, RTX_HIT_NB>(pl, list, n8, lane, rsrc, voff);
#else
            if (n8) {
                uint32_t idv = list[lane & 31u];
                uint4 A[8], B[8];
                load8v<0>(A, rsrc, voff, idv);
                for (uint32_t g = 0; g < n32; g++) {
                    const uint32_t idn = list[(g + 1) * 32 + (lane & 31u)];
                    load8v<8>(B, rsrc, voff, idv);
                    const uint4 c3a = tree8<NP>(pl, A);
                    load8v<16>(A, rsrc, voff, idv);
                    const uint4 c3b = tree8<NP>(pl, B);
                    const uint4 c4a = csa_plane<NP, 3>(pl, c3a, c3b);
                    load8v<24>(B, rsrc, voff, idv);
                    const uint4 c3c = tree8<NP>(pl, A);
                    load8v<0>(A, rsrc, voff, idn);  // first rows of the next group, or of the tail
                    const uint4 c3d = tree8<NP>(pl, B);
                    const uint4 c4b = csa_plane<NP, 3>(pl, c3c, c3d);
                    const uint4 c5 = csa_plane<NP, 4>(pl, c4a, c4b);
                    planes_ripple<NP, 5>(pl[0], c5.x);
                    planes_ripple<NP, 5>(pl[1], c5.y);
                    planes_ripple<NP, 5>(pl[2], c5.z);
                    planes_ripple<NP, 5>(pl[3], c5.w);
                    idv = idn;
                }
                if (ntail) {  // A holds the first eight rows of the tail; the others are loaded one group at a time
                    uint4 c3 = tree8<NP>(pl, A);
                    planes_ripple<NP, 3>(pl[0], c3.x);
                    planes_ripple<NP, 3>(pl[1], c3.y);
                    planes_ripple<NP, 3>(pl[2], c3.z);
                    planes_ripple<NP, 3>(pl[3], c3.w);
                    if (ntail > 1) {
                        load8v<8>(A, rsrc, voff, idv);
                        c3 = tree8<NP>(pl, A);
                        planes_ripple<NP, 3>(pl[0], c3.x);
                        planes_ripple<NP, 3>(pl[1], c3.y);
                        planes_ripple<NP, 3>(pl[2], c3.z);
                        planes_ripple<NP, 3>(pl[3], c3.w);
                    }
                    if (ntail > 2) {
                        load8v<16>(A, rsrc, voff, idv);
                        c3 = tree8<NP>(pl, A);
                        planes_ripple<NP, 3>(pl[0], c3.x);
                        planes_ripple<NP, 3>(pl[1], c3.y);
                        planes_ripple<NP, 3>(pl[2], c3.z);
                        planes_ripple<NP, 3>(pl[3], c3.w);
                    }
                }
            }
#endif
            __syncthreads();  // the list is rewritten (next round) or becomes the histogram
        }
    }
    hit_epilogue<NP, kPacked, false, kGlobalHist>(p, pl, q, tile, lane, t, active, hist_lds, cnt8, ns, srows);
}

template __global__ void hit_count_kernel<10, true>(HitParams);
template __global__ void hit_count_kernel<10, false>(HitParams);
template __global__ void hit_count_kernel<12, false>(HitParams);
template __global__ void hit_count_kernel<16, false>(HitParams);
template __global__ void hit_count_kernel<16, false, true>(HitParams);

// ---------------------------------------------------------------------------
// prob_table (src/prob.rs:8-103): one workgroup of kProbWaves waves per query.
// Lanes <-> distinct hit counts m in DESCENDING order (group g = 64 consecutive counts, handled
// by wave g mod kProbWaves), sequential over i = 0..n with the linear-domain recurrence of rtx_math.hpp.
//   pre-pass: i_lo from the largest count M (P(i) := 0 below it), per-group skip test.
//   pass 1 : P(i) = prod_m cmf_m(i)^hist[m] = exp(prod(i)) of prob.rs:62-73 for i >= i_lo, kept as a
//            product (integer powers by square-and-multiply, no logarithms), reduced over lanes by
//            a fixed DPP order and over waves through per-wave LDS slots (deterministic, unlike the
//            reference's ahash iteration order).  A group starts where its smallest count's pmf
//            reaches e^-100 and stops when all its cmfs have saturated; from there on it
//            contributes the constant `base`.
//   pass 2 : table[m] = sum_i pmf_m(i) * exp(prod(i)) / cmf_m(i)              (prob.rs:74-90)
// then Z = sum_m hist[m] table[m] (= probs_sum, prob.rs:97), table/Z (prob.rs:99-102) and the
// global signal ||p - 1/N||_2 (lineage.rs:86-90) from the histogram.
// LDS (dynamic): slots[4][n1max] | P[n1max] | inv[tmax+n1max+1] | red[16] | gbase[ng] | gbw[ng], gstart[ng] u32 | ms u16
// ---------------------------------------------------------------------------
static constexpr uint32_t kGroupSkipped = 0xFFFFFFFFu;

static constexpr uint32_t kProbWaves = 2;  // waves per query: groups g = wave, wave+2, ...
static constexpr uint32_t kProbThreads = kProbWaves * 64;

// kGlobal: reads of tens of kilobases (t up to 65 535) -- the arrays (3.4 bytes per k-mer ... 1.6 MB at the limit) do not fit LDS: the
// workgroup works in a stretch of global memory of its own (ProbParams::gscratch), the same code and the same arithmetic through flat
// addresses; __syncthreads orders the waves of the workgroup through it as it does through LDS (same CU, write-through L1).
template <bool kGlobal>
__global__ __launch_bounds__(kProbThreads) void prob_table_kernel(ProbParams p) {
    extern __shared__ double smem_lds[];
    double *const smem = kGlobal ? p.gscratch + (size_t)blockIdx.x * p.gstride : smem_lds;
    __shared__ uint32_t s_D, s_ilo;
    const uint32_t q = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint64_t gq = p.q0 + q;
    const uint32_t t = p.t[q];
    const uint32_t n = t >> 1;  // num_trials = k_mers.len() / 2, raxtax.rs:57
    const uint32_t n1 = n + 1;
    const uint32_t n1max = p.n1max;
    const uint32_t ngmax = (p.tmax + 64) / 64 + 1;
    double *slots = smem;
    double *Pi = slots + kProbWaves * (size_t)n1max;
    double *inv = Pi + n1max;
    double *red = inv + (p.tmax + n1max + 1);
    double *gbase = red + 16;
    uint32_t *gbw = reinterpret_cast<uint32_t *>(gbase + ngmax);
    uint32_t *gstart = gbw + ngmax;
    uint16_t *ms = reinterpret_cast<uint16_t *>(gstart + ngmax);
    const uint32_t *hist = p.hist + (size_t)q * p.hstride;
    double *tz = p.table_z + (size_t)q * p.hstride;
    const double *lf = p.lnfact;

    if (t == 0) {  // reference: u64 underflow at prob.rs:21
        if (tid == 0) { p.status[gq] = RTX_Q_NO_KMERS; p.z[gq] = 0.0; p.gs[gq] = 0.0; p.ndist[gq] = 0; }
        return;
    }
    // distinct counts, ascending (wave 0 compacts)
    if (wave == 0) {
        uint32_t D = 0;
        for (uint32_t m0 = 0; m0 <= t; m0 += 64) {
            const uint32_t m = m0 + lane;
            const bool has = m <= t && hist[m] != 0;
            const unsigned long long bal = __ballot(has);
            if (has) ms[D + __popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)m;
            D += (uint32_t)__popcll(bal);
        }
        if (lane == 0) { s_D = D; s_ilo = n; p.ndist[gq] = D; }
    }
    if (!kGlobal)
        for (uint32_t x = tid + 1; x <= t + n; x += kProbThreads) inv[x] = 1.0 / (double)x;
    for (uint32_t i = tid; i < kProbWaves * n1; i += kProbThreads) slots[i] = 1.0;
    __syncthreads();
    const uint32_t D = s_D;
    const double ln_total = ln_binom_tab(lf, t + n - 1, n);  // prob.rs:20-23
    const uint32_t M = ms[D - 1];
    const bool any_full = M == t;  // prob.rs:24-26

    if (any_full) {  // prob.rs:27-41
        for (uint32_t j = tid; j < D; j += kProbThreads) {
            const uint32_t m = ms[j];
            tz[m] = only_last_pmf_tab(lf, t, n, m, ln_total);
        }
    } else {
        if (n == 0) {  // reference: zip_eq length mismatch at prob.rs:162
            if (tid == 0) { p.status[gq] = RTX_Q_NO_KMERS; p.z[gq] = 0.0; p.gs[gq] = 0.0; }
            return;
        }
        // ---- pre-pass: first i at which P(i) can matter
        if (M > 0) {
            for (uint32_t i = tid; i <= n; i += kProbThreads)
                if (ln_pmf_tab(lf, t, n, M, i, ln_total) >= kLnNegligibleP) atomicMin(&s_ilo, i);
        } else if (tid == 0) {
            s_ilo = 0;
        }
        __syncthreads();
        const uint32_t i_lo = s_ilo;
        const uint32_t ngroups = (D + 63u) >> 6;
        // ---- pass 1: P(i) = prod_m cmf_m(i)^hist[m], accumulated multiplicatively (no logarithms)
        for (uint32_t g = wave; g < ngroups; g += kProbWaves) {
            const uint32_t j = g * 64 + lane;
            const uint32_t m = j < D ? ms[D - 1 - j] : 0u;
            const bool act = m != 0;
            const uint32_t m_hi = ms[D - 1 - g * 64];
            if (m_hi == 0 || group_negligible(lf, t, n, m_hi, i_lo, ln_total)) {  // wave-uniform
                if (lane == 0) { gbw[g] = kGroupSkipped; gbase[g] = 1.0; gstart[g] = 0; }
                continue;
            }
            // group start: first i at which the group's smallest count reaches e^-100 (<= i_lo)
            const uint32_t nl = D - g * 64 < 64u ? D - g * 64 : 64u;
            uint32_t m_lo = ms[D - 1 - (g * 64 + nl - 1)];
            if (m_lo == 0) m_lo = nl > 1 ? ms[D - 1 - (g * 64 + nl - 2)] : m_hi;
            uint32_t i_s = i_lo;
            for (uint32_t i0 = 0; i0 < i_lo; i0 += 64) {
                const uint32_t i = i0 + lane;
                const bool hit = i < i_lo && ln_pmf_tab(lf, t, n, m_lo, i, ln_total) >= kLnNegligibleP;
                const unsigned long long bal = __ballot(hit);
                if (bal) { i_s = i0 + (uint32_t)__ffsll((long long)bal) - 1u; break; }
            }
            const uint32_t h = act ? hist[m] : 0u;
            PmfState st{1.0, 1.0, 0};
            if (act) st = pmf_start_at(lf, t, n, m, i_s, ln_total);
            uint32_t bw = n + 1;
            double base = 1.0;
            // kGlobal: slots lives in global memory -- a read-modify-write per step would be a round trip per step.  The factors of 64
            // consecutive i are collected in a register (lane i & 63) and multiplied in with ONE coalesced access per 64 steps.
            double fbuf = 1.0;
            uint32_t f_lo = 0xFFFFFFFFu;  // first i of the chunk in flight that has a factor (wave-uniform)
            auto flush = [&](uint32_t i_last) {
                if (!kGlobal || f_lo == 0xFFFFFFFFu) return;
                const uint32_t idx = (i_last & ~63u) + lane;
                if (idx >= f_lo && idx <= i_last) slots[wave * n1 + idx] *= fbuf;
                f_lo = 0xFFFFFFFFu;
            };
            for (uint32_t i = i_s; i <= n; i++) {
                bool sat = !act;
                if (i > i_s && act) {
                    const double c_old = st.c;
                    const int k_old = st.k;
                    if (kGlobal) pmf_step(st, InvDiv(), t, n, m, i);
                    else pmf_step(st, inv, t, n, m, i);
                    sat = st.c == c_old && k_old == 0 && st.k == 0;  // pmf < 2^-53 cmf: cmf is final
                }
                if (i > i_s && __all(sat)) {
                    if (i > 0) flush(i - 1u);
                    base = wave_prod_f64_dpp(act ? pmf_cmf_pow(st, h) : 1.0);
                    bw = i;
                    break;
                }
                if (i >= i_lo) {
                    const double f = wave_prod_f64_dpp(act ? pmf_cmf_pow(st, h) : 1.0);
                    if (kGlobal) {
                        if (lane == (i & 63u)) fbuf = f;
                        if (f_lo == 0xFFFFFFFFu) f_lo = i;
                        if ((i & 63u) == 63u || i == n) flush(i);
                    } else if (lane == 0) {
                        slots[wave * n1 + i] *= f;
                    }
                }
            }
            if (lane == 0) { gbw[g] = bw; gbase[g] = base; gstart[g] = i_s; }
        }
        __syncthreads();
        for (uint32_t i = tid; i <= n; i += kProbThreads) {
            double P = 0.0;
            if (i >= i_lo) {
                P = slots[i];
#pragma unroll
                for (uint32_t w = 1; w < kProbWaves; w++) P *= slots[w * n1 + i];
                for (uint32_t g = 0; g < ngroups; g++) {
                    const uint32_t b = gbw[g];
                    if (b != kGroupSkipped && i >= b) P *= gbase[g];
                }
            }
            Pi[i] = P;
        }
        __syncthreads();
        // ---- pass 2
        for (uint32_t g = wave; g < ngroups; g += kProbWaves) {
            const uint32_t j = g * 64 + lane;
            if (!kGlobal && j >= D) continue;  // (kGlobal: every lane stays for the shuffles below, the spare ones on a stand-in count)
            const bool valid = j < D;
            const uint32_t m_raw = valid ? ms[D - 1 - j] : 1u;
            if (valid && m_raw == 0) tz[0] = Pi[0];  // pmf = [1,0,...], cmf = 1: table[0] = P(0)
            if (!kGlobal && m_raw == 0) continue;
            const uint32_t m = m_raw ? m_raw : 1u;
            const uint32_t b = gbw[g];
            if (b == kGroupSkipped) {  // wave-uniform
                if (valid && m_raw) tz[m] = 0.0;
                continue;
            }
            const uint32_t i_s = gstart[g];
            const uint32_t last = b == 0 ? 0u : (b - 1 < n ? b - 1 : n);
            PmfState st = pmf_start_at(lf, t, n, m, i_s, ln_total);
            double acc = 0.0;
            if (kGlobal) {  // P(i) 64 at a time: one coalesced load per 64 steps, handed out with a shuffle (the loop is wave-uniform)
                double Pc = 0.0;
                const uint32_t i_first = i_s > i_lo ? i_s : i_lo;
                for (uint32_t i = i_s; i <= last; i++) {
                    if (i > i_s) pmf_step(st, InvDiv(), t, n, m, i);
                    if (i < i_lo) continue;
                    if ((i & 63u) == 0u || i == i_first) { const uint32_t idx = (i & ~63u) + lane; Pc = Pi[idx <= n ? idx : n]; }
                    const double P = __shfl(Pc, (int)(i & 63u), 64);
                    if (P > 0.0 && st.k == 0 && st.c > 0.0) acc += st.v * P / st.c;
                }
            } else {
                for (uint32_t i = i_s; i <= last; i++) {
                    if (i > i_s) pmf_step(st, inv, t, n, m, i);
                    if (i < i_lo) continue;
                    const double P = Pi[i];
                    if (P > 0.0 && st.k == 0 && st.c > 0.0) acc += st.v * P / st.c;
                }
            }
            if (!(valid && m_raw)) continue;
            tz[m] = acc;
        }
    }
    __syncthreads();
    // Z = probs_sum (prob.rs:97) grouped by count value; fixed reduction order
    double part = 0.0;
    for (uint32_t j = tid; j < D; j += kProbThreads) {
        const uint32_t m = ms[j];
        part += (double)hist[m] * tz[m];
    }
    part = wave_sum_f64(part);
    if (lane == 0) red[wave] = part;
    __syncthreads();
    double Z = red[0];
#pragma unroll
    for (uint32_t w = 1; w < kProbWaves; w++) Z += red[w];
    __syncthreads();
    const double inv_n = 1.0 / (double)p.n_refs;
    double gsum = 0.0;
    for (uint32_t j = tid; j < D; j += kProbThreads) {
        const uint32_t m = ms[j];
        const double v = tz[m] / Z;  // prob.rs:99-102
        tz[m] = v;
        const double d = v - inv_n;
        gsum += (double)hist[m] * d * d;
    }
    gsum = wave_sum_f64(gsum);
    if (lane == 0) red[8 + wave] = gsum;
    __syncthreads();
    if (tid == 0) {
        p.z[gq] = Z;
        double gtot = red[8];
#pragma unroll
        for (uint32_t w = 1; w < kProbWaves; w++) gtot += red[8 + w];
        p.gs[gq] = sqrt(gtot);
        p.status[gq] = RTX_Q_OK;
    }
}

// ---------------------------------------------------------------------------
// lineage_walk (src/lineage.rs:114-179): one wave per query, explicit DFS stack; the 64
// lanes evaluate 64 children of the current node at once (confidence = P[hi]-P[lo],
// lineage.rs:114-117; rounding lineage.rs:128-129).  Rows (node id + per-level rounded
// confidences in hundredths) are staged in LDS and appended to a global arena with one
// atomic per query.  Sorting (lineage.rs:91-93) and the local signal happen on the host.
//
// A walk is a chain of dependent loads, so the kernel is built to keep that chain short: a node is one 16-byte
// record {blo, bhi, first_child, n_children | type << 30}, so the load that fetches the children's ranges also
// brings what is needed to descend into any of them (children records -> P -> decision: two dependent loads per
// level instead of three); everything about the nodes on the stack (record, position, the still unvisited
// significant children of the 64 scanned last) lives in LDS, so returning to a parent costs no global load and
// no second scan.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void lineage_walk_kernel(WalkParams p) {
    __shared__ WalkLds L;
    lineage_walk_wave(p, blockIdx.x, threadIdx.x, L, GapPrefix{p.prefix + (size_t)blockIdx.x * p.n_bnd, nullptr});
}

// ---------------------------------------------------------------------------
// taxon_prefix (src/lineage.rs:61-66): P[j] = sum_{r < bnd[j]} p_r, p_r = table[count_r]/Z,
// sampled at the taxonomy boundaries only (every node range is [bnd[a], bnd[b])).
// One workgroup of NW waves per query, 8 references per thread per sweep (NW*512 per sweep); the
// sweeps are sequential (running carry), so wide workgroups = fewer, better overlapped sweeps.
// ---------------------------------------------------------------------------
// eight counts, read once: non-temporal so that they do not displace bitmap rows / table rows in L2
__device__ __forceinline__ uint4 load_counts8(const uint16_t *p) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

// a reference whose probability is below this adds nothing that could be seen in a confidence (taxon_prefix, tile skipping)
static constexpr double kLiveEps = 1e-30;

template <int NW, bool TZ_LDS, bool PACKED>
__global__ __launch_bounds__(NW * 64) void taxon_prefix_kernel(PrefixParams p) {
    extern __shared__ double tz_lds[];
    __shared__ double wsum[2][NW];  // double-buffered: one barrier per sweep
    const uint32_t q = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint64_t gq = p.q0 + q;
    if (p.rec_nslots && p.rec_nslots[q] != 0u) return;  // a query on the records path: records_tail_kernel has it (rtx_records.hip)
    // a query of the dense path that was left without a row of the counts buffer (HitParams::cnt_row: the rows ran out, the host repeats the
    // run with more of them): nothing to sum -- no rows, so that nothing is made of counts that were never written
    if (p.status[gq] != RTX_Q_OK || (p.cnt_row && p.cnt_row[q] == 0xFFFFFFFFu)) {
        if (p.fuse_walk && tid == 0) { p.walk.n_rows[gq] = 0; p.walk.row_start[gq] = 0; }
        return;
    }
    const uint32_t crow = p.cnt_row ? p.cnt_row[q] : q;  // its row of the counts buffer and of the prefix sums (HitParams::cnt_row)
    double *__restrict__ P = p.prefix + (size_t)crow * p.n_bnd;
    const double *__restrict__ tzg = p.table_z + (size_t)q * p.hstride;
    // Most references of a large database share too few k-mers with the query to get any probability at all:
    // prob_lookup gives every count whose row has saturated below i_lo the value 0.0 EXACTLY (rtx_prob_tables.hip).
    // m_lo = the smallest count with a non-zero table entry; a wave whose 512 references all lie below it adds
    // nothing (+0.0) to the prefix and skips the look-ups, the sums and the scan (bit-identical results).
    // Tile skipping: hit_count leaves the largest count of every tile of 8192 references (p.tile_max).  m_live = the
    // smallest count whose probability reaches kLiveEps; a tile whose largest count lies below it holds no reference
    // with p >= kLiveEps and is not read at all: the boundaries inside it get the running sum.  What is dropped is
    // below n_refs * kLiveEps in every prefix value (confidences are differences of those, rounded to 1e-2).
    __shared__ uint32_t s_mlo, s_mlive;
    __shared__ PrefixGaps s_gaps;  // boundaries of unswept runs of tiles (fused walk only): not written, the walk is told
    if (tid == 0) s_gaps.n = 0;
    {
        const uint32_t t1 = p.t[q] + 1;
        if (tid == 0) { s_mlo = 0xFFFFFFFFu; s_mlive = 0xFFFFFFFFu; }
        __syncthreads();
        uint32_t mine = 0xFFFFFFFFu, mine_live = 0xFFFFFFFFu;
        for (uint32_t m = tid; m < t1; m += NW * 64) {
            const double v = tzg[m];
            if (TZ_LDS) tz_lds[m] = v;  // 8 random look-ups per reference chunk: serve them from LDS
            if (v != 0.0 && m < mine) mine = m;   // entries of absent counts are never written: whatever they hold only lowers m_lo
            if (v >= kLiveEps && m < mine_live) mine_live = m;
        }
        if (mine != 0xFFFFFFFFu) atomicMin(&s_mlo, mine);
        if (mine_live != 0xFFFFFFFFu) atomicMin(&s_mlive, mine_live);
        __syncthreads();
    }
    const uint32_t m_lo = (PACKED && TZ_LDS) ? s_mlo : 0u;
    // "some byte of x is >= m_lo" without unpacking: byte + (256 - m_lo) carries out of the byte (low 7 bits added
    // without crossing bytes, the carry out is the majority of the two top bits and the carry into them)
    const uint32_t ge_add = ((256u - (m_lo & 0xFFu)) & 0x7Fu) * 0x01010101u;
    const uint32_t ge_top = ((256u - (m_lo & 0xFFu)) & 0x80u) ? 0xFFFFFFFFu : 0u;
    // counts of this query: u16 per reference, or packed (low byte per reference + 2 high bits x 8 references per u16)
    const uint16_t *__restrict__ cnt = PACKED ? nullptr : p.counts + (size_t)crow * p.npad;
    const uint8_t *__restrict__ cnt_lo = PACKED ? p.counts_lo + (size_t)crow * p.npad : nullptr;
    const uint16_t *__restrict__ cnt_hi = PACKED ? p.counts_hi + (size_t)crow * (p.npad >> 3) : nullptr;
    if (tid == 0) P[0] = 0.0;
    double carry = 0.0;
    const uint32_t n = (uint32_t)p.n_refs;  // references of this handle (< 2^32)
    constexpr uint32_t kSweep = NW * 512;
    // Counts, boundary flags and boundary ranks of the next sweep are requested before the current one is scanned (the
    // sweeps are a serial chain through `carry`).  The loads are UNCONDITIONAL (clamped addresses, values
    // masked afterwards) and nothing else is loaded inside the loop: gfx9 counts loads and stores in one in-order
    // counter, and only with a fixed number of younger operations can the wait for this sweep's data leave the
    // next sweep's loads (and this sweep's boundary stores) in flight -- with predicated loads or a rank look-up
    // behind the scan every sweep waited for everything (vmcnt(0)) and exposed a full HBM round trip.
    const uint32_t last_chunk = (n - 1u) >> 3;  // n >= 1: every query slot of an index holds references
    uint4 cv_l;             // raw values of the chunk requested last (clamped address) ...
    uint32_t hi_l = 0, bits_l, rank_l;
    auto request = [&](uint32_t r) {
        const uint32_t ch = r >> 3 < last_chunk ? r >> 3 : last_chunk;  // a valid chunk (counts rows are padded to 8)
        if (PACKED) {
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            const u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(cnt_lo + (size_t)ch * 8u));
            cv_l = make_uint4(v.x, v.y, 0u, 0u);
            hi_l = __builtin_nontemporal_load(cnt_hi + ch);
        } else {
            cv_l = load_counts8(cnt + (size_t)ch * 8u);
        }
        bits_l = p.bnd_bits[ch];
        rank_l = p.bnd_rank[ch];
    };
    uint4 cv_next;          // ... and what they mean once masked
    uint32_t hi_next = 0, bits_next, rank_next;
    auto accept = [&](uint32_t r) {
        const bool in = r < n;
        cv_next = in ? cv_l : make_uint4(0, 0, 0, 0);  // past the end: count 0 (a valid table index, masked below), no boundary
        hi_next = in ? hi_l : 0u;
        bits_next = in ? bits_l : 0u;
        rank_next = rank_l;
    };
    const uint16_t *__restrict__ tmx = p.tile_max ? p.tile_max + (size_t)q * p.ntiles : nullptr;
    // a query that tile pruning gave a threshold u: every count up to u has probability 0 (prob_lookup), so a tile whose largest
    // count is at most u stays out -- among them the tiles hit_count left out, which keep a largest count of 0 and hold stale
    // counts (entries of absent counts in table_z may hold anything: the bound is enforced here, not read from the table)
    const uint32_t u_thr = p.prune_thr ? p.prune_thr[q] : 0u;
    const uint32_t m_live = u_thr && s_mlive <= u_thr ? u_thr + 1u : s_mlive;
    if (p.prune_stats && u_thr && wave == 0 && tmx) {  // reporting: how many tiles of this query hold a count above its threshold
        uint32_t need = 0;
        for (uint32_t T0 = 0; T0 < p.ntiles; T0 += 64) need += (uint32_t)__popcll(__ballot(T0 + lane < p.ntiles && (uint32_t)tmx[T0 + lane] > u_thr));
        if (lane < 2u) atomicAdd(&p.prune_stats[(size_t)(q & (kPruneStatCopies - 1u)) * 8u + lane], lane == 0u ? (unsigned long long)need : 1ull);
    }
    const uint32_t ntiles = (n + 8191u) >> 13;
    unsigned long long live_mask = ~0ull;  // liveness of the 64 tiles of group live_group (wave-uniform, the same in every wave)
    uint32_t live_group = 0xFFFFFFFFu;
    auto tile_live = [&](uint32_t T) -> bool {
        if (!tmx) return true;
        if ((T >> 6) != live_group) {
            live_group = T >> 6;
            const uint32_t Tl = live_group * 64u + lane;
            live_mask = __ballot(Tl < ntiles && (uint32_t)tmx[Tl] >= m_live);
        }
        return (live_mask >> (T & 63u)) & 1ull;
    };
    uint32_t filled = 1;  // P[0 .. filled) are written
    // boundaries [a, b) in front of / behind / between the swept runs: the running sum -- as a gap for the fused walk (while
    // the gap table has room), else written out (the walk as a kernel of its own, the exchange of the sharded modes)
    uint32_t n_gaps = 0;  // the same in every thread
    auto fill = [&](uint32_t a, uint32_t b) {
        if (a >= b) return;
        if (p.fuse_walk && n_gaps < kMaxPrefixGaps) {
            if (tid == 0) { s_gaps.lo[n_gaps] = a; s_gaps.hi[n_gaps] = b; s_gaps.val[n_gaps] = carry; s_gaps.n = n_gaps + 1u; }
            n_gaps++;
            return;
        }
        for (uint32_t i = a + tid; i < b; i += NW * 64) P[i] = carry;
    };
    uint32_t buf = 0;
    uint32_t T = 0;
    while (T < ntiles) {
        // the next run of live tiles [Ts, Te)
        while (T < ntiles && !tile_live(T)) T++;
        if (T >= ntiles) break;
        const uint32_t Ts = T;
        while (T < ntiles && tile_live(T)) T++;
        const uint32_t Te = T;
        // boundaries inside the dead tiles in front of the run: the running sum
        const uint32_t rb = p.bnd_rank[Ts * 1024u];
        fill(filled, rb);
        filled = Te * 1024u <= last_chunk ? p.bnd_rank[Te * 1024u] : p.n_bnd;
        const uint32_t span_end = Te * 8192u < n ? Te * 8192u : n;
        request(Ts * 8192u + tid * 8u);
        accept(Ts * 8192u + tid * 8u);
    for (uint32_t base = Ts * 8192u; base < span_end; base += kSweep, buf ^= 1u) {
        const uint32_t r0 = base + tid * 8u;
        const uint4 cv = cv_next;
        const uint32_t hi_cur = hi_next;
        const uint32_t bits_cur = bits_next, rank_cur = rank_next;
        request(r0 + kSweep);
        const uint32_t cw[4] = {cv.x, cv.y, cv.z, cv.w};
        bool live = true;  // wave-uniform: some reference of this wave's 512 may have a non-zero probability
        if (PACKED && TZ_LDS && m_lo != 0u) {
            bool mine = hi_cur != 0u;  // a count of 256 or more (conservative where m_lo is larger still)
            if (m_lo < 256u) {
                const uint32_t tx = (cv.x & 0x7F7F7F7Fu) + ge_add, ty = (cv.y & 0x7F7F7F7Fu) + ge_add;
                const uint32_t gx = (cv.x & ge_top) | (cv.x & tx) | (ge_top & tx), gy = (cv.y & ge_top) | (cv.y & ty) | (ge_top & ty);
                mine = mine || (((gx | gy) & 0x80808080u) != 0u);
            }
            live = __ballot(mine) != 0ull;
        }
        double s[8];
        double run = 0.0, incl = 0.0;
        if (live) {
            double v[8];
            if (PACKED && __ballot(hi_cur != 0u) == 0ull) {  // no count of this wave reaches 256 (the rule): the low bytes are the counts
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const uint32_t c = (cw[j >> 2] >> ((j & 3) * 8)) & 0xFFu;
                    v[j] = TZ_LDS ? tz_lds[c] : tzg[c];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    uint32_t c;  // lanes past the end hold 0: a valid index
                    if (PACKED) c = ((cw[j >> 2] >> ((j & 3) * 8)) & 0xFFu) | (((hi_cur >> (2 * j)) & 3u) << 8);
                    else c = (cw[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu;
                    v[j] = TZ_LDS ? tz_lds[c] : tzg[c];
                }
            }
            if (base + kSweep > n) {  // last sweep (wave-uniform): references past the end contribute nothing
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = (r0 + j < n) ? v[j] : 0.0;
            }
            s[0] = v[0];
#pragma unroll
            for (int j = 1; j < 8; j++) s[j] = s[j - 1] + v[j];
            run = s[7];
            incl = wave_incl_scan_f64_dpp(run);
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) s[j] = 0.0;
        }
        if (lane == 63) wsum[buf][wave] = incl;
        __syncthreads();
        double off = carry + (incl - run);
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const double ws = wsum[buf][w];
            if ((uint32_t)w < wave) off += ws;
            tot += ws;
        }
        carry += tot;
        // boundary j of this chunk goes to P[rank + number of boundaries below it]: eight predicated stores, no loop (a
        // loop of stores makes the compiler drain the memory counter in front of it -- and with it the prefetch)
        // The data of the next sweep are taken in HERE, before this sweep's stores: behind them the wait would also cover
        // the stores (their number is not known at compile time, so the compiler could not leave them in flight).
        if (PACKED) asm volatile("" : "+v"(cv_l.x), "+v"(cv_l.y), "+v"(hi_l), "+v"(bits_l), "+v"(rank_l)::"memory");
        else asm volatile("" : "+v"(cv_l.x), "+v"(cv_l.y), "+v"(cv_l.z), "+v"(cv_l.w), "+v"(bits_l), "+v"(rank_l)::"memory");
        const uint32_t bits = bits_cur;  // 0 for chunks past the end
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (bits & (1u << j)) P[rank_cur + (uint32_t)__popc(bits & ((1u << j) - 1u))] = off + s[j];
        accept(r0 + kSweep);
    }
    }
    fill(filled, p.n_bnd);
    if (p.fuse_walk) {
        // The walk of this query by wave 0 while the prefix sums are still in this XCD's L2 (a walk on its own is a
        // chain of ~1.5 us misses: the prefix arrays of a sub-batch are 10x the L2); the other waves retire, and the
        // walking wave hides under the streaming workgroups that take their place.  The dynamic LDS (the table copy,
        // dead now) becomes the walk state.
        __syncthreads();  // workgroup-scope release/acquire of the P stores (same CU: no cache maintenance needed)
        if (wave == 0) lineage_walk_wave(p.walk, q, lane, *reinterpret_cast<WalkLds *>(tz_lds), GapPrefix{P, &s_gaps});
    }
}

// histogram of prob.rs:13-19 from the u16 counts of a sub-batch (k-mer-sharded database: counts summed over the ranks),
// and the largest count of every tile of 8192 references (what hit_count wrote covered this rank's k-mers only)
__global__ __launch_bounds__(256) void rehist_kernel(const uint16_t *__restrict__ counts, uint64_t npad, uint64_t n_refs,
                                                     const uint32_t *__restrict__ t, uint32_t *__restrict__ hist, uint32_t hstride,
                                                     uint16_t *__restrict__ tile_max, uint32_t ntiles) {
    extern __shared__ uint32_t h_lds[];  // [hstride] histogram | [ntiles] tile maxima
    uint32_t *tm_lds = h_lds + hstride;
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    const uint32_t tq = t[q];
    for (uint32_t m = tid; m <= tq; m += 256) h_lds[m] = 0;
    for (uint32_t i = tid; i < ntiles; i += 256) tm_lds[i] = 0;
    __syncthreads();
    const uint16_t *c = counts + (size_t)q * npad;
    for (uint64_t r0 = 0; r0 < n_refs; r0 += 8192) {  // the 256 threads stay inside one tile
        const uint64_t r1 = r0 + 8192 < n_refs ? r0 + 8192 : n_refs;
        uint32_t mx = 0;
        for (uint64_t r = r0 + tid; r < r1; r += 256) {
            uint32_t v = c[r];
            v = v <= tq ? v : tq;  // a count cannot exceed t (clamped against corrupt input)
            atomicAdd(&h_lds[v], 1u);
            mx = v > mx ? v : mx;
        }
        if (mx) atomicMax(&tm_lds[r0 >> 13], mx);
    }
    __syncthreads();
    uint32_t *h = hist + (size_t)q * hstride;
    for (uint32_t m = tid; m <= tq; m += 256) h[m] = h_lds[m];
    if (tile_max)
        for (uint32_t i = tid; i < ntiles; i += 256) tile_max[(size_t)q * ntiles + i] = (uint16_t)tm_lds[i];
}

// debug taps: the packed counts of one query as u16 (rtx_debug_hit_counts, rtx_debug_probs)
__global__ void counts_unpack_kernel(const uint8_t *lo, const uint16_t *hi, uint64_t n, uint16_t *out) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) out[r] = (uint16_t)(lo[r] | (((hi[r >> 3] >> (2u * (r & 7u))) & 3u) << 8));
}

// ---------------------------------------------------------------------------
// debug tap: p_r = table[count_r]/Z for one query (rtx_debug_probs)
// ---------------------------------------------------------------------------
__global__ void probs_expand_kernel(const uint16_t *counts, const double *tz, uint64_t n, double *out) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) out[r] = tz[counts[r]];
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
void launch_bitmap_build(hipStream_t s, const uint64_t *off, const uint32_t *post, const uint32_t *row_of,
                         uint32_t *bitmap, uint32_t stride_words, uint32_t n_rows1, uint32_t ref_lo, uint32_t ref_hi, uint32_t shift) {
    hipLaunchKernelGGL(bitmap_build_kernel, dim3(RTX_NUM_KMERS), dim3(256), 0, s, off, post, row_of, bitmap,
                       stride_words, n_rows1, ref_lo, ref_hi, shift);
}
void launch_ref_kmer_mark(hipStream_t s, const uint8_t *bases, const uint64_t *off, uint64_t n_refs, uint32_t *present) {
    hipLaunchKernelGGL(ref_kmer_mark_kernel, dim3(4096), dim3(64), 0, s, bases, off, n_refs, present);
}
void launch_ref_bitmap_set(hipStream_t s, const uint8_t *bases, const uint64_t *off, uint64_t n_refs,
                           const uint32_t *row_of, uint32_t *bitmap, uint32_t stride_words, uint32_t n_rows1, uint32_t shift) {
    hipLaunchKernelGGL(ref_bitmap_set_kernel, dim3((unsigned)n_refs), dim3(64), 0, s, bases, off, n_refs, row_of, bitmap,
                       stride_words, n_rows1, shift);
}
void launch_row_popcount(hipStream_t s, const uint32_t *row_of, const uint32_t *bitmap, uint32_t stride_words, uint32_t n_rows1,
                         uint32_t *list_len) {
    hipLaunchKernelGGL(row_popcount_kernel, dim3(RTX_NUM_KMERS), dim3(256), 0, s, row_of, bitmap, stride_words, n_rows1, list_len);
}
__global__ __launch_bounds__(256) void row_len_pack_kernel(const uint32_t *__restrict__ row_of, const uint32_t *__restrict__ list_len, uint2 *__restrict__ out) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    out[k] = make_uint2(row_of[k], list_len[k]);
}
void launch_row_len_pack(hipStream_t s, const uint32_t *row_of, const uint32_t *list_len, uint2 *out) {
    hipLaunchKernelGGL(row_len_pack_kernel, dim3(RTX_NUM_KMERS / 256), dim3(256), 0, s, row_of, list_len, out);
}
void launch_kmer_extract(hipStream_t s, const KmerParams &p, uint32_t nq) {
    if (p.mode == 2u) hipLaunchKernelGGL(kmer_extract_kernel<true>, dim3(nq), dim3(64), 0, s, p);
    else hipLaunchKernelGGL(kmer_extract_kernel<false>, dim3(nq), dim3(64), 0, s, p);
}
// a histogram row of more entries than this does not share LDS with the row list: hit_count adds to it in global memory
constexpr uint32_t kHitLdsHistMax = 12288;
void launch_hit_count(hipStream_t s, const HitParams &p_in, uint32_t nq, uint32_t ntiles, int planes) {
    HitParams p = p_in;
    const bool ghist = p.hstride > kHitLdsHistMax;
    const uint32_t first = std::max<uint32_t>(ghist ? 0u : (p.hstride + 3u) & ~3u, kHitListCap + (RTX_HIT_NB > 2 ? 192u : 64u));  // histogram / row-id list
    p.lds_cnt8_off = first;
    // 8.4 KB per wave: measured flat up to there, +7 % at 10.4 KB, +14 % at 12.5 KB (16 waves per CU must fit in 160 KB)
    const size_t lds = (size_t)first * sizeof(uint32_t) + 4096;  // ... | byte counters
    if (ghist) hipLaunchKernelGGL((hit_count_kernel<16, false, true>), dim3(nq, ntiles), dim3(64), lds, s, p);
    else if (planes <= 10 && p.counts_lo) hipLaunchKernelGGL((hit_count_kernel<10, true>), dim3(nq, ntiles), dim3(64), lds, s, p);
    else if (planes <= 10) hipLaunchKernelGGL((hit_count_kernel<10, false>), dim3(nq, ntiles), dim3(64), lds, s, p);
    else if (planes <= 12) hipLaunchKernelGGL((hit_count_kernel<12, false>), dim3(nq, ntiles), dim3(64), lds, s, p);
    else hipLaunchKernelGGL((hit_count_kernel<16, false>), dim3(nq, ntiles), dim3(64), lds, s, p);
}
size_t prob_table_lds_bytes(uint32_t tmax) {
    const size_t n1max = tmax / 2 + 1;
    const size_t ngmax = (tmax + 64) / 64 + 1;
    return sizeof(double) * (2 * n1max + n1max + (tmax + n1max + 1) + 16 + ngmax) + sizeof(uint32_t) * (2 * ngmax + 2) +
           sizeof(uint16_t) * ((size_t)tmax + 2);
}
void launch_prob_table(hipStream_t s, const ProbParams &p, uint32_t nq) {
    if (p.gscratch) hipLaunchKernelGGL(prob_table_kernel<true>, dim3(nq), dim3(kProbThreads), 0, s, p);
    else hipLaunchKernelGGL(prob_table_kernel<false>, dim3(nq), dim3(kProbThreads), prob_table_lds_bytes(p.tmax), s, p);
}
template <int NW>
static void launch_taxon_prefix_nw(hipStream_t s, const PrefixParams &p, uint32_t nq, size_t lds) {
    if (p.tz_in_lds && p.packed) hipLaunchKernelGGL((taxon_prefix_kernel<NW, true, true>), dim3(nq), dim3(NW * 64), lds, s, p);
    else if (p.tz_in_lds) hipLaunchKernelGGL((taxon_prefix_kernel<NW, true, false>), dim3(nq), dim3(NW * 64), lds, s, p);
    else if (p.packed) hipLaunchKernelGGL((taxon_prefix_kernel<NW, false, true>), dim3(nq), dim3(NW * 64), lds, s, p);
    else hipLaunchKernelGGL((taxon_prefix_kernel<NW, false, false>), dim3(nq), dim3(NW * 64), lds, s, p);
}
void launch_taxon_prefix(hipStream_t s, const PrefixParams &p, uint32_t nq) {
    size_t lds = p.tz_in_lds ? (size_t)p.hstride * sizeof(double) : 0;
    if (p.fuse_walk) lds = std::max(lds, sizeof(WalkLds));
#ifndef RTX_PREFIX_NW
#define RTX_PREFIX_NW 2
#endif
#ifndef RTX_PREFIX_NW_PRUNED
#define RTX_PREFIX_NW_PRUNED 2
#endif
    // waves per query: NW * 512 references per sweep.  A query of a pruned run sweeps one or two tiles: two waves (more queries
    // in flight) beat four (N = 500k, per 1 M queries: 18.2 ms with four, 14.5 with two, 15.2 with one).  A run that sweeps every tile: two as
    // well since round 5 (131 072 short barcodes against 14 tiles: 11.7 ms with eight, 8.7 with four, 7.6 with two, 8.8 with one; 65 536 COI reads
    // against 62 tiles, most of them skipped by their largest count: 1.2 with four, 0.8 with two) -- the kernel is short of issue slots, not of waves
    if (p.prune_thr) launch_taxon_prefix_nw<RTX_PREFIX_NW_PRUNED>(s, p, nq, lds);
    else launch_taxon_prefix_nw<RTX_PREFIX_NW>(s, p, nq, lds);
}
void launch_lineage_walk(hipStream_t s, const WalkParams &p, uint32_t nq) {
    hipLaunchKernelGGL(lineage_walk_kernel, dim3(nq), dim3(64), 0, s, p);
}
void launch_rehist(hipStream_t s, const uint16_t *counts, uint64_t npad, uint64_t n_refs, const uint32_t *t, uint32_t *hist,
                   uint32_t hstride, uint16_t *tile_max, uint32_t ntiles, uint32_t nq) {
    hipLaunchKernelGGL(rehist_kernel, dim3(nq), dim3(256), ((size_t)hstride + ntiles) * 4, s, counts, npad, n_refs, t, hist, hstride,
                       tile_max, ntiles);
}
void launch_counts_unpack(hipStream_t s, const uint8_t *lo, const uint16_t *hi, uint64_t n, uint16_t *out) {
    hipLaunchKernelGGL(counts_unpack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, lo, hi, n, out);
}
void launch_probs_expand(hipStream_t s, const uint16_t *counts, const double *tz, uint64_t n, double *out) {
    hipLaunchKernelGGL(probs_expand_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, counts, tz, n, out);
}

}  // namespace rtx
