// Processing order of the queries of a batch.
//
// hit_count reads one bitmap row per (query, k-mer); queries that share k-mers share rows.  Related
// queries are therefore brought next to each other before the batch is cut into sub-batches: rows are then
// reused out of L2, and hit_count_pair_kernel (rtx_kernels.hip) loads the rows two neighbouring queries
// have in common only once.  The order is a pure scheduling decision: every query is still classified
// exactly as raxtax.rs:39-88 does and results are returned in input order.
//
//   sketch_kernel : per query three min-hashes over its 12-mers (bottom-1 sketches, 21 bits each)
//                   -> 63-bit key; related sequences agree on a min-hash with probability = their
//                   12-mer Jaccard similarity, so sorting by the key puts most of them side by side
//   locator       : (only when the index was built from the reference sequences) where in the lineage-ordered database
//                   the relatives of a query sit.  At index build every 12-mer of every reference is entered into a
//                   direct-mapped table (2^24 entries: lowest reference position holding it, number of occurrences);
//                   12-mers that occur more than kLocMaxCount times say nothing about a clade and are dropped.  A query
//                   looks its 12-mers up and votes: coarse bins of `binw` references (the best pair of neighbouring
//                   bins wins), then bins of binw / 16 inside the winner.  The position leads the sort key, the
//                   min-hashes break ties: neighbouring sub-batches, XCD slices and pairs then hold queries of the same
//                   genus / species, the min-hash buckets alone are ordered at random.  Measured at BASELINE configs[2]
//                   (tools/exp_order_potential2.py, hit_count per 262 144 queries): input order 324 ms, min-hash order
//                   255 ms, queries sorted by their true source reference 210 ms.
//   sort          : rocPRIM device radix sort of (key, query index) pairs -> perm[position] = query
//   invert        : inv[query] = position
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include "rtx_kernels.hpp"
#include "rtx_wave.hpp"

namespace rtx {

static constexpr int kSketchK = 12;
static constexpr uint32_t kSketchBits = 21;

__global__ __launch_bounds__(64) void sketch_kernel(const uint8_t *__restrict__ bases, const uint64_t *__restrict__ off,
                                                    uint64_t *__restrict__ keys, uint32_t *__restrict__ idx) {
    const uint32_t q = blockIdx.x, lane = threadIdx.x;
    const uint64_t b0 = off[q], len = off[q + 1] - b0;
    const uint8_t *seq = bases + b0;
    const uint32_t none = (1u << kSketchBits) - 1u;
    uint32_t m1 = none, m2 = none, m3 = none;
    if (len >= (uint64_t)kSketchK) {
        // lane l rolls over the windows [l*wpl, (l+1)*wpl), sixteen at a time: their 27 bases come in as four unaligned 8-byte words
        // (one round trip instead of 27 dependent byte loads; the batch buffer is padded by 64 bytes behind its end) and the
        // code rolls over them in registers, as in kmer_extract_kernel
        const uint64_t nwin = len - kSketchK + 1;
        const uint64_t wpl = (nwin + 63) / 64;
        const uint64_t w0 = (uint64_t)lane * wpl;
        const uint64_t w1 = w0 + wpl < nwin ? w0 + wpl : nwin;
        for (uint64_t i = w0; i < w1; i += 16) {
            unsigned long long v[4];
            __builtin_memcpy(v, seq + i, 32);
            const uint32_t nw = w1 - i < 16 ? (uint32_t)(w1 - i) : 16u;
            uint32_t code = 0, run = 0;
#pragma unroll
            for (int b = 0; b < 16 + kSketchK - 1; b++) {
                const uint32_t c = (uint32_t)(v[b >> 3] >> ((b & 7) * 8)) & 0xFFu;
                const bool ok = c == 1u || c == 2u || c == 4u || c == 8u;
                code = ((code << 2) | (((uint32_t)__ffs((int)c) - 1u) & 3u)) & 0xFFFFFFu;
                run = ok ? run + 1u : 0u;
                if (b >= kSketchK - 1 && (uint32_t)(b - (kSketchK - 1)) < nw && run >= (uint32_t)kSketchK) {
                    const uint64_t x = code;
                    const uint32_t h1 = (uint32_t)((x * 0x9E3779B97F4A7C15ull) >> (64 - kSketchBits));
                    const uint32_t h2 = (uint32_t)((x * 0xC2B2AE3D27D4EB4Full) >> (64 - kSketchBits));
                    const uint32_t h3 = (uint32_t)((x * 0x165667B19E3779F9ull) >> (64 - kSketchBits));
                    m1 = h1 < m1 ? h1 : m1;
                    m2 = h2 < m2 ? h2 : m2;
                    m3 = h3 < m3 ? h3 : m3;
                }
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o1 = __shfl_xor(m1, d, 64), o2 = __shfl_xor(m2, d, 64), o3 = __shfl_xor(m3, d, 64);
        m1 = o1 < m1 ? o1 : m1;
        m2 = o2 < m2 ? o2 : m2;
        m3 = o3 < m3 ? o3 : m3;
    }
    if (lane == 0) {
        keys[q] = ((uint64_t)m1 << (2 * kSketchBits)) | ((uint64_t)m2 << kSketchBits) | m3;
        idx[q] = q;
    }
}

// ---------------------------------------------------------------------------
// Locator table (index build).  pos_min[c] = lowest reference id (lineage order) that contains 12-mer c, cnt[c] = its
// occurrences over all references; loc_finish folds both into one word per 12-mer: the position, or kLocNone where the
// 12-mer is absent or too common to tell clades apart.
// ---------------------------------------------------------------------------
static constexpr uint32_t kLocNone = 0xFFFFFFFFu;
static constexpr uint32_t kLocMaxCount = 512;

__device__ __forceinline__ bool base_code(uint32_t c, uint32_t &two) {
    two = ((uint32_t)__ffs((int)c) - 1u) & 3u;
    return c == 1u || c == 2u || c == 4u || c == 8u;
}

__global__ __launch_bounds__(64) void loc_mark_kernel(const uint8_t *__restrict__ bases, const uint64_t *__restrict__ off,
                                                      uint64_t n_refs, uint32_t *__restrict__ pos_min, uint32_t *__restrict__ cnt) {
    const uint32_t lane = threadIdx.x;
    for (uint64_t r = blockIdx.x; r < n_refs; r += gridDim.x) {
        const uint64_t b0 = off[r], len = off[r + 1] - b0;
        if (len < (uint64_t)kSketchK) continue;
        const uint8_t *seq = bases + b0;
        const uint64_t nwin = len - kSketchK + 1;
        for (uint64_t w = lane; w < nwin; w += 64) {
            uint32_t code = 0;
            bool ok = true;
#pragma unroll
            for (int j = 0; j < kSketchK; j++) {
                uint32_t two;
                ok = base_code(seq[w + j], two) && ok;
                code = (code << 2) | two;
            }
            if (ok) {
                atomicMin(&pos_min[code], (uint32_t)r);
                atomicAdd(&cnt[code], 1u);
            }
        }
    }
}

__global__ __launch_bounds__(256) void loc_finish_kernel(uint32_t *__restrict__ pos_min, const uint32_t *__restrict__ cnt) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const uint32_t c = cnt[i];
    if (c == 0u || c > kLocMaxCount) pos_min[i] = kLocNone;
}

// ---------------------------------------------------------------------------
// Locator of a query: one wave per query.
//   1. lane l rolls over the windows [l * wpl, (l + 1) * wpl) of the first kLocWin windows (their wpl + 11 bases in one round trip,
//      as sketch_kernel) and leaves their 12-mer codes in LDS;
//   2. lane l looks up the windows 2 l, 2 (l + 64), ... (kLocStride; four table loads in flight), keeps the positions in
//      LDS and votes for coarse bins of 2^bin_shift references.  Most lanes vote for the SAME bin -- that is the point --
//      and LDS atomics on one address are served lane by lane (about 100 cycles for 64 of them): the lanes that agree
//      with the first voter are counted with a ballot and added by one lane, two such rounds, the rest votes alone;
//   3. the best pair of neighbouring coarse bins wins, then bins of a sixteenth inside it.
// Votes are u16 pairs in u32 words (a bin gets at most kLocWin of them).  Queries longer than kLocWin + 11 bases are
// located by their first kLocWin windows.  keys[q] = loc << 39 | (min-hash key >> 24): 24 bits of position, m1 and the
// high 18 bits of m2.
// ---------------------------------------------------------------------------
static constexpr uint32_t kLocMaxBins = 8192;    // coarse bins (LDS: 16 KiB)
static constexpr uint32_t kLocFineDiv = 16;      // fine bins per coarse bin
static constexpr uint32_t kLocWin = 1024;        // windows of a query that vote
#ifndef RTX_LOC_STRIDE
#define RTX_LOC_STRIDE 2
#endif
// ... every kLocStride-th of them.  A look-up is a random word of a 64 MB table; at configs[2] every window / every second / every third:
// order stage 5.55 / 3.57 / 3.22 ms per 1 M queries, the stages that live on the order (bounds + counting) 42.5 / 42.85 / 43.5 ms.
static constexpr uint32_t kLocStride = RTX_LOC_STRIDE;

// one vote per lane with `valid` for bin b: lanes that agree with the first pending voter are added by one atomic
__device__ __forceinline__ void vote_bins(uint32_t *h, uint32_t b, bool valid, uint32_t lane) {
    unsigned long long pending = __ballot(valid);
#pragma unroll
    for (int r = 0; r < 2; r++) {
        if (!pending) return;  // wave-uniform
        const int first = __builtin_ctzll(pending);
        const uint32_t lead = (uint32_t)__builtin_amdgcn_readlane((int)b, first);
        const unsigned long long same = __ballot(valid && b == lead) & pending;
        if ((int)lane == first) atomicAdd(&h[lead >> 1], (uint32_t)__popcll(same) << ((lead & 1u) * 16u));
        pending &= ~same;
    }
    if ((pending >> lane) & 1ull) atomicAdd(&h[b >> 1], 1u << ((b & 1u) * 16u));
}

__global__ __launch_bounds__(64) void locator_kernel(const uint8_t *__restrict__ bases, const uint64_t *__restrict__ off,
                                                     const uint32_t *__restrict__ table, uint32_t bin_shift, uint32_t n_bins,
                                                     uint64_t *__restrict__ keys) {
    extern __shared__ __attribute__((aligned(16))) uint32_t h[];  // n_bins / 2 + 2 words (rounded up to four): the coarse bins as u16 pairs
    __shared__ uint32_t lcode[kLocWin / kLocStride];  // 12-mer codes of the windows that vote, then the positions found for them
    __shared__ uint32_t hf[kLocFineDiv + 2];   // 2 * kLocFineDiv + 1 fine bins as u16 pairs
    const uint32_t q = blockIdx.x, lane = threadIdx.x;
    const uint64_t b0 = off[q], len = off[q + 1] - b0;
    if (len < (uint64_t)kSketchK) return;  // wave-uniform: the min-hash key stays
    const uint64_t old_key = keys[q];
    const uint8_t *seq = bases + b0;
    const uint32_t nwin = (uint32_t)(len - kSketchK + 1 < (uint64_t)kLocWin ? len - kSketchK + 1 : kLocWin);
    const uint32_t words = n_bins / 2 + 2;
    for (uint32_t i = lane; i < (words + 3u) / 4u; i += 64) reinterpret_cast<uint4 *>(h)[i] = make_uint4(0u, 0u, 0u, 0u);
    if (lane < kLocFineDiv + 2) hf[lane] = 0;
    {   // 1. codes of this lane's run of windows (at most 16: nwin <= kLocWin): its 27 bases as four unaligned 8-byte words, as sketch_kernel
        const uint32_t wpl = (nwin + 63u) / 64u;
        const uint32_t w0 = lane * wpl, w1 = w0 + wpl < nwin ? w0 + wpl : nwin;
        static_assert(kLocWin <= 1024u && kLocWin % kLocStride == 0u, "a lane's run of windows fits one piece of 16");
        if (w0 < w1) {
            unsigned long long v[4];
            __builtin_memcpy(v, seq + w0, 32);
            const uint32_t nw = w1 - w0;
            uint32_t code = 0, run = 0;
#pragma unroll
            for (int b = 0; b < 16 + kSketchK - 1; b++) {
                uint32_t two;
                const bool ok = base_code((uint32_t)(v[b >> 3] >> ((b & 7) * 8)) & 0xFFu, two);
                code = ((code << 2) | two) & 0xFFFFFFu;
                run = ok ? run + 1u : 0u;
                if (b >= kSketchK - 1 && (uint32_t)(b - (kSketchK - 1)) < nw) {
                    const uint32_t w = w0 + (uint32_t)(b - (kSketchK - 1));
                    if (w % kLocStride == 0u) lcode[w / kLocStride] = run >= (uint32_t)kSketchK ? code : kLocNone;
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // 2. look-ups and the coarse vote, four windows per lane and turn (all sixteen of a lane at once: 0.3 ms per 1 M queries slower)
    uint32_t any = 0;
    const uint32_t nlook = (nwin + kLocStride - 1u) / kLocStride;  // every kLocStride-th window votes
    for (uint32_t j0 = 0; j0 < nlook; j0 += 256) {
        uint32_t pos[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t j = j0 + (uint32_t)u * 64u + lane;
            const uint32_t code = j < nlook ? lcode[j] : kLocNone;
            pos[u] = table[code == kLocNone ? 0u : code];  // unconditional: the four loads leave together
            if (code == kLocNone) pos[u] = kLocNone;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t j = j0 + (uint32_t)u * 64u + lane;
            const bool valid = pos[u] != kLocNone;
            vote_bins(h, pos[u] >> bin_shift, valid, lane);
            if (j < nlook) lcode[j] = pos[u];
            any |= valid ? 1u : 0u;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (!__ballot(any != 0u)) return;  // no vote: the min-hash key stays
    // 3. best pair of neighbouring coarse bins (B, B + 1): value << 16 | (0xFFFF - B), the lowest B among equals
    auto bin = [&](uint32_t b) { return (h[b >> 1] >> ((b & 1u) * 16u)) & 0xFFFFu; };
    uint32_t best = 0;
    for (uint32_t k = lane; 2u * k < n_bins; k += 64) {  // the bins 2k and 2k + 1 from two words (h[words - 1] = 0 stands behind the last bin)
        const uint32_t w0 = h[k], w1 = h[k + 1u];
        const uint32_t v0 = (w0 & 0xFFFFu) + (w0 >> 16), v1 = (w0 >> 16) + (w1 & 0xFFFFu);
        const uint32_t c0 = (v0 << 16) | (0xFFFFu - 2u * k), c1 = 2u * k + 1u < n_bins ? (v1 << 16) | (0xFFFFu - (2u * k + 1u)) : 0u;
        best = c0 > best ? c0 : best;
        best = c1 > best ? c1 : best;
    }
    best = wave_max_u32(best);
    const uint32_t B = 0xFFFFu - (best & 0xFFFFu);
    const uint32_t lo = B << bin_shift;
    uint32_t fine = 0;
    if (bin_shift >= 4u) {
        const uint32_t fshift = bin_shift - 4u;  // fine bins of 2^bin_shift / 16 references over [lo, lo + 2 * 2^bin_shift)
        for (uint32_t w0 = 0; w0 < nlook; w0 += 64) {  // wave-uniform trip count (vote_bins works with ballots)
            const uint32_t w = w0 + lane;
            const uint32_t p = w < nlook ? lcode[w] : kLocNone;
            const bool in = p != kLocNone && p >= lo && ((p - lo) >> bin_shift) < 2u;
            vote_bins(hf, in ? (p - lo) >> fshift : 0u, in, lane);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t fb = 0;
        if (lane < 2u * kLocFineDiv) {
            const uint32_t v = ((hf[lane >> 1] >> ((lane & 1u) * 16u)) & 0xFFFFu) + ((hf[(lane + 1u) >> 1] >> (((lane + 1u) & 1u) * 16u)) & 0xFFFFu);
            fb = (v << 16) | (0xFFFFu - lane);
        }
        fb = wave_max_u32(fb);
        fine = 0xFFFFu - (fb & 0xFFFFu);
    }
    const uint64_t loc = ((uint64_t)B * kLocFineDiv + fine) & 0xFFFFFFull;
    if (lane == 0) keys[q] = (loc << 39) | (old_key >> 24);
}

// Length classes (rtx_index.hpp: BatchClass): the class of a query -- the number of the limits its length exceeds -- leads the sort key, so
// that the processing order holds the classes one after the other; inside a class the order is the one the keys gave (or, with the
// processing order switched off, the input order: index keys).
__global__ __launch_bounds__(256) void class_keys_kernel(uint64_t *__restrict__ keys, const uint64_t *__restrict__ off, uint32_t n, uint64_t lim0,
                                                         uint64_t lim1, uint64_t lim2, uint64_t lim3, uint32_t shift, uint32_t from_index, uint32_t *__restrict__ idx) {
    const uint32_t q = blockIdx.x * 256 + threadIdx.x;
    if (q >= n) return;
    const uint64_t len = off[q + 1] - off[q];
    const uint64_t rank = (len > lim0 ? 1u : 0u) + (len > lim1 ? 1u : 0u) + (len > lim2 ? 1u : 0u) + (len > lim3 ? 1u : 0u);  // five classes: three bits
    const uint64_t k = from_index ? (uint64_t)q : keys[q] >> shift;
    keys[q] = (rank << 61) | k;
    if (from_index) idx[q] = q;
}

__global__ __launch_bounds__(256) void invert_perm_kernel(const uint32_t *__restrict__ perm, uint32_t n, uint32_t *__restrict__ inv) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) inv[perm[i]] = i;
}

__global__ __launch_bounds__(256) void identity_perm_kernel(uint32_t n, uint32_t *__restrict__ perm, uint32_t *__restrict__ inv) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { perm[i] = i; inv[i] = i; }
}

void launch_sketch(hipStream_t s, const uint8_t *bases, const uint64_t *off, uint32_t n_q, uint64_t *keys, uint32_t *idx) {
    hipLaunchKernelGGL(sketch_kernel, dim3(n_q), dim3(64), 0, s, bases, off, keys, idx);
}
// Index build: enters the 12-mers of all references into the table
void launch_loc_mark(hipStream_t s, const uint8_t *bases, const uint64_t *off, uint64_t n_refs, uint32_t *pos_min, uint32_t *cnt) {
    const uint32_t grid = (uint32_t)(n_refs < 65536u ? (n_refs ? n_refs : 1u) : 65536u);
    hipLaunchKernelGGL(loc_mark_kernel, dim3(grid), dim3(64), 0, s, bases, off, n_refs, pos_min, cnt);
}
void launch_loc_finish(hipStream_t s, uint32_t *pos_min, const uint32_t *cnt) {
    hipLaunchKernelGGL(loc_finish_kernel, dim3(kLocTableEntries / 256), dim3(256), 0, s, pos_min, cnt);
}
// bins of 2^bin_shift references: at most kLocMaxBins of them, at least 256 references each
uint32_t loc_bin_shift(uint64_t n_refs) {
    uint32_t sh = 8;
    while (((n_refs + (1ull << sh) - 1) >> sh) + 1 > kLocMaxBins) sh++;
    return sh;
}
void launch_locator(hipStream_t s, const uint8_t *bases, const uint64_t *off, uint32_t n_q, const uint32_t *table, uint64_t n_refs,
                    uint64_t *keys) {
    const uint32_t sh = loc_bin_shift(n_refs);
    const uint32_t n_bins = (uint32_t)((n_refs + (1ull << sh) - 1) >> sh);
    hipLaunchKernelGGL(locator_kernel, dim3(n_q), dim3(64), ((n_bins / 2 + 2 + 3) & ~3u) * 4, s, bases, off, table, sh, n_bins, keys);
}
void launch_invert_perm(hipStream_t s, const uint32_t *perm, uint32_t n, uint32_t *inv) {
    hipLaunchKernelGGL(invert_perm_kernel, dim3((n + 255) / 256), dim3(256), 0, s, perm, n, inv);
}
void launch_identity_perm(hipStream_t s, uint32_t n, uint32_t *perm, uint32_t *inv) {
    hipLaunchKernelGGL(identity_perm_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, perm, inv);
}

void launch_class_keys(hipStream_t s, uint64_t *keys, const uint64_t *off, uint32_t n, const uint64_t lim[4], bool from_index, uint32_t *idx) {
    // (the keys hold 63 bits: two of the third min-hash's make room for the class)
    hipLaunchKernelGGL(class_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, s, keys, off, n, lim[0], lim[1], lim[2], lim[3], 2u, from_index ? 1u : 0u, idx);
}

// tmp == nullptr: only reports the temporary storage needed.  with_class: the keys carry the length class in their three top bits.
int cluster_sort(hipStream_t s, void *tmp, size_t *tmp_bytes, const uint64_t *keys_in, uint64_t *keys_out, const uint32_t *idx_in,
                 uint32_t *perm_out, size_t n, bool with_class) {
    const hipError_t e = rocprim::radix_sort_pairs(tmp, *tmp_bytes, keys_in, keys_out, idx_in, perm_out, n, 0, with_class ? 64u : 3 * kSketchBits, s);
    return e == hipSuccess ? 0 : 1;
}

}  // namespace rtx
