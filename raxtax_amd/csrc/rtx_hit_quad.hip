// hit_count for four neighbouring queries at a time (src/raxtax.rs:41,58-68, prob.rs:13-19).
//
// hit_count_kernel (rtx_kernels.hip) runs at the rate at which an XCD's L2 can hand 1-KiB row segments to the CUs
// (DESIGN.md section 3): every (query, tile) wave fetches every row of its query itself.  But the queries of a batch
// are processed in an order that puts related ones next to each other (rtx_cluster.hip), and four neighbours ask
// for largely the same rows: at BASELINE.json configs[2] the union of their dense segments is a third of the sum
// (tools/exp_quad_sim.py).  Here a workgroup of four waves takes four consecutive queries of the processing order
// and ONE tile:
//
//   prologue   every wave marks the dense rows of its query in a 65 536-bit set in LDS; a rank scan over the set
//              gives the union in ascending row order (U), and every wave the positions of its own rows in it (Lw)
//   row loop   rounds of kQuadRR union rows.  Each wave brings 1/4 of a round's rows from L2 straight into an LDS
//              ring (buffer_load ... lds: no VGPR staging), kQuadAhead rounds ahead; one s_barrier per round says
//              "round r has landed, round r-2 may be overwritten".  Each wave folds the rows of ITS query out of the
//              ring (ds_read_b128) into its bit planes, eight at a time with the same Harley-Seal tree as
//              hit_count_kernel; rows left over at the end of a round (fewer than eight) wait for the next round
//              (the ring keeps a round longer for that), so that next to no fold is padded (1.04 slots per row)
//   epilogue   per wave, unchanged (rtx_hit_common.hpp)
//
// Bytes leave L2 once per workgroup instead of once per query; the VALU work per query is what it was.  Results
// are those of hit_count_kernel bit for bit (integer counts).
#include <hip/hip_runtime.h>

#include "rtx_hit_common.hpp"

namespace rtx {

#ifndef RTX_QUAD_AHEAD
#define RTX_QUAD_AHEAD 3
#endif
#ifndef RTX_QUAD_WAVES_PER_SIMD
#define RTX_QUAD_WAVES_PER_SIMD 3
#endif
#ifndef RTX_QUAD_RR
#define RTX_QUAD_RR 8
#endif
constexpr uint32_t kQuadRR = RTX_QUAD_RR;                   // union rows per round: kQuadRR / 4 loads per wave and round
constexpr uint32_t kQuadLoads = kQuadRR / 4u;
constexpr uint32_t kQuadAhead = RTX_QUAD_AHEAD;             // rounds in flight behind the one being folded
constexpr uint32_t kQuadDepth = kQuadAhead + 2;             // ring: in flight + current + one round of left-overs
constexpr uint32_t kQuadSlots = kQuadRR * kQuadDepth;       // 40 KiB
constexpr uint32_t kQuadUCap = 1024;                        // union rows per pass (a second pass is rare: DESIGN.md)
constexpr uint32_t kQuadRingBytes = kQuadSlots * 1024u;
constexpr uint32_t kQuadUOff = kQuadRingBytes;                            // u16 U[kQuadUCap + 64]
constexpr uint32_t kQuadLOff = kQuadUOff + (kQuadUCap + 64u) * 2u;        // u16 Lw[4][kQuadUCap + 64]
constexpr uint32_t kQuadZeroOff = kQuadLOff + 4u * (kQuadUCap + 64u) * 2u;  // 1 KiB of zeros: the rows a padded fold lacks
constexpr uint32_t kQuadLdsBytes = kQuadZeroOff + 1024u;
constexpr uint32_t kQuadRankOff = 8192u;                    // prologue only (inside the ring): u16 wordrank[1024 + 1]
constexpr uint32_t kQuadEpiBytes = 8448u;                   // per wave: histogram / byte counters of the epilogue

typedef int v4i_t __attribute__((ext_vector_type(4)));

// One 1-KiB row segment from global memory straight into LDS (LDS-DMA): lane l's 16 bytes land at lds_off + 16 l.
// Raw buffer load as in hit_count_kernel (row base in SGPRs, num_records = bytes per row: lanes beyond the row deliver
// zeros).  Inline asm on purpose: the compiler would put s_waitcnt vmcnt(0) in front of the next LDS read of ANY
// address (it does not track where a DMA lands); the waits are counted by hand in the row loop instead.  M0 carries
// the LDS address of the instruction and is put back afterwards.
__device__ __forceinline__ void dma_row(const char *rowbase, uint32_t stride, uint32_t col, uint32_t lds_off) {
    const uint64_t b = reinterpret_cast<uint64_t>(rowbase);
    v4i_t rsrc;
    rsrc.x = (int)(uint32_t)b;
    rsrc.y = (int)(uint32_t)((b >> 32) & 0xFFFFu);
    rsrc.z = (int)stride;
    rsrc.w = 0x00027000;
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %1\n\t"
                 "s_nop 0\n\t"
                 "buffer_load_dwordx4 %2, %3, 0 offen lds\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_off), "v"(col), "s"(rsrc)
                 : "memory");
}

template <int NP, bool kPacked>
__global__ __launch_bounds__(256, RTX_QUAD_WAVES_PER_SIMD) void hit_count_quad_kernel(HitParams p) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds32[];
    __shared__ uint32_t s_wtot[4];
    __shared__ uint32_t s_hi_word;
    char *lds = reinterpret_cast<char *>(lds32);
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)lds;  // byte address inside the workgroup's LDS
    const uint32_t tile = blockIdx.y, tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    // groups of four consecutive slots; XCD x takes a contiguous slice of the groups (as hit_count_kernel does with slots)
    const uint32_t ng8 = gridDim.x >> 3;
    const uint32_t g = blockIdx.x < ng8 * 8u ? (blockIdx.x & 7u) * ng8 + (blockIdx.x >> 3) : blockIdx.x;
    const uint32_t q = g * 4u + wave;
    const bool valid = q < p.nq;  // the last group of a sub-batch may be short: such a wave only loads and keeps the barriers
    const uint32_t qc = valid ? q : p.nq - 1u;
    const uint32_t t = p.t[qc];
    const uint32_t ns = valid ? p.nsparse[(size_t)qc * p.ntiles + tile] : 0u;
    const uint32_t *srows = p.srows + ((size_t)qc * p.ntiles + tile) * (kSegMaxSparseRows + 1);
    const uint32_t col = tile * 1024u + lane * 16u;
    const bool active = col < p.stride_bytes;
    // the tile's region of the tile-major bitmap (rtx_math.hpp: bitmap_word): row r at r KiB
    const char *bitmap = reinterpret_cast<const char *>(p.bitmap) + (size_t)tile * p.n_rows1 * 1024u;
    const uint32_t voff = lane * 16u;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    uint32_t pl[4][NP];
#pragma unroll
    for (int w = 0; w < 4; w++)
#pragma unroll
        for (int b = 0; b < NP; b++) pl[w][b] = 0;
    unsigned long long *bits64 = reinterpret_cast<unsigned long long *>(lds);          // [1024] prologue: union of the dense rows
    uint32_t *bits32 = reinterpret_cast<uint32_t *>(lds);
    uint16_t *wordrank = reinterpret_cast<uint16_t *>(lds + kQuadRankOff);             // [1025] prologue: rows below word w
    uint16_t *U = reinterpret_cast<uint16_t *>(lds + kQuadUOff);                       // union rows of the pass, ascending
    uint16_t *Lw = reinterpret_cast<uint16_t *>(lds + kQuadLOff) + wave * (kQuadUCap + 64u);  // this wave's rows as positions in U

    const uint32_t *rows = p.rows + (size_t)qc * p.rstride;
    const unsigned long long *masks = p.dmask + ((size_t)qc * p.ntiles + tile) * (p.rstride >> 6);
    const uint32_t nchunks = valid ? (p.nrows[qc] + 63u) >> 6 : 0u;  // <= 16: the quad kernel runs with t <= 1023

    if (tid < 64u) reinterpret_cast<uint4 *>(lds + kQuadZeroOff)[tid] = make_uint4(0, 0, 0, 0);  // visible behind the first barrier
    uint32_t rows_loaded = 0;  // union rows of this (group, tile): the work accounting of the launch
#ifdef RTX_QUAD_STAMP
    unsigned long long stamp_acc = 0, stamp_t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long stamp_k0 = stamp_t0;
#define STAMP_MARK(phase) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); if (RTX_QUAD_STAMP == (phase)) stamp_acc += now_ - stamp_t0; stamp_t0 = now_; }
#else
#define STAMP_MARK(phase)
#endif
    uint32_t row_lo = 0;       // rows below it were folded by earlier passes
    for (;;) {
        // ---- prologue a: the dense rows (>= row_lo) of the four queries as a bit set; own rows parked in Lw as row ids
        for (uint32_t i = tid; i < 512u; i += 256u) reinterpret_cast<uint4 *>(lds)[i] = make_uint4(0, 0, 0, 0);
        __syncthreads();
        uint32_t n_own = 0;
        {
            const unsigned long long mv = lane < nchunks ? masks[lane] : 0ull;
            for (uint32_t c0 = 0; c0 < nchunks; c0 += 4) {
                uint32_t rowv[4];
#pragma unroll
                for (int u = 0; u < 4; u++) rowv[u] = c0 + u < nchunks ? rows[(c0 + u) * 64 + lane] : 0u;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (c0 + u >= nchunks) break;
                    const unsigned long long m = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(mv >> 32), (int)(c0 + u)) << 32) |
                                                 (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mv, (int)(c0 + u));
                    const bool keep = ((m >> lane) & 1ull) && rowv[u] >= row_lo;
                    const unsigned long long km = __ballot(keep);
                    if (keep) {
                        Lw[n_own + (uint32_t)__popcll(km & lt_mask)] = (uint16_t)rowv[u];  // real rows are < 65536 (the zero row is no dense row)
                        atomicOr(&bits32[rowv[u] >> 5], 1u << (rowv[u] & 31u));
                    }
                    n_own += (uint32_t)__popcll(km);
                }
            }
        }
        __syncthreads();
        // ---- prologue b: rank of every 64-bit word of the set (thread <-> four consecutive words)
        unsigned long long w4[4];
        uint32_t c4[4], tot = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            w4[k] = bits64[tid * 4u + k];
            c4[k] = (uint32_t)__popcll(w4[k]);
            tot += c4[k];
        }
        const uint32_t incl = wave_incl_scan_u32(tot);
        if (lane == 63u) s_wtot[wave] = incl;
        if (tid == 0) s_hi_word = 1024u;
        __syncthreads();
        uint32_t base = incl - tot, total = 0;
#pragma unroll
        for (uint32_t w = 0; w < 4; w++) {
            const uint32_t wt = s_wtot[w];
            if (w < wave) base += wt;
            total += wt;
        }
        {
            uint32_t run = base;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                wordrank[tid * 4u + k] = (uint16_t)run;
                // more union rows than a pass takes: the pass ends in front of the word in which rank kQuadUCap falls
                if (run <= kQuadUCap && run + c4[k] > kQuadUCap) s_hi_word = tid * 4u + k;
                run += c4[k];
            }
            if (tid == 255u) wordrank[1024] = (uint16_t)run;
        }
        __syncthreads();
        const uint32_t hi_word = s_hi_word;                       // 1024: everything fits
        const uint32_t hi_row = hi_word * 64u;
        const uint32_t un = hi_word < 1024u ? wordrank[hi_word] : total;  // union rows of this pass
        // ---- prologue c: own rows -> positions in U; U itself (thread <-> the same four words)
        uint32_t n_pass = 0;
        for (uint32_t i0 = 0; i0 < n_own; i0 += 64) {
            const uint32_t i = i0 + lane;
            uint32_t row = i < n_own ? (uint32_t)Lw[i] : 0xFFFFFFFFu;
            const bool in = row < hi_row;
            if (in) {
                const uint32_t wd = row >> 6;
                Lw[i] = (uint16_t)(wordrank[wd] + (uint32_t)__popcll(bits64[wd] & ((1ull << (row & 63u)) - 1ull)));
            }
            n_pass += (uint32_t)__popcll(__ballot(in));          // rows ascend: the rows of this pass are a prefix
        }
        for (uint32_t i = n_pass + lane; i < n_pass + 64u; i += 64) Lw[i] = 0xFFFFu;  // behind the end: never lands
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t wd = tid * 4u + k;
            if (wd >= hi_word) break;
            unsigned long long x = w4[k];
            uint32_t pos = base;
            while (x) {
                U[pos++] = (uint16_t)(wd * 64u + (uint32_t)__builtin_ctzll(x));
                x &= x - 1;
            }
            base += c4[k];
        }
        __syncthreads();  // U and Lw complete; the bit set and the ranks are dead: the ring may be filled
        rows_loaded += un;
        STAMP_MARK(1)

        // ---- row loop
        const uint32_t n_rounds = (un + kQuadRR - 1u) / kQuadRR;
        // round rr: this wave loads union rows rr * 8 + wave * 2 + {0, 1} into ring slots (rr mod depth) * 8 + wave * 2 + {0, 1};
        // always two loads (the zero row behind the end of U), so that the number of loads in flight is a constant
        auto issue = [&](uint32_t rr) {
            const uint32_t u0 = rr * kQuadRR + wave * kQuadLoads;
            const uint32_t slot0 = (rr % kQuadDepth) * kQuadRR + wave * kQuadLoads;
#pragma unroll
            for (uint32_t j = 0; j < kQuadLoads; j++) {
                uint32_t row = p.zero_row;
                if (u0 + j < un) row = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)U[u0 + j]);
#ifndef RTX_QUAD_NO_DMA
                dma_row(bitmap + (size_t)row * 1024u, 1024u, voff, lds_base + (slot0 + j) * 1024u);
#else
                if (row == 0xFFFFFFFEu) dma_row(bitmap, 1024u, voff, lds_base);  // experiment: the loop without its loads
#endif
            }
        };
        uint32_t head = 0;      // own rows [0, head) are folded
        uint32_t wb = 0;        // the window: lane i holds own row wb + i
        uint32_t wu = 0, woff = 0;
        auto window = [&](uint32_t from) {
            wb = from;
            wu = (uint32_t)Lw[from + lane];                                  // position in U, 0xFFFF behind the end
            woff = (wu % kQuadSlots) * 1024u;                                // byte offset of its ring slot
        };
        uint32_t r = 0;         // barriers passed: the rows of U below r * kQuadRR are in the ring
        // barrier of round r -- own loads of that round have landed (those of the kQuadAhead - 1 younger rounds may still
        // fly), own reads of the round that is overwritten next are done; behind the barrier that holds for every wave --
        // then the loads of round r + kQuadAhead
        auto next_round = [&]() {
#if defined(RTX_QUAD_STAMP) && RTX_QUAD_STAMP >= 5
            const unsigned long long ta_ = __builtin_amdgcn_s_memtime();
#endif
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"n"(kQuadLoads * (kQuadAhead - 1)) : "memory");
#if defined(RTX_QUAD_STAMP) && RTX_QUAD_STAMP == 7
            stamp_acc += __builtin_amdgcn_s_memtime() - ta_;   // own loads
#endif
            __builtin_amdgcn_s_barrier();
#if defined(RTX_QUAD_STAMP) && RTX_QUAD_STAMP >= 5
            const unsigned long long tb_ = __builtin_amdgcn_s_memtime();
            if (RTX_QUAD_STAMP == 5) stamp_acc += tb_ - ta_;   // own loads + barrier
#endif
            issue(r + kQuadAhead);
#if defined(RTX_QUAD_STAMP) && RTX_QUAD_STAMP == 6
            stamp_acc += __builtin_amdgcn_s_memtime() - tb_;   // issuing the loads
#endif
            r++;
        };
        // How many own rows the next fold takes: eight as soon as eight have landed (rounds pass while it waits); fewer
        // only when the oldest of them would be overwritten behind the next barrier (it is of round r - 2: left-overs
        // wait one round, no longer) or when everything has landed.  0: no rows left and all rounds done.  The planes are
        // not touched in here, so that the folds below stay one straight line of code (with the folds under
        // data-dependent branches the compiler copied the 40 plane registers at every join: 2 moves per useful op).
        auto acquire = [&]() -> uint32_t {
            for (;;) {
                const uint32_t navail = wb + (uint32_t)__popcll(__ballot(wu < r * kQuadRR));
                const uint32_t left = navail - head;
                if (left >= 8u) return 8u;
                if (r == n_rounds) return left;
                if (left && r >= 2u) {
                    const uint32_t u_head = (uint32_t)__builtin_amdgcn_readlane((int)wu, (int)(head - wb));
                    if (u_head < (r - 1u) * kQuadRR) return left;
                }
                next_round();
            }
        };
        // own rows [head, head + n) out of the ring (n <= 8; the missing ones count as zero rows)
        auto take = [&](uint4 (&A)[8], uint32_t n) {
            const uint32_t l0 = head - wb;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)woff, (int)((l0 + j) & 63u));
                A[j] = *reinterpret_cast<const uint4 *>(lds + ((uint32_t)j < n ? off : kQuadZeroOff) + lane * 16u);
            }
            head += n;
            if (head - wb + 8u > 64u) window(head);
        };
        if (n_rounds) {
            window(0);
            for (uint32_t rr = 0; rr < kQuadAhead; rr++) issue(rr);
            for (;;) {  // 32 own rows per turn: four folds of eight, carries combined as in hit_count_kernel
                uint4 A[8];
                const uint32_t n0 = acquire();
                if (n0 == 0u) break;
                take(A, n0);
                const uint4 c3a = tree8<NP>(pl, A);
                __builtin_amdgcn_sched_barrier(0);  // one set of eight rows in registers at a time (168 VGPRs: three workgroups per CU)
                take(A, acquire());
                const uint4 c3b = tree8<NP>(pl, A);
                const uint4 c4a = csa_plane<NP, 3>(pl, c3a, c3b);
                __builtin_amdgcn_sched_barrier(0);
                take(A, acquire());
                const uint4 c3c = tree8<NP>(pl, A);
                __builtin_amdgcn_sched_barrier(0);
                take(A, acquire());
                const uint4 c3d = tree8<NP>(pl, A);
                const uint4 c4b = csa_plane<NP, 3>(pl, c3c, c3d);
                const uint4 c5 = csa_plane<NP, 4>(pl, c4a, c4b);
                planes_ripple<NP, 5>(pl[0], c5.x);
                planes_ripple<NP, 5>(pl[1], c5.y);
                planes_ripple<NP, 5>(pl[2], c5.z);
                planes_ripple<NP, 5>(pl[3], c5.w);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the loads behind the end of U (zero rows)
        }
        __syncthreads();  // every wave is done with the ring: the next pass or the epilogue may overwrite it
        STAMP_MARK(2)
        if (hi_word >= 1024u) break;
        row_lo = hi_row;
    }
#ifdef RTX_QUAD_STAMP
    if (valid) {
        uint32_t *hist_lds = reinterpret_cast<uint32_t *>(lds + wave * kQuadEpiBytes);
        uint32_t *cnt8 = hist_lds + (kQuadEpiBytes - 4096u) / 4u;
        hit_epilogue<NP, kPacked>(p, pl, q, tile, lane, t, active, hist_lds, cnt8, ns, srows);
    }
    STAMP_MARK(3)
    if (RTX_QUAD_STAMP == 4) stamp_acc = __builtin_amdgcn_s_memtime() - stamp_k0;
    if (tid == 0 && p.group_rows) atomicAdd(&p.group_rows[p.group_base + g], (uint32_t)(stamp_acc >> 6));
    return;
#endif
    if (tid == 0 && p.group_rows) atomicAdd(&p.group_rows[p.group_base + g], rows_loaded);
    if (!valid) return;
    uint32_t *hist_lds = reinterpret_cast<uint32_t *>(lds + wave * kQuadEpiBytes);
    uint32_t *cnt8 = hist_lds + (kQuadEpiBytes - 4096u) / 4u;
    hit_epilogue<NP, kPacked>(p, pl, q, tile, lane, t, active, hist_lds, cnt8, ns, srows);
}

template __global__ void hit_count_quad_kernel<10, true>(HitParams);
template __global__ void hit_count_quad_kernel<10, false>(HitParams);

void launch_hit_count_quad(hipStream_t s, const HitParams &p, uint32_t nq, uint32_t ntiles) {
    static_assert(kQuadLdsBytes >= 4u * kQuadEpiBytes, "the epilogue regions alias the ring and the lists");
    static_assert(kQuadRankOff + 2u * 1025u <= kQuadRingBytes, "prologue scratch inside the ring");
    const uint32_t ng = (nq + 3u) / 4u;
    if (p.counts_lo) hipLaunchKernelGGL((hit_count_quad_kernel<10, true>), dim3(ng, ntiles), dim3(256), kQuadLdsBytes, s, p);
    else hipLaunchKernelGGL((hit_count_quad_kernel<10, false>), dim3(ng, ntiles), dim3(256), kQuadLdsBytes, s, p);
}

}  // namespace rtx
