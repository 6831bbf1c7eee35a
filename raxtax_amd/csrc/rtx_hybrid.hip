// Hybrid dense/sparse reference index and the hit_count kernel that reads it (src/raxtax.rs:41,58-68).
//
// A bitmap row restricted to one quarter tile (2048 references = 256 B = 16 lanes x 16 B) is usually
// either well filled or nearly empty: on the phylo workload 39 % of the (query row, quarter) chunks hold
// more than 64 references, 23 % none and 29 % at most 16 (tools/ measurements).  Chunks with
// at most kSparseMax set bits are therefore moved out of the bitmap into short lists of u16 offsets:
//   qmask[tile][row] : bit q = quarter q of that tile is stored densely in the bitmap
//   soff [tile][row] : CSR offsets into `sent` (offsets 0..8191 inside the tile) for the sparse quarters
// hit_count_hybrid then (1) adds the sparse entries of the query's rows into 8192 packed u16 counters in
// LDS, (2) streams only the dense quarters through the bit-plane adders -- lanes of non-dense quarters
// get an out-of-range buffer offset, which returns 0 without a memory transaction, and rows without a
// dense quarter are dropped from the list -- and (3) adds both in the epilogue.  count[r] is unchanged:
// dense and sparse parts partition the set bits.
#include <hip/hip_runtime.h>

#include "rtx_kernels.hpp"
#include "rtx_math.hpp"
#include "rtx_wave.hpp"

namespace rtx {

static constexpr uint32_t kEmptyRowH = 0xFFFFFFFFu;

typedef uint32_t u32x4h __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------
// build pass A: classify the four quarters of every (row, tile); count the sparse entries
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void hybrid_classify_kernel(const uint32_t *__restrict__ bitmap, uint32_t stride_bytes,
                                                             uint32_t n_rows1, uint32_t sparse_max,
                                                             uint8_t *__restrict__ qmask, uint32_t *__restrict__ scount) {
    const uint32_t row = blockIdx.x, tile = blockIdx.y, lane = threadIdx.x;
    const uint32_t col = tile * 1024u + lane * 16u;
    uint32_t pc = 0;
    if (col < stride_bytes) {
        const uint4 v = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(bitmap) + (size_t)row * stride_bytes + col);
        pc = __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
    }
    // sum inside each group of 16 lanes
    uint32_t qs = pc;
    qs += __shfl_xor(qs, 1, 64);
    qs += __shfl_xor(qs, 2, 64);
    qs += __shfl_xor(qs, 4, 64);
    qs += __shfl_xor(qs, 8, 64);
    uint32_t mask = 0, sparse = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t c = __shfl(qs, q * 16, 64);
        if (c > sparse_max) mask |= 1u << q;
        else sparse += c;
    }
    if (lane == 0) {
        qmask[(size_t)tile * n_rows1 + row] = (uint8_t)mask;
        scount[(size_t)tile * n_rows1 + row] = sparse;
    }
}

// ---------------------------------------------------------------------------
// build pass B: write the sparse entries and clear them from the bitmap
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void hybrid_emit_kernel(uint32_t *__restrict__ bitmap, uint32_t stride_bytes, uint32_t n_rows1,
                                                         const uint8_t *__restrict__ qmask, const uint32_t *__restrict__ soff,
                                                         uint16_t *__restrict__ sent) {
    const uint32_t row = blockIdx.x, tile = blockIdx.y, lane = threadIdx.x;
    const size_t idx = (size_t)tile * n_rows1 + row;
    const uint32_t s0 = soff[idx], s1 = soff[idx + 1];
    if (s0 == s1) return;
    const uint32_t mask = qmask[idx];
    const uint32_t col = tile * 1024u + lane * 16u;
    const bool sparse_lane = col < stride_bytes && !((mask >> (lane >> 4)) & 1u);
    uint4 *p = reinterpret_cast<uint4 *>(reinterpret_cast<char *>(bitmap) + (size_t)row * stride_bytes + col);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (sparse_lane) v = *p;
    const uint32_t pc = __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
    const uint32_t incl = wave_incl_scan_u32(pc);
    uint32_t pos = s0 + incl - pc;
    if (pc) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t x = w[k];
            while (x) {
                const uint32_t b = __ffs((int)x) - 1;
                x &= x - 1;
                sent[pos++] = (uint16_t)(lane * 128u + k * 32u + b);
            }
        }
        *p = make_uint4(0, 0, 0, 0);
    }
}

// ---------------------------------------------------------------------------
// hit_count on the hybrid index: one wave per (query, tile of 8192 references)
// LDS (dynamic, u32 units): hist[hstride] | sp[4096] (8192 packed u16) | drows[rcap] | dmask[rcap/4]
// ---------------------------------------------------------------------------
__device__ __forceinline__ void load8h(uint4 (&buf)[8], const char *__restrict__ bitmap, uint32_t col, uint32_t qbit,
                                       uint32_t stride, const uint32_t *drows, const uint8_t *dmask) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t row = __builtin_amdgcn_readfirstlane(drows[j]);
        const uint32_t m = __builtin_amdgcn_readfirstlane((uint32_t)dmask[j]);
        const char *rowbase = bitmap + (size_t)row * stride;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(rowbase), 0, stride, 0x00027000);
        // lanes of quarters that are not stored densely read out of range: zero, no memory transaction
        const uint32_t voff = (m & qbit) ? col : 0xFFFFFF00u;
        const u32x4h v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
        buf[j] = make_uint4(v.x, v.y, v.z, v.w);
    }
}

template <int NP>
__device__ __forceinline__ uint4 tree8h(uint32_t (&pl)[4][NP], const uint4 (&a)[8]) {
    uint4 e;
    e.x = planes_tree8<NP>(pl[0], a[0].x, a[1].x, a[2].x, a[3].x, a[4].x, a[5].x, a[6].x, a[7].x);
    e.y = planes_tree8<NP>(pl[1], a[0].y, a[1].y, a[2].y, a[3].y, a[4].y, a[5].y, a[6].y, a[7].y);
    e.z = planes_tree8<NP>(pl[2], a[0].z, a[1].z, a[2].z, a[3].z, a[4].z, a[5].z, a[6].z, a[7].z);
    e.w = planes_tree8<NP>(pl[3], a[0].w, a[1].w, a[2].w, a[3].w, a[4].w, a[5].w, a[6].w, a[7].w);
    return e;
}

template <int NP, int P>
__device__ __forceinline__ uint4 csa_plane_h(uint32_t (&pl)[4][NP], const uint4 &a, const uint4 &b) {
    uint4 c;
    csa(pl[0][P], a.x, b.x, pl[0][P], c.x);
    csa(pl[1][P], a.y, b.y, pl[1][P], c.y);
    csa(pl[2][P], a.z, b.z, pl[2][P], c.z);
    csa(pl[3][P], a.w, b.w, pl[3][P], c.w);
    return c;
}

template <int NP>
__global__ __launch_bounds__(64) void hit_count_hybrid_kernel(HitParams p, HybridIndex hy) {
    extern __shared__ uint32_t lds[];
    uint32_t *hist_lds = lds;
    uint32_t *sp = hist_lds + p.hstride;
    uint32_t *drows = sp + 4096;
    uint8_t *dmask = reinterpret_cast<uint8_t *>(drows + hy.rcap);
    const uint32_t q = blockIdx.x, tile = blockIdx.y, lane = threadIdx.x;
    const uint32_t t = p.t[q];
    for (uint32_t i = lane; i <= t; i += 64) hist_lds[i] = 0;
    for (uint32_t i = lane; i < 4096; i += 64) sp[i] = 0;
    __syncthreads();

    // ---- phase 0: sparse entries -> LDS counters; compact the rows that have a dense quarter
    const uint32_t *rows = p.rows + (size_t)q * p.rstride;
    const uint32_t nrows = p.nrows[q];
    const uint8_t *qm = hy.qmask + (size_t)tile * hy.n_rows1;
    const uint32_t *so = hy.soff + (size_t)tile * hy.n_rows1;
    uint32_t nd = 0;
    for (uint32_t i0 = 0; i0 < nrows; i0 += 64) {
        const uint32_t i = i0 + lane;
        uint32_t row = 0, m = 0, s0 = 0, s1 = 0;
        if (i < nrows) {
            row = rows[i];
            m = qm[row];
            s0 = so[row];
            s1 = so[row + 1];
        }
        for (uint32_t e = s0; e < s1; e++) {
            const uint32_t loc = hy.sent[e];
            atomicAdd(&sp[loc >> 1], 1u << (16u * (loc & 1u)));
        }
        const unsigned long long bal = __ballot(m != 0);
        if (m) {
            const uint32_t pos = nd + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            drows[pos] = row;
            dmask[pos] = (uint8_t)m;
        }
        nd += (uint32_t)__popcll(bal);
    }
    const uint32_t padded = ((nd + 31u) & ~31u) + 8u;
    for (uint32_t i = nd + lane; i < padded; i += 64) { drows[i] = hy.zero_row; dmask[i] = 0; }
    __syncthreads();

    // ---- phase 1: dense quarters through the bit-plane adders (same tree as hit_count_kernel)
    const uint32_t col = tile * 1024u + lane * 16u;
    const uint32_t qbit = 1u << (lane >> 4);
    const bool active = col < p.stride_bytes;
    uint32_t pl[4][NP];
#pragma unroll
    for (int w = 0; w < 4; w++)
#pragma unroll
        for (int b = 0; b < NP; b++) pl[w][b] = 0;
    {
        const uint32_t n32 = (nd + 31u) >> 5;
        const char *bitmap = reinterpret_cast<const char *>(p.bitmap);
        const uint32_t stride = p.stride_bytes;
        uint4 A[8], B[8];
        if (n32) load8h(A, bitmap, col, qbit, stride, drows, dmask);
        for (uint32_t g = 0; g < n32; g++) {
            const uint32_t *r = drows + g * 32;
            const uint8_t *mk = dmask + g * 32;
            load8h(B, bitmap, col, qbit, stride, r + 8, mk + 8);
            const uint4 c3a = tree8h<NP>(pl, A);
            load8h(A, bitmap, col, qbit, stride, r + 16, mk + 16);
            const uint4 c3b = tree8h<NP>(pl, B);
            const uint4 c4a = csa_plane_h<NP, 3>(pl, c3a, c3b);
            load8h(B, bitmap, col, qbit, stride, r + 24, mk + 24);
            const uint4 c3c = tree8h<NP>(pl, A);
            load8h(A, bitmap, col, qbit, stride, r + 32, mk + 32);  // look-ahead group (zero rows past the end)
            const uint4 c3d = tree8h<NP>(pl, B);
            const uint4 c4b = csa_plane_h<NP, 3>(pl, c3c, c3d);
            const uint4 c5 = csa_plane_h<NP, 4>(pl, c4a, c4b);
            planes_ripple<NP, 5>(pl[0], c5.x);
            planes_ripple<NP, 5>(pl[1], c5.y);
            planes_ripple<NP, 5>(pl[2], c5.z);
            planes_ripple<NP, 5>(pl[3], c5.w);
        }
    }

    // ---- phase 2: counts = planes + sparse counters; exact-match zeroing; store; histogram
    if (active) {
        const uint32_t ref0 = tile * 8192u + lane * 128u;
        const uint32_t nvalid =
            ref0 >= p.n_refs ? 0u : ((p.n_refs - ref0) < 128u ? (uint32_t)(p.n_refs - ref0) : 128u);
        if (p.flags & RTX_SKIP_EXACT_MATCHES) {  // raxtax.rs:65-68: clear the reference in the planes AND in the sparse counters
            const uint64_t e0 = p.exact_off[p.q0 + q], e1 = p.exact_off[p.q0 + q + 1];
            for (uint64_t e = e0; e < e1; e++) {
                const uint32_t id = p.exact_ids[e] - p.ref_base;  // local id; other shards' ids wrap out of range
                if (id >= ref0 && id < ref0 + 128u) {
                    const uint32_t l = id - ref0;
                    const uint32_t w = l >> 5, msk = ~(1u << (l & 31u));
#pragma unroll
                    for (int ww = 0; ww < 4; ww++)
                        if ((uint32_t)ww == w) {
#pragma unroll
                            for (int bb = 0; bb < NP; bb++) pl[ww][bb] &= msk;
                        }
                    sp[lane * 64 + (l >> 1)] &= (l & 1u) ? 0x0000FFFFu : 0xFFFF0000u;  // this lane owns these words
                }
            }
        }
        uint16_t *out = p.counts + (size_t)q * p.npad + ref0;
        const uint4 *spv = reinterpret_cast<const uint4 *>(sp + lane * 64);
#pragma unroll
        for (int w = 0; w < 4; w++) {
#pragma unroll
            for (int g2 = 0; g2 < 4; g2++) {  // 8 references per 16-byte store
                uint32_t lo0, hi0, lo1, hi1;
                planes_unpack4<NP>(pl[w], 2 * g2, lo0, hi0);
                planes_unpack4<NP>(pl[w], 2 * g2 + 1, lo1, hi1);
                const uint4 s4 = spv[w * 4 + g2];
                uint4 st;
                st.x = __builtin_amdgcn_perm(hi0, lo0, 0x05010400u) + s4.x;  // packed u16 pairs: no carry (counts <= t)
                st.y = __builtin_amdgcn_perm(hi0, lo0, 0x07030602u) + s4.y;
                st.z = __builtin_amdgcn_perm(hi1, lo1, 0x05010400u) + s4.z;
                st.w = __builtin_amdgcn_perm(hi1, lo1, 0x07030602u) + s4.w;
                const uint32_t rbase = w * 32 + g2 * 8;
                const uint32_t cw[4] = {st.x, st.y, st.z, st.w};
                *reinterpret_cast<uint4 *>(out + rbase) = st;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const uint32_t c = (cw[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu;
                    if (rbase + j < nvalid) atomicAdd(&hist_lds[c], 1u);
                }
            }
        }
    }
    __syncthreads();
    uint32_t *hist = p.hist + (size_t)q * p.hstride;
    for (uint32_t m = lane; m <= t; m += 64) {
        const uint32_t v = hist_lds[m];
        if (v) atomicAdd(&hist[m], v);
    }
}

template __global__ void hit_count_hybrid_kernel<10>(HitParams, HybridIndex);
template __global__ void hit_count_hybrid_kernel<12>(HitParams, HybridIndex);
template __global__ void hit_count_hybrid_kernel<16>(HitParams, HybridIndex);

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
void launch_hybrid_classify(hipStream_t s, const uint32_t *bitmap, uint32_t stride_bytes, uint32_t n_rows1, uint32_t ntiles,
                            uint32_t sparse_max, uint8_t *qmask, uint32_t *scount) {
    hipLaunchKernelGGL(hybrid_classify_kernel, dim3(n_rows1, ntiles), dim3(64), 0, s, bitmap, stride_bytes, n_rows1, sparse_max,
                       qmask, scount);
}
void launch_hybrid_emit(hipStream_t s, uint32_t *bitmap, uint32_t stride_bytes, uint32_t n_rows1, uint32_t ntiles,
                        const uint8_t *qmask, const uint32_t *soff, uint16_t *sent) {
    hipLaunchKernelGGL(hybrid_emit_kernel, dim3(n_rows1, ntiles), dim3(64), 0, s, bitmap, stride_bytes, n_rows1, qmask, soff, sent);
}
size_t hit_count_hybrid_lds_bytes(uint32_t hstride, uint32_t rcap) {
    return sizeof(uint32_t) * ((size_t)hstride + 4096 + rcap) + rcap + 16;
}
void launch_hit_count_hybrid(hipStream_t s, const HitParams &p, const HybridIndex &hy, uint32_t nq, uint32_t ntiles, int planes) {
    const size_t lds = hit_count_hybrid_lds_bytes(p.hstride, hy.rcap);
    if (planes <= 10) hipLaunchKernelGGL(hit_count_hybrid_kernel<10>, dim3(nq, ntiles), dim3(64), lds, s, p, hy);
    else if (planes <= 12) hipLaunchKernelGGL(hit_count_hybrid_kernel<12>, dim3(nq, ntiles), dim3(64), lds, s, p, hy);
    else hipLaunchKernelGGL(hit_count_hybrid_kernel<16>, dim3(nq, ntiles), dim3(64), lds, s, p, hy);
}

}  // namespace rtx
