// Segment classes of the bitmap index.
//
// A bitmap row is read in segments of one tile (8192 references, 1 KiB).  The references are in lineage order, so a
// k-mer that is typical for a clade fills a few tiles and leaves the others empty or nearly empty: 11 % of the
// (row, tile) segments a query asks for are empty and another 20 % hold at most 32 references (bench workload),
// and those are the segments no other query has just pulled into L2.  At index creation every segment is classified
//     0 = empty (never loaded), 1 = dense (loaded as a 1-KiB row segment), s + 2 = sparse, slot s
// and the references of a sparse segment (at most 16) are written to a 32-byte slot of 16 local ids (u16); unused entries
// hold ids >= kSegPad (pad words of hit_count).  kmer_extract turns the classes into per-(query, tile) row lists; hit_count
// adds the sparse slots through byte counters in LDS.  (Lists for segments of 17 .. 128 references were built twice in round 2:
// fewer requested bytes, bit-exact, and slower both times -- DESIGN.md section 3; removed in round 3.)
#include <hip/hip_runtime.h>

#include "rtx_kernels.hpp"
#include "rtx_math.hpp"
#include "rtx_wave.hpp"

namespace rtx {

// pop[row][tile] = number of references in the segment (saturated to 65535); one wave per row
__global__ __launch_bounds__(64) void seg_popcount_kernel(const uint32_t *__restrict__ bitmap, uint32_t stride_bytes,
                                                          uint32_t ntiles, uint16_t *__restrict__ pop) {
    const uint32_t row = blockIdx.x, lane = threadIdx.x, n_rows1 = gridDim.x;
    for (uint32_t tile = 0; tile < ntiles; tile++) {
        const uint32_t col = tile * 1024u + lane * 16u;
        uint32_t c = 0;
        if (col < stride_bytes) {
            const uint4 v = *reinterpret_cast<const uint4 *>(bitmap + bitmap_word(row, tile * 256u + lane * 4u, n_rows1));
            c = __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
        if (lane == 0) pop[(size_t)row * ntiles + tile] = (uint16_t)(c > 65535u ? 65535u : c);
    }
}

// writes the local reference ids of every sparse segment into its slot; one wave per row
__global__ __launch_bounds__(64) void seg_emit_kernel(const uint32_t *__restrict__ bitmap, uint32_t stride_bytes, uint32_t ntiles,
                                                      const uint32_t *__restrict__ seginfo, uint32_t seg_stride,
                                                      uint16_t *__restrict__ slots) {
    const uint32_t row = blockIdx.x, lane = threadIdx.x, n_rows1 = gridDim.x;
    for (uint32_t tile = 0; tile < ntiles; tile++) {
        const uint32_t code = seginfo[(size_t)row * seg_stride + tile];
        if (code < 2u) continue;  // wave-uniform
        uint16_t *out = slots + (size_t)(code - 2u) * kSegSlotEntries;
        const uint32_t cap = kSegSlotEntries;
        const uint32_t col = tile * 1024u + lane * 16u;
        uint32_t w[4] = {0, 0, 0, 0};
        if (col < stride_bytes) {
            const uint4 v = *reinterpret_cast<const uint4 *>(bitmap + bitmap_word(row, tile * 256u + lane * 4u, n_rows1));
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        }
        const uint32_t cnt = __popc(w[0]) + __popc(w[1]) + __popc(w[2]) + __popc(w[3]);
        uint32_t pos = wave_incl_scan_u32(cnt) - cnt;
        const uint32_t L = tile_lanes(stride_bytes, tile);
#pragma unroll
        for (uint32_t ww = 0; ww < 4; ww++) {
            uint32_t x = w[ww];
            while (x) {
                const uint32_t b = (uint32_t)__ffs((int)x) - 1u;
                x &= x - 1u;
                const uint32_t g = ww * 4u + (b >> 3);  // inverse of ref_slot (rtx_math.hpp)
                if (pos < cap) out[pos] = (uint16_t)((g * L + lane) * 8u + (b & 7u));
                pos++;
            }
        }
        {   // the entries behind the last reference: hit_count adds them into pad words behind its byte counters without
            // looking (different words for neighbouring slots and entries: one word would serialise the LDS atomics)
            const uint32_t total = (uint32_t)__shfl((int)pos, 63, 64);  // lane 63 ends behind the last one
            const uint32_t salt = (code - 2u) * 5u;
            for (uint32_t i = lane; i < cap; i += 64)
                if (i >= total) out[i] = (uint16_t)seg_pad(i + salt);
        }
    }
}

void launch_seg_popcount(hipStream_t s, const uint32_t *bitmap, uint32_t stride_bytes, uint32_t n_rows1, uint32_t ntiles, uint16_t *pop) {
    hipLaunchKernelGGL(seg_popcount_kernel, dim3(n_rows1), dim3(64), 0, s, bitmap, stride_bytes, ntiles, pop);
}
void launch_seg_emit(hipStream_t s, const uint32_t *bitmap, uint32_t stride_bytes, uint32_t n_rows1, uint32_t ntiles,
                     const uint32_t *seginfo, uint32_t seg_stride, uint16_t *slots) {
    hipLaunchKernelGGL(seg_emit_kernel, dim3(n_rows1), dim3(64), 0, s, bitmap, stride_bytes, ntiles, seginfo, seg_stride, slots);
}

}  // namespace rtx
