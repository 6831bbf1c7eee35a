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
//   sort          : rocPRIM device radix sort of (key, query index) pairs -> perm[position] = query
//   invert        : inv[query] = position
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include "rtx_kernels.hpp"

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
        // lane l rolls over the windows [l*wpl, (l+1)*wpl): wpl + 11 sequential byte reads
        const uint64_t nwin = len - kSketchK + 1;
        const uint64_t wpl = (nwin + 63) / 64;
        const uint64_t w0 = (uint64_t)lane * wpl;
        const uint64_t w1 = w0 + wpl < nwin ? w0 + wpl : nwin;
        uint32_t code = 0, run = 0;
        for (uint64_t i = w0; i < w1 + kSketchK - 1 && w0 < w1; i++) {
            const uint32_t c = seq[i];
            const bool ok = c == 1u || c == 2u || c == 4u || c == 8u;
            code = ((code << 2) | (((uint32_t)__ffs((int)c) - 1u) & 3u)) & 0xFFFFFFu;
            run = ok ? run + 1u : 0u;
            if (run >= (uint32_t)kSketchK) {
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
void launch_invert_perm(hipStream_t s, const uint32_t *perm, uint32_t n, uint32_t *inv) {
    hipLaunchKernelGGL(invert_perm_kernel, dim3((n + 255) / 256), dim3(256), 0, s, perm, n, inv);
}
void launch_identity_perm(hipStream_t s, uint32_t n, uint32_t *perm, uint32_t *inv) {
    hipLaunchKernelGGL(identity_perm_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, perm, inv);
}

// tmp == nullptr: only reports the temporary storage needed
int cluster_sort(hipStream_t s, void *tmp, size_t *tmp_bytes, const uint64_t *keys_in, uint64_t *keys_out, const uint32_t *idx_in,
                 uint32_t *perm_out, size_t n) {
    const hipError_t e = rocprim::radix_sort_pairs(tmp, *tmp_bytes, keys_in, keys_out, idx_in, perm_out, n, 0, 3 * kSketchBits, s);
    return e == hipSuccess ? 0 : 1;
}

}  // namespace rtx
