// Exact-match lookup on the device: `tree.sequences.get(sequence)` (src/raxtax.rs:42, type src/tree.rs:40) for every query of a
// batch.  The reference hashes the 658 encoded bytes of a query (ahash) and probes a HashMap<Vec<u8>, Vec<u32>>; here the distinct
// reference sequences ("groups": a sequence and the ids of every reference that has it, ascending as Tree::new pushes them,
// tree.rs:109-112) sit in an open-addressing table in HBM, keyed by a 64-bit hash of the bytes (rtx_math.hpp: em_mix_word), and one
// wave per query hashes, probes and VERIFIES byte by byte -- a hash collision can cost a compare, never a wrong id.  The result is
// the group of the query (or none); hit_count's zeroing under --skip-exact-matches (raxtax.rs:65-68), prune_kernel and the host
// (override raxtax.rs:73-84, warning raxtax.rs:43-53) read the ids through it.
#include <hip/hip_runtime.h>

#include "rtx_kernels.hpp"
#include "rtx_math.hpp"

namespace rtx {

// 8 bytes at seq + 8 j, zero beyond len (the buffers are padded: reading up to 7 bytes past the end is safe)
__device__ __forceinline__ uint64_t em_load_word(const uint8_t *seq, uint64_t len, uint64_t j) {
    uint64_t w;
    __builtin_memcpy(&w, seq + 8u * j, 8);
    const uint64_t rest = len - 8u * j;  // > 0
    if (rest < 8u) w &= (1ull << (8u * rest)) - 1ull;
    return w;
}

__global__ __launch_bounds__(256) void exact_match_kernel(ExactParams p) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t q = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (q >= p.n_q) return;  // wave-uniform
    const uint64_t b0 = p.base_off[q], len = p.base_off[q + 1] - b0;
    const uint8_t *seq = p.bases + b0;
    const uint64_t nw = (len + 7u) >> 3;
    uint64_t sum = 0;
    for (uint64_t j = lane; j < nw; j += 64) sum += em_mix_word(em_load_word(seq, len, j), j);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
    const uint64_t h = em_finish(sum, len) & p.hash_mask;
    const uint32_t tag = em_tag(h), mask = (1u << p.bits) - 1u;
    uint32_t slot = em_slot(h, p.bits);
    uint32_t found = 0xFFFFFFFFu;
    for (uint32_t probe = 0; probe <= mask; probe++) {  // wave-uniform; the table is at most half full
        const uint2 e = p.table[slot];
        if (e.y == 0u) break;  // empty slot: no reference has this sequence
        if (e.x == tag) {
            const uint32_t g = e.y - 1u;
            const uint64_t r0 = p.rep_off[g], rlen = p.rep_off[g + 1] - r0;
            if (rlen == len) {
                bool differ = false;
                for (uint64_t j = lane; j < nw; j += 64) differ = differ || em_load_word(seq, len, j) != em_load_word(p.rep_bytes + r0, len, j);
                if (__ballot(differ) == 0ull) { found = g; break; }
            }
        }
        slot = (slot + 1u) & mask;
    }
    if (lane == 0) p.grp_out[q] = found;
}

void launch_exact_match(hipStream_t s, const ExactParams &p) {
    if (p.n_q) hipLaunchKernelGGL(exact_match_kernel, dim3((p.n_q + 3u) / 4u), dim3(256), 0, s, p);
}

}  // namespace rtx
