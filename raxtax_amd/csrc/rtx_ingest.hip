// Query ingest: the reference hands raxtax() its queries one byte per base (parser.rs:11-34: 4-bit one-hot codes, ambiguity codes
// are unions of bits, N = 15), and that is what the kernels read.  Over PCIe the bases travel two per byte: the host packs them
// into page-locked staging memory (threads of the library's budget; packing costs no more than the copy into pinned memory that an
// asynchronous transfer needs anyway), the transfer runs on a stream of its own while the previous batch is classified, and this
// kernel unpacks them into the batch's base array (0.3 ms per million 658-bp queries: 0.33 GB read, 0.66 GB written).
#include <hip/hip_runtime.h>

#if !defined(__HIP_DEVICE_COMPILE__)
#include <emmintrin.h>
#endif

#include <thread>
#include <vector>

#include "rtx_kernels.hpp"

namespace rtx {

// packed byte i = base 2i | base 2i+1 << 4; n_bases bases out, then `pad` zero bytes (the kernels read up to 63 bytes behind a batch)
__global__ __launch_bounds__(256) void unpack_nibbles_kernel(const uint8_t *__restrict__ packed, uint8_t *__restrict__ bases, uint64_t n_bases,
                                                             uint64_t n_out) {
    const uint64_t i16 = ((uint64_t)blockIdx.x * 256u + threadIdx.x) * 16u;  // 16 packed bytes -> 32 bases per thread
    if (i16 * 2u >= n_out) return;
    const uint64_t n_packed = (n_bases + 1u) >> 1;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (i16 + 16u <= n_packed) {
        v = *reinterpret_cast<const uint4 *>(packed + i16);  // the staging buffers are 16-byte aligned and padded
    } else {
        uint8_t b[16];
#pragma unroll
        for (int k = 0; k < 16; k++) b[k] = i16 + (uint64_t)k < n_packed ? packed[i16 + k] : (uint8_t)0;
        __builtin_memcpy(&v, b, 16);
    }
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t o[8];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        // bytes p0 p1 p2 p3 -> (p0 & 15, p0 >> 4, p1 & 15, p1 >> 4), (p2 ..)
        const uint32_t lo = w[k] & 0x0F0F0F0Fu, hi = (w[k] >> 4) & 0x0F0F0F0Fu;
        o[2 * k] = __builtin_amdgcn_perm(hi, lo, 0x05010400u);      // lo.b0, hi.b0, lo.b1, hi.b1
        o[2 * k + 1] = __builtin_amdgcn_perm(hi, lo, 0x07030602u);  // lo.b2, hi.b2, lo.b3, hi.b3
    }
    const uint64_t o0 = i16 * 2u;
    if (o0 + 32u <= n_bases) {
        uint4 *dst = reinterpret_cast<uint4 *>(bases + o0);
        dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
        dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
    } else {  // the end of the batch: an odd last nibble and the padding are zero
        const uint8_t *ob = reinterpret_cast<const uint8_t *>(o);
        for (uint32_t k = 0; k < 32u && o0 + k < n_out; k++) bases[o0 + k] = o0 + k < n_bases ? ob[k] : (uint8_t)0;
    }
}

void launch_unpack_nibbles(hipStream_t s, const uint8_t *packed, uint8_t *bases, uint64_t n_bases, uint64_t n_out) {
    const uint64_t threads = (n_out + 31u) / 32u;
    if (threads) hipLaunchKernelGGL(unpack_nibbles_kernel, dim3((unsigned)((threads + 255u) / 256u)), dim3(256), 0, s, packed, bases, n_bases, n_out);
}

// Host side: out[i] = in[2i] | in[2i+1] << 4 for n bases (an odd last base alone in its byte).  Returns false if a byte above 15
// was seen (not a code of parser.rs:11-34: the caller then sends the batch unpacked, so that whatever the kernels made of such a
// byte before they still make of it).  SSE2: 32 bases per step.
#if defined(__HIP_DEVICE_COMPILE__)
bool pack_nibbles_mt(const uint8_t *, uint64_t, uint8_t *, unsigned);  // host only
#else
static bool pack_range(const uint8_t *in, uint64_t n, uint8_t *out) {
    __m128i seen = _mm_setzero_si128();
    const __m128i low = _mm_set1_epi16(0x00FF);
    uint64_t i = 0;
    for (; i + 32 <= n; i += 32) {
        const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i *>(in + i));
        const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i *>(in + i + 16));
        seen = _mm_or_si128(seen, _mm_or_si128(a, b));
        const __m128i pa = _mm_and_si128(_mm_or_si128(a, _mm_srli_epi16(a, 4)), low);  // per 16-bit lane: b0 | b1 << 4
        const __m128i pb = _mm_and_si128(_mm_or_si128(b, _mm_srli_epi16(b, 4)), low);
        _mm_storeu_si128(reinterpret_cast<__m128i *>(out + i / 2), _mm_packus_epi16(pa, pb));
    }
    uint8_t tail = 0;
    for (; i < n; i += 2) {
        const uint8_t b0 = in[i], b1 = i + 1 < n ? in[i + 1] : (uint8_t)0;
        tail |= b0 | b1;
        out[i / 2] = (uint8_t)((b0 & 15u) | (b1 << 4));
    }
    alignas(16) uint8_t s[16];
    _mm_store_si128(reinterpret_cast<__m128i *>(s), seen);
    for (int k = 0; k < 16; k++) tail |= s[k];
    return (tail & 0xF0u) == 0;
}

bool pack_nibbles_mt(const uint8_t *in, uint64_t n, uint8_t *out, unsigned nt) {
    if (nt <= 1 || n < (1u << 20)) return pack_range(in, n, out);
    std::vector<std::thread> th;
    std::vector<uint8_t> ok(nt, 1);
    for (unsigned k = 0; k < nt; k++) {
        const uint64_t a = (n * k / nt) & ~63ull, b = k + 1 == nt ? n : ((n * (k + 1) / nt) & ~63ull);  // even cuts: a pair never straddles two ranges
        th.emplace_back([=, &ok] { ok[k] = pack_range(in + a, b - a, out + a / 2) ? 1 : 0; });
    }
    for (auto &t : th) t.join();
    for (unsigned k = 0; k < nt; k++)
        if (!ok[k]) return false;
    return true;
}
#endif

}  // namespace rtx
