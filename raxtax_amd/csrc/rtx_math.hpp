// Arithmetic shared by the HIP kernels and the host-side emulation used by the CPU
// tests (tests/test_device_math_cpu.py drives it through rtx_emul_* in rtx_emul.cpp).
// Everything here is `RTX_HD` so that the exact same source runs on gfx950 and on x86.
#pragma once

#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define RTX_HD __host__ __device__ __forceinline__
#else
#define RTX_HD inline
#endif

namespace rtx {

// ---------------------------------------------------------------------------
// Bit-sliced ("vertical") counters.
//
// hit_count adds one 32-reference bitmap word per query k-mer into NP bit planes:
// plane p holds bit p of the 32 per-reference counters.  Eight words are folded per
// step with a carry-save adder tree (7 CSAs -> one carry of weight 8) followed by a
// ripple into planes 3..NP-1.  count[r] = |K(q) ∩ K(r)| (src/raxtax.rs:58-64) exactly:
// the planes are an exact binary representation as long as count < 2^NP.
// ---------------------------------------------------------------------------

// full adder on bit vectors: (a + b + c) -> sum (weight 1), carry (weight 2)
RTX_HD void csa(uint32_t a, uint32_t b, uint32_t c, uint32_t &sum, uint32_t &carry) {
    uint32_t u = a ^ b;
    sum = u ^ c;
    carry = (a & b) | (u & c);  // v_bfi-able majority
}

template <int NP>
RTX_HD void planes_add8(uint32_t (&pl)[NP], uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4,
                        uint32_t a5, uint32_t a6, uint32_t a7) {
    uint32_t t1a, t1b, t1c, t1d, t2a, t2b, e;
    csa(pl[0], a0, a1, pl[0], t1a);
    csa(pl[0], a2, a3, pl[0], t1b);
    csa(pl[1], t1a, t1b, pl[1], t2a);
    csa(pl[0], a4, a5, pl[0], t1c);
    csa(pl[0], a6, a7, pl[0], t1d);
    csa(pl[1], t1c, t1d, pl[1], t2b);
    csa(pl[2], t2a, t2b, pl[2], e);
#pragma unroll
    for (int p = 3; p < NP; p++) {  // ripple the weight-8 carry upwards
        uint32_t c = pl[p] & e;
        pl[p] ^= e;
        e = c;
    }
}

// Spreads the 4 bits x[3:0] into the low bit of the 4 bytes of the result.
RTX_HD uint32_t spread4(uint32_t x) { return ((x & 0xFu) * 0x00204081u) & 0x01010101u; }

// Counters of references 4g..4g+3 of a 32-reference word: low 8 bits of each counter in
// the bytes of `lo`, bits 8.. in the bytes of `hi`.
template <int NP>
RTX_HD void planes_unpack4(const uint32_t (&pl)[NP], int g, uint32_t &lo, uint32_t &hi) {
    lo = 0;
    hi = 0;
#pragma unroll
    for (int p = 0; p < NP; p++) {
        uint32_t s = spread4(pl[p] >> (4 * g));
        if (p < 8) lo |= s << p;
        else hi |= s << (p - 8);
    }
}

// ---------------------------------------------------------------------------
// prob.rs restated for the device: the reference builds ln pmf_m(i) for every distinct
// hit count m and every i in 0..=n (prob.rs:121-170), exponentiates, accumulates ln cmf,
// sums hist[m]*ln cmf over m (prob.rs:62-73) and evaluates
//     table[m] = sum_i exp(ln pmf_m(i) + prod(i) - ln cmf_m(i))        (prob.rs:74-90).
// Here pmf_m(i) = C(m+i-1,i) C(t-m+n-i-1,n-i) / C(t+n-1,n) is advanced in the linear
// domain by its exact ratio
//     pmf_m(i)/pmf_m(i-1) = (m+i-1)(n-i+1) / (i (t-m+n-i))
// starting from pmf_m(0) = exp(lnC(t-m+n-1,n) - lnC(t+n-1,n)).  Because the start can be
// far below DBL_MIN for long sequences the value is carried as v * 2^(-512 k): whenever v
// exceeds 2^100 it is rescaled (to ~2^-412, still a normal double) and k decremented.
// While k > 0 the true pmf and cmf are < 2^-412 (~1e-124); such terms are treated as 0
// (ln cmf = -inf), which is what exp() underflow does to them in the reference at a slightly
// lower threshold, and cannot matter: Z = sum_r table[count_r] >= 1.
// ---------------------------------------------------------------------------
struct PmfState {
    double v;  // scaled pmf_m(i)
    double c;  // scaled cmf_m(i) = sum_{j<=i} pmf_m(j)
    int k;     // scale exponent: true value = v * 2^(-512 k)
};

constexpr double kScaleLn = 354.89135644669199;  // 512 * ln 2
constexpr double kScaleUp = 1.2676506002282294e30;    // 2^100: rescale threshold
constexpr double kScaleDown = 7.458340731200207e-155; // 2^-512

// ln_total = lnC(t+n-1, n); lf = ln-factorial table, lf[x] = ln(x!)
RTX_HD double ln_binom_tab(const double *lf, uint32_t n, uint32_t k) { return lf[n] - lf[k] - lf[n - k]; }

RTX_HD PmfState pmf_start(const double *lf, uint32_t t, uint32_t n, uint32_t m, double ln_total) {
    // 0 < m < t.  ln pmf_m(0) = lnC(t-m+n-1, n) - ln_total  (prob.rs:143-146,158)
    double x0 = ln_binom_tab(lf, t - m + n - 1, n) - ln_total;
    PmfState s;
    s.k = 0;
    if (x0 < -600.0) {
        s.k = (int)ceil((-600.0 - x0) / kScaleLn);
        x0 += (double)s.k * kScaleLn;
    }
    s.v = exp(x0);
    s.c = s.v;
    return s;
}

// advance from i-1 to i (1 <= i <= n); inv[x] = 1.0/x
RTX_HD void pmf_step(PmfState &s, const double *inv, uint32_t t, uint32_t n, uint32_t m, uint32_t i) {
    double num = (double)(m + i - 1) * (double)(n - i + 1);
    s.v = s.v * num * inv[i] * inv[t - m + n - i];
    s.c += s.v;
    if (s.k > 0 && s.v > kScaleUp) {
        s.v *= kScaleDown;
        s.c *= kScaleDown;
        s.k -= 1;
    }
}

RTX_HD double neg_inf() { return -INFINITY; }

// ln cmf, or -inf where the true value is below 2^-512 (or exactly 0)
RTX_HD double pmf_ln_cmf(const PmfState &s) {
    if (s.k > 0 || !(s.c > 0.0)) return neg_inf();
    return log(s.c);
}

// prob.rs:105-119 with the table
RTX_HD double only_last_pmf_tab(const double *lf, uint32_t t, uint32_t n, uint32_t m, double ln_total) {
    if (m == t) return 1.0;
    if (m == 0) return 0.0;
    return exp(ln_binom_tab(lf, m + n - 1, n) - ln_total);
}

}  // namespace rtx
