// Host-side emulation of the device arithmetic in rtx_math.hpp, for the CPU test-suite
// (tests/test_device_math_cpu.py).  TEST SUPPORT ONLY: it is not part of libraxtax_hip.so
// and never runs on the product path.  It executes the very same inline functions the HIP
// kernels call (bit-plane adders/unpack, pmf recurrence), sequentially on x86, so that
// their logic is checked against the oracle without a GPU.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "rtx_math.hpp"

using namespace rtx;

extern "C" {

// rows: n_rows x 1 words (one 32-reference column).  n_rows must be a multiple of 8.
// out: 32 counters.
void emul_planes_count(const uint32_t *rows, uint32_t n_rows, int planes, uint32_t *out) {
    auto run = [&](auto tag) {
        constexpr int NP = decltype(tag)::value;
        uint32_t pl[NP];
        for (int p = 0; p < NP; p++) pl[p] = 0;
        for (uint32_t r = 0; r + 8 <= n_rows; r += 8)
            planes_add8<NP>(pl, rows[r], rows[r + 1], rows[r + 2], rows[r + 3], rows[r + 4], rows[r + 5], rows[r + 6],
                            rows[r + 7]);
        for (int g = 0; g < 8; g++) {
            uint32_t lo, hi;
            planes_unpack4<NP>(pl, g, lo, hi);
            for (int j = 0; j < 4; j++) out[4 * g + j] = ((lo >> (8 * j)) & 0xFF) | (((hi >> (8 * j)) & 0xFF) << 8);
        }
    };
    if (planes == 10) run(std::integral_constant<int, 10>{});
    else if (planes == 12) run(std::integral_constant<int, 12>{});
    else run(std::integral_constant<int, 16>{});
}

// Sequential emulation of prob_table_kernel.  hist has t+1 entries.  Returns 0, or 1 when the
// kernel would flag RTX_Q_NO_KMERS.
int emul_prob_table(uint32_t t, const uint32_t *hist, uint64_t n_refs, const double *lf, double *table_z, double *z,
                    double *gs) {
    const uint32_t n = t >> 1;
    if (t == 0) return 1;
    std::vector<uint32_t> ms;
    for (uint32_t m = 0; m <= t; m++)
        if (hist[m]) ms.push_back(m);
    std::vector<double> inv(t + n + 2, 0.0);
    for (uint32_t x = 1; x <= t + n; x++) inv[x] = 1.0 / (double)x;
    const double ln_total = ln_binom_tab(lf, t + n - 1, n);
    std::vector<double> tab(t + 1, 0.0);
    if (ms.back() == t) {
        for (uint32_t m : ms) tab[m] = only_last_pmf_tab(lf, t, n, m, ln_total);
    } else {
        if (n == 0) return 1;
        std::vector<double> prod(n + 1, 0.0), Pi(n + 1);
        for (uint32_t m : ms) {
            if (m == 0) continue;
            const double h = (double)hist[m];
            PmfState st = pmf_start(lf, t, n, m, ln_total);
            double L = pmf_ln_cmf(st);
            for (uint32_t i = 0; i <= n; i++) {
                if (i > 0) {
                    const double c_old = st.c;
                    const int k_old = st.k;
                    pmf_step(st, inv.data(), t, n, m, i);
                    if (st.k > 0) L = neg_inf();
                    else if (st.c != c_old || k_old != 0) L = log(st.c);
                }
                prod[i] += h * L;
            }
        }
        for (uint32_t i = 0; i <= n; i++) Pi[i] = exp(prod[i]);
        for (uint32_t m : ms) {
            if (m == 0) { tab[0] = Pi[0]; continue; }
            PmfState st = pmf_start(lf, t, n, m, ln_total);
            double acc = 0.0;
            for (uint32_t i = 0; i <= n; i++) {
                if (i > 0) pmf_step(st, inv.data(), t, n, m, i);
                const double P = Pi[i];
                if (P > 0.0 && st.k == 0 && st.c > 0.0) acc += st.v * P / st.c;
            }
            tab[m] = acc;
        }
    }
    double Z = 0.0;
    for (uint32_t m : ms) Z += (double)hist[m] * tab[m];
    const double inv_n = 1.0 / (double)n_refs;
    double g = 0.0;
    for (uint32_t m = 0; m <= t; m++) table_z[m] = 0.0;
    for (uint32_t m : ms) {
        const double v = tab[m] / Z;
        table_z[m] = v;
        g += (double)hist[m] * (v - inv_n) * (v - inv_n);
    }
    *z = Z;
    *gs = sqrt(g);
    return 0;
}

}  // extern "C"
