// Host-side emulation of the device arithmetic in rtx_math.hpp, for the CPU test-suite
// (tests/test_device_math_cpu.py).  TEST SUPPORT ONLY: it is not part of libraxtax_hip.so
// and never runs on the product path.  It executes the very same inline functions the HIP
// kernels call (bit-plane adders/unpack, pmf recurrence), sequentially on x86, so that
// their logic is checked against the oracle without a GPU.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <array>
#include <vector>

#include "rtx_math.hpp"

using namespace rtx;

extern "C" {

// The arithmetic of the two-level bounds pass (rtx_bounds2.hip) on one 32-counter column: the rows are dealt to `groups` lane groups as the
// load instructions deal them (unit of 8 * groups rows: group g takes rows 8 g .. 8 g + 7 of the unit), every group folds its own (tree8 +
// ripple), the partial plane sets are added pairwise as bit-sliced numbers (the butterfly of reduce_rows) and the largest counter is taken on
// the planes.  n_rows must be a multiple of 8 * groups.  out: 32 counters; returns max << 8 | lowest counter that holds it.
uint32_t emul_planes_grouped(const uint32_t *rows, uint32_t n_rows, uint32_t groups, uint32_t *out) {
    constexpr int NP = 10;
    std::vector<std::array<uint32_t, NP>> part(groups);
    for (auto &p : part) p.fill(0);
    for (uint32_t u = 0; u + 8 * groups <= n_rows; u += 8 * groups)
        for (uint32_t g = 0; g < groups; g++) {
            const uint32_t *r = rows + u + 8 * g;
            uint32_t pl[NP];
            for (int p = 0; p < NP; p++) pl[p] = part[g][p];
            planes_ripple<NP, 3>(pl, planes_tree8<NP>(pl, r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]));
            for (int p = 0; p < NP; p++) part[g][p] = pl[p];
        }
    for (uint32_t d = 1; d < groups; d <<= 1)  // butterfly: every group ends with the total
        for (uint32_t g = 0; g < groups; g++)
            if (!(g & d)) {
                uint32_t a[NP], b[NP];
                for (int p = 0; p < NP; p++) { a[p] = part[g][p]; b[p] = part[g | d][p]; }
                planes_add<NP>(a, b);
                for (int p = 0; p < NP; p++) part[g][p] = part[g | d][p] = a[p];
            }
    uint32_t tot[NP];
    for (int p = 0; p < NP; p++) tot[p] = part[groups - 1][p];
    for (int b = 0; b < 32; b++) {
        uint32_t c = 0;
        for (int p = 0; p < NP; p++) c |= ((tot[p] >> b) & 1u) << p;
        out[b] = c;
    }
    uint32_t cand;
    const uint32_t m = planes_max<NP>(tot, cand);
    return (m << 8) | (uint32_t)__builtin_ctz(cand);
}

// rows: n_rows x 1 words (one 32-reference column).  n_rows must be a multiple of 8.
// out: 32 counters.
void emul_planes_count(const uint32_t *rows, uint32_t n_rows, int planes, uint32_t *out) {
    auto run = [&](auto tag) {
        constexpr int NP = decltype(tag)::value;
        uint32_t pl[NP];
        for (int p = 0; p < NP; p++) pl[p] = 0;
        uint32_t r = 0;
        // the kernel's schedule: 32 rows = four tree8 + two CSAs on plane 3 + one on plane 4 + ripple
        for (; r + 32 <= n_rows; r += 32) {
            auto t8 = [&](uint32_t o) {
                return planes_tree8<NP>(pl, rows[o], rows[o + 1], rows[o + 2], rows[o + 3], rows[o + 4], rows[o + 5],
                                        rows[o + 6], rows[o + 7]);
            };
            uint32_t c3a = t8(r), c3b = t8(r + 8), c4a, c4b, c5;
            csa(pl[3], c3a, c3b, pl[3], c4a);
            c3a = t8(r + 16);
            c3b = t8(r + 24);
            csa(pl[3], c3a, c3b, pl[3], c4b);
            csa(pl[4], c4a, c4b, pl[4], c5);
            planes_ripple<NP, 5>(pl, c5);
        }
        for (; r + 8 <= n_rows; r += 8)
            planes_add8<NP>(pl, rows[r], rows[r + 1], rows[r + 2], rows[r + 3], rows[r + 4], rows[r + 5], rows[r + 6],
                            rows[r + 7]);
        for (int g = 0; g < 8; g++) {
            uint32_t lo, hi;
            planes_unpack4<NP>(pl, g, lo, hi);
            for (int j = 0; j < 4; j++) out[4 * g + j] = ((lo >> (8 * j)) & 0xFF) | (((hi >> (8 * j)) & 0xFF) << 8);
        }
        // the eight-at-a-time unpack of the epilogue (bit-matrix transpose) must give the same counters
        for (int b = 0; b < 4; b++) {
            uint32_t lo0, hi0, lo1, hi1;
            planes_unpack8<NP>(pl, b, lo0, hi0, lo1, hi1);
            for (int j = 0; j < 4; j++) {
                const uint32_t c0 = ((lo0 >> (8 * j)) & 0xFF) | (((hi0 >> (8 * j)) & 0xFF) << 8);
                const uint32_t c1 = ((lo1 >> (8 * j)) & 0xFF) | (((hi1 >> (8 * j)) & 0xFF) << 8);
                if (c0 != out[8 * b + j] || c1 != out[8 * b + 4 + j]) out[8 * b + j] = 0xFFFFFFFFu;  // poison: the test fails
            }
        }
        // ... and so must the unpack of the whole word at once (delta swaps between the plane registers)
        uint32_t lo[4][2];
        planes_unpack32<NP>(pl, lo);
        for (int b = 0; b < 4; b++) {
            uint32_t hi0, hi1;
            planes_unpack8_hi<NP>(pl, b, hi0, hi1);
            for (int j = 0; j < 4; j++) {
                const uint32_t c0 = ((lo[b][0] >> (8 * j)) & 0xFF) | (((hi0 >> (8 * j)) & 0xFF) << 8);
                const uint32_t c1 = ((lo[b][1] >> (8 * j)) & 0xFF) | (((hi1 >> (8 * j)) & 0xFF) << 8);
                if (c0 != out[8 * b + j] || c1 != out[8 * b + 4 + j]) out[8 * b + j] = 0xFFFFFFFEu;
            }
        }
    };
    if (planes == 10) run(std::integral_constant<int, 10>{});
    else if (planes == 12) run(std::integral_constant<int, 12>{});
    else run(std::integral_constant<int, 16>{});
}

// Sequential emulation of prob_table_kernel (same lane grouping, pruning and arithmetic; only the
// order of the cross-lane sums differs).  hist has t+1 entries.  Returns 0, or 1 when the kernel
// would flag RTX_Q_NO_KMERS.  stats (may be null): [0] heavy group-steps, [1] light group-steps,
// [2] skipped groups, [3] i_lo.
int emul_prob_table(uint32_t t, const uint32_t *hist, uint64_t n_refs, const double *lf, double *table_z, double *z,
                    double *gs, uint64_t *stats) {
    const uint32_t n = t >> 1;
    if (t == 0) return 1;
    std::vector<uint32_t> ms;
    for (uint32_t m = 0; m <= t; m++)
        if (hist[m]) ms.push_back(m);
    const uint32_t D = (uint32_t)ms.size();
    std::vector<double> inv(t + n + 2, 0.0);
    for (uint32_t x = 1; x <= t + n; x++) inv[x] = 1.0 / (double)x;
    const double ln_total = ln_binom_tab(lf, t + n - 1, n);
    std::vector<double> tab(t + 1, 0.0);
    uint64_t st_heavy = 0, st_light = 0, st_skip = 0, st_ilo = 0;
    if (ms.back() == t) {
        for (uint32_t m : ms) tab[m] = only_last_pmf_tab(lf, t, n, m, ln_total);
    } else {
        if (n == 0) return 1;
        const uint32_t M = ms.back();
        uint32_t i_lo = 0;
        if (M > 0) {
            i_lo = n;
            for (uint32_t i = 0; i <= n; i++)
                if (ln_pmf_tab(lf, t, n, M, i, ln_total) >= kLnNegligibleP) { i_lo = i; break; }
        }
        st_ilo = i_lo;
        const uint32_t ngroups = (D + 63) / 64;  // lane groups (waves) in descending order of m
        std::vector<double> prod(n + 1, 1.0), Pi(n + 1, 0.0), base(ngroups, 1.0);
        std::vector<uint32_t> bw(ngroups, 0), istart(ngroups, 0);
        std::vector<uint8_t> skipped(ngroups, 0);
        for (uint32_t g = 0; g < ngroups; g++) {
            const uint32_t m_hi = ms[D - 1 - g * 64];
            if (m_hi == 0 || group_negligible(lf, t, n, m_hi, i_lo, ln_total)) {
                skipped[g] = 1;
                st_skip++;
                continue;
            }
            const uint32_t nl = std::min<uint32_t>(64, D - g * 64);
            // start index of the group: first i at which its smallest count's pmf reaches e^-100
            uint32_t m_lo = ms[D - 1 - (g * 64 + nl - 1)];
            if (m_lo == 0) m_lo = nl > 1 ? ms[D - 1 - (g * 64 + nl - 2)] : m_hi;
            uint32_t i_s = i_lo;
            for (uint32_t i = 0; i < i_lo; i++)
                if (ln_pmf_tab(lf, t, n, m_lo, i, ln_total) >= kLnNegligibleP) { i_s = i; break; }
            istart[g] = i_s;
            std::vector<PmfState> st(nl);
            std::vector<uint32_t> h(nl, 0);
            std::vector<uint8_t> act(nl, 0), sat(nl, 0);
            for (uint32_t l = 0; l < nl; l++) {
                const uint32_t m = ms[D - 1 - (g * 64 + l)];
                act[l] = m != 0;
                if (act[l]) { st[l] = pmf_start_at(lf, t, n, m, i_s, ln_total); h[l] = hist[m]; }
            }
            bw[g] = n + 1;
            for (uint32_t i = i_s; i <= n; i++) {
                bool all_sat = i > i_s;
                for (uint32_t l = 0; l < nl; l++) {
                    if (!act[l]) continue;
                    const uint32_t m = ms[D - 1 - (g * 64 + l)];
                    if (i > i_s) {
                        const double c_old = st[l].c;
                        const int k_old = st[l].k;
                        pmf_step(st[l], inv.data(), t, n, m, i);
                        sat[l] = st[l].c == c_old && k_old == 0 && st[l].k == 0;
                    }
                    all_sat = all_sat && sat[l];
                }
                if (all_sat) {
                    double b = 1.0;
                    for (uint32_t l = 0; l < nl; l++)
                        if (act[l]) b *= pmf_cmf_pow(st[l], h[l]);
                    base[g] = b;
                    bw[g] = i;
                    break;
                }
                if (i >= i_lo) {
                    double f = 1.0;
                    for (uint32_t l = 0; l < nl; l++)
                        if (act[l]) f *= pmf_cmf_pow(st[l], h[l]);
                    prod[i] *= f;
                    st_heavy++;
                } else {
                    st_light++;
                }
            }
        }
        for (uint32_t i = i_lo; i <= n; i++) {
            double p = prod[i];
            for (uint32_t g = 0; g < ngroups; g++)
                if (!skipped[g] && i >= bw[g]) p *= base[g];
            Pi[i] = p;
        }
        for (uint32_t g = 0; g < ngroups; g++) {
            const uint32_t nl = std::min<uint32_t>(64, D - g * 64);
            for (uint32_t l = 0; l < nl; l++) {
                const uint32_t m = ms[D - 1 - (g * 64 + l)];
                if (m == 0) { tab[0] = Pi[0]; continue; }
                if (skipped[g]) { tab[m] = 0.0; continue; }
                const uint32_t i_s = istart[g];
                PmfState st = pmf_start_at(lf, t, n, m, i_s, ln_total);
                double acc = 0.0;
                const uint32_t last = std::min(n, bw[g] == 0 ? 0 : bw[g] - 1);
                for (uint32_t i = i_s; i <= last; i++) {
                    if (i > i_s) pmf_step(st, inv.data(), t, n, m, i);
                    if (i < i_lo) continue;
                    const double P = Pi[i];
                    if (P > 0.0 && st.k == 0 && st.c > 0.0) acc += st.v * P / st.c;
                }
                tab[m] = acc;
            }
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
    if (stats) { stats[0] = st_heavy; stats[1] = st_light; stats[2] = st_skip; stats[3] = st_ilo; }
    return 0;
}


// Sequential emulation of prob_lookup_kernel + prob_tables_build_kernel (rtx_prob_tables.hip): the rows
// C = ln cmf, R = pmf/cmf, sat and ilo are produced by the same recurrence (here on demand instead of from the
// memoised table), then P(i) and table[m] are formed exactly as the lookup kernel does.
// u_thr > 0: the query as tile pruning treats it (prob_lookup_kernel with ProbParams::prune_thr / prune_i1): the counts up to u_thr
// become references without a hit (bin 0) with probability 0, and the sums over i start at i1 if that lies beyond i_lo.
static int emul_prob_lookup_x(uint32_t t, const uint32_t *hist_in, uint64_t n_refs, const double *lf, uint32_t u_thr, uint32_t i1,
                              double *table_z, double *z, double *gs, uint64_t *stats) {
    const uint32_t n = t >> 1;
    if (t == 0) return 1;
    std::vector<uint32_t> hist_v(hist_in, hist_in + t + 1);
    for (uint32_t m = 1; m <= u_thr && m <= t; m++) { hist_v[0] += hist_v[m]; hist_v[m] = 0; }
    const uint32_t *hist = hist_v.data();
    std::vector<uint32_t> ms;
    for (uint32_t m = 0; m <= t; m++)
        if (hist[m]) ms.push_back(m);
    std::vector<double> inv(t + n + 2, 0.0);
    for (uint32_t x = 1; x <= t + n; x++) inv[x] = 1.0 / (double)x;
    const double ln_total = ln_binom_tab(lf, t + n - 1, n);
    std::vector<double> tab(t + 1, 0.0);
    uint64_t st_rows = 0, st_points = 0;
    auto build_row = [&](uint32_t m, std::vector<double> &C, std::vector<double> &R, uint32_t &sat, uint32_t &ilo) {
        C.assign(n + 1, 0.0);
        R.assign(n + 1, 0.0);
        PmfState st = pmf_start(lf, t, n, m, ln_total);
        sat = n + 1;
        ilo = n;
        bool found = false;
        for (uint32_t i = 0; i <= n; i++) {
            if (i > 0) {
                const double c_old = st.c;
                const int k_old = st.k;
                pmf_step(st, inv.data(), t, n, m, i);
                if (sat == n + 1 && st.c == c_old && k_old == 0 && st.k == 0) sat = i;
            }
            const bool live = st.k == 0 && st.c > 0.0;
            C[i] = live ? log(st.c) : -INFINITY;
            R[i] = live ? st.v / st.c : 0.0;
            if (!found && ln_pmf_tab(lf, t, n, m, i, ln_total) >= kLnNegligibleP) { ilo = i; found = true; }
        }
    };
    if (ms.back() == t) {
        for (uint32_t m : ms) tab[m] = only_last_pmf_tab(lf, t, n, m, ln_total);
    } else {
        if (n == 0) return 1;
        const uint32_t M = ms.back();
        std::vector<double> C, R;
        uint32_t sat = 0, ilo = 0, i_lo = 0;
        if (M > 0) { build_row(M, C, R, sat, ilo); i_lo = ilo; }
        if (u_thr && i1 > i_lo && i1 <= n) i_lo = i1;
        std::vector<double> Pi(n + 1, 0.0);  // sum_m hist[m] ln cmf_m(i), then exp
        struct Row { uint32_t m, sat; std::vector<double> R; };
        std::vector<Row> rows;
        for (size_t j = ms.size(); j-- > 0;) {
            const uint32_t m = ms[j];
            if (m == 0) continue;
            build_row(m, C, R, sat, ilo);
            if (sat <= i_lo) continue;  // saturated before i_lo: factor 1, table 0
            for (uint32_t i = i_lo; i <= n; i++)
                if (i < sat) { Pi[i] = fma((double)hist[m], C[i], Pi[i]); st_points++; }
            rows.push_back(Row{m, sat, R});
            st_rows++;
        }
        for (uint32_t i = 0; i <= n; i++) Pi[i] = exp(Pi[i]);
        for (const Row &r : rows) {
            const uint32_t last = std::min(n, r.sat - 1);
            double acc = 0.0;
            for (uint32_t i = i_lo; i <= last; i++) acc += r.R[i] * Pi[i];
            tab[r.m] = acc;
        }
        if (ms[0] == 0) tab[0] = i_lo == 0 && u_thr == 0 ? Pi[0] : 0.0;
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
    if (stats) { stats[0] = st_rows; stats[1] = st_points; stats[2] = 0; stats[3] = 0; }
    return 0;
}
int emul_prob_lookup(uint32_t t, const uint32_t *hist, uint64_t n_refs, const double *lf, double *table_z, double *z,
                     double *gs, uint64_t *stats) {
    return emul_prob_lookup_x(t, hist, n_refs, lf, 0, 0, table_z, z, gs, stats);
}
int emul_prob_lookup_pruned(uint32_t t, const uint32_t *hist, uint64_t n_refs, const double *lf, uint32_t u_thr, uint32_t i1,
                            double *table_z, double *z, double *gs) {
    return emul_prob_lookup_x(t, hist, n_refs, lf, u_thr, i1, table_z, z, gs, nullptr);
}

// ln cmf_m(i), i = 0 .. n, exactly as prob_tables_build_kernel stores it (the rows prune_kernel reads for G)
static void emul_ln_cmf_row(uint32_t t, uint32_t m, const double *lf, std::vector<double> &C) {
    const uint32_t n = t >> 1;
    C.assign(n + 1, 0.0);
    if (m == 0) return;
    std::vector<double> inv(t + n + 2, 0.0);
    for (uint32_t x = 1; x <= t + n; x++) inv[x] = 1.0 / (double)x;
    const double ln_total = ln_binom_tab(lf, t + n - 1, n);
    PmfState st = pmf_start(lf, t, n, m, ln_total);
    for (uint32_t i = 0; i <= n; i++) {
        if (i > 0) pmf_step(st, inv.data(), t, n, m, i);
        C[i] = st.k == 0 && st.c > 0.0 ? log(st.c) : -INFINITY;
    }
}

// Sequential restatement of step 3 of prune_kernel (rtx_prune.hip): the threshold u (the largest count a reference may have
// and still be treated as a reference without a hit) and i* + 1 from the exact counts `hm` of the 64 references of the block
// with the largest bound (0: no reference / zeroed).  Same inequalities, same tables; only the order of the sums over the
// lanes differs.  The GPU tests hold the kernel's (u, i* + 1) against this, the CPU tests hold this against the oracle on
// adversarial histograms (tests/test_prune_threshold_cpu.py).
// n_groups = 0: the criterion "(3)" -- every reference of the database priced at the candidate threshold (reference shards: every
// shard must arrive at the same threshold, and a shard knows only its own tiles).  n_groups > 0: the tile-aware criterion "(4)" of a
// whole-database handle: gub[g] = the largest bound of group g of tiles (every count of its gn[g] references is at most that), the
// dropped references are priced at min(u, gub), the kept ones number at most the references of the groups with gub > u.
static void emul_prune_threshold_x(uint32_t t, uint64_t n_refs, const uint32_t *hm_in, const double *lf, uint32_t tab_tmax, uint32_t n_groups,
                                   const uint32_t *gub, const uint64_t *gn, uint32_t *u_out, uint32_t *i1_out) {
    const double kLnEps = kPruneLnEpsHD;
    const uint32_t n = t >> 1;
    uint32_t hm[64], M = 0;
    for (int l = 0; l < 64; l++) { hm[l] = hm_in[l]; M = std::max(M, hm[l]); }
    *u_out = 0;
    *i1_out = 0;
    if (!(t >= 16u && t <= tab_tmax && M >= 1u && n >= 2u)) return;
    const double ln_n = log((double)n_refs), ln_total = ln_binom_tab(lf, t + n - 1, n);
    if (M >= t) {
        uint32_t u_max = 0;
        for (uint32_t u = 1; u < t; u++)
            if (ln_binom_tab(lf, u + n - 1, n) - ln_total + ln_n <= kLnEps) u_max = std::max(u_max, u);
        *u_out = u_max;
        return;
    }
    uint32_t n_h = 0, h_min = 0xFFFFFFFFu;
    std::vector<std::vector<double>> rows(64);
    for (int l = 0; l < 64; l++) {
        if (hm[l] * 5u < M * 4u) hm[l] = 0;
        if (hm[l]) { n_h++; h_min = std::min(h_min, hm[l]); emul_ln_cmf_row(t, hm[l], lf, rows[l]); }
    }
    auto passes = [&](uint32_t i) {
        double s = 0.0;
        for (int l = 0; l < 64; l++)
            if (hm[l]) s += rows[l][i];
        return s + log((double)n_h + (double)n_refs * (double)(i + 1u)) <= kLnEps;
    };
    if (!passes(0u)) return;
    uint32_t lo = 0, hi = n - 2u;
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1u) >> 1;
        if (passes(mid)) lo = mid; else hi = mid - 1u;
    }
    const uint32_t i1 = lo + 1u;
    const double ln_len = log((double)(n - i1 + 1u));
    uint32_t first_fail = h_min;
    for (uint32_t u = 1; u < h_min; u++) {
        const double up = (double)(u + i1) * (double)(n - i1), dn = (double)(i1 + 1u) * (double)(t - u + n - i1 - 1u);
        if (!(up < dn && ln_len + ln_pmf_tab(lf, t, n, u, i1, ln_total) + ln_n <= kLnEps)) { first_fail = u; break; }
    }
    uint32_t u_max = first_fail - 1u;
    {   // the tighter criteria ("(3)" and "(4)" in rtx_prune.hip): window W = i1 .. i1 + 62 (lanes 0 .. 62), lane 63 = the tail point i1 + 63
        // (4): the kept references number at most those of the groups whose bound lies above the threshold of (2) -- every later candidate is larger
        double kept = (double)n_refs;
        if (n_groups) {
            kept = 0.0;
            for (uint32_t g = 0; g < n_groups; g++)
                if (gub[g] > u_max) kept += (double)gn[g];
        }
        const double ln_kept = log(kept > 1.0 ? kept : 1.0);
        double WA[64], WB[64], ww[64], gw[64];
        for (uint32_t l = 0; l < 64; l++) {
            const uint32_t iw = i1 + l;
            gw[l] = ww[l] = 0.0;
            if (iw > n || l == 63) continue;
            double lnG = 0.0;
            for (int h = 0; h < 64; h++)
                if (hm[h]) lnG += rows[h][iw];
            gw[l] = exp(lnG);
            ww[l] = exp(std::min(0.0, ln_kept + lnG));
        }
        double incl = 0.0, a_tot = 0.0;
        for (uint32_t l = 0; l < 64; l++) a_tot += ww[l];
        for (uint32_t l = 0; l < 64; l++) {
            const uint32_t iw = i1 + l;
            const bool vi = iw <= n;
            incl += ww[l];
            const double len_tail = vi ? (double)(n - iw + 1u) : 0.0;
            WA[l] = l == 63 ? len_tail * (a_tot + 1.0) : (vi ? incl - ww[l] : 0.0);
            WB[l] = l == 63 ? len_tail : gw[l];
        }
        const double nn = (double)n_refs;
        const bool has_tail = i1 + 63u <= n;
        auto sums_at = [&](uint32_t m, double &a, double &b) {
            a = b = 0.0;
            for (uint32_t l = 0; l < 64; l++) {
                const double P = i1 + l <= n && m != 0u ? exp(ln_pmf_tab(lf, t, n, m, i1 + l, ln_total)) : 0.0;
                a += P * WA[l];
                b += P * WB[l];
            }
        };
        // (4): S_A, S_B of the groups at their own bounds (capped below min H: a group above it is never dead); the groups far below the
        // threshold of (2) together at S(u_(2) - kPruneFarGap) unless too many lie near (prune_kernel: same rule)
        std::vector<double> gsa(n_groups, 0.0), gsb(n_groups, 0.0);
        std::vector<uint8_t> near(n_groups, 0), mid(n_groups, 0);
        double far_a = 0.0, far_b = 0.0, n_far = 0.0, mid_a = 0.0, mid_b = 0.0, n_mid = 0.0;
        if (n_groups) {
            const uint32_t m_far = u_max > kPruneFarGap ? u_max - kPruneFarGap : 0u;
            uint32_t n_near = 0;
            // near: above the threshold of (2), a window sum of their own; mid: within kPruneFarGap below it, together at S(u_(2))
            for (uint32_t g = 0; g < n_groups; g++) {
                near[g] = gn[g] > 0 && gub[g] > u_max;
                mid[g] = gn[g] > 0 && gub[g] > m_far && gub[g] <= u_max;
                n_near += near[g];
            }
            if (n_near > kPruneMaxNear) {
                std::vector<double> inv(t + n + 2, 0.0);
                for (uint32_t x = 1; x <= t + n; x++) inv[x] = 1.0 / (double)x;
                for (uint32_t g = 0; g < n_groups; g++) {
                    near[g] = gn[g] > 0;
                    prune_window_sums(lf, inv.data(), t, n, std::min(gub[g], h_min - 1u), i1, ln_total, [&](uint32_t l) { return WA[l]; },
                                      [&](uint32_t l) { return WB[l]; }, gsa[g], gsb[g]);
                }
            } else {
                for (uint32_t g = 0; g < n_groups; g++) {
                    if (mid[g]) n_mid += (double)gn[g];
                    else if (!near[g]) n_far += (double)gn[g];
                }
                if (m_far) sums_at(m_far, far_a, far_b);
                if (n_mid > 0.0) sums_at(u_max, mid_a, mid_b);
                for (uint32_t g = 0; g < n_groups; g++)
                    if (near[g]) sums_at(std::min(gub[g], h_min - 1u), gsa[g], gsb[g]);
            }
        }
        auto crit = [&](uint32_t u) {
            double a, b;
            sums_at(u, a, b);
            bool falling = true;
            if (has_tail) {
                const uint32_t j = i1 + 63u;
                falling = (double)(u + j) * (double)(n - j) < (double)(j + 1u) * (double)(t - u + n - j - 1u);
            }
            if (!n_groups) return falling && nn * a <= kPruneHalfEpsHD && nn * b <= kPruneHalfEpsHD;
            double ta = n_far * far_a + n_mid * mid_a, tb = n_far * far_b + n_mid * mid_b;
            for (uint32_t g = 0; g < n_groups; g++) {
                if (!near[g]) continue;
                const bool dead = gub[g] <= u;
                ta += (double)gn[g] * (dead ? gsa[g] : a);
                tb += (double)gn[g] * (dead ? gsb[g] : b);
            }
            return falling && ta <= kPruneHalfEpsHD && tb <= kPruneHalfEpsHD;
        };
        uint32_t lo2 = u_max, hi2 = h_min - 1u;
        while (lo2 < hi2) {
            const uint32_t mid = (lo2 + hi2 + 1u) >> 1;
            if (crit(mid)) lo2 = mid; else hi2 = mid - 1u;
        }
        u_max = lo2;
    }
    *u_out = u_max;
    *i1_out = u_max ? i1 : 0u;
}

void emul_prune_threshold(uint32_t t, uint64_t n_refs, const uint32_t *hm_in, const double *lf, uint32_t tab_tmax, uint32_t *u_out,
                          uint32_t *i1_out) {
    emul_prune_threshold_x(t, n_refs, hm_in, lf, tab_tmax, 0, nullptr, nullptr, u_out, i1_out);
}
// The tile-aware criterion from the largest bound of every TILE of 8192 references (what the bounds pass leaves): the tiles are taken in
// groups of ceil(ntiles / 64) consecutive ones, as the 64 lanes of prune_kernel take them.
void emul_prune_threshold_tiles(uint32_t t, uint64_t n_refs, const uint32_t *hm_in, const double *lf, uint32_t tab_tmax, uint32_t ntiles,
                                const uint16_t *tile_ub, uint32_t *u_out, uint32_t *i1_out) {
    const uint32_t per = (ntiles + 63u) / 64u, ng = (ntiles + per - 1u) / per;
    std::vector<uint32_t> gub(ng, 0);
    std::vector<uint64_t> gn(ng, 0);
    for (uint32_t T = 0; T < ntiles; T++) {
        const uint32_t g = T / per;
        gub[g] = std::max<uint32_t>(gub[g], tile_ub[T]);
        const uint64_t lo = (uint64_t)T * 8192u, hi = std::min<uint64_t>(lo + 8192u, n_refs);
        gn[g] += hi > lo ? hi - lo : 0u;
    }
    emul_prune_threshold_x(t, n_refs, hm_in, lf, tab_tmax, ng, gub.data(), gn.data(), u_out, i1_out);
}

// Bit layout of the bitmap rows (rtx_math.hpp ref_slot) and its inverse as hit_count / seg_emit use it:
// word, bit of local reference r; and the reference of (tile, lane, group g, j) = the j-th count of a lane's g-th store.
void emul_ref_slot(uint32_t r, uint32_t stride_bytes, uint32_t *word, uint32_t *bit) { ref_slot(r, stride_bytes, *word, *bit); }
uint32_t emul_slot_ref(uint32_t tile, uint32_t lane, uint32_t g, uint32_t j, uint32_t stride_bytes) {
    return tile * 8192u + (g * tile_lanes(stride_bytes, tile) + lane) * 8u + j;
}

// The SWAR merge of hit_count's epilogue: eight byte counters (two dwords) added to eight u16 counts (four dwords).
void emul_merge_bytes(const uint32_t *sb, uint32_t *st) {
    st[0] += (sb[0] & 0xFFu) | ((sb[0] & 0xFF00u) << 8);
    st[1] += ((sb[0] >> 16) & 0xFFu) | ((sb[0] >> 24) << 16);
    st[2] += (sb[1] & 0xFFu) | ((sb[1] & 0xFF00u) << 8);
    st[3] += ((sb[1] >> 16) & 0xFFu) | ((sb[1] >> 24) << 16);
}

// One lane's view of the six butterfly stages of transpose64 (rtx_kernels.hip): x[64] in, columns out.
void emul_transpose64(const unsigned long long *in, unsigned long long *out) {
    unsigned long long x[64], y[64];
    for (int l = 0; l < 64; l++) x[l] = in[l];
    for (int s = 32; s >= 1; s >>= 1) {
        const unsigned long long m = s == 32 ? 0x00000000FFFFFFFFull : s == 16 ? 0x0000FFFF0000FFFFull : s == 8 ? 0x00FF00FF00FF00FFull
                                   : s == 4 ? 0x0F0F0F0F0F0F0F0Full : s == 2 ? 0x3333333333333333ull : 0x5555555555555555ull;
        for (int l = 0; l < 64; l++) y[l] = x[l ^ s];
        for (int l = 0; l < 64; l++) x[l] = (l & s) ? (((y[l] >> s) & m) | (x[l] & ~m)) : ((x[l] & m) | ((y[l] & m) << s));
    }
    for (int l = 0; l < 64; l++) out[l] = x[l];
}

// The finalisation of one query's result rows as finalise_kernel (rtx_finalise.hip) does it: for every row the number of rows that come
// before it in the order of lineage.rs:91-93 (fin_row_before: the rank count of the kernel's lanes) and its local signal
// (fin_local_signal).  k: [n][D] hundredths, depth: [n], size: [n][D] references below the row's ancestor per level.
// Returns the number of row pairs on which the word-wise compare of the kernel (fin_row_before_words over zero-padded big-endian words)
// and the byte-wise statement of lineage.rs:91-93 (fin_row_before) disagree: 0.
uint32_t emul_finalise_rows(uint32_t n, uint32_t D, const uint8_t *k, const uint32_t *depth, const uint32_t *size, double n_total, uint32_t *rank,
                            double *local) {
    const uint32_t kw = (D + 3u) / 4u;
    std::vector<uint32_t> words((size_t)n * kw, 0u);
    for (uint32_t r = 0; r < n; r++)
        for (uint32_t w = 0; w < kw; w++) {
            uint32_t v = 0;  // the word as the device loads it (little-endian bytes of the DevRow, zero beyond the depth)
            for (uint32_t b = 0; b < 4u; b++) {
                const uint32_t d = 4u * w + b;
                v |= (uint32_t)(d < depth[r] && d < D ? k[(size_t)r * D + d] : 0u) << (8u * b);
            }
            words[(size_t)r * kw + w] = fin_be32(v);
        }
    uint32_t disagree = 0;
    for (uint32_t r = 0; r < n; r++) {
        uint32_t before = 0;
        for (uint32_t x = 0; x < n; x++) {
            const bool a = x != r && fin_row_before(k + (size_t)x * D, depth[x], x, k + (size_t)r * D, depth[r], r);
            const bool b = x != r && fin_row_before_words(&words[(size_t)x * kw], depth[x], x, &words[(size_t)r * kw], depth[r], r, kw);
            disagree += a != b;
            before += b ? 1u : 0u;
        }
        rank[r] = before;
        double eb[64];
        const uint32_t s0 = fin_node_expected(size + (size_t)r * D, depth[r], n_total, eb);  // (per node in the library: node_tables)
        const uint8_t *kr = k + (size_t)r * D;
        local[r] = fin_local_signal([&](uint32_t d) { return kr[d]; }, eb, s0, depth[r]);
    }
    return disagree;
}

}  // extern "C"
