// prob_table through memoised cmf tables (src/prob.rs:8-103).
//
// pmf_m(i) and cmf_m(i) of prob.rs:121-170 depend on (t, n = t/2, m, i) only -- not on the query
// beyond its number of distinct k-mers t.  For queries of t <= 2047 (barcodes; since round 6 full-length 16S) the library
// therefore builds, once per index handle, for every t <= tmax and every 0 < m < t
//     L[t][m][i] = ln cmf_m(i)           R[t][m][i] = pmf_m(i) / cmf_m(i)
// with the very recurrence of rtx_math.hpp (same arithmetic as prob_table_kernel), plus per (t, m)
//     ilo[t][m] = first i with ln pmf_m(i) >= -100          (i_lo when m is the largest count)
//     sat[t][m] = first i at which cmf_m stops changing      (cmf == its final value from there on)
// A query then needs no recurrence at all:
//     P(i)     = exp(sum_m hist[m] L[t][m][i])      (lanes <-> i, rows streamed coalesced; this is
//                                                    prod[i] of prob.rs:62-73, one FMA per row and i)
//     table[m] = sum_i R[t][m][i] * P(i)            (one wave reduction per distinct count)
// restricted to i >= i_lo and to the rows that are not yet saturated at i_lo (a saturated row
// contributes the constant factor cmf_final^h = 1 + O(h 1e-16), identical for every i >= i_lo,
// which cancels in the normalisation by Z).  Longer queries use prob_table_kernel.
#include <hip/hip_runtime.h>

#include "rtx_kernels.hpp"
#include "rtx_math.hpp"
#include "rtx_wave.hpp"

namespace rtx {

// ---------------------------------------------------------------------------
// table build: one thread per (t, m), sequential over i (same recurrence as prob_table_kernel)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prob_tables_build_kernel(ProbTables tb, const double *__restrict__ lf,
                                                                const double *__restrict__ inv) {
    const uint32_t t = blockIdx.y + 2;  // t = 2 .. tmax
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > tb.tmax || m >= t) return;
    const uint32_t n = t >> 1, n1 = n + 1;
    double *C = tb.cmf + tb.off[t] + (size_t)m * n1;
    double *R = tb.ratio + tb.off[t] + (size_t)m * n1;
    uint16_t *meta_ilo = tb.ilo + tb.moff[t];
    uint16_t *meta_sat = tb.sat + tb.moff[t];
    if (m == 0) {  // pmf = [1, 0, ...], cmf = 1, ln cmf = 0 (prob.rs:134-137)
        for (uint32_t i = 0; i <= n; i++) { C[i] = 0.0; R[i] = i == 0 ? 1.0 : 0.0; }
        meta_ilo[0] = 0;
        meta_sat[0] = 1;
        return;
    }
    const double ln_total = ln_binom_tab(lf, t + n - 1, n);
    PmfState st = pmf_start(lf, t, n, m, ln_total);
    uint32_t sat = n + 1, ilo = n;
    bool ilo_found = false;
    for (uint32_t i = 0; i <= n; i++) {
        if (i > 0) {
            const double c_old = st.c;
            const int k_old = st.k;
            pmf_step(st, inv, t, n, m, i);
            if (sat == n + 1 && st.c == c_old && k_old == 0 && st.k == 0) sat = i;
        }
        const bool live = st.k == 0 && st.c > 0.0;  // below 2^-412 the cmf counts as 0 (rtx_math.hpp)
        C[i] = live ? log(st.c) : -INFINITY;
        R[i] = live ? st.v / st.c : 0.0;
        if (!ilo_found && ln_pmf_tab(lf, t, n, m, i, ln_total) >= kLnNegligibleP) { ilo = i; ilo_found = true; }
    }
    meta_ilo[m] = (uint16_t)ilo;
    meta_sat[m] = (uint16_t)sat;
}

// One f64 of a table row through a raw buffer load.  The descriptor covers the table of this t from its base up to
// the END of the wanted part of the row (`end` bytes), the row start goes into the SGPR offset and the only vector
// operand is the byte offset inside the row.  On gfx950 the bounds check of a raw buffer includes the SGPR offset
// (tools/micro/soffset_bounds.hip), so offsets at or beyond end - off read 0.0: per row the descriptor changes in
// one dword (num_records) -- no scalar or vector address arithmetic at all.
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double row_load_f64(const double *table, uint32_t off, uint32_t end, uint32_t voff) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(table), 0, end, 0x00027000);
    const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, off, 0);
    return __hiloint2double((int)v.y, (int)v.x);
}

// ---------------------------------------------------------------------------
// per-query lookup kernel: 256 threads (4 waves) per query
// LDS (dynamic): Pi[n1max] f64 | red[16] f64 | Ppart[64] f64 | tzl[tmax+2] f64 | row_h[tmax+18] u32 | row_m, row_sat, ms [tmax+18] u16 each | hl[tmax+4] u32 | row_off, row_end [tmax+18] u32
// ---------------------------------------------------------------------------
#ifndef RTX_PROB_WAVES
#define RTX_PROB_WAVES 6  // waves per SIMD (<= 80 VGPRs, no spills; the kernel is a chain of round trips: 10.9 / 10.2 / 9.1 / 9.1 ms per 1 M queries at 4 / 5 / 6 / 7 -- unbounded it takes 100 VGPRs and runs four)
#endif
__global__ __launch_bounds__(256, RTX_PROB_WAVES) void prob_lookup_kernel(ProbParams p, ProbTables tb) {
    extern __shared__ double smem[];
    __shared__ uint32_t s_D, s_nact;
    __shared__ uint32_t s_nlive[16];  // per 64-wide slice of i (n <= 1023: sixteen at most): rows [0, s_nlive) include every row not yet saturated there
    const uint32_t q = p.order ? p.order[blockIdx.x] : blockIdx.x;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint64_t gq = p.q0 + q;
    const uint32_t t = p.t[q];
    const uint32_t n = t >> 1;  // num_trials = k_mers.len() / 2, raxtax.rs:57
    const uint32_t n1 = n + 1;
    double *Pi = smem;
    double *red = Pi + p.n1max;
    double *Ppart = red + 16;  // [64] second half of the row sum of the first slice (pass 1)
    double *tzl = Ppart + 64;  // [tmax+2] table[m] until it has been divided by Z: written by several waves, summed by all -- in LDS, not
                               // through global memory (the sum and the division each waited for a store-to-load round trip through L2)
    uint32_t *row_h = reinterpret_cast<uint32_t *>(tzl + (p.tmax + 2));
    uint16_t *row_m = reinterpret_cast<uint16_t *>(row_h + (p.tmax + 18));
    uint16_t *row_sat = row_m + (p.tmax + 18);
    uint16_t *ms = row_sat + (p.tmax + 18);
    uint32_t *hl = reinterpret_cast<uint32_t *>(ms + ((p.tmax + 18 + 1) & ~1u));  // [tmax+2] histogram copy
    uint32_t *row_off = hl + (p.tmax + 4);      // [tmax+18] byte offset of row m in the tables of this t
    uint32_t *row_end = row_off + (p.tmax + 18);  // [tmax+18] byte offset of the end of its part below saturation
    const uint32_t *hist = p.hist + (size_t)q * p.hstride;
    double *tz = p.table_z + (size_t)q * p.hstride;
    const double *lf = p.lnfact;

    {   // the histogram row, as far as a count can reach in this batch -- not "up to t": the loads then leave together with the load of t
        // (kmer_extract zeroes the whole row; bins above t stay 0)
        const uint32_t hlim = p.hstride < p.tmax + 4u ? p.hstride : p.tmax + 4u;
        for (uint32_t m = tid; m < hlim; m += 256) hl[m] = hist[m];
    }
    if (t == 0) {  // reference: u64 underflow at prob.rs:21
        if (tid == 0) { p.status[gq] = RTX_Q_NO_KMERS; p.z[gq] = 0.0; p.gs[gq] = 0.0; p.ndist[gq] = 0; }
        return;
    }
    __syncthreads();
    // Tile pruning gave the query a threshold u: the references with a count up to u hold less than eps = 1e-10 of probability
    // together and move no product by more than that (rtx_prune.hip).  They all become references without a hit (cmf = 1,
    // probability 0) -- those of the tiles that were not counted already sit in bin 0 -- so that the result does not depend on
    // which of them happened to be counted (a tile is counted if EITHER query of its pair needs it).
    const uint32_t u_thr = p.prune_thr ? p.prune_thr[q] : 0u;
    if (u_thr) {  // workgroup-uniform
        uint32_t low = 0;
        for (uint32_t m = 1u + tid; m <= u_thr && m <= t; m += 256) {
            low += hl[m];
            hl[m] = 0;
            tz[m] = 0.0;  // taxon_prefix looks these up for the references that were counted
        }
        if (low) atomicAdd(&hl[0], low);
        __syncthreads();
    }
    if (wave == 0) {  // distinct counts, ascending
        uint32_t D = 0;
        for (uint32_t m0 = 0; m0 <= t; m0 += 64) {
            const uint32_t m = m0 + lane;
            const bool has = m <= t && hl[m] != 0;
            const unsigned long long bal = __ballot(has);
            if (has) ms[D + __popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)m;
            D += (uint32_t)__popcll(bal);
        }
        if (lane == 0) { s_D = D; p.ndist[gq] = D; }
    }
    __syncthreads();
    const uint32_t D = s_D;
    const uint32_t M = ms[D - 1];
    const double ln_total = ln_binom_tab(lf, t + n - 1, n);  // prob.rs:20-23
    if (M == t) {  // prob.rs:24-41
        for (uint32_t j = tid; j < D; j += 256) {
            const uint32_t m = ms[j];
            tzl[m] = only_last_pmf_tab(lf, t, n, m, ln_total);
        }
    } else {
        if (n == 0) {  // reference: zip_eq length mismatch at prob.rs:162
            if (tid == 0) { p.status[gq] = RTX_Q_NO_KMERS; p.z[gq] = 0.0; p.gs[gq] = 0.0; }
            return;
        }
        const uint16_t *t_ilo = tb.ilo + tb.moff[t];
        const uint16_t *t_sat = tb.sat + tb.moff[t];
        const double *Ct = tb.cmf + tb.off[t];
        const double *Rt = tb.ratio + tb.off[t];
        uint32_t i_lo = M > 0 ? t_ilo[M] : 0u;
        // a pruned query: everything at i <= i* holds less than eps = 1e-10 of Z (rtx_prune.hip, step 2 of the threshold) -- the sums
        // start behind it: one slice of i instead of three on the bench workload, and fewer rows still moving there
        if (u_thr && p.prune_i1) { const uint32_t i1 = p.prune_i1[q]; i_lo = i1 > i_lo && i1 <= n ? i1 : i_lo; }
        // saturation index of every distinct count, gathered by all threads at once (row_h as staging)
        for (uint32_t j = tid; j < D; j += 256) {
            const uint32_t m = ms[D - 1 - j];
            row_h[j] = m ? t_sat[m] : 0u;
        }
        __syncthreads();
        // rows still moving at i_lo, in descending order of m; everything else gets table = 0
        if (wave == 0) {
            uint32_t na = 0;
            if (lane < 16) s_nlive[lane] = 0;
            for (uint32_t j0 = 0; j0 < D; j0 += 64) {
                const uint32_t j = j0 + lane;
                uint32_t m = 0, sat = 0;
                if (j < D) {
                    m = ms[D - 1 - j];
                    sat = row_h[j];
                    if (m && sat <= i_lo) tzl[m] = 0.0;
                }
                const bool keep = m != 0 && sat > i_lo;
                const unsigned long long bal = __ballot(keep);
                if (keep) {
                    const uint32_t pos = na + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                    row_m[pos] = (uint16_t)m;
                    row_sat[pos] = (uint16_t)sat;
                    row_h[pos] = hl[m];  // pos <= j: the staged entries still needed lie at indices > j
                    row_off[pos] = m * n1 * 8u;
                    row_end[pos] = (m * n1 + (sat < n1 ? sat : n1)) * 8u;
                    // last slice of 64 values of i (from i_lo) in which this row still moves
                    const uint32_t nsl = ((sat < n1 ? sat : n1) - i_lo + 63u) >> 6;
                    atomicMax(&s_nlive[(nsl < 16u ? nsl : 16u) - 1u], pos + 1u);
                }
                na += (uint32_t)__popcll(bal);
            }
            // pad to a multiple of 16 with neutral rows (m = 0: ln cmf = 0, sat = 0, h = 0)
            const uint32_t padded = (na + 15u) & ~15u;
            if (na + lane < padded) { row_m[na + lane] = 0; row_sat[na + lane] = 0; row_h[na + lane] = 0; row_off[na + lane] = 0; row_end[na + lane] = 0; }
            if (lane == 0) {
                s_nact = na;
                for (int sl = 14; sl >= 0; sl--) s_nlive[sl] = max(s_nlive[sl], s_nlive[sl + 1]);  // live in slice s' => live in every s <= s'
            }
        }
        __syncthreads();
        const uint32_t nact = s_nact;
        // ---- pass 1: P(i) = exp(sum_m hist[m] ln cmf_m(i))  (prob.rs:62-73), lanes <-> i.  Rows are taken
        // sixteen at a time: sixteen independent 512-byte row-slice loads in flight per wave, one FMA each.
        // Rows are read through row_load_f64: lanes past saturation (factor 1, ln = 0) or past n get 0.0 from the
        // bounds check -- no per-lane compare, select or address arithmetic.
        // Work is dealt by what is live: every row moves in the first slice, few do further out (231 / 129 / 41 rows in
        // slices 0 / 1 / 2 on the bench workload), so waves 0 and 1 share the first slice (alternate batches; the two
        // partial sums meet through 64 doubles of LDS) and waves 2 and 3 take the other slices, each over the rows
        // [0, s_nlive[slice]) only.  (One slice per wave over all rows: the first wave did 231 batches-worth of rows
        // while the others mostly loaded zeros.)
        auto slice_sum = [&](uint32_t sl, uint32_t rbeg, uint32_t rstep) -> double {
            const uint32_t voff = (i_lo + sl * 64u + lane) * 8u;
            const uint32_t rend = s_nlive[sl];
            double S = 0.0;
            for (uint32_t r0 = rbeg; r0 < rend; r0 += rstep) {
                const uint32_t l16 = r0 + (lane & 15u);
                const uint32_t ov = row_off[l16], ev = row_end[l16], hv = row_h[l16];
                double c[16];
#pragma unroll
                for (int k = 0; k < 16; k++)
                    c[k] = row_load_f64(Ct, (uint32_t)__builtin_amdgcn_readlane((int)ov, k), (uint32_t)__builtin_amdgcn_readlane((int)ev, k), voff);
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const uint32_t h = (uint32_t)__builtin_amdgcn_readlane((int)hv, k);
                    S = fma((double)h, c[k], S);
                }
            }
            return S;
        };
        const uint32_t nslices = ((n - i_lo) >> 6) + 1u;  // <= 8 for t <= 1023, <= 16 for t <= 2047
        double S0 = 0.0;
        if (wave < 2) {
            S0 = slice_sum(0, wave * 16u, 32u);
            if (wave == 1) Ppart[lane] = S0;
        } else {
            for (uint32_t sl = wave - 1u; sl < nslices; sl += 2) {
                const double S = slice_sum(sl, 0, 16u);
                const uint32_t i = i_lo + sl * 64u + lane;
                if (i <= n) Pi[i - i_lo] = exp(S);
            }
        }
        __syncthreads();
        if (wave == 0) {
            const uint32_t i = i_lo + lane;
            if (i <= n) Pi[i - i_lo] = exp(S0 + Ppart[lane]);
        }
        __syncthreads();
        // ---- pass 2: table[m] = sum_i pmf_m(i) P(i) / cmf_m(i)  (prob.rs:74-90); a wave takes eight
        // rows per turn and two 64-wide slices of i at a time (sixteen loads in flight), then one joint
        // reduction of the eight rows.  Rows through row_load_f64 as in pass 1 (a padding row has end = off: all zeros).
        for (uint32_t r0 = wave * 8; r0 < nact; r0 += 32) {
            double acc[8];
            uint32_t off[8], end[8];
            uint32_t lmax = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                acc[k] = 0.0;
                off[k] = (uint32_t)__builtin_amdgcn_readfirstlane((int)row_off[r0 + k]);
                end[k] = (uint32_t)__builtin_amdgcn_readfirstlane((int)row_end[r0 + k]);
                const uint32_t cnt = (end[k] - off[k]) >> 3;  // entries below saturation (0 for a padding row)
                lmax = max(lmax, cnt ? cnt - 1u : 0u);
            }
            for (uint32_t ib = i_lo; ib <= lmax; ib += 128) {
                // the second 64-wide slice only if some row of the batch reaches it (44 % of the rows of a bench query end
                // inside the first slice: their second loads would all be clipped to zeros, but still be issued)
                const bool two = ib + 64u <= lmax;
                double rv[2][8], P2[2];
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    if (c == 1 && !two) break;
                    const uint32_t i = ib + c * 64 + lane;
                    P2[c] = i <= lmax ? Pi[i - i_lo] : 0.0;
#pragma unroll
                    for (int k = 0; k < 8; k++) rv[c][k] = row_load_f64(Rt, off[k], end[k], i * 8u);
                }
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    if (c == 1 && !two) break;
#pragma unroll
                    for (int k = 0; k < 8; k++) acc[k] = fma(rv[c][k], P2[c], acc[k]);
                }
            }
            const double v = wave_sum8_f64(acc);  // lane l: total of row r0 + (l & 7)
            if (lane < 8) {
                const uint32_t m = row_m[r0 + lane];
                if (m != 0) tzl[m] = v;
            }
        }
        if (tid == 0 && ms[0] == 0) tzl[0] = i_lo == 0 && u_thr == 0u ? Pi[0] : 0.0;  // m = 0: table[0] = P(0)
    }
    __syncthreads();
    // Z = probs_sum (prob.rs:97) grouped by count value; fixed reduction order
    double part = 0.0;
    for (uint32_t j = tid; j < D; j += 256) {
        const uint32_t m = ms[j];
        part += (double)hl[m] * tzl[m];
    }
    part = wave_sum_f64_dpp(part);
    if (lane == 0) red[wave] = part;
    __syncthreads();
    const double Z = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    const double inv_n = 1.0 / (double)p.n_refs;
    double gsum = 0.0;
    for (uint32_t j = tid; j < D; j += 256) {
        const uint32_t m = ms[j];
        const double v = tzl[m] / Z;  // prob.rs:99-102
        tz[m] = v;
        const double d = v - inv_n;
        gsum += (double)hl[m] * d * d;
    }
    gsum = wave_sum_f64_dpp(gsum);
    if (lane == 0) red[8 + wave] = gsum;
    __syncthreads();
    if (tid == 0) {
        p.z[gq] = Z;
        p.gs[gq] = sqrt((red[8] + red[9]) + (red[10] + red[11]));
        p.status[gq] = RTX_Q_OK;
    }
}

// ---------------------------------------------------------------------------
// processing order of a sub-batch for prob_lookup: queries grouped by t (descending), so that
// concurrently resident workgroups read the same (t) tables out of L2.  One workgroup: counting sort
// over t <= 2047 in LDS (histogram, exclusive scan, scatter).  The order inside one t is arbitrary;
// it only decides which workgroup runs when, never a result.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void prob_order_kernel(const uint32_t *__restrict__ t, uint32_t nq,
                                                          uint32_t *__restrict__ order) {
    constexpr uint32_t kBins = 2048;  // t <= 2047 (the class of the memoised tables); a thread takes two neighbouring bins
    __shared__ uint32_t bin[kBins];
    __shared__ uint32_t wsum[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    bin[2u * tid] = 0;
    bin[2u * tid + 1u] = 0;
    __syncthreads();
    for (uint32_t q = tid; q < nq; q += 1024) atomicAdd(&bin[t[q] < kBins - 1u ? t[q] : kBins - 1u], 1u);
    __syncthreads();
    const uint32_t v0 = bin[2u * tid], v1 = bin[2u * tid + 1u];
    const uint32_t incl = wave_incl_scan_u32(v0 + v1);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t off = 0;
    for (uint32_t w = 0; w < wave; w++) off += wsum[w];
    bin[2u * tid] = off + incl - v0 - v1;
    bin[2u * tid + 1u] = off + incl - v1;
    __syncthreads();
    // descending t: the longest queries start first and the short ones fill the tail of the launch
    for (uint32_t q = tid; q < nq; q += 1024) order[nq - 1u - atomicAdd(&bin[t[q] < kBins - 1u ? t[q] : kBins - 1u], 1u)] = q;
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
size_t prob_lookup_lds_bytes(uint32_t tmax) {
    const size_t n1max = tmax / 2 + 1;
    return sizeof(double) * (n1max + 16 + 64 + ((size_t)tmax + 2)) + (sizeof(uint32_t) + 3 * sizeof(uint16_t)) * ((size_t)tmax + 18) + 8 +
           sizeof(uint32_t) * ((size_t)tmax + 4) + 2 * sizeof(uint32_t) * ((size_t)tmax + 18);
}
void launch_prob_tables_build(hipStream_t s, const ProbTables &tb, const double *lf, const double *inv) {
    hipLaunchKernelGGL(prob_tables_build_kernel, dim3((tb.tmax + 255) / 256, tb.tmax - 1), dim3(256), 0, s, tb, lf, inv);
}
void launch_prob_order(hipStream_t s, const uint32_t *t, uint32_t nq, uint32_t *order) {
    hipLaunchKernelGGL(prob_order_kernel, dim3(1), dim3(1024), 0, s, t, nq, order);
}
void launch_prob_lookup(hipStream_t s, const ProbParams &p, const ProbTables &tb, uint32_t nq) {
    hipLaunchKernelGGL(prob_lookup_kernel, dim3(nq), dim3(256), prob_lookup_lds_bytes(p.tmax), s, p, tb);
}

}  // namespace rtx
