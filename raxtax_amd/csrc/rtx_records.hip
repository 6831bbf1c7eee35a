// records_tail_kernel: the back end of a pruned query on the RECORDS path (RecordRef, rtx_kernels.hpp) -- src/lineage.rs:61-179 over
// the references that can carry probability at all.
//
// Tile pruning gives a query a threshold u: every reference with a count up to u is a reference without a hit (probability 0,
// rtx_prune.hip).  For a query that was left a few live tiles the epilogue of hit_count has written nothing but the references ABOVE u
// as (reference, count) records in reference order -- on the bench workload the ~40 members of the query's own species out of 500 000.
// Lineage::new (lineage.rs:61-77) is a running sum of p_r = table[count_r] / Z over ALL references, and a node's confidence the
// difference of that sum at the ends of its range (lineage.rs:114-117): with p_r = 0 outside the records both are sums over records.
// taxon_prefix_kernel sweeps the 8192 references of every live tile to find them again; here ONE wave per query
//   1. reads the records (64 per turn): p = table[count] / Z (prob_lookup), f = the first taxonomy boundary behind the reference
//      (bnd_rank / bnd_bits: the tables of taxon_prefix);
//   2. keeps the running sum at the END of every run of records that share f: entries (f_e, S_e) in LDS, ascending --
//      prefix(b) = S of the last entry with f <= b (0 in front of the first), whatever the number of boundaries in between;
//   3. walks the lineage (rtx_walk.hpp) with that prefix: a binary search over a few LDS words per look-up where the walk over
//      taxon_prefix's array paid a round trip to L2 -- and no prefix array is written at all.
// More entries than fit LDS (kTailEntries: a query above whose threshold hundreds of species lie): the prefix row of the query is
// written out as taxon_prefix would have (every boundary its running sum) and walked from memory -- slower, same rows.
// The sums run in reference order (a wave scan per 64 records, the carry in front): the association differs from the reference's
// one-by-one loop in the last bits only, as taxon_prefix's block scan does.
#include <hip/hip_runtime.h>

#include "rtx_kernels.hpp"
#include "rtx_walk.hpp"
#include "rtx_wave.hpp"

namespace rtx {

constexpr uint32_t kTailEntries = 256;  // boundary entries of a query kept in LDS (3 KB)

struct EntryPrefix {  // prefix(b) = S of the last entry with f <= b
    const uint32_t *f;
    const double *S;
    uint32_t n;
    __device__ __forceinline__ double operator()(uint32_t b) const {
        uint32_t lo = 0;  // entries [0, lo) have f <= b
#pragma unroll
        for (uint32_t step = kTailEntries; step >= 1u; step >>= 1) {
            const uint32_t probe = lo + step;
            const bool ok = probe <= n && f[probe <= n ? probe - 1u : 0u] <= b;
            lo = ok ? probe : lo;
        }
        return lo ? S[lo - 1u] : 0.0;
    }
};

__global__ __launch_bounds__(64) void records_tail_kernel(TailParams p) {
    extern __shared__ double tail_lds[];  // WalkLds | ent_S[kTailEntries] f64 | ent_f[kTailEntries] u32
    const uint32_t q = blockIdx.x, lane = threadIdx.x;
    const uint32_t ns = p.rec.nslots[q];
    if (ns == 0u) return;  // a query of the dense path: taxon_prefix_kernel has it
    const uint64_t gq = p.walk.q0 + q;
    if (p.walk.status[gq] != RTX_Q_OK) {
        if (lane == 0) { p.walk.n_rows[gq] = 0; p.walk.row_start[gq] = 0; }
        return;
    }
    WalkLds &L = *reinterpret_cast<WalkLds *>(tail_lds);
    double *ent_S = reinterpret_cast<double *>(reinterpret_cast<char *>(tail_lds) + ((sizeof(WalkLds) + 15u) & ~(size_t)15u));
    uint32_t *ent_f = reinterpret_cast<uint32_t *>(ent_S + kTailEntries);
    const double *__restrict__ tz = p.table_z + (size_t)q * p.hstride;
    const uint32_t tq = p.t[q];
    const uint32_t tile_v = lane < ns ? (uint32_t)p.rec.slots[(size_t)q * kRecMaxSlots + lane] : 0u;
    const uint32_t cnt_v = lane < ns ? p.rec.cnt[(size_t)q * kRecMaxSlots + lane] : 0u;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    double *__restrict__ Prow = p.prefix + (size_t)q * p.n_bnd;  // (the slow path; with the rows on their diet it takes one first, below)
    // pass 0: entries into LDS; if they do not fit, pass 1: the whole prefix row into memory (every boundary its running sum)
    uint32_t E = 0, n_rec = 0;
    bool slow = false;
    for (uint32_t pass = 0; pass < 2u; pass++) {
        double carry = 0.0;
        uint32_t filled = 0;  // pass 1: boundaries [0, filled) are written
        E = 0;
        n_rec = 0;
        for (uint32_t k = 0; k < ns; k++) {
            const uint32_t tile = (uint32_t)__builtin_amdgcn_readlane((int)tile_v, (int)k);
            uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cnt_v, (int)k);
            c = c < p.rec.seg_len ? c : p.rec.seg_len;
            const uint32_t *seg = p.rec.rec + ((size_t)q * p.rec.stride + k) * p.rec.seg_len;
            n_rec += c;
            for (uint32_t i0 = 0; i0 < c; i0 += 64) {
                const uint32_t i = i0 + lane;
                const bool valid = i < c;
                const uint32_t r = seg[valid ? i : 0u];
                const uint32_t ref = tile * 8192u + (r & 8191u);
                uint32_t cnt = r >> 13;
                cnt = cnt <= tq ? cnt : tq;  // (a count cannot exceed t)
                const uint32_t ch = ref >> 3, j = ref & 7u;
                const double tv = tz[cnt];
                const uint32_t rk = p.bnd_rank[ch], bits = p.bnd_bits[ch];
                const double pv = valid ? tv : 0.0;
                const uint32_t f = valid ? rk + (uint32_t)__popc(bits & ((1u << j) - 1u)) : 0xFFFFFFFFu;  // first boundary behind the reference
                const double incl = carry + wave_incl_scan_f64_dpp(pv);
                // the end of a run of equal f (the end of the turn counts as one: two entries with the same f -- the later one, with
                // the larger sum, is the one a look-up finds)
                const uint32_t f_next = (uint32_t)__shfl_down((int)f, 1, 64);
                const bool is_end = valid && (lane == 63u || f_next != f);
                const unsigned long long em = __ballot(is_end);
                if (pass == 0u) {
                    const uint32_t pos = E + (uint32_t)__popcll(em & lt_mask);
                    if (is_end && pos < kTailEntries) { ent_f[pos] = f; ent_S[pos] = incl; }
                } else {
                    // boundaries [filled, f_first) of this turn: the running sum in front of the turn; then, per run end, [f, next run's f)
                    // gets this run's sum -- the last run of the turn is closed by the next turn (or behind the loop)
                    const uint32_t f_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)f);
                    for (uint32_t b = filled + lane; b < f_first; b += 64) Prow[b] = carry;
                    // next run end's f for every run end: the f of the following run = f_next of this lane (f_next != f there)
                    unsigned long long rest = em;
                    while (rest) {  // wave-uniform: run by run (a run's stretch of boundaries is written by the whole wave)
                        const int e = __builtin_ctzll(rest);
                        rest &= rest - 1ull;
                        const uint32_t fa = (uint32_t)__builtin_amdgcn_readlane((int)f, e);
                        const double sa = readlane_f64(incl, e);
                        if (rest) {
                            const uint32_t fb = (uint32_t)__builtin_amdgcn_readlane((int)f_next, e);  // (e < 63 here: another run end follows)
                            for (uint32_t b = fa + lane; b < fb; b += 64) Prow[b] = sa;
                            filled = fb;
                        } else {
                            if (lane == 0) Prow[fa] = sa;  // closed by whatever comes next
                            filled = fa;                   // [fa, ..) still open: rewritten from fa on with the sum at that time
                        }
                    }
                }
                E += (uint32_t)__popcll(em);
                carry = readlane_f64(incl, 63);
            }
        }
        if (pass == 0u) {
            if (E <= kTailEntries) break;
            slow = true;
            if (p.cnt_cursor) {  // a row of the prefix buffer for this query (the dense queries got theirs from prune_kernel: PruneParams::cnt_row)
                uint32_t r = 0;
                if (lane == 0) r = atomicAdd(p.cnt_cursor, 1u);
                r = (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
                if (r >= p.cnt_cap) {  // none left: the host enlarges the buffers and repeats the run
                    if (lane == 0) { atomicOr(p.flags_out, 4u); p.walk.n_rows[gq] = 0; p.walk.row_start[gq] = 0; }
                    return;
                }
                Prow = p.prefix + (size_t)r * p.n_bnd;
            }
        } else {
            for (uint32_t b = filled + lane; b < p.n_bnd; b += 64) Prow[b] = carry;  // from the last run on: everything
        }
    }
    if (p.prefix_stats) {  // what taxon_prefix reports for the queries it sweeps: tiles with a count above the threshold (here: segments with a record), queries
        const uint32_t need = (uint32_t)__popcll(__ballot(lane < ns && cnt_v != 0u));
        if (lane < 2u) atomicAdd(&p.prefix_stats[(size_t)(q & (kPruneStatCopies - 1u)) * 8u + lane], lane == 0u ? (unsigned long long)need : 1ull);
    }
    if (p.stats && lane < 4u) {
        const unsigned long long v = lane == 0u ? n_rec : (lane == 1u ? 1u : (lane == 2u ? E : (slow ? 1u : 0u)));
        if (v) atomicAdd(&p.stats[(size_t)(q & (kPruneStatCopies - 1u)) * 8u + lane], v);
    }
    wave_lds_sync();
    if (!slow) {
        lineage_walk_wave(p.walk, q, lane, L, EntryPrefix{ent_f, ent_S, E});
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the row was written by this wave: visible to its own loads (same CU, write-through L1)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        lineage_walk_wave(p.walk, q, lane, L, GapPrefix{Prow, nullptr});
    }
}

void launch_records_tail(hipStream_t s, const TailParams &p, uint32_t nq) {
    const size_t lds = ((sizeof(WalkLds) + 15u) & ~(size_t)15u) + (size_t)kTailEntries * 12u;
    hipLaunchKernelGGL(records_tail_kernel, dim3(nq), dim3(64), lds, s, p);
}

}  // namespace rtx
