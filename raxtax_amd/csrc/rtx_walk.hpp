// The lineage walk of one query by one wave (src/lineage.rs:114-179), shared by lineage_walk_kernel / taxon_prefix_kernel
// (rtx_kernels.hip: prefix sums in global memory, gaps in LDS) and records_tail_kernel (rtx_records.hip: prefix sums of the few
// boundary intervals that hold a record, in LDS).  The walk is written against a PREFIX FUNCTOR pf(b) = sum of p_r over the references
// in front of boundary b.
#pragma once

#include <hip/hip_runtime.h>

#include "rtx_hit_common.hpp"
#include "rtx_kernels.hpp"
#include "rtx_wave.hpp"

namespace rtx {

// Boundaries inside runs of tiles that taxon_prefix did not sweep (no reference there reaches 1e-30) all hold the running
// sum of that moment: the fused kernel does not write them (at N = 500k they were 100 KB of stores per query, what bounded
// it) but hands the walk the few gaps [lo, hi) with their value; a boundary inside a gap reads the gap's value instead of
// memory.  Lives in LDS of the workgroup; n = 0 (or a null pointer): every boundary is in memory.
constexpr uint32_t kMaxPrefixGaps = 6;
struct PrefixGaps {
    uint32_t n;
    uint32_t lo[kMaxPrefixGaps], hi[kMaxPrefixGaps];
    double val[kMaxPrefixGaps];
};
__device__ __forceinline__ double prefix_at(const double *__restrict__ P, const PrefixGaps *G, uint32_t b) {
    double v = P[b];  // inside a gap: whatever the buffer holds, replaced below
    if (G) {
        const uint32_t n = G->n;  // wave-uniform
        for (uint32_t g = 0; g < n; g++)
            if (b >= G->lo[g] && b < G->hi[g]) v = G->val[g];
    }
    return v;
}

struct GapPrefix {  // prefix sums in global memory + the gaps taxon_prefix left (null: none)
    const double *__restrict__ P;
    const PrefixGaps *G;
    __device__ __forceinline__ double operator()(uint32_t b) const { return prefix_at(P, G, b); }
};

template <class PF>
__device__ __forceinline__ int rounded_conf(const PF &pf, uint32_t blo, uint32_t bhi) {
    const double conf = pf(bhi) - pf(blo);
    const double r = round(conf * 100.0);  // f64::round: half away from zero
    return r > 255.0 ? 255 : (r < -255.0 ? -255 : (int)r);
}

struct WalkLds {  // LDS state of one walking wave
    unsigned long long st_mask[RTX_MAX_DEPTH + 1];  // which of the 64 children scanned last are significant and unvisited ...
    uint32_t st_mbase[RTX_MAX_DEPTH + 1];           // ... and the index of the first of them
    uint32_t st_node[RTX_MAX_DEPTH + 1];
    uint32_t st_fc[RTX_MAX_DEPTH + 1];              // first child / number of children / type of the node
    uint32_t st_nch[RTX_MAX_DEPTH + 1];
    uint32_t st_next[RTX_MAX_DEPTH + 1];            // first child not scanned yet
    uint8_t st_type[RTX_MAX_DEPTH + 1];
    uint8_t st_nosig[RTX_MAX_DEPTH + 1];
    uint8_t st_pushed[RTX_MAX_DEPTH + 1];
    uint8_t kpath[RTX_MAX_DEPTH + 1];
    DevRow rows[kWalkMaxRows];
};

// The walk of query slot q by the calling wave (all 64 lanes).
template <class PF>
__device__ __forceinline__ void lineage_walk_wave(const WalkParams &p, uint32_t q, uint32_t lane, WalkLds &L, const PF &pf) {
    auto &st_mask = L.st_mask;
    auto &st_mbase = L.st_mbase;
    auto &st_node = L.st_node;
    auto &st_fc = L.st_fc;
    auto &st_nch = L.st_nch;
    auto &st_next = L.st_next;
    auto &st_type = L.st_type;
    auto &st_nosig = L.st_nosig;
    auto &st_pushed = L.st_pushed;
    auto &kpath = L.kpath;
    auto &rows = L.rows;
    const uint64_t gq = p.q0 + q;
    if (p.status[gq] != RTX_Q_OK) {
        if (lane == 0) { p.n_rows[gq] = 0; p.row_start[gq] = 0; }
        return;
    }
    const uint4 *__restrict__ rec = p.rec;
    uint32_t nrows = 0;
    bool overflow = false;

    auto emit = [&](uint32_t node, uint32_t depth) {
        if (nrows < kWalkMaxRows) {
            if (lane == 0) rows[nrows].node = node;
            if (lane < RTX_MAX_DEPTH) rows[nrows].k[lane] = lane < depth ? kpath[lane] : 0;
        } else {
            overflow = true;
        }
        nrows++;
    };

    int depth = 0;
    {
        const uint4 root = rec[0];
        if (lane == 0) {
            st_node[0] = 0; st_fc[0] = root.z; st_nch[0] = root.w & 0x3FFFFFFFu; st_type[0] = (uint8_t)(root.w >> 30);
            st_next[0] = 0; st_mbase[0] = 0; st_mask[0] = 0; st_nosig[0] = 1; st_pushed[0] = 0;
        }
    }
    wave_lds_sync();
    while (depth >= 0) {
        const uint32_t node = st_node[depth], fc = st_fc[depth], nch = st_nch[depth], type = st_type[depth];
        uint32_t next = st_next[depth], mbase = st_mbase[depth];
        unsigned long long mask = st_mask[depth];
        int found = -1;
        int kf = 0;
        uint32_t cfc = 0, cnt = 0;  // first child and n_children | type of the child found
        if (mask) {  // a significant child of the chunk scanned before: its record again (one uniform load)
            const int bit = __builtin_ctzll(mask);
            mask &= mask - 1;
            found = (int)mbase + bit;
            const uint4 cr = rec[fc + (uint32_t)found];
            kf = rounded_conf(pf, cr.x, cr.y);
            cfc = cr.z;
            cnt = cr.w;
        } else {
            while (next < nch) {
                const uint32_t idx = next + lane;
                int kk = 0;
                uint4 cr = make_uint4(0, 0, 0, 0);
                if (idx < nch) {
                    cr = rec[fc + idx];
                    kk = rounded_conf(pf, cr.x, cr.y);
                }
                const unsigned long long bal = __ballot(kk != 0);
                mbase = next;
                next += 64;
                if (bal) {
                    const int first = __builtin_ctzll(bal);
                    mask = bal & (bal - 1);
                    found = (int)mbase + first;
                    kf = __shfl(kk, first, 64);
                    cfc = (uint32_t)__shfl((int)cr.z, first, 64);
                    cnt = (uint32_t)__shfl((int)cr.w, first, 64);
                    break;
                }
            }
        }
        wave_lds_sync();
        if (found >= 0) {
            if (kf < 0 || kf > 200) overflow = true;  // cannot happen for probabilities
            if (depth + 1 > (int)RTX_MAX_DEPTH) {  // guarded at index creation
                overflow = true;
                break;
            }
            if (lane == 0) {
                st_next[depth] = next;
                st_mbase[depth] = mbase;
                st_mask[depth] = mask;
                st_nosig[depth] = 0;
                kpath[depth] = (uint8_t)(kf < 0 ? 255 : kf);
                st_node[depth + 1] = fc + (uint32_t)found;
                st_fc[depth + 1] = cfc;
                st_nch[depth + 1] = cnt & 0x3FFFFFFFu;
                st_type[depth + 1] = (uint8_t)(cnt >> 30);
                st_next[depth + 1] = 0;
                st_mbase[depth + 1] = 0;
                st_mask[depth + 1] = 0;
                st_nosig[depth + 1] = 1;
                st_pushed[depth + 1] = 0;
            }
            depth++;
            wave_lds_sync();
            continue;
        }
        // children exhausted
        bool pushed = st_pushed[depth] != 0;
        if (st_nosig[depth] && type == kInner) {  // lineage.rs:151-177
            uint32_t cn = node, ctype = type, cfirst = fc, cnch = nch;
            uint32_t d = (uint32_t)depth;
            while (ctype == kInner && cnch > 0) {
                // Iterator::max_by keeps the LAST maximum
                double best = -INFINITY;
                uint32_t besti = 0, bz = 0, bw = 0;
                bool have = false;
                for (uint32_t idx = lane; idx < cnch; idx += 64) {
                    const uint4 cr = rec[cfirst + idx];
                    const double v = pf(cr.y) - pf(cr.x);
                    if (!have || !(v < best)) { best = v; besti = idx; bz = cr.z; bw = cr.w; have = true; }
                }
#pragma unroll
                for (int s = 32; s >= 1; s >>= 1) {
                    const double ov = __shfl_xor(best, s, 64);
                    const uint32_t oi = __shfl_xor(besti, s, 64);
                    const int oh = __shfl_xor((int)have, s, 64);
                    if (oh && (!have || ov > best || (ov == best && oi > besti))) { best = ov; besti = oi; have = true; }
                }
                // the winner is the local best of lane besti & 63, which still holds its record
                cn = cfirst + besti;
                cfirst = (uint32_t)__shfl((int)bz, (int)(besti & 63u), 64);
                const uint32_t w = (uint32_t)__shfl((int)bw, (int)(besti & 63u), 64);
                cnch = w & 0x3FFFFFFFu;
                ctype = w >> 30;
                if (d >= RTX_MAX_DEPTH) { overflow = true; break; }
                if (lane == 0) kpath[d] = 1;  // 1.0 / rounding_factor
                d++;
            }
            wave_lds_sync();
            emit(cn, d);
            pushed = true;
        }
        depth--;
        if (depth >= 0) {
            // back in the parent: lineage.rs:141-149
            if (!pushed && type == kTaxon) {
                emit(node, (uint32_t)depth + 1);
                pushed = true;
            }
            if (pushed && lane == 0) st_pushed[depth] = 1;
        }
        wave_lds_sync();
    }
    wave_lds_sync();
    const uint32_t keep = nrows < kWalkMaxRows ? nrows : kWalkMaxRows;
    unsigned long long start = 0;
    if (lane == 0 && keep) {
        if (p.sub_alloc) {
            unsigned long long *w = p.sub_alloc + (size_t)(q & (kWalkSubAllocs - 1u)) * kWalkSubStride;
            const unsigned long long old = atomicAdd(w, (unsigned long long)keep);  // (the cursor half: a piece never ends above 2^32 rows)
            const unsigned long long cur = old & 0xFFFFFFFFull, end = old >> 32;
            if (cur + keep <= end) {
                start = cur;
            } else {  // the piece is used up (or this is the first walk behind the reset): a new one, what this walk leaves of it is published
                const unsigned long long take = keep > kWalkChunkRows ? keep : kWalkChunkRows;
                start = atomicAdd(p.arena_cursor, take);
                if (start + take < (1ull << 32)) atomicExch(w, ((start + take) << 32) | (start + keep));
            }
        } else {
            start = atomicAdd(p.arena_cursor, (unsigned long long)keep);
        }
    }
    start = __shfl(start, 0, 64);
    if (start + keep <= p.arena_cap) {
        // DevRow = 9 dwords
        const uint32_t *src = reinterpret_cast<const uint32_t *>(rows);
        uint32_t *dst = reinterpret_cast<uint32_t *>(p.arena + start);
        for (uint32_t i = lane; i < keep * (uint32_t)(sizeof(DevRow) / 4); i += 64) dst[i] = src[i];
    } else if (lane == 0) {
        atomicOr(p.flags_out, 1u);  // arena overflow: host re-runs with a larger arena
    }
    if (lane == 0) {
        p.n_rows[gq] = keep;
        p.row_start[gq] = start;
        if (overflow) atomicOr(p.flags_out, 2u);
    }
}

}  // namespace rtx
