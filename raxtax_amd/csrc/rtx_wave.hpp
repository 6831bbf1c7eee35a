// Wave-level (wave64) device helpers shared by the HIP translation units.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace rtx {

// ---------------------------------------------------------------------------
// wave helpers (wave64)
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// Inclusive scan over the 64 lanes: DPP row shifts inside each row of 16 lanes (zero fill at the row start),
// then the three row totals through v_readlane -- no LDS round trips (six ds_bpermute before).
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);  // row_shr:8
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 15), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 31),
                   r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 47);
    const uint32_t row = lane_id() >> 4;
    return v + (row == 0 ? 0u : (row == 1 ? r0 : (row == 2 ? r0 + r1 : r0 + r1 + r2)));
}

__device__ __forceinline__ uint32_t row16_max_u32(uint32_t v) {  // maximum over the 16 lanes of a DPP row, in every lane of the row
    uint32_t o;
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true); v = o > v ? o : v;   // quad_perm [1,0,3,2]
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true); v = o > v ? o : v;   // quad_perm [2,3,0,1]
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true); v = o > v ? o : v;  // row_half_mirror
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true); v = o > v ? o : v;  // row_mirror
    return v;
}

// Maximum over the 64 lanes, in every lane: DPP inside the rows, the four row maxima through v_readlane
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    v = row16_max_u32(v);
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16),
                   c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    const uint32_t ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;  // every lane holds the total (fixed butterfly order: deterministic)
}

// Sum over the 64 lanes, result in every lane.  DPP moves inside each row of 16 lanes
// (quad_perm xor 1, xor 2, row_half_mirror, row_mirror), then the four row totals through
// v_readlane: ~10x lower latency than six ds_bpermute round trips, and a fixed order.
__device__ __forceinline__ double dpp_mov_f64(double v, int ctrl_sel) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    switch (ctrl_sel) {
        case 0: lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true); break;   // quad_perm [1,0,3,2]
        case 1: lo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, true); break;   // quad_perm [2,3,0,1]
        case 2: lo = __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xF, 0xF, true); break; // row_half_mirror
        case 3: lo = __builtin_amdgcn_update_dpp(0, lo, 0x140, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x140, 0xF, 0xF, true); break; // row_mirror
        default: lo = __builtin_amdgcn_update_dpp(0, lo, 0x128, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x128, 0xF, 0xF, true); break; // row_ror:8 (lane ^ 8)
    }
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double wave_sum_f64_dpp(double v) {
    v += dpp_mov_f64(v, 0);
    v += dpp_mov_f64(v, 1);
    v += dpp_mov_f64(v, 2);
    v += dpp_mov_f64(v, 3);
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

// Eight sums over the 64 lanes at once: lane l returns the total of a[l & 7].  Three halving exchanges (partner
// l ^ 7 while every lane still holds all eight values, then l ^ 2 and l ^ 1 inside the quad: each lane keeps the
// half its lane bits select and hands the other half to its partner), one row rotation (l ^ 8) and two cross-row
// exchanges: ~60 VALU instructions for eight rows instead of eight separate reductions (~35 each).  Fixed order.
__device__ __forceinline__ double wave_sum8_f64(const double (&a)[8]) {
    const uint32_t lane = lane_id();
    const bool b2 = lane & 4u, b1 = lane & 2u, b0 = lane & 1u;
    double b[4], c[2];
#pragma unroll
    for (int j = 0; j < 4; j++) b[j] = (b2 ? a[j + 4] : a[j]) + dpp_mov_f64(b2 ? a[j] : a[j + 4], 2);
#pragma unroll
    for (int j = 0; j < 2; j++) c[j] = (b1 ? b[j + 2] : b[j]) + dpp_mov_f64(b1 ? b[j] : b[j + 2], 1);
    double d = (b0 ? c[1] : c[0]) + dpp_mov_f64(b0 ? c[0] : c[1], 0);
    d += dpp_mov_f64(d, 4);
    d += __shfl_xor(d, 16, 64);
    d += __shfl_xor(d, 32, 64);
    return d;
}

__device__ __forceinline__ double wave_prod_f64_dpp(double v) {
    v *= dpp_mov_f64(v, 0);
    v *= dpp_mov_f64(v, 1);
    v *= dpp_mov_f64(v, 2);
    v *= dpp_mov_f64(v, 3);
    return (readlane_f64(v, 0) * readlane_f64(v, 16)) * (readlane_f64(v, 32) * readlane_f64(v, 48));
}

// Inclusive scan over the 64 lanes with DPP row shifts (Hillis-Steele inside each row of 16 lanes, zero
// fill at the row start), then the three row totals through v_readlane.  Fixed order: deterministic.
__device__ __forceinline__ double dpp_shr_f64(double v, int n) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    switch (n) {
        case 1: lo = __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xF, 0xF, true); break;
        case 2: lo = __builtin_amdgcn_update_dpp(0, lo, 0x112, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x112, 0xF, 0xF, true); break;
        case 4: lo = __builtin_amdgcn_update_dpp(0, lo, 0x114, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x114, 0xF, 0xF, true); break;
        default: lo = __builtin_amdgcn_update_dpp(0, lo, 0x118, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x118, 0xF, 0xF, true); break;
    }
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_incl_scan_f64_dpp(double v) {
    v += dpp_shr_f64(v, 1);
    v += dpp_shr_f64(v, 2);
    v += dpp_shr_f64(v, 4);
    v += dpp_shr_f64(v, 8);
    const double r0 = readlane_f64(v, 15), r1 = readlane_f64(v, 31), r2 = readlane_f64(v, 47);
    const uint32_t row = lane_id() >> 4;
    const double add = row == 0 ? 0.0 : (row == 1 ? r0 : (row == 2 ? r0 + r1 : (r0 + r1) + r2));
    return v + add;
}

__device__ __forceinline__ double wave_incl_scan_f64(double v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        double o = __shfl_up(v, d, 64);
        if ((int)lane_id() >= d) v += o;
    }
    return v;
}


}  // namespace rtx
