// rbpf_device.h -- device-side helpers shared by the kernel files (gfx950).
#pragma once
#include "rbpf_internal.h"

namespace rbpf {

// HybridMap.get_odds_at (hybridmap.py:85-93): value in quanta, or false for None.
__device__ __forceinline__ bool lookup_cell(const DevView& v, const int32_t* __restrict__ tab,
                                            double gx, double gy, int& val) {
    int lx, ly;
    if (!tile_of_coord(gx, v.tile_len, v.R, lx)) return false;   // hybridmap.py:44-45 is_in_map
    if (!tile_of_coord(gy, v.tile_len, v.R, ly)) return false;
    int t = tab[(lx + v.R) * v.L + (ly + v.R)];
    if (t < 0) return false;
    double rx = gx - (double)lx * v.tile_len;                    // hybridmap.py:88
    double ry = gy - (double)ly * v.tile_len;
    int ix, iy;
    if (!get_cell_index(ry, v.tile_len, v.dim, iy)) return false; // gridmap.py:121-122
    if (!get_cell_index(rx, v.tile_len, v.dim, ix)) return false; // gridmap.py:123-124
    val = v.pool[(size_t)t * v.dim * v.dim + (size_t)ix * v.dim + iy];
    return true;
}

// ---- LUT helpers ------------------------------------------------------------------------------
__device__ __forceinline__ bool lut_valid_g(const DevView& v, int g) {
    return g >= v.g_min && g < v.g_min + v.n_lut;
}
__device__ __forceinline__ uint32_t lut_at(const DevView& v, int g) { return v.lut[g - v.g_min]; }

// first global index whose packed entry is >= key (entries are non-decreasing in g)
__device__ inline int lut_lower_bound(const DevView& v, uint32_t key) {
    int lo = 0, hi = v.n_lut;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (v.lut[mid] < key) lo = mid + 1; else hi = mid;
    }
    return v.g_min + lo;
}

__device__ __forceinline__ void unpack_end(int32_t e, int x0, int y0, int& x1, int& y1) {
    x1 = x0 + (int)(int16_t)(e & 0xFFFF);
    y1 = y0 + (int)(int16_t)((uint32_t)e >> 16);
}


}  // namespace rbpf
