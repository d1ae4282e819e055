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

// Same result as lookup_cell with one multiply-add per axis instead of four float64 divisions: the quotients are
// evaluated approximately and the reference's exact expressions (gridmap.py:126, hybridmap.py:44-45) are used only
// when an approximate value lies within 1e-6 of an integer, i.e. when the two could truncate differently.
__device__ __forceinline__ bool lookup_cell_fast(const DevView& v, const int32_t* __restrict__ tab,
                                                 double gx, double gy, int& val) {
    const double half = v.tile_len * 0.5, inv_len = 1.0 / v.tile_len, scale = (double)v.dim * inv_len, hd = (double)v.dim * 0.5;
    const double tx = (gx + half) * inv_len, ty = (gy + half) * inv_len;
    const double fxl = __builtin_floor(tx), fyl = __builtin_floor(ty);
    const double qx = (gx - fxl * v.tile_len) * scale + hd, qy = (gy - fyl * v.tile_len) * scale + hd;   // in [0, dim)
    const double rx = qx - __builtin_floor(qx), ry = qy - __builtin_floor(qy);
    const double ux = tx - fxl, uy = ty - fyl;
    const double eps = 1e-6;
    if (!(rx > eps && rx < 1 - eps && ry > eps && ry < 1 - eps && ux > eps && ux < 1 - eps && uy > eps && uy < 1 - eps))
        return lookup_cell(v, tab, gx, gy, val);
    const int lx = (int)fxl, ly = (int)fyl;
    if (lx < -v.R || lx > v.R || ly < -v.R || ly > v.R) return false;
    const int t = tab[(lx + v.R) * v.L + (ly + v.R)];
    if (t < 0) return false;
    val = v.pool[(size_t)t * v.dim * v.dim + (size_t)(int)qx * v.dim + (int)qy];
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
