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

// Same result as lookup_cell with two multiplications per axis instead of four float64 divisions.  A position is
// resolved through its global cell number floor(x / cell_size): every bound the reference compares against - the tile
// bounds of hybridmap.py:44-45 and the integer steps of gridmap.py:126 - is a multiple of cell_size up to rounding
// (tile_len = dim * cell_size), so the exact expressions are needed only when x / cell_size lies within 1e-6 of an
// integer, i.e. when the two could pick different cells.
__device__ __forceinline__ bool lookup_cell_fast(const DevView& v, const int32_t* __restrict__ tab,
                                                 double gx, double gy, int& val) {
    const double inv_cs = (double)v.dim / v.tile_len, inv_dim = 1.0 / (double)v.dim, hd = (double)(v.dim / 2);
    const double cx = gx * inv_cs, cy = gy * inv_cs;                               // in cells
    const double fx = __builtin_floor(cx), fy = __builtin_floor(cy);
    const double rx = cx - fx, ry = cy - fy;
    const double eps = 1e-6;
    if (!(rx > eps && rx < 1 - eps && ry > eps && ry < 1 - eps) || (v.dim & 1))
        return lookup_cell(v, tab, gx, gy, val);
    const double lxf = __builtin_floor((fx + hd) * inv_dim), lyf = __builtin_floor((fy + hd) * inv_dim);   // tile lattice coordinate
    const int lx = (int)lxf, ly = (int)lyf;
    if (lx < -v.R || lx > v.R || ly < -v.R || ly > v.R) return false;
    const int t = tab[(lx + v.R) * v.L + (ly + v.R)];
    if (t < 0) return false;
    const int ix = (int)fx - lx * v.dim + v.dim / 2, iy = (int)fy - ly * v.dim + v.dim / 2;   // 0 .. dim - 1
    val = v.pool[(size_t)t * v.dim * v.dim + (size_t)ix * v.dim + iy];
    return true;
}

// The same with the integer side kept off the quarter-rate multiplier: `base` holds the byte offset of every tile of
// the particle's lattice in the pool (tile index * dim * dim, ~0 = no tile), products of small factors go through
// the 24-bit multiplier.  (Per lookup the plain form spends six 32-bit and four 64-bit multiplies on addresses.)
__device__ __forceinline__ bool lookup_cell_fast_b(const DevView& v, const int32_t* __restrict__ tab,
                                                   const unsigned long long* __restrict__ base, double gx, double gy, int& val) {
    const double inv_cs = (double)v.dim / v.tile_len, inv_dim = 1.0 / (double)v.dim, hd = (double)(v.dim / 2);
    const double cx = gx * inv_cs, cy = gy * inv_cs;                               // in cells
    const double fx = __builtin_floor(cx), fy = __builtin_floor(cy);
    const double rx = cx - fx, ry = cy - fy;
    const double eps = 1e-6;
    if (!(rx > eps && rx < 1 - eps && ry > eps && ry < 1 - eps) || (v.dim & 1))
        return lookup_cell(v, tab, gx, gy, val);
    const double lxf = __builtin_floor((fx + hd) * inv_dim), lyf = __builtin_floor((fy + hd) * inv_dim);   // tile lattice coordinate
    const int lx = (int)lxf, ly = (int)lyf;
    if (lx < -v.R || lx > v.R || ly < -v.R || ly > v.R) return false;
    const unsigned long long b = base[__mul24(lx + v.R, v.L) + (ly + v.R)];
    if (b == ~0ull) return false;
    const int ix = (int)fx - __mul24(lx, v.dim) + v.dim / 2, iy = (int)fy - __mul24(ly, v.dim) + v.dim / 2;   // 0 .. dim - 1
    val = v.pool[b + (unsigned)(__umul24(ix, v.dim) + iy)];
    return true;
}

// The same again with the common case first: the position lies in the particle's HOME tile (the one that holds its pose:
// almost every end point of a scan does).  The global cell number floor(x / cell_size) minus the tile's first cell IS the
// reference's index whenever it is safely inside a cell (see lookup_cell_fast), so the lattice arithmetic and the table
// read are skipped; everything else goes the general way.
struct HomeTile { const int8_t* base; int off_x, off_y; int ok; };
__device__ __forceinline__ HomeTile home_tile(const DevView& v, const int32_t* __restrict__ tab, double px, double py) {
    HomeTile h; h.base = nullptr; h.off_x = h.off_y = 0; h.ok = 0;
    int lx, ly;
    if ((v.dim & 1) || !tile_of_coord(px, v.tile_len, v.R, lx) || !tile_of_coord(py, v.tile_len, v.R, ly)) return h;
    const int t = tab[(lx + v.R) * v.L + (ly + v.R)];
    if (t < 0) return h;
    h.base = v.pool + (size_t)t * v.dim * v.dim;
    h.off_x = v.dim / 2 - lx * v.dim; h.off_y = v.dim / 2 - ly * v.dim;
    h.ok = 1;
    return h;
}
__device__ __forceinline__ bool lookup_cell_home(const DevView& v, const HomeTile& h, const int32_t* __restrict__ tab,
                                                 const unsigned long long* __restrict__ base, double gx, double gy, int& val) {
    const double inv_cs = (double)v.dim / v.tile_len;
    const double cx = gx * inv_cs, cy = gy * inv_cs;                               // in cells
    const double fx = __builtin_floor(cx), fy = __builtin_floor(cy);
    const double rx = cx - fx, ry = cy - fy;
    const double eps = 1e-6;
    const int ix = (int)fx + h.off_x, iy = (int)fy + h.off_y;
    if (h.ok && rx > eps && rx < 1 - eps && ry > eps && ry < 1 - eps && (unsigned)ix < (unsigned)v.dim && (unsigned)iy < (unsigned)v.dim) {
        val = h.base[__umul24(ix, v.dim) + iy];
        return true;
    }
    return lookup_cell_fast_b(v, tab, base, gx, gy, val);
}

// robot.py:75-77 for a particle on the NaN-covariance branch: weight += (1 + sum of the log-odds under the scan at the
// latest pose) * 1, evaluated on the map AFTER its update.  Whole workgroup (contains barriers).
__device__ inline void nan_branch_weight(const DevView& v, int p, int tid, int nthreads) {
    __shared__ int s_sum_nb;
    __shared__ int s_tab_nb[49];
    const int LL = v.L * v.L;
    const int32_t* tab = v.tile_tab + (size_t)v.slot[p] * LL;
    for (int i = tid; i < LL; i += nthreads) s_tab_nb[i] = tab[i];
    if (tid == 0) s_sum_nb = 0;
    __syncthreads();
    double sn, cs;
    sincos(v.pth[p], &sn, &cs);
    const double tx = v.px[p], ty = v.py[p];
    int acc = 0;
    for (int b = tid; b < v.B; b += nthreads) {
        if (!(v.bflags[b] & BF_WEIGHT)) continue;                        // robot.py:130
        const double x = v.bx[b], y = v.by[b];
        double gx = (cs * x + (-sn) * y) + tx, gy = (sn * x + cs * y) + ty;
        int val;
        if (lookup_cell_fast(v, s_tab_nb, gx, gy, val)) acc += val;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((tid & 63) == 0) atomicAdd(&s_sum_nb, acc);
    __syncthreads();
    if (tid == 0) {
        double obs = v.inv_quantum > 0 ? (v.inv_quantum + (double)s_sum_nb) / v.inv_quantum : 1.0 + (double)s_sum_nb * v.quantum;
        v.weight[p] = obs * 1.0 + v.weight[p];
    }
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
