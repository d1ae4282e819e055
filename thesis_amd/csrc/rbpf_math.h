// rbpf_math.h -- small exact helpers shared by host code and gfx950 kernels.
//
// Everything here is plain integer / IEEE-754 double arithmetic with a fixed operation
// order (the library is built with -ffp-contract=off), so that host (LUT construction) and
// device evaluate the reference's float64 index expressions identically.
// Reference = amansanghvi/Thesis, cited file:line.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

#define RBPF_HD __host__ __device__ __forceinline__

namespace rbpf {

// ---------------------------------------------------------------------------------------------
// Ray = the reference's Bresenham variant (hybridmap.py:274-301) in closed form.
//   point j (0 <= j < n):  major = m0 + s_maj * j,   minor = n0 + s_min * ((2*dmin*j + dmaj) / (2*dmaj))
// n = 0 reproduces the reference's degenerate cases (hybridmap.py:278-281: straight-down and
// straight-left rays yield no points).
// ---------------------------------------------------------------------------------------------
struct Ray {
    int x0, y0, x1, y1;
    int n;        // number of points
    int dmaj, dmin;
    int sx, sy;   // +-1 (hybridmap.py:282-283: "1 if d > 0 else -1")
    int steep;    // dy > dx: y is the major axis
};

RBPF_HD Ray ray_make(int x0, int y0, int x1, int y1) {
    Ray r;
    r.x0 = x0; r.y0 = y0; r.x1 = x1; r.y1 = y1;
    int dx = x1 - x0, dy = y1 - y0;
    int adx = dx < 0 ? -dx : dx, ady = dy < 0 ? -dy : dy;
    r.sx = dx > 0 ? 1 : -1;
    r.sy = dy > 0 ? 1 : -1;
    r.steep = ady > adx;
    r.dmaj = r.steep ? ady : adx;
    r.dmin = r.steep ? adx : ady;
    bool empty = (adx == 0 && y1 < y0) || (adx != 0 && ady == 0 && x1 < x0);
    r.n = empty ? 0 : r.dmaj + 1;
    return r;
}

RBPF_HD int ray_minor_at(const Ray& r, int j) {
    return r.dmaj > 0 ? (2 * r.dmin * j + r.dmaj) / (2 * r.dmaj) : 0;
}

RBPF_HD void ray_point(const Ray& r, int j, int& gx, int& gy) {
    int m = ray_minor_at(r, j);
    if (r.steep) { gx = r.x0 + r.sx * m; gy = r.y0 + r.sy * j; }
    else         { gx = r.x0 + r.sx * j; gy = r.y0 + r.sy * m; }
}

// first j with minor offset >= m (m >= 1), assuming dmin > 0:  (2*dmin*j + dmaj) / (2*dmaj) >= m
RBPF_HD int ray_first_j_minor_ge(const Ray& r, int m) {
    long long num = 2LL * r.dmaj * m - r.dmaj;      // need 2*dmin*j >= num
    long long den = 2LL * r.dmin;
    long long j = (num + den - 1) / den;            // ceil, num > 0 for m >= 1
    return (int)(j < 0 ? 0 : j);
}
// last j with minor offset <= m:  2*dmin*j + dmaj < 2*dmaj*(m+1)
RBPF_HD int ray_last_j_minor_le(const Ray& r, int m) {
    if (r.dmin == 0) return r.dmaj;                 // minor offset stays 0
    long long num = 2LL * r.dmaj * (m + 1) - r.dmaj - 1;  // 2*dmin*j <= num
    if (num < 0) return -1;
    long long j = num / (2LL * r.dmin);
    return (int)(j > r.dmaj ? r.dmaj : j);
}

// ---------------------------------------------------------------------------------------------
// Python int() : truncation toward zero of a float64.
// ---------------------------------------------------------------------------------------------
RBPF_HD int trunc_to_int(double v) { return (int)v; }

// ---------------------------------------------------------------------------------------------
// Global-index LUT entry: for global cell index g (hybridmap.py:123  pos = g * cell_size)
//   lat  = lattice coordinate of the tile whose [c-len/2, c+len/2) contains pos
//          (hybridmap.py:44-45, 193-208), biased by +R
//   cidx = int(rel/cell_size + dim/2.0) with rel = pos - lat*len   (gridmap.py:93, "set" formula)
// packed as (lat+R) << 16 | cidx ; LUT_INVALID outside the addressable lattice.
// ---------------------------------------------------------------------------------------------
static const uint32_t LUT_INVALID = 0xFFFFFFFFu;
RBPF_HD int lut_lat(uint32_t e) { return (int)(e >> 16); }     // biased lattice coordinate
RBPF_HD int lut_cidx(uint32_t e) { return (int)(e & 0xFFFFu); }

// ---------------------------------------------------------------------------------------------
// "get" formula for a continuous coordinate (gridmap.py:120-128 via hybridmap.py:85-93):
// lattice coordinate by exact comparisons against tile bounds, then
//   idx = int(rel / size * dim + dim / 2)
// Returns false where the reference yields None (outside every addressable tile).
// ---------------------------------------------------------------------------------------------
RBPF_HD bool tile_of_coord(double v, double tile_len, int R, int& lat_out) {
    double half = tile_len * 0.5;                   // 20.0, exact
    int l = (int)__builtin_floor((v + half) / tile_len);
    // exact fix-up with the reference's comparisons  v >= c - half  and  v < c + half
    while (v < (double)l * tile_len - half) --l;
    while (v >= (double)l * tile_len + half) ++l;
    lat_out = l;
    return l >= -R && l <= R;
}

RBPF_HD bool get_cell_index(double rel, double tile_len, int dim, int& idx) {
    double half = tile_len / 2;                     // gridmap.py:121  -self._size/2
    if (rel < -half || rel >= half) return false;
    double t = rel / tile_len * (double)dim + (double)dim / 2;   // gridmap.py:126
    idx = (int)t;
    return true;
}

// ---------------------------------------------------------------------------------------------
// Clamped log-odds adds on the int8 lattice (gridmap.py:86-117).
// ---------------------------------------------------------------------------------------------
struct CellConsts {
    int occ, nearby, emp;    // +8, +2, -3   (units of quantum)
    int vmax, vmin;          // +30, -30
    int thr;                 // occupied threshold in quanta (10); "occupied" is cell > thr
};
RBPF_HD int cell_occ(int v, const CellConsts& c)  { int t = v + c.occ;    return t < c.vmax ? t : c.vmax; }
RBPF_HD int cell_near(int v, const CellConsts& c) { int t = v + c.nearby; return t < c.vmax ? t : c.vmax; }
RBPF_HD int cell_emp(int v, const CellConsts& c)  { int t = v + c.emp;    return t > c.vmin ? t : c.vmin; }
RBPF_HD int cell_emp_n(int v, int n, const CellConsts& c) {
    long long t = (long long)v + (long long)n * c.emp;
    return t > c.vmin ? (int)t : c.vmin;
}

// event ranks inside one (cell, beam): ascending j, NEARBY after the end cell's OCCUPIED
enum { EV_E_FAR = 0, EV_E_3 = 1, EV_E_2 = 2, EV_E_LAST = 3, EV_OCC = 4, EV_NEAR = 5 };
RBPF_HD int cell_apply_rank(int v, int rank, const CellConsts& c) {
    if (rank == EV_OCC) return cell_occ(v, c);
    if (rank == EV_NEAR) return cell_near(v, c);
    return cell_emp(v, c);
}

// ---------------------------------------------------------------------------------------------
// double-double helpers (error-free transforms; need -ffp-contract=off)
// ---------------------------------------------------------------------------------------------
struct dd { double hi, lo; };
RBPF_HD dd dd_two_sum(double a, double b) {
    double s = a + b, bb = s - a;
    double e = (a - (s - bb)) + (b - bb);
    return {s, e};
}
RBPF_HD dd dd_add(dd a, dd b) {
    dd s = dd_two_sum(a.hi, b.hi);
    double e = s.lo + (a.lo + b.lo);
    double hi = s.hi + e;
    return {hi, e - (hi - s.hi)};
}
RBPF_HD dd dd_add_d(dd a, double b) { return dd_add(a, dd{b, 0.0}); }

}  // namespace rbpf
