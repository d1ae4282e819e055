// rbpf_mapupdate.h -- helpers shared by the two map-update kernels (kernels_mapupdate.hip: 128x128 LDS windows,
// kernels_mapfan.hip: the whole ray fan in one LDS window).  gfx950 device code.
#pragma once
#include <limits.h>

#include "rbpf_internal.h"
#include "rbpf_device.h"

namespace rbpf {

// Workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() also drains vmcnt, i.e. every
// outstanding global load AND store; inside the window loop no thread reads or overwrites a global cell another
// thread of the workgroup wrote in the same kernel, so the HBM read-modify-writes may stay in flight across it.
#define BAR_LDS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// a value every lane agrees on, moved to a scalar register (values read from LDS are not known to be uniform)
#define UNI(x) __builtin_amdgcn_readfirstlane((int)(x))

static const int CHUNK = 16;           // ray steps per work item of the walk

// per-ray info byte
enum { RI_VALID = 1, RI_OCC = 2, RI_NEAR = 4 };   // bits 3-4: near dx + 1, bits 5-6: near dy + 1

// same-tile test of hybridmap.py:141 (m.is_in_map(nearby_pos) with m = tile of the end cell)
__device__ __forceinline__ bool same_tile(const DevView& v, int xa, int ya, int xb, int yb) {
    return lut_lat(lut_at(v, xa)) == lut_lat(lut_at(v, xb)) && lut_lat(lut_at(v, ya)) == lut_lat(lut_at(v, yb));
}

// wave-level reductions (all 64 lanes take part): one LDS atomic per wave instead of one per lane
__device__ __forceinline__ int wave_min(int v) { for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64)); return v; }
__device__ __forceinline__ int wave_max(int v) { for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64)); return v; }
__device__ __forceinline__ int wave_sum(int v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); return v; }
__device__ __forceinline__ int wave_excl_scan(int v, int lane) {   // exclusive prefix sum over the wave
    int incl = v;
    for (int o = 1; o < 64; o <<= 1) { int n = __shfl_up(incl, o, 64); if (lane >= o) incl += n; }
    return incl - v;
}

// first j with minor offset >= m (m >= 1, dmin > 0), 32-bit (2*dmaj*m < 2^31 for rays shorter than a tile)
__device__ __forceinline__ int first_j_minor_ge(const Ray& r, int m) {
    int num = 2 * r.dmaj * m - r.dmaj, den = 2 * r.dmin;
    return (num + den - 1) / den;
}
__device__ __forceinline__ int last_j_minor_le(const Ray& r, int m) {
    int num = 2 * r.dmaj * (m + 1) - r.dmaj - 1;
    if (num < 0) return -1;
    int j = num / (2 * r.dmin);
    return j > r.dmaj ? r.dmaj : j;
}

// Clamped-add functions v -> min(max(v + a, lo), hi) are closed under composition, so the ordered sequence of
// a cell's events folds associatively: each lane folds the events of one beam, the wave folds 64 beams in
// beam order with a shuffle tree.
struct Caf { int a, lo, hi; };
__device__ __forceinline__ Caf caf_then(Caf f, Caf g) {          // g after f
    Caf r;
    r.a = f.a + g.a;
    int lo = f.lo + g.a; lo = lo < g.lo ? g.lo : lo; r.lo = lo > g.hi ? g.hi : lo;
    int hi = f.hi + g.a; hi = hi < g.lo ? g.lo : hi; r.hi = hi > g.hi ? g.hi : hi;
    return r;
}
__device__ __forceinline__ int caf_apply(Caf f, int x) { int t = x + f.a; t = t < f.lo ? f.lo : t; return t > f.hi ? f.hi : t; }

// One lane replays a bucket of up to N events: bitonic sorting network on registers (padded with 0xFFFF), then the
// clamped adds in ascending (beam, rank) order.
template <int N>
__device__ __forceinline__ int replay_sorted(const uint16_t* __restrict__ evp, int m, int val, const CellConsts& cc) {
    uint32_t ev[N];
#pragma unroll
    for (int e = 0; e < N; ++e) ev[e] = e < m ? (uint32_t)evp[e] : 0xFFFFu;
#pragma unroll
    for (int k = 2; k <= N; k <<= 1)
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1)
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const int l = i ^ j;
                if (l > i) {
                    const uint32_t a = ev[i], b = ev[l];
                    const bool up = (i & k) == 0;
                    ev[i] = up ? min(a, b) : max(a, b);
                    ev[l] = up ? max(a, b) : min(a, b);
                }
            }
#pragma unroll
    for (int e = 0; e < N; ++e)
        if (e < m) val = cell_apply_rank(val, (int)(ev[e] & 7u), cc);
    return val;
}

__device__ inline int replay_cell_wave(const DevView& v, const uint8_t* r_info, const int32_t* r_end, int x0, int y0, const int* gxc, int ngx,
                                const int* gyc, int ngy, int val, int lane) {
    const int BIG = 1000000;
    const Caf fE = {v.cc.emp, v.cc.vmin, BIG}, fO = {v.cc.occ, -BIG, v.cc.vmax}, fN = {v.cc.nearby, -BIG, v.cc.vmax};
    for (int base = 0; base < v.B; base += 64) {
        const int b = base + lane;
        Caf f = {0, -BIG, BIG};
        bool has = false;
        if (b < v.B && (r_info[b] & RI_VALID)) {
            const int info = r_info[b];
            int x1, y1;
            unpack_end(r_end[b], x0, y0, x1, y1);
            Ray r = ray_make(x0, y0, x1, y1);
            const bool occ = info & RI_OCC;
            int js[4], nj = 0;
            for (int ix = 0; ix < ngx; ++ix)
                for (int iy = 0; iy < ngy; ++iy) {
                    int gx = gxc[ix], gy = gyc[iy];
                    int j = r.steep ? (gy - y0) * r.sy : (gx - x0) * r.sx;
                    if (j < 0 || j >= r.n) continue;
                    int qx, qy;
                    ray_point(r, j, qx, qy);
                    if (qx == gx && qy == gy) js[nj++] = j;
                }
            for (int a = 1; a < nj; ++a) {
                int key = js[a], c = a - 1;
                while (c >= 0 && js[c] > key) { js[c + 1] = js[c]; --c; }
                js[c + 1] = key;
            }
            bool near_here = false;
            for (int a = 0; a < nj; ++a) {
                int j = js[a];
                f = caf_then(f, (j == r.n - 1 && occ) ? fO : fE);
                if (j == r.n - 2 && (info & RI_NEAR)) near_here = true;
            }
            if (near_here) f = caf_then(f, fN);
            has = nj > 0;
        }
        if (__ballot(has) == 0ull) continue;
        for (int off = 1; off < 64; off <<= 1) {
            Caf g;
            g.a = __shfl_down(f.a, off, 64); g.lo = __shfl_down(f.lo, off, 64); g.hi = __shfl_down(f.hi, off, 64);
            if ((lane & (2 * off - 1)) == 0) f = caf_then(f, g);
        }
        val = caf_apply(f, val);        // lane 0 holds the fold of the whole chunk
        val = __shfl(val, 0, 64);
    }
    return val;
}

}  // namespace rbpf
