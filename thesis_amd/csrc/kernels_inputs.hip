// kernels_inputs.hip -- a6: the point lists HybridMap.get_scan_match hands to the matcher (hybridmap.py:210-242),
// built from ONE particle's map.  Inspection / parity entry (rbpf_match_inputs), not on the per-step path: the
// built-in matcher reads the map directly (kernels_match.hip).
//
//   curr   endpoints of beams with 1e-3 < range < 11 at the guess pose, snapped to the lower corner of their cell
//          (get_cell + index_to_distance, hybridmap.py:216-228), in beam order;
//   ref    every cell with log-odds > threshold inside the [c-w, c+w) window (w = int(1.8/cell), gridmap.py:142-155)
//          of some curr point, for every tile, as np.unique returns them: sorted by x, then y (hybridmap.py:230-237);
//   both translated by -guess.xy and filtered by radius (11.0 / 11.5 m, hybridmap.py:239-240).
#include "rbpf_internal.h"
#include "rbpf_device.h"

namespace rbpf {

struct InputsArgs {
    int particle;
    double gx, gy, gth;
    double* all_curr;      // [B][2] unfiltered curr points (global frame), compacted in beam order
    int* n_all;            // [1]
    uint32_t* mask;        // [L*L][dim][ow] candidate bits
    int* row_cnt;          // [L*dim] ref points per (lattice x, cell x) row, then exclusive offsets
    double* ref_xy; int* n_ref; int cap_ref;
    double* curr_xy; int* n_curr;
    int win;               // int(1.8 / cell_size)
    double match_max;      // 11.0
};

__device__ __forceinline__ double index_to_distance(int i, int dim, double tile_len) {     // gridmap.py:333-334
    return ((double)i - (double)dim / 2) * tile_len / (double)dim;
}

// curr points in beam order (single workgroup)
__global__ __launch_bounds__(1024) void inputs_curr_kernel(DevView v, InputsArgs a) {
    __shared__ int s_scan[1024];
    __shared__ int s_base;
    const int tid = threadIdx.x;
    const int32_t* tab = v.tile_tab + (size_t)v.slot[a.particle] * v.L * v.L;
    double sn, cs;
    sincos(a.gth, &sn, &cs);
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int b0 = 0; b0 < v.B; b0 += 1024) {
        const int b = b0 + tid;
        bool ok = false;
        double px = 0, py = 0;
        if (b < v.B && (v.bflags[b] & BF_MATCH)) {                              // hybridmap.py:218
            const double x = (cs * v.bx[b] + (-sn) * v.by[b]) + a.gx, y = (sn * v.bx[b] + cs * v.by[b]) + a.gy;
            int lx, ly;
            if (tile_of_coord(x, v.tile_len, v.R, lx) && tile_of_coord(y, v.tile_len, v.R, ly) &&
                tab[(lx + v.R) * v.L + (ly + v.R)] >= 0) {                      // hybridmap.py:220-221
                const double cx = (double)lx * v.tile_len, cy = (double)ly * v.tile_len;
                int ix, iy;
                if (get_cell_index(y - cy, v.tile_len, v.dim, iy) && get_cell_index(x - cx, v.tile_len, v.dim, ix)) {
                    px = index_to_distance(ix, v.dim, v.tile_len) + cx;         // hybridmap.py:226-227
                    py = index_to_distance(iy, v.dim, v.tile_len) + cy;
                    ok = true;
                }
            }
        }
        s_scan[tid] = ok;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            int t = s_scan[tid];
            if (tid >= off) t += s_scan[tid - off];
            __syncthreads();
            s_scan[tid] = t;
            __syncthreads();
        }
        if (ok) { const int k = s_base + s_scan[tid] - 1; a.all_curr[2 * k] = px; a.all_curr[2 * k + 1] = py; }
        __syncthreads();
        if (tid == 1023) s_base += s_scan[1023];
        __syncthreads();
    }
    if (tid == 0) *a.n_all = s_base;
}

// candidate windows: one workgroup per curr point, all tiles (hybridmap.py:230-234)
__global__ void inputs_window_kernel(DevView v, InputsArgs a) {
    const int k = blockIdx.x;
    if (k >= *a.n_all) return;
    const double px = a.all_curr[2 * k], py = a.all_curr[2 * k + 1];
    const int32_t* tab = v.tile_tab + (size_t)v.slot[a.particle] * v.L * v.L;
    const size_t cells = (size_t)v.dim * v.dim;
    for (int t = 0; t < v.L * v.L; ++t) {
        const int tile = tab[t];
        if (tile < 0) continue;
        const double cx = (double)(t / v.L - v.R) * v.tile_len, cy = (double)(t % v.L - v.R) * v.tile_len;
        const double rx = px - cx, ry = py - cy, half = v.tile_len / 2;
        int dec_x = 0, dec_y = 0;                                               // gridmap.py:130-140 _get_rel_cell
        if (ry < -half) dec_y = 1; else if (rx < -half) dec_x = 1;
        const int c_x = (int)(rx / v.tile_len * (double)v.dim + (double)v.dim / 2) - dec_x;
        const int c_y = (int)(ry / v.tile_len * (double)v.dim + (double)v.dim / 2) - dec_y;
        const int sx = max(0, c_x - a.win), sy = max(0, c_y - a.win);          // gridmap.py:146-149
        const int ex = min(v.dim, c_x + a.win), ey = min(v.dim, c_y + a.win);
        if (ex <= sx || ey <= sy) continue;
        const int nxw = ex - sx, nyw = ey - sy;
        for (int q = threadIdx.x; q < nxw * nyw; q += blockDim.x) {
            const int x = sx + q / nyw, y = sy + q % nyw;
            if ((int)v.pool[(size_t)tile * cells + (size_t)x * v.dim + y] > v.cc.thr)     // gridmap.py:153
                atomicOr(&a.mask[((size_t)t * v.dim + x) * v.ow + (y >> 5)], 1u << (y & 31));
        }
    }
}

// ref points of one (lattice x, cell x) row, in (lattice y, cell y) order; pass 0 counts, pass 1 writes
__global__ void inputs_rows_kernel(DevView v, InputsArgs a, int pass) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= v.L * v.dim) return;
    const int la = row / v.dim, ix = row % v.dim;
    const int32_t* tab = v.tile_tab + (size_t)v.slot[a.particle] * v.L * v.L;
    int n = 0;
    int out = pass ? a.row_cnt[row] : 0;
    const double x = index_to_distance(ix, v.dim, v.tile_len) + (double)(la - v.R) * v.tile_len - a.gx;
    for (int lb = 0; lb < v.L; ++lb) {
        const int t = la * v.L + lb;
        if (tab[t] < 0) continue;
        const uint32_t* mrow = a.mask + ((size_t)t * v.dim + ix) * v.ow;
        for (int w = 0; w < v.ow; ++w) {
            uint32_t bits = mrow[w];
            while (bits) {
                const int bit = __ffs(bits) - 1;
                bits &= bits - 1;
                const int iy = w * 32 + bit;
                const double y = index_to_distance(iy, v.dim, v.tile_len) + (double)(lb - v.R) * v.tile_len - a.gy;
                if (sqrt(x * x + y * y) < a.match_max + 0.5) {                   // hybridmap.py:239
                    if (pass && out < a.cap_ref) { a.ref_xy[2 * out] = x; a.ref_xy[2 * out + 1] = y; }
                    ++out; ++n;
                }
            }
        }
    }
    if (!pass) a.row_cnt[row] = n;
}

__global__ __launch_bounds__(1024) void inputs_scan_kernel(int n_rows, int* row_cnt, int* n_ref) {
    __shared__ int s_buf[1024];
    const int tid = threadIdx.x;
    const int chunk = (n_rows + 1023) / 1024;
    const int i0 = min(tid * chunk, n_rows), i1 = min(i0 + chunk, n_rows);
    int sum = 0;
    for (int i = i0; i < i1; ++i) sum += row_cnt[i];
    s_buf[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int t = s_buf[tid];
        if (tid >= off) t += s_buf[tid - off];
        __syncthreads();
        s_buf[tid] = t;
        __syncthreads();
    }
    int run = s_buf[tid] - sum;
    for (int i = i0; i < i1; ++i) { int c = row_cnt[i]; row_cnt[i] = run; run += c; }
    if (tid == 1023) *n_ref = s_buf[1023];
}

// curr points translated by -guess and filtered by radius (hybridmap.py:236,240), order kept
__global__ __launch_bounds__(1024) void inputs_curr_filter_kernel(InputsArgs a) {
    __shared__ int s_scan[1024];
    __shared__ int s_base;
    const int tid = threadIdx.x, n = *a.n_all;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int k0 = 0; k0 < n; k0 += 1024) {
        const int k = k0 + tid;
        bool ok = false;
        double x = 0, y = 0;
        if (k < n) {
            x = a.all_curr[2 * k] - a.gx; y = a.all_curr[2 * k + 1] - a.gy;
            ok = sqrt(x * x + y * y) < a.match_max;
        }
        s_scan[tid] = ok;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            int t = s_scan[tid];
            if (tid >= off) t += s_scan[tid - off];
            __syncthreads();
            s_scan[tid] = t;
            __syncthreads();
        }
        if (ok) { const int o = s_base + s_scan[tid] - 1; a.curr_xy[2 * o] = x; a.curr_xy[2 * o + 1] = y; }
        __syncthreads();
        if (tid == 1023) s_base += s_scan[1023];
        __syncthreads();
    }
    if (tid == 0) *a.n_curr = s_base;
}

void launch_match_inputs(const DevView& v, int particle, const double* guess3, double* d_all_curr, int* d_counts,
                         uint32_t* d_mask, int* d_row_cnt, double* d_ref, int cap_ref, double* d_curr, int win,
                         double match_max, hipStream_t s) {
    InputsArgs a;
    a.particle = particle; a.gx = guess3[0]; a.gy = guess3[1]; a.gth = guess3[2];
    a.all_curr = d_all_curr; a.n_all = d_counts; a.n_curr = d_counts + 1; a.n_ref = d_counts + 2;
    a.mask = d_mask; a.row_cnt = d_row_cnt; a.ref_xy = d_ref; a.cap_ref = cap_ref; a.curr_xy = d_curr;
    a.win = win; a.match_max = match_max;
    const int n_rows = v.L * v.dim;
    hipLaunchKernelGGL(inputs_curr_kernel, dim3(1), dim3(1024), 0, s, v, a);
    hipLaunchKernelGGL(inputs_window_kernel, dim3(v.B), dim3(256), 0, s, v, a);
    hipLaunchKernelGGL(inputs_rows_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, s, v, a, 0);
    hipLaunchKernelGGL(inputs_scan_kernel, dim3(1), dim3(1024), 0, s, n_rows, d_row_cnt, a.n_ref);
    hipLaunchKernelGGL(inputs_rows_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, s, v, a, 1);
    hipLaunchKernelGGL(inputs_curr_filter_kernel, dim3(1), dim3(1024), 0, s, a);
}

}  // namespace rbpf
