// kernels_weight.hip -- a4: Robot._generate_sample_weight (robot.py:118-139) for all particles.
//
// One 256-thread workgroup per particle.  Threads own beams (b = tid, tid+256, ...) and loop the
// K sample poses innermost: the K endpoints of one beam fall into the same one or two 128-byte
// lines of the particle's tile, so the K-fold reuse is served by the CU's L1 and each touched
// line crosses HBM once per particle.  Cells are int8 lattice values, so the per-sample sum is an
// exact int32; the only float work is the reference's float64 index expression
// (gridmap.py:126) evaluated with its operation order.
#include "rbpf_internal.h"
#include "rbpf_device.h"

namespace rbpf {

static const int KMAX = 32;

__global__ __launch_bounds__(BLOCK) void weight_samples_kernel(DevView v, const double* __restrict__ guesses,
                                                               const double* __restrict__ prs, int K,
                                                               double* __restrict__ out_w) {
    __shared__ double s_c[KMAX], s_s[KMAX], s_tx[KMAX], s_ty[KMAX];
    __shared__ int s_sum[KMAX];
    __shared__ int s_tab[49];                      // lattice table of this particle (L*L <= 49)
    const int p = blockIdx.x, tid = threadIdx.x;
    const int LL = v.L * v.L;
    if (tid < K) {
        const double* g = guesses + ((size_t)p * K + tid) * 3;
        double th = g[2];
        double s, c;
        sincos(th, &s, &c);                        // lidar.py:115-116 (libm on the host side)
        s_c[tid] = c; s_s[tid] = s; s_tx[tid] = g[0]; s_ty[tid] = g[1];
        s_sum[tid] = 0;
    }
    const int32_t* tab = v.tile_tab + (size_t)v.slot[p] * LL;
    for (int i = tid; i < LL; i += BLOCK) s_tab[i] = tab[i];
    __syncthreads();

    int acc[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) acc[k] = 0;

    for (int b = tid; b < v.B; b += BLOCK) {
        if (!(v.bflags[b] & BF_WEIGHT)) continue;  // robot.py:130  0.01 < dist < 25
        const double x = v.bx[b], y = v.by[b];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            if (k < K) {
                // lidar.py:123 matmul row: c*x + (-s)*y + tx*1, left to right
                double gx = (s_c[k] * x + (-s_s[k]) * y) + s_tx[k];
                double gy = (s_s[k] * x + s_c[k] * y) + s_ty[k];
                int val;
                if (lookup_cell(v, s_tab, gx, gy, val)) acc[k] += val;
            }
        }
    }
    // block reduction: wave shuffle then one LDS atomic per wave and sample
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        if (k < K) {
            int a = acc[k];
            for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
            if ((tid & 63) == 0) atomicAdd(&s_sum[k], a);
        }
    }
    __syncthreads();
    if (tid < K) {
        // robot.py:124-138: (1 + sum of log-odds) * motion_pr ; the lattice sum is exact
        double obs = v.inv_quantum > 0 ? (v.inv_quantum + (double)s_sum[tid]) / v.inv_quantum
                                       : 1.0 + (double)s_sum[tid] * v.quantum;
        out_w[(size_t)p * K + tid] = obs * prs[(size_t)p * K + tid];
    }
}

void launch_weight_samples(const DevView& v, const double* d_guesses, const double* d_prs, int K,
                           double* d_out_w, hipStream_t s) {
    hipLaunchKernelGGL(weight_samples_kernel, dim3(v.P), dim3(BLOCK), 0, s, v, d_guesses, d_prs, K, d_out_w);
}

__global__ void get_odds_kernel(DevView v, int particle, const double* __restrict__ xy, int n,
                                double* __restrict__ vals, uint8_t* __restrict__ none) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t* tab = v.tile_tab + (size_t)v.slot[particle] * v.L * v.L;
    int val;
    bool ok = lookup_cell(v, tab, xy[2 * i], xy[2 * i + 1], val);
    none[i] = ok ? 0 : 1;
    vals[i] = ok ? (double)val * v.quantum : 0.0;
}

void launch_get_odds(const DevView& v, int particle, const double* d_xy, int n, double* d_vals,
                     uint8_t* d_none, hipStream_t s) {
    hipLaunchKernelGGL(get_odds_kernel, dim3((n + 255) / 256), dim3(256), 0, s, v, particle, d_xy, n, d_vals, d_none);
}

}  // namespace rbpf
