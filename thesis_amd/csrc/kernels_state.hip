// kernels_state.hip -- a2: Robot.imu_update (robot.py:45-57) with the reference's three motion-model
// families evaluated per particle in float64, operation order kept (no FMA contraction).
#include "rbpf_internal.h"

namespace rbpf {

struct M3 { double m[3][3]; };

// np.matmul order: out[i][j] = a[i][0]*b[0][j] + a[i][1]*b[1][j] + a[i][2]*b[2][j]
__device__ __forceinline__ M3 mat_mul(const M3& a, const M3& b) {
    M3 o;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            o.m[i][j] = (a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j]) + a.m[i][2] * b.m[2][j];
    return o;
}
__device__ __forceinline__ M3 mat_t(const M3& a) {
    M3 o;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) o.m[i][j] = a.m[j][i];
    return o;
}

__global__ void imu_update_kernel(DevView v, int model, double d0, double d1, double d2, double dt_ticks,
                                  double a0, double a1, double b0, double b1) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= v.P) return;
    const double PI = 3.141592653589793;
    double x = v.px[p], y = v.py[p], th = v.pth[p];
    M3 C, F, Q;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            C.m[i][j] = v.cov[(size_t)(3 * i + j) * v.P + p];
            F.m[i][j] = i == j ? 1.0 : 0.0;
            Q.m[i][j] = 0.0;
        }
    double nx, ny, nth;
    const double nz0 = 0.01 * 0.01, nz2 = (0.2 * PI / 180) * (0.2 * PI / 180);
    if (model == RBPF_IMU_UNICYCLE) {                 // DefaultIMUData.py:26-54
        double dt = dt_ticks / 1e4;
        nth = th + dt * d1;
        nx = x + dt * d0 * cos(nth);
        ny = y + dt * d0 * sin(nth);
        double c = cos(th), s = sin(th);
        F.m[0][2] = dt * d0 * c;
        F.m[1][2] = dt * d0 * s;
        double su[3][2] = {{dt * c, 0.0}, {dt * s, 0.0}, {0.0, dt}};
        double mag[2] = {0.05 * 0.05, (PI / 180 / 2) * (PI / 180 / 2)};
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                // (su @ diag(mag)) @ su^T, two-term sums in index order
                double a_i0 = su[i][0] * mag[0] + su[i][1] * 0.0;
                double a_i1 = su[i][0] * 0.0 + su[i][1] * mag[1];
                Q.m[i][j] = fabs(a_i0 * su[j][0] + a_i1 * su[j][1]);
            }
        Q.m[0][0] += nz0; Q.m[1][1] += nz0; Q.m[2][2] += nz2;
    } else if (model == RBPF_IMU_ABSOLUTE) {          // IntelIMUData.py:23-36 (callback roles as in the reference)
        nx = d0; ny = d1; nth = d2;
        F.m[0][0] = nz0; F.m[1][1] = nz0; F.m[2][2] = nz2;
        Q.m[0][0] = 1.0; Q.m[1][1] = 1.0; Q.m[2][2] = 1.0;
        Q.m[0][2] = d0 - x;
        Q.m[1][2] = d1 - y;
    } else {                                          // Freid101IMUData.py:34-55
        double dt = dt_ticks / 1e4;
        nx = x + d0 * dt; ny = y + d1 * dt; nth = th + d2 * dt;
        double q0 = a0 + a1 * fabs(d0) * dt, q1 = a0 + a1 * fabs(d1) * dt;
        double q2 = b0 * PI / 180 + b1 * fabs(d2) * dt;
        Q.m[0][0] = fabs(q0 * q0); Q.m[1][1] = fabs(q1 * q1); Q.m[2][2] = fabs(q2 * q2);
    }
    M3 R = mat_mul(mat_mul(F, C), mat_t(F));          // robot.py:50
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) v.cov[(size_t)(3 * i + j) * v.P + p] = R.m[i][j] + Q.m[i][j];   // robot.py:51
    v.px[p] = nx; v.py[p] = ny; v.pth[p] = nth;
}

void launch_imu_update(const DevView& v, int model, double d0, double d1, double d2, double dt_ticks,
                       const double* vn, hipStream_t s) {
    hipLaunchKernelGGL(imu_update_kernel, dim3((v.P + 255) / 256), dim3(256), 0, s, v, model, d0, d1, d2,
                       dt_ticks, vn[0], vn[1], vn[2], vn[3]);
}

// main.py:167-168: last_scan = scan.from_global_reference(particles[0].get_latest_pose()) - the current scan in the
// global frame of one particle's pose (lidar.py:111-128), kept on the device for the next get_scan_adj calls
__global__ void last_scan_kernel(DevView v, int particle, double* __restrict__ out_xy) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= v.B) return;
    double sn, cs;
    sincos(v.pth[particle], &sn, &cs);
    const double x = v.bx[b], y = v.by[b];
    out_xy[2 * b] = (cs * x + (-sn) * y) + v.px[particle];                          // lidar.py:123
    out_xy[2 * b + 1] = (sn * x + cs * y) + v.py[particle];
}

// The per-step scan block (17 KB) is pulled out of pinned, device-mapped host memory by a kernel: a DMA copy of this
// size spends ~20 us in the copy engine's scheduling, during which the stream's next kernel cannot start.
__global__ void ingest_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int n16) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}

void launch_ingest(const void* mapped_src, void* d_dst, size_t bytes, hipStream_t s) {
    const int n16 = (int)((bytes + 15) / 16);
    hipLaunchKernelGGL(ingest_kernel, dim3((n16 + 255) / 256), dim3(256), 0, s, static_cast<const uint4*>(mapped_src),
                       static_cast<uint4*>(d_dst), n16);
}

// two ingests in one launch (the sharded resample hands over two index vectors)
__global__ void ingest2_kernel(const int32_t* __restrict__ src_a, int32_t* __restrict__ dst_a, const int32_t* __restrict__ src_b,
                               int32_t* __restrict__ dst_b, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { dst_a[i] = src_a[i]; dst_b[i] = src_b[i]; }
}
void launch_ingest2(const int32_t* mapped_a, int32_t* d_a, const int32_t* mapped_b, int32_t* d_b, int n, hipStream_t s) {
    hipLaunchKernelGGL(ingest2_kernel, dim3((n + 255) / 256), dim3(256), 0, s, mapped_a, d_a, mapped_b, d_b, n);
}

// The early resample's read-back written straight into pinned, device-mapped host memory by one kernel:
// [0] the NaN-branch element of the reduced weight vector (8 bytes), [8] the did flag, [16...] the ancestors.
__global__ void readback_kernel(unsigned char* __restrict__ dst, const double* __restrict__ nan_elem, const int32_t* __restrict__ did,
                                const int32_t* __restrict__ idx, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) reinterpret_cast<int32_t*>(dst + 16)[i] = idx[i];
    if (i == 0) { *reinterpret_cast<double*>(dst) = *nan_elem; *reinterpret_cast<int32_t*>(dst + 8) = *did; }
}
void launch_readback(void* mapped_dst, const double* d_nan_elem, const int32_t* d_did, const int32_t* d_idx, int n, hipStream_t s) {
    hipLaunchKernelGGL(readback_kernel, dim3((n + 255) / 256), dim3(256), 0, s, static_cast<unsigned char*>(mapped_dst), d_nan_elem, d_did, d_idx, n);
}

void launch_last_scan(const DevView& v, int particle, double* d_out_xy, hipStream_t s) {
    hipLaunchKernelGGL(last_scan_kernel, dim3((v.B + 255) / 256), dim3(256), 0, s, v, particle, d_out_xy);
}

}  // namespace rbpf
