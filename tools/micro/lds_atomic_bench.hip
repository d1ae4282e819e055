// Developer microbenchmark (not part of the product): cycles per LDS wave-instruction for the access patterns the map
// update uses.  One 1024-thread workgroup per CU, 160 KB LDS.  hipcc --offload-arch=gfx950 -O3 -o /tmp/ldsb lds_atomic_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define N_ITER 512
enum { P_WORDS = 0, P_BYTES4 = 1, P_ODDROW = 2, P_RANDOM = 3, P_SAMEWORD = 4 };

template <int MODE>   // 0 = ds_add no return, 1 = ds_add returning (value used), 2 = ds_write_b8, 3 = ds_write_b32, 4 = ds_read_b32, 5 = ds_read_u8 + ds_write_b8 (rmw, non-atomic)
__global__ __launch_bounds__(1024) void k(int pattern, unsigned long long* out, uint32_t* sink) {
    extern __shared__ uint32_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 32768; i += 1024) lds[i] = 0;
    __syncthreads();
    uint32_t addr;   // byte address
    const uint32_t wbase = wave * 8192;   // each wave its own 8 KB
    if (pattern == P_WORDS) addr = wbase + lane * 4;
    else if (pattern == P_BYTES4) addr = wbase + lane;             // 4 lanes per word
    else if (pattern == P_ODDROW) addr = (lane * 332 + wave * 7) % 100000;   // rows 83 words apart (odd)
    else if (pattern == P_RANDOM) addr = ((uint32_t)(tid * 2654435761u) >> 8) % 120000;
    else addr = wbase;                                             // all lanes one word
    uint32_t acc = 0;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < N_ITER; ++it) {
        const uint32_t a = addr & ~3u, sh = (addr & 3u) * 8u;
        if (MODE == 0) atomicAdd(&lds[a >> 2], 1u << sh);
        else if (MODE == 1) acc += atomicAdd(&lds[a >> 2], 1u << sh);
        else if (MODE == 2) reinterpret_cast<volatile uint8_t*>(lds)[addr] = (uint8_t)it;
        else if (MODE == 3) reinterpret_cast<volatile uint32_t*>(lds)[a >> 2] = it;
        else if (MODE == 4) acc += reinterpret_cast<volatile uint32_t*>(lds)[a >> 2];
        else { uint8_t x = reinterpret_cast<volatile uint8_t*>(lds)[addr]; reinterpret_cast<volatile uint8_t*>(lds)[addr] = x + 1; }
        addr += (pattern == P_RANDOM) ? 1021u * 4u : 4u * 64u * 0u + ((pattern == P_ODDROW) ? 1u : 0u);
        if (pattern == P_RANDOM && addr >= 120000u) addr -= 120000u;
    }
    __syncthreads();
    const long long t1 = clock64();
    if (tid == 0 && blockIdx.x == 0) out[0] = (unsigned long long)(t1 - t0);
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE>
static void run(const char* name, int pattern, const char* pname, unsigned long long* d_out, uint32_t* d_sink) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 131072, 0, pattern, d_out, d_sink);
    hipDeviceSynchronize();
    unsigned long long c = 0;
    hipMemcpy(&c, d_out, 8, hipMemcpyDeviceToHost);
    // 16 waves x N_ITER wave-instructions per CU
    printf("%-22s %-10s %8.1f cycles per wave-instruction per CU (%.2f lanes/cycle)\n", name, pname, (double)c / (16.0 * N_ITER), 64.0 * 16.0 * N_ITER / (double)c);
}

int main() {
    unsigned long long* d_out; uint32_t* d_sink;
    hipMalloc(&d_out, 8); hipMalloc(&d_sink, 4);
    const char* pn[5] = {"words", "bytes4", "oddrow", "random", "sameword"};
    for (int p = 0; p < 5; ++p) {
        run<0>("ds_add (no return)", p, pn[p], d_out, d_sink);
        run<1>("ds_add_rtn", p, pn[p], d_out, d_sink);
        run<2>("ds_write_b8", p, pn[p], d_out, d_sink);
        run<3>("ds_write_b32", p, pn[p], d_out, d_sink);
        run<4>("ds_read_b32", p, pn[p], d_out, d_sink);
        run<5>("read_u8+write_b8", p, pn[p], d_out, d_sink);
    }
    return 0;
}
