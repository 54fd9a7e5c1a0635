// How many 256-thread workgroups does a CU hold at once as a function of the LDS a workgroup asks for?  Every workgroup spins for a fixed
// number of clock ticks; the time of a grid of 64 workgroups per CU divided by the time of one workgroup per CU gives the number of
// rounds, hence the concurrency.  hipcc --offload-arch=gfx950 -O2 -o lds_occupancy lds_occupancy.hip && ./lds_occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
template <int V>
__device__ __forceinline__ void touch_vgpr()
{
    if (V == 144) asm volatile("v_mov_b32 v143, 0" ::: "v143");
    if (V == 168) asm volatile("v_mov_b32 v167, 0" ::: "v167");
    if (V == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
}
template <int V>
__global__ __launch_bounds__(256) void spin_v(long long ticks, int* sink)
{
    extern __shared__ int lds[];
    lds[threadIdx.x] = threadIdx.x;
    touch_vgpr<V>();
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (lds[(threadIdx.x + 1) & 255] == -1) *sink = 1;
}
__global__ __launch_bounds__(256) void spin(long long ticks, int* sink)
{
    extern __shared__ int lds[];
    lds[threadIdx.x] = threadIdx.x;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (lds[(threadIdx.x + 1) & 255] == -1) *sink = 1;
}
int main(int argc, char** argv)
{
    const bool optin = argc > 1;             // any argument: raise hipFuncAttributeMaxDynamicSharedMemorySize first

    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s: %d CUs, sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu, regsPerBlock %d\n", p.name, p.multiProcessorCount, p.sharedMemPerBlock,
           p.maxSharedMemoryPerMultiProcessor, p.regsPerBlock);
    int* sink;
    hipMalloc(&sink, 4);
    const int cus = p.multiProcessorCount;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const long long ticks = 100 * 200;            // wall_clock64 runs at 100 MHz: 200 us
    for (int kb : {1, 8, 12, 16, 20, 24, 28, 32, 40, 48, 64}) {
        if (optin) hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
        int nb = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, spin, 256, (size_t)kb * 1024);
        float ms1 = 0, msN = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0); hipLaunchKernelGGL(spin, dim3(cus), dim3(256), kb * 1024, 0, ticks, sink); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms1, e0, e1);
            hipEventRecord(e0); hipLaunchKernelGGL(spin, dim3(cus * 64), dim3(256), kb * 1024, 0, ticks, sink); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&msN, e0, e1);
        }
        printf("%sLDS %2d KB per workgroup: runtime says %2d workgroups per CU; 1 per CU %.3f ms, 64 per CU %.3f ms -> %.1f concurrent per CU (%s)\n", optin ? "(opt-in) " : "", kb, nb, ms1, msN,
               64.0 * ms1 / msN, hipGetErrorString(hipGetLastError()));
    }
    // the same with the vector registers of a 3- or 4-waves-per-SIMD kernel
    auto run = [&](auto kern, const char* what, int kb) {
        float ms1 = 0, msN = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(cus), dim3(256), kb * 1024, 0, ticks, sink); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms1, e0, e1);
            hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(cus * 64), dim3(256), kb * 1024, 0, ticks, sink); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&msN, e0, e1);
        }
        printf("%s, LDS %2d KB: %.1f concurrent workgroups per CU (%s)\n", what, kb, 64.0 * ms1 / msN, hipGetErrorString(hipGetLastError()));
    };
    for (int kb : {8, 28}) {
        run(spin_v<128>, "128 VGPRs", kb);
        run(spin_v<144>, "144 VGPRs", kb);
        run(spin_v<168>, "168 VGPRs", kb);
    }
    return 0;
}
