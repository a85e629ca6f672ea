// census.hip -- where does the dispatcher put the workgroups of a 512-block launch?
// Each block records XCC id, HW_ID (se/cu) and start/end realtime; host prints blocks-per-CU histograms.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <map>
#include <vector>
__global__ void census(unsigned *out, int spin)
{
    extern __shared__ float sm[];
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20); // HW_REG_XCC_ID
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        unsigned long long t = t0;
        while (t - t0 < (unsigned long long)spin) t = __builtin_amdgcn_s_memrealtime(); // 100 MHz ticks
        const int b = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        out[b * 4 + 0] = hw; out[b * 4 + 1] = xcc; out[b * 4 + 2] = (unsigned)t0; out[b * 4 + 3] = (unsigned)t;
        sm[0] = 1.f;
    }
}
static void run(int threads, int lds, dim3 grid, const char *name)
{
    const int nb = grid.x * grid.y * grid.z;
    unsigned *d; hipMalloc(&d, nb * 16); hipMemset(d, 0, nb * 16);
    hipFuncSetAttribute((const void *)census, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(census, grid, dim3(threads), lds, 0, d, 1000 /* 10 us */);
    hipDeviceSynchronize();
    std::vector<unsigned> h(nb * 4); hipMemcpy(h.data(), d, nb * 16, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu;
    unsigned tmin = ~0u;
    for (int b = 0; b < nb; ++b) tmin = h[b * 4 + 2] < tmin ? h[b * 4 + 2] : tmin;
    for (int b = 0; b < nb; ++b) {
        const unsigned hw = h[b * 4], key = ((h[b * 4 + 1] & 15) << 16) | (((hw >> 13) & 7) << 8) | ((hw >> 8) & 15) | (((hw >> 12) & 1) << 4);
        cu[key].push_back(b);
    }
    int hist[8] = {0}, mixed = 0, late = 0;
    for (auto &kv : cu) {
        hist[kv.second.size() < 7 ? kv.second.size() : 7]++;
        bool z0 = false, z1 = false;
        for (int b : kv.second) { (b < nb / 2 ? z0 : z1) = true; }
        mixed += z0 && z1;
    }
    for (int b = 0; b < nb; ++b) late += (h[b * 4 + 2] - tmin) > 500; // started > 5 us after the first block
    printf("%-34s blocks %d distinct CUs %zu  blocks/CU hist [1:%d 2:%d 3:%d 4:%d 5+:%d]  CUs with both halves %d  late starters %d\n",
           name, nb, cu.size(), hist[1], hist[2], hist[3], hist[4], hist[5] + hist[6] + hist[7], mixed, late);
    hipFree(d);
}
int main()
{
    run(1024, 69632, dim3(16, 16, 2), "1024thr 68KB 512 blocks");
    run(1024, 69632, dim3(16, 16, 1), "1024thr 68KB 256 blocks");
    run(512, 40960, dim3(16, 16, 2), "512thr 40KB 512 blocks");
    run(512, 40960, dim3(16, 16, 1), "512thr 40KB 256 blocks");
    run(512, 81920, dim3(16, 16, 1), "512thr 80KB 256 blocks");
    run(256, 33792, dim3(32, 8, 1), "256thr 33KB 256 blocks");
    run(256, 33792, dim3(32, 32, 1), "256thr 33KB 1024 blocks");
    return 0;
}
