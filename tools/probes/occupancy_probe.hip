// How many workgroups of a given LDS size / thread count are resident per CU on gfx950?  (tuning probe, not product code)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <map>
#include <algorithm>

__global__ void probe(long long* rec, int spin_us) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (long long)spin_us * 100) { }
    if (threadIdx.x == 0) {
        rec[blockIdx.x * 4 + 0] = t0;
        rec[blockIdx.x * 4 + 1] = wall_clock64();
        rec[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
        rec[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));
    }
    if (lds[(threadIdx.x + 1) % blockDim.x] < 0) rec[0] = 0;
}

int main() {
    const int nblk = 8192;
    long long* d;
    hipMalloc(&d, nblk * 4 * sizeof(long long));
    std::vector<long long> h(nblk * 4);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int threads[] = {256, 512};
    const int ldss[] = {1024, 16384, 32768, 40960, 41216, 49152, 53248, 65536, 81920, 98304};
    for (int t : threads)
        for (int lds : ldss) {
            hipMemset(d, 0, nblk * 4 * sizeof(long long));
            hipLaunchKernelGGL(probe, dim3(nblk), dim3(t), lds, 0, d, 20);
            hipError_t e = hipDeviceSynchronize();
            if (e != hipSuccess) { printf("threads %d lds %d: %s\n", t, lds, hipGetErrorString(e)); continue; }
            hipMemcpy(h.data(), d, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
            // max overlap per CU: sweep events
            std::map<long long, std::vector<std::pair<long long, int>>> ev;
            for (int b = 0; b < nblk; ++b) {
                const long long key = ((h[b * 4 + 3] & 0xf) << 16) | ((h[b * 4 + 2] >> 8) & 0xff);
                ev[key].push_back({h[b * 4 + 0], +1});
                ev[key].push_back({h[b * 4 + 1], -1});
            }
            int gmax = 0; double mean = 0;
            for (auto& kv : ev) {
                auto& v = kv.second;
                std::sort(v.begin(), v.end());
                int cur = 0, mx = 0;
                for (auto& x : v) { cur += x.second; mx = std::max(mx, cur); }
                gmax = std::max(gmax, mx); mean += mx;
            }
            printf("threads %3d  lds %6d B : %zu CUs, max resident workgroups per CU: max %d mean %.2f\n", t, lds, ev.size(), gmax, mean / ev.size());
        }
    return 0;
}
