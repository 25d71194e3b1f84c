// Resident workgroups per CU as a function of VGPR count (256 threads, 41216 B of LDS).  Tuning probe, not product code.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <map>
#include <algorithm>

#define PROBE(NAME, REG)                                                                         \
    __global__ __launch_bounds__(256) void NAME(long long* rec, int spin_us) {                   \
        extern __shared__ float lds[];                                                           \
        lds[threadIdx.x] = threadIdx.x;                                                          \
        __syncthreads();                                                                         \
        asm volatile("v_mov_b32 " REG ", 0" ::: REG);                                            \
        const long long t0 = wall_clock64();                                                     \
        while (wall_clock64() - t0 < (long long)spin_us * 100) { }                               \
        if (threadIdx.x == 0) {                                                                  \
            rec[blockIdx.x * 4 + 0] = t0;                                                        \
            rec[blockIdx.x * 4 + 1] = wall_clock64();                                            \
            rec[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));     \
            rec[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));    \
        }                                                                                        \
        if (lds[(threadIdx.x + 1) % blockDim.x] < 0) rec[0] = 0;                                 \
    }
PROBE(p63, "v63") PROBE(p79, "v79") PROBE(p95, "v95") PROBE(p103, "v103") PROBE(p111, "v111") PROBE(p127, "v127")
PROBE(p135, "v135") PROBE(p155, "v155") PROBE(p167, "v167") PROBE(p175, "v175") PROBE(p255, "v255")

typedef void (*kern_t)(long long*, int);

int main() {
    const int nblk = 8192;
    long long* d;
    (void)hipMalloc(&d, nblk * 4 * sizeof(long long));
    std::vector<long long> h(nblk * 4);
    struct { kern_t k; int regs; } ks[] = {{p63, 64}, {p79, 80}, {p95, 96}, {p103, 104}, {p111, 112}, {p127, 128}, {p135, 136}, {p155, 156}, {p167, 168}, {p175, 176}, {p255, 256}};
    for (auto& kk : ks)
        for (int lds : {1024, 40960, 41216}) {
            (void)hipMemset(d, 0, nblk * 4 * sizeof(long long));
            hipLaunchKernelGGL(kk.k, dim3(nblk), dim3(256), lds, 0, d, 20);
            if (hipDeviceSynchronize() != hipSuccess) { printf("fail\n"); continue; }
            (void)hipMemcpy(h.data(), d, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
            std::map<long long, std::vector<std::pair<long long, int>>> ev;
            for (int b = 0; b < nblk; ++b) {
                const long long key = ((h[b * 4 + 3] & 0xf) << 16) | ((h[b * 4 + 2] >> 8) & 0xff);
                ev[key].push_back({h[b * 4 + 0], +1});
                ev[key].push_back({h[b * 4 + 1], -1});
            }
            int gmax = 0;
            for (auto& kv : ev) {
                auto& v = kv.second;
                std::sort(v.begin(), v.end());
                int cur = 0, mx = 0;
                for (auto& x : v) { cur += x.second; mx = std::max(mx, cur); }
                gmax = std::max(gmax, mx);
            }
            printf("vgprs %3d  lds %6d B  256 threads: max resident workgroups per CU %d\n", kk.regs, lds, gmax);
        }
    return 0;
}
