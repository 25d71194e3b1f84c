// What does rocprofv3's FETCH_SIZE report for a KNOWN number of bytes in the access patterns of this library's kernels?  (MI355X_MICROARCH.md: "FETCH_SIZE reports exactly
// 1/2 of the bytes of a wide coalesced streaming read ... other access widths are uncalibrated: calibrate on a known byte count in your own access pattern".)
// Every kernel below reads each byte of a 1 GiB buffer (4 x the Infinity Cache) exactly once per launch — the true fabric traffic is the buffer size — through
//   stream16        : global_load_dwordx4, consecutive lanes on consecutive 16 bytes                                 (register loads, the guide's case)
//   stream4 / rows4<ROWF> : global_load_dword, consecutive lanes on consecutive floats, straight or in rows of ROWF floats (the masked kernels' activation rows)
//   dma_rows<ROWB>  : LDS-DMA global_load_lds_dwordx4 (the chain / mx kernels' loads): a wave request = 64 pieces of 16 B that walk down rows of ROWB bytes of a
//                     "tile column" (row pitch PITCH = 16 KB, the 512-pixel split-plane row) — ROWB = 1024: one row per request (the weight slabs, a chain-conv
//                     patch row is 1 088 B); 512 / 256 / 128 / 64: two to sixteen rows per request (the half-composed up kernel's patch rows are 576 B)
//   dma_rows_off<ROWB, OFF> : the same with every row shifted by OFF bytes (a halo pixel in front: rows start 32 B before a line) — lines shared between horizontally
//                     adjacent tile columns: the true traffic is then between 1 x and (ROWB + 128) / ROWB x the buffer
// usage: rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o calib -- ./fetch_calib_probe        (tools/fetch_calib.sh prints reported / true per kernel)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

constexpr size_t BYTES = (size_t)1 << 30;
constexpr int PITCH = 16384;                     // bytes per image row
constexpr int NT = 256;

__global__ __launch_bounds__(NT) void stream16(float* out, const uint4* x, size_t n16) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n16; i += (size_t)gridDim.x * NT) {
        const uint4 v = x[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = 1.f;
}

// dword register loads, consecutive lanes on consecutive floats (the masked kernels' channels-first activation rows: a wave reads 256 B = two lines per request)
__global__ __launch_bounds__(NT) void stream4(float* out, const unsigned* x, size_t n4) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n4; i += (size_t)gridDim.x * NT) acc += x[i];
    if (acc == 0x12345678u) out[0] = 1.f;
}
// the same in rows of ROWF floats at PITCH (a 32-pixel tile row + halo of a channels-first map is 34 floats = 136 B; 32 floats = one line)
template <int ROWF>
__global__ __launch_bounds__(NT) void rows4(float* out, const unsigned* x, unsigned rows) {
    constexpr int NCOLF = PITCH / 4 / ROWF;                       // tile columns per image row
    unsigned acc = 0;
    const size_t n = (size_t)rows * NCOLF * ROWF;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
        const size_t col = i / ((size_t)rows * ROWF), in_col = i - col * (size_t)rows * ROWF;
        const size_t row = in_col / ROWF, f = in_col - row * ROWF;
        acc += x[row * (PITCH / 4) + col * ROWF + f];
    }
    if (acc == 0x12345678u) out[0] = 1.f;
}

__device__ __forceinline__ void dma16(const void* gbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(gbase), "s"(lds_dst)
                 : "memory");
}

// The buffer as an image of PITCH-byte rows; a tile column is ROWB bytes wide.  Wave request r of a workgroup's share covers pieces [64 r, 64 r + 64) of the sequence
// (tile column, row, 16-byte piece of the row) — row-major inside a column, columns side by side: every byte once.
template <int ROWB, int OFF>
__global__ __launch_bounds__(NT) void dma_rows(float* out, const unsigned char* x, unsigned rows) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];       // 32 KB landing zone, overwritten again and again
    constexpr int PPR = ROWB / 16;                                              // pieces per row
    constexpr int NCOL = PITCH / ROWB;                                          // tile columns
    const unsigned lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned long long npieces = (unsigned long long)rows * (PITCH / 16);
    const unsigned long long nreq = npieces / 64;
    for (unsigned long long r = (unsigned long long)blockIdx.x * (NT / 64) + wave; r < nreq; r += (unsigned long long)gridDim.x * (NT / 64)) {
        const unsigned long long piece = r * 64 + lane;
        const unsigned long long col_pieces = (unsigned long long)rows * PPR;   // pieces of one tile column
        const unsigned col = (unsigned)(piece / col_pieces);
        const unsigned long long in_col = piece - (unsigned long long)col * col_pieces;
        const unsigned row = (unsigned)(in_col / PPR), pc = (unsigned)(in_col - (unsigned long long)row * PPR);
        long long byte = (long long)row * PITCH + (long long)col * ROWB + pc * 16 - OFF;
        if (byte < 0) byte += PITCH;                                            // (the first column's shifted pieces wrap to the end of their row: still every byte once)
        dma16(x, (unsigned)byte, (unsigned)(((unsigned)(r & 7u)) * 4096u + wave * 1024u));      // (the buffer is 1 GiB + one row: the byte offset fits 32 bits)
        (void)NCOL;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (lds[threadIdx.x] == 0x5a && lds[threadIdx.x + 1] == 0xa5 && lds[threadIdx.x + 2] == 0x77 && lds[threadIdx.x + 3] == 0x33) out[0] = 1.f;
}

#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); exit(1); } } while (0)

template <int ROWB, int OFF>
void run_rows(float* out, const unsigned char* x) {
    const unsigned rows = (unsigned)(BYTES / PITCH);
    hipLaunchKernelGGL((dma_rows<ROWB, OFF>), dim3(2048), dim3(NT), 32768, 0, out, x, rows);
    CHECK(hipDeviceSynchronize());
}

int main() {
    unsigned char* x;
    float* out;
    CHECK(hipMalloc(&x, BYTES + PITCH));
    CHECK(hipMalloc(&out, 4096));
    CHECK(hipMemset(x, 1, BYTES + PITCH));
    CHECK(hipDeviceSynchronize());
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(stream16, dim3(4096), dim3(NT), 0, 0, out, reinterpret_cast<const uint4*>(x), BYTES / 16);
        CHECK(hipDeviceSynchronize());
        hipLaunchKernelGGL(stream4, dim3(4096), dim3(NT), 0, 0, out, reinterpret_cast<const unsigned*>(x), BYTES / 4);
        CHECK(hipDeviceSynchronize());
        hipLaunchKernelGGL((rows4<32>), dim3(4096), dim3(NT), 0, 0, out, reinterpret_cast<const unsigned*>(x), (unsigned)(BYTES / PITCH));
        CHECK(hipDeviceSynchronize());
        hipLaunchKernelGGL((rows4<16>), dim3(4096), dim3(NT), 0, 0, out, reinterpret_cast<const unsigned*>(x), (unsigned)(BYTES / PITCH));
        CHECK(hipDeviceSynchronize());
        run_rows<1024, 0>(out, x);
        run_rows<512, 0>(out, x);
        run_rows<256, 0>(out, x);
        run_rows<128, 0>(out, x);
        run_rows<64, 0>(out, x);
        run_rows<1024, 32>(out, x);
        run_rows<512, 32>(out, x);
    }
    printf("true bytes per launch: %zu\n", BYTES);
    return 0;
}
