// Masked up-sampling StyledConv, region-uniform output blocks (a3/a4, reference models/stylegan2/model.py:287-300, 389-398): the MAP of the 16 x 16 output blocks whose
// pixels all carry one region.  Inside such a block the layer is the single-region form (1x the transposed conv's MACs instead of the composed form's 4x): the blocks
// are computed by csrc/modconv_upblock_mx.hip (round 5: f16 + fp6, operands prepared at staging; it replaced this file's round-2 split-bf16 kernel and its
// four-sub-block variant, which were deleted), everything else by the composed kernels with the same map — the same launches for every mask, each block computed by
// exactly one of them.
#include "sb_common.h"

using namespace e4s;

namespace {

constexpr int MB_OUT = 16;                 // output pixels per block side
constexpr int MB_QUAD = 254;               // block map value: four region-uniform 8 x 8 sub-blocks with different regions (SUB = 2 variant)

// One workgroup per ROW OF FOUR 16 x 16 output blocks (= the 64 x 16 output pixels of one tile of the composed kernel, modconv_sb.hip: an
// 8 x 32 input tile at the four parities), one thread per pixel column of each block (labels sampled 'nearest' at ho x wo exactly as the
// masked kernels do):
//   sub[b][2 by + sy][2 bx + sx] = the region shared by the 8 x 8 sub-block's pixels, 255 if they differ or are no region (>= nreg);
//   blocks[b][by][bx] = that region if all four sub-blocks share one, MB_QUAD (with want_quad) if each sub-block is uniform but they differ,
//   else 255 — and 255 for ALL FOUR blocks of the row unless every one of them qualifies: the composed kernel can only leave a tile out as
//   a whole, so a tile with one mixed block is cheaper entirely in the composed form than partly in both.
__global__ __launch_bounds__(256) void uniform_blocks_kernel(uint8_t* __restrict__ blocks, uint8_t* __restrict__ sub, const uint8_t* __restrict__ labels, int lh,
                                                             int lw, int ho, int wo, float lsy, float lsx, int nreg, int want_quad, int* __restrict__ ctrl,
                                                             int min_percent) {
    // wave k = block k of the row; lane = (sub-block q = lane >> 4, its pixel column x8 = lane & 7, rows 4 yh .. 4 yh + 3 with yh = (lane >> 3) & 1):
    // four labels per lane, everything else with wave ballots — one barrier in the whole kernel
    __shared__ int flag[4];
    const int b = blockIdx.z, by = blockIdx.y, tile = blockIdx.x;
    const int nbx = wo / MB_OUT, nby = ho / MB_OUT;
    const int lane = threadIdx.x & 63, k = threadIdx.x >> 6;
    const int q = lane >> 4, x8 = lane & 7, yh = (lane >> 3) & 1;
    const int bxk = tile * 4 + k;
    int ox = bxk * MB_OUT + (q & 1) * 8 + x8;
    ox = ox < wo ? ox : wo - 1;
    const int sx = nearest_src(ox, lsx, lw);
    int c0 = -1;
    bool same = true;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int oy = by * MB_OUT + (q >> 1) * 8 + yh * 4 + i;
        oy = oy < ho ? oy : ho - 1;
        const int c = labels[((size_t)b * lh + nearest_src(oy, lsy, lh)) * lw + sx];
        if (i == 0) c0 = c;
        same = same && c == c0;
    }
    const int refq = __shfl(c0, q * 16, 64);                              // the sub-block's first label
    const unsigned long long okm = __ballot(same && c0 == refq);
    int r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rj = __shfl(c0, j * 16, 64);
        r[j] = (((okm >> (16 * j)) & 0xffffull) == 0xffffull && rj < nreg) ? rj : 255;
    }
    int f = 255;
    if (bxk < nbx) {
        const bool all_uni = r[0] != 255 && r[1] != 255 && r[2] != 255 && r[3] != 255;
        const bool one = all_uni && r[0] == r[1] && r[0] == r[2] && r[0] == r[3];
        f = one ? r[0] : ((all_uni && want_quad) ? MB_QUAD : 255);
        if (lane < 4) sub[((size_t)b * 2 * nby + 2 * by + (lane >> 1)) * (2 * nbx) + 2 * bxk + (lane & 1)] = (uint8_t)r[lane];
    }
    if (lane == 0) flag[k] = bxk < nbx ? f : 0;                           // (a block beyond the edge does not veto its row)
    __syncthreads();
    const bool all = flag[0] != 255 && flag[1] != 255 && flag[2] != 255 && flag[3] != 255;
    if (lane == 0 && bxk < nbx) blocks[((size_t)b * nby + by) * nbx + bxk] = (uint8_t)(all ? f : 255);
    // ctrl (zeroed once by the caller, left zeroed by every launch): [0] = rows of four blocks that qualify, [1] = workgroups done, [2] = the verdict for the two consumers: 1 if
    // at least min_percent of the rows qualify.  Below that the transposed-conv form does not pay (measured: scattered blocks run at half
    // the rate of contiguous ones in both kernels) and the whole layer stays in the composed form.
    if (threadIdx.x == 0) {
        if (all) atomicAdd(&ctrl[0], 1);
        __threadfence();
        const int total = gridDim.x * gridDim.y * gridDim.z;
        if (atomicAdd(&ctrl[1], 1) == total - 1) {
            __threadfence();
            const int q = atomicAdd(&ctrl[0], 0);
            ctrl[2] = (q * 100 >= min_percent * total) ? 1 : 0;
            ctrl[0] = 0;        // ready for the next layer on this stream (its map kernel runs after this layer's consumers)
            ctrl[1] = 0;
        }
    }
}

}  // namespace

extern "C" int e4s_uniform_blocks(uint8_t* blocks, uint8_t* sub, int* ctrl, const uint8_t* labels, int bs, int lh, int lw, int ho, int wo, int nreg, int want_quad,
                                  int min_percent, void* stream) {
    E4S_REQUIRE(blocks && sub && labels && ctrl, "uniform_blocks: null tensor");
    E4S_REQUIRE(min_percent >= 0 && min_percent <= 100, "uniform_blocks: min_percent in 0..100");
    E4S_REQUIRE(bs >= 0 && bs <= 65535 && lh >= 1 && lw >= 1 && ho >= 16 && wo >= 16 && (ho % 16) == 0 && (wo % 16) == 0 && nreg >= 1 && nreg <= E4S_MAX_REGIONS,
                "uniform_blocks: bad size (output height / width multiples of 16)");
    if (bs == 0) return 0;
    hipLaunchKernelGGL(uniform_blocks_kernel, dim3(cdiv(wo, 64), ho / 16, bs), dim3(256), 0, (hipStream_t)stream, blocks, sub, labels, lh, lw, ho, wo,
                       (float)lh / (float)ho, (float)lw / (float)wo, nreg, want_quad, ctrl, min_percent);
    return check_launch("uniform_blocks");
}

