// Bodies of the latency-form kernels that more than one translation unit launches (decoder.hip: sttode_gru_cols; encoder.hip: the fused
// per-agent stage).  See decoder.hip for the design notes.
#pragma once
#include "chain.hpp"

// Latency form of the same conv + GRU for FEW columns (a single scene: <= 640 trajectories): the throughput kernel above gives every
// wave a whole 16-column tile, i.e. 8 steps x 584 dependent MFMAs = 62 us however few tiles there are.  Here a workgroup owns ONE
// 16-column tile and its six waves split the 96 hidden units: wave w keeps the 24 weight fragments of ITS 16 units (3 gates x (2 + 6)
// k-tiles, 96 VGPRs) in registers for all steps, computes h'[16w .. 16w+16) and publishes it through a double-buffered LDS tile set
// (one barrier per step); the small conv is recomputed by every wave.  104 instead of 584 MFMAs per wave and step.
// sH: [2][6][64] f32x4 of LDS (h as B-operand fragments: [buffer][k-tile][lane]); `tile` = the workgroup's 16-column tile.
template <int TPX>
__device__ __forceinline__ void gru_lat_body(const float* __restrict__ xin, const f32x4* __restrict__ convP,
                                             const float* __restrict__ convB, const f32x4* __restrict__ wihP,
                                             const f32x4* __restrict__ whhP, const float* __restrict__ gbias,
                                             float* __restrict__ state, int ncols, int Tp, int tile, f32x4 (*sH)[6][64]) {
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int w = threadIdx.x >> 6;
    const int col = tile * 16 + c;
    const int colc = col < ncols ? col : ncols - 1;
    f32x4 wi[3][2], wh[3][6], b0[4];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
#pragma unroll
        for (int T = 0; T < 2; ++T) wi[g][T] = wihP[((g * 6 + w) * 2 + T) * 64 + lane];
#pragma unroll
        for (int T = 0; T < 6; ++T) wh[g][T] = whhP[((g * 6 + w) * 6 + T) * 64 + lane];
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) b0[g] = ld4(gbias + g * 96 + 16 * w + 4 * q);
    const f32x4 cb0 = ld4(convB + 4 * q), cb1 = ld4(convB + 16 + 4 * q);
    f32x4 d[TPX];
#pragma unroll
    for (int T = 0; T < TPX; ++T) d[T] = ld4(xin + (size_t)colc * (16 * TPX) + 16 * T + 4 * q);
    f32x4 hn = splat4(0.f);
    sH[0][w][lane] = hn;
    // conv fragments of step t are fetched one step ahead (an L2 round trip per step would otherwise sit on the critical path)
    f32x4 cw[2][TPX];
#pragma unroll
    for (int io = 0; io < 2; ++io)
#pragma unroll
        for (int T = 0; T < TPX; ++T) cw[io][T] = convP[(io * TPX + T) * 64 + lane];
    __syncthreads();
    int cur = 0;
#pragma unroll 1
    for (int t = 0; t < Tp; ++t) {
        f32x4 e[2] = {cb0, cb1};
#pragma unroll
        for (int io = 0; io < 2; ++io) {
#pragma unroll
            for (int T = 0; T < TPX; ++T) e[io] = mfma_k16(e[io], cw[io][T], d[T]);
            e[io] = relu4(e[io]);
        }
        {
            const int tn = (t + 1 < Tp) ? t + 1 : 0;
#pragma unroll
            for (int io = 0; io < 2; ++io)
#pragma unroll
                for (int T = 0; T < TPX; ++T) cw[io][T] = convP[((2 * tn + io) * TPX + T) * 64 + lane];
        }
        f32x4 ar = b0[0], az = b0[1], ai = b0[2], ah = b0[3];
#pragma unroll
        for (int T = 0; T < 2; ++T) {
            ar = mfma_k16(ar, wi[0][T], e[T]);
            az = mfma_k16(az, wi[1][T], e[T]);
            ai = mfma_k16(ai, wi[2][T], e[T]);
        }
        const f32x4 hp = sH[cur][w][lane];
#pragma unroll
        for (int T = 0; T < 6; ++T) {
            const f32x4 hb = sH[cur][T][lane];
            ar = mfma_k16(ar, wh[0][T], hb);
            az = mfma_k16(az, wh[1][T], hb);
            ah = mfma_k16(ah, wh[2][T], hb);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float rg = sigmoid_prescaled(ar[r]);
            const float zg = sigmoid_prescaled(az[r]);
            const float ng = tanh_prescaled(fmaf(rg, ah[r], ai[r]));
            hn[r] = fmaf(zg, hp[r] - ng, ng);  // (1-z) n + z h
        }
        sH[cur ^ 1][w][lane] = hn;
        __syncthreads();
        cur ^= 1;
    }
    if (col < ncols) st4(state + (size_t)col * 96 + 16 * w + 4 * q, hn);
}

