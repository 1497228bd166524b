// Decoder hot path: the residual-decomposition decoder of model/STTODE.py:16-77,302-347 (reference),
// re-designed for CDNA4 as three kernel families over "columns" (agents or trajectories):
//
//   gru_cols      conv1d(k=3) + relu + GRU(32->96) over Tp steps, recurrent weights RESIDENT IN LDS
//                 (W_ih 36 KiB + W_hh 108 KiB PK16-packed, loaded once per persistent workgroup), hidden
//                 state in registers, every gate GEMM on v_mfma_f32_16x16x4_f32.
//                 block 0 runs once per AGENT (x_hat = 0 => the K samples of an agent share it),
//                 block 1 runs once per TRAJECTORY.
//   mlp_cols      the 256->512->256->{2Tp,2Tf} relu MLPs.  Layer 1 is split algebraically:
//                     W1 * [pf | z | state] = (W1[:, pf] * pf (+ W1[:, state] * state0) + b1)   <- per AGENT (linear_cols)
//                                           +  W1[:, z] * z (+ W1[:, state] * state1)           <- per TRAJECTORY
//                 so the per-trajectory K of layer 1 drops from 256 to 32 (block 0) / 128 (block 1).
//                 Layer 2's 512 KiB weight streams through LDS in fragment-ordered chunks shared by the
//                 workgroup's waves; the 512-wide hidden activation never exists outside registers.
//   linear_cols   generic  Y^T = W * [X1 | X2]^T + b  (per-agent pre-activations).
//
// All arithmetic fp32 (MFMA fp32-in is exact fmaf chaining).  Launch geometry is chosen by the host
// wrappers at the bottom (persistent grids sized to the 256 CUs, interleaved tile assignment for tail balance).
#include "chain.hpp"
#include "api_util.hpp"

// ---------------------------------------------------------------------------------------------------
// conv + GRU over columns
// ---------------------------------------------------------------------------------------------------
#define GRU_WIH_F4 (18 * 2 * 64)
#define GRU_WHH_F4 (18 * 6 * 64)
#define GRU_LDS_BYTES ((GRU_WIH_F4 + GRU_WHH_F4) * 16 + 4 * 96 * 4)

template <int TPX>
__global__ __launch_bounds__(512) void gru_cols_kernel(
    const float* __restrict__ xin,    // [ncols][16*TPX]  flattened (t,c) input sequence, zero padded
    const f32x4* __restrict__ convP,  // PK16 Toeplitz conv  [2*Tp row tiles][TPX][64]
    const float* __restrict__ convB,  // [32]
    const f32x4* __restrict__ wihP,   // PK16 [18][2][64]
    const f32x4* __restrict__ whhP,   // PK16 [18][6][64]
    const float* __restrict__ gbias,  // [4][96] : b_ir+b_hr, b_iz+b_hz, b_in, b_hn
    float* __restrict__ state,        // [ncols][96]
    int ncols, int Tp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* sWih = reinterpret_cast<f32x4*>(smem);
    f32x4* sWhh = sWih + GRU_WIH_F4;
    float* sB = reinterpret_cast<float*>(sWhh + GRU_WHH_F4);
    for (int i = threadIdx.x; i < GRU_WIH_F4; i += blockDim.x) sWih[i] = wihP[i];
    for (int i = threadIdx.x; i < GRU_WHH_F4; i += blockDim.x) sWhh[i] = whhP[i];
    for (int i = threadIdx.x; i < 4 * 96; i += blockDim.x) sB[i] = gbias[i];
    __syncthreads();

    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int ntiles = (ncols + 15) >> 4;
    // interleaved assignment: consecutive tiles go to different CUs first, then to different waves
    for (int tile = blockIdx.x + gridDim.x * wave; tile < ntiles; tile += gridDim.x * nw) {
        const int col = tile * 16 + c;
        const int colc = col < ncols ? col : ncols - 1;
        f32x4 d[TPX];
#pragma unroll
        for (int T = 0; T < TPX; ++T) d[T] = ld4(xin + (size_t)colc * (16 * TPX) + 16 * T + 4 * q);
        f32x4 h[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) h[j] = splat4(0.f);

        for (int t = 0; t < Tp; ++t) {
            f32x4 e[2];
#pragma unroll
            for (int io = 0; io < 2; ++io) {
                f32x4 a = ld4(convB + 16 * io + 4 * q);
#pragma unroll
                for (int T = 0; T < TPX; ++T) a = mfma_k16(a, convP[((2 * t + io) * TPX + T) * 64 + lane], d[T]);
                e[io] = relu4(a);
            }
            f32x4 hn[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                STT_FENCE();
                f32x4 ar = ld4(sB + 0 * 96 + 16 * j + 4 * q);
                f32x4 az = ld4(sB + 1 * 96 + 16 * j + 4 * q);
                f32x4 ai = ld4(sB + 2 * 96 + 16 * j + 4 * q);
                f32x4 ah = ld4(sB + 3 * 96 + 16 * j + 4 * q);
#pragma unroll
                for (int T = 0; T < 2; ++T) {
                    ar = mfma_k16(ar, sWih[((0 + j) * 2 + T) * 64 + lane], e[T]);
                    az = mfma_k16(az, sWih[((6 + j) * 2 + T) * 64 + lane], e[T]);
                    ai = mfma_k16(ai, sWih[((12 + j) * 2 + T) * 64 + lane], e[T]);
                }
#pragma unroll
                for (int T = 0; T < 6; ++T) {
                    ar = mfma_k16(ar, sWhh[((0 + j) * 6 + T) * 64 + lane], h[T]);
                    az = mfma_k16(az, sWhh[((6 + j) * 6 + T) * 64 + lane], h[T]);
                    ah = mfma_k16(ah, sWhh[((12 + j) * 6 + T) * 64 + lane], h[T]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float rg = sigmoidf_(ar[r]);
                    const float zg = sigmoidf_(az[r]);
                    const float ng = tanhf_(ai[r] + rg * ah[r]);
                    hn[j][r] = (1.0f - zg) * ng + zg * h[j][r];
                }
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) h[j] = hn[j];
        }
        if (col < ncols) {
#pragma unroll
            for (int j = 0; j < 6; ++j) st4(state + (size_t)col * 96 + 16 * j + 4 * q, h[j]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// generic per-column linear:  out[col][row0 + ...] = act( W * [X1 | X2] + b )
// ---------------------------------------------------------------------------------------------------
template <int RT>
__global__ __launch_bounds__(256) void linear_cols_kernel(
    const float* __restrict__ X1, int ld1, int KT1, const float* __restrict__ X2, int ld2, int KT2,
    const f32x4* __restrict__ WP,  // PK16 [NT][KT1+KT2][64]
    const float* __restrict__ bias, float* __restrict__ out, int ldo, int ncols, int NT, int relu) {
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int wave = threadIdx.x >> 6;
    const int ctile = blockIdx.x * 4 + wave;
    const int rt0 = blockIdx.y * RT;
    if (ctile * 16 >= ncols) return;
    const int col = ctile * 16 + c;
    const int colc = col < ncols ? col : ncols - 1;
    const int KT = KT1 + KT2;
    f32x4 acc[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) acc[i] = (rt0 + i < NT && bias) ? ld4(bias + 16 * (rt0 + i) + 4 * q) : splat4(0.f);
    for (int T = 0; T < KT; ++T) {
        const f32x4 b = T < KT1 ? ld4(X1 + (size_t)colc * ld1 + 16 * T + 4 * q) : ld4(X2 + (size_t)colc * ld2 + 16 * (T - KT1) + 4 * q);
#pragma unroll
        for (int i = 0; i < RT; ++i)
            if (rt0 + i < NT) acc[i] = mfma_k16(acc[i], WP[((size_t)(rt0 + i) * KT + T) * 64 + lane], b);
    }
    if (col < ncols) {
#pragma unroll
        for (int i = 0; i < RT; ++i)
            if (rt0 + i < NT) st4(out + (size_t)col * ldo + 16 * (rt0 + i) + 4 * q, relu ? relu4(acc[i]) : acc[i]);
    }
}

// ---------------------------------------------------------------------------------------------------
// MLP over columns:  out = W3 * relu( W2 * relu( A0[agent] + W1v * B ) + b2 ) + b3
// ---------------------------------------------------------------------------------------------------
struct MlpDesc {
    const float* A0;      // [nagents][512] per-agent layer-1 pre-activation (bias folded in)
    const f32x4* chunks;  // chunk stream: NCH x { W1v tiles [CHT][KTV][64] , W2 tiles [CHT][16][64] }
    const float* b2;      // [256]
    const f32x4* w3;      // PK16 [NO][16][64]
    const float* b3;      // [16*NO] zero padded
};

template <int KTV, int CHT, int NO>
__device__ __forceinline__ void run_mlp(const MlpDesc& m, const f32x4 (&B)[KTV], int agent, f32x4 (&out)[NO],
                                        f32x4* __restrict__ lds, int lane, int q) {
    constexpr int CHW = CHT * (KTV + 16) * 64;  // float4 per chunk
    constexpr int NCH = 32 / CHT;
    static_assert(CHW % 256 == 0, "chunk must split evenly over 256 threads");
    constexpr int PER = CHW / 256;
    f32x4 acc2[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) acc2[it] = ld4(m.b2 + 16 * it + 4 * q);
    f32x4 stage[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) stage[i] = m.chunks[i * 256 + threadIdx.x];
#pragma unroll
    for (int i = 0; i < PER; ++i) lds[i * 256 + threadIdx.x] = stage[i];
    __syncthreads();
    const float* a0 = m.A0 + (size_t)agent * 512 + 4 * q;
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
        if (ch + 1 < NCH) {
            const f32x4* src = m.chunks + (size_t)(ch + 1) * CHW;
#pragma unroll
            for (int i = 0; i < PER; ++i) stage[i] = src[i * 256 + threadIdx.x];
        }
        const f32x4* buf = lds + (ch & 1) * CHW;
#pragma unroll
        for (int hf = 0; hf < CHT; ++hf) {
            STT_FENCE();
            f32x4 h1 = ld4(a0 + (ch * CHT + hf) * 16);
#pragma unroll
            for (int T = 0; T < KTV; ++T) h1 = mfma_k16(h1, buf[(hf * KTV + T) * 64 + lane], B[T]);
            h1 = relu4(h1);
            const f32x4* w2 = buf + (CHT * KTV + hf * 16) * 64 + lane;
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                if ((it & 7) == 0) STT_FENCE();
                acc2[it] = mfma_k16(acc2[it], w2[it * 64], h1);
            }
        }
        if (ch + 1 < NCH) {
            f32x4* dst = lds + ((ch + 1) & 1) * CHW;
#pragma unroll
            for (int i = 0; i < PER; ++i) dst[i * 256 + threadIdx.x] = stage[i];
        }
        __syncthreads();
    }
#pragma unroll
    for (int it = 0; it < 16; ++it) acc2[it] = relu4(acc2[it]);
#pragma unroll
    for (int o = 0; o < NO; ++o) {
        STT_FENCE();
        f32x4 a = ld4(m.b3 + 16 * o + 4 * q);
#pragma unroll
        for (int T = 0; T < 16; ++T) {
            if ((T & 7) == 0) STT_FENCE();
            a = mfma_k16(a, m.w3[(o * 16 + T) * 64 + lane], acc2[T]);
        }
        out[o] = a;
    }
}

// block 0: x and y MLPs per trajectory.  d = x_true - x_hat0 -> dbuf ; y_hat0 -> ybuf
template <int TPX, int NOY>
__global__ __launch_bounds__(256, 2) void mlp_block0_kernel(
    MlpDesc mx, MlpDesc my, const float* __restrict__ z,  // [ncols][32]
    const float* __restrict__ xpad,                       // [nagents][16*TPX] normalised past (t,c), zero padded
    float* __restrict__ dbuf,                             // [ncols][16*TPX]
    float* __restrict__ ybuf,                             // [ncols][16*NOY]
    int ncols, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* lds = reinterpret_cast<f32x4*>(smem);
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4, wave = threadIdx.x >> 6;
    const int ngroups = (ncols + 63) >> 6;  // 4 waves x 16 columns per workgroup step
    for (int g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const int col = g * 64 + wave * 16 + c;
        const int colc = col < ncols ? col : ncols - 1;
        const int agent = colc / K;
        f32x4 B[2];
        B[0] = ld4(z + (size_t)colc * 32 + 4 * q);
        B[1] = ld4(z + (size_t)colc * 32 + 16 + 4 * q);
        f32x4 xo[TPX];
        run_mlp<2, 2, TPX>(mx, B, agent, xo, lds, lane, q);
        if (col < ncols) {
#pragma unroll
            for (int o = 0; o < TPX; ++o) {
                const f32x4 xt = ld4(xpad + (size_t)agent * (16 * TPX) + 16 * o + 4 * q);
                st4(dbuf + (size_t)col * (16 * TPX) + 16 * o + 4 * q, xt - xo[o]);
            }
        }
        f32x4 yo[NOY];
        run_mlp<2, 2, NOY>(my, B, agent, yo, lds, lane, q);
        if (col < ncols) {
#pragma unroll
            for (int o = 0; o < NOY; ++o) st4(ybuf + (size_t)col * (16 * NOY) + 16 * o + 4 * q, yo[o]);
        }
    }
}

// block 1: y MLP per trajectory with the per-trajectory GRU state; final epilogue
//   pred[col][t][c] = ((y_hat0 + y_hat1) + cur[agent][c]) + orig[agent][c]      (model/STTODE.py:338,344,622)
template <int NOY>
__global__ __launch_bounds__(256, 2) void mlp_block1_kernel(
    MlpDesc my, const float* __restrict__ z,  // [ncols][32]
    const float* __restrict__ state1,         // [ncols][96]
    const float* __restrict__ ybuf,           // [ncols][16*NOY]  y_hat0
    const float* __restrict__ cur,            // [nagents][2]
    const float* __restrict__ orig,           // [nagents][2]
    float* __restrict__ pred,                 // [ncols][2*Tf]
    int ncols, int K, int Tf2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* lds = reinterpret_cast<f32x4*>(smem);
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4, wave = threadIdx.x >> 6;
    const int ngroups = (ncols + 63) >> 6;
    for (int g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const int col = g * 64 + wave * 16 + c;
        const int colc = col < ncols ? col : ncols - 1;
        const int agent = colc / K;
        f32x4 B[8];
        B[0] = ld4(z + (size_t)colc * 32 + 4 * q);
        B[1] = ld4(z + (size_t)colc * 32 + 16 + 4 * q);
#pragma unroll
        for (int T = 0; T < 6; ++T) B[2 + T] = ld4(state1 + (size_t)colc * 96 + 16 * T + 4 * q);
        f32x4 yo[NOY];
        run_mlp<8, 1, NOY>(my, B, agent, yo, lds, lane, q);
        if (col < ncols) {
            const float cx = cur[2 * agent], cy = cur[2 * agent + 1];
            const float ox = orig[2 * agent], oy = orig[2 * agent + 1];
#pragma unroll
            for (int o = 0; o < NOY; ++o) {
                const int row0 = 16 * o + 4 * q;
                if (row0 < Tf2) {
                    const f32x4 y0 = ld4(ybuf + (size_t)col * (16 * NOY) + row0);
                    f32x4 v;
                    v[0] = ((y0[0] + yo[o][0]) + cx) + ox;
                    v[1] = ((y0[1] + yo[o][1]) + cy) + oy;
                    v[2] = ((y0[2] + yo[o][2]) + cx) + ox;
                    v[3] = ((y0[3] + yo[o][3]) + cy) + oy;
                    float* p = pred + (size_t)col * Tf2 + row0;
                    if (row0 + 3 < Tf2 && (Tf2 & 3) == 0) {
                        st4(p, v);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (row0 + r < Tf2) p[r] = v[r];
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------
static int g_num_cu = 0;
static int num_cus() {
    if (!g_num_cu) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) g_num_cu = p.multiProcessorCount;
        if (g_num_cu <= 0) g_num_cu = 256;
    }
    return g_num_cu;
}

extern "C" int sttode_gru_cols(const float* xin, const float* convP, const float* convB, const float* wihP, const float* whhP,
                               const float* gbias, float* state, int ncols, int Tp, int TPX, void* stream) {
    STT_REQUIRE(xin && convP && convB && wihP && whhP && gbias && state, "sttode_gru_cols: null pointer");
    STT_REQUIRE(ncols > 0 && Tp > 0 && (TPX == 1 || TPX == 2) && 2 * Tp <= 16 * TPX, "sttode_gru_cols: bad ncols/Tp/TPX");
    hipStream_t s = (hipStream_t)stream;
    const int ntiles = (ncols + 15) / 16;
    int grid = num_cus();
    if (grid > ntiles) grid = ntiles;
    if (TPX == 1) {
        STT_HIP(hipFuncSetAttribute((const void*)gru_cols_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, GRU_LDS_BYTES));
        hipLaunchKernelGGL(gru_cols_kernel<1>, dim3(grid), dim3(512), GRU_LDS_BYTES, s, xin, (const f32x4*)convP, convB,
                           (const f32x4*)wihP, (const f32x4*)whhP, gbias, state, ncols, Tp);
    } else {
        STT_HIP(hipFuncSetAttribute((const void*)gru_cols_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, GRU_LDS_BYTES));
        hipLaunchKernelGGL(gru_cols_kernel<2>, dim3(grid), dim3(512), GRU_LDS_BYTES, s, xin, (const f32x4*)convP, convB,
                           (const f32x4*)wihP, (const f32x4*)whhP, gbias, state, ncols, Tp);
    }
    STT_HIP(hipGetLastError());
    return 0;
}

extern "C" int sttode_linear_cols(const float* X1, int ld1, int K1, const float* X2, int ld2, int K2, const float* WP,
                                  const float* bias, float* out, int ldo, int ncols, int N, int relu, void* stream) {
    STT_REQUIRE(X1 && WP && out, "sttode_linear_cols: null pointer");
    STT_REQUIRE(ncols > 0 && N > 0 && N % 16 == 0 && K1 > 0 && K1 % 16 == 0 && K2 >= 0 && K2 % 16 == 0, "sttode_linear_cols: N, K1, K2 must be multiples of 16");
    STT_REQUIRE(ld1 % 4 == 0 && ld2 % 4 == 0 && ldo % 4 == 0 && (K2 == 0 || X2), "sttode_linear_cols: leading dims must be multiples of 4");
    const int NT = N / 16;
    dim3 grid((ncols + 63) / 64, (NT + 7) / 8);
    hipLaunchKernelGGL(linear_cols_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, X1, ld1, K1 / 16, X2, ld2, K2 / 16,
                       (const f32x4*)WP, bias, out, ldo, ncols, NT, relu);
    STT_HIP(hipGetLastError());
    return 0;
}

static MlpDesc mk(const float* A0, const float* chunks, const float* b2, const float* w3, const float* b3) {
    MlpDesc m;
    m.A0 = A0; m.chunks = (const f32x4*)chunks; m.b2 = b2; m.w3 = (const f32x4*)w3; m.b3 = b3;
    return m;
}

#define MLP0_LDS (2 * 2 * (2 + 16) * 64 * 16)  // double buffer, CHT=2, KTV=2 : 73728 B
#define MLP1_LDS (2 * 1 * (8 + 16) * 64 * 16)  // double buffer, CHT=1, KTV=8 : 49152 B

extern "C" int sttode_mlp_block0(const float* A0x, const float* chunks_x, const float* b2x, const float* w3x, const float* b3x,
                                 const float* A0y, const float* chunks_y, const float* b2y, const float* w3y, const float* b3y,
                                 const float* z, const float* xpad, float* dbuf, float* ybuf, int ncols, int K, int TPX, int NOY,
                                 void* stream) {
    STT_REQUIRE(A0x && chunks_x && b2x && w3x && b3x && A0y && chunks_y && b2y && w3y && b3y && z && xpad && dbuf && ybuf,
                "sttode_mlp_block0: null pointer");
    STT_REQUIRE(ncols > 0 && K > 0, "sttode_mlp_block0: ncols and K must be positive");
    const int ngroups = (ncols + 63) / 64;
    int grid = 2 * num_cus();
    if (grid > ngroups) grid = ngroups;
    hipStream_t s = (hipStream_t)stream;
    MlpDesc mx = mk(A0x, chunks_x, b2x, w3x, b3x), my = mk(A0y, chunks_y, b2y, w3y, b3y);
#define L0(TX, NY)                                                                                                            \
    do {                                                                                                                      \
        STT_HIP(hipFuncSetAttribute((const void*)mlp_block0_kernel<TX, NY>, hipFuncAttributeMaxDynamicSharedMemorySize, MLP0_LDS)); \
        hipLaunchKernelGGL((mlp_block0_kernel<TX, NY>), dim3(grid), dim3(256), MLP0_LDS, s, mx, my, z, xpad, dbuf, ybuf, ncols, K); \
    } while (0)
    if (TPX == 1 && NOY == 2) L0(1, 2);
    else if (TPX == 2 && NOY == 5) L0(2, 5);
    else if (TPX == 1 && NOY == 1) L0(1, 1);
    else if (TPX == 1 && NOY == 3) L0(1, 3);
    else if (TPX == 2 && NOY == 2) L0(2, 2);
    else if (TPX == 2 && NOY == 3) L0(2, 3);
    else STT_REQUIRE(false, "sttode_mlp_block0: unsupported (TPX, NOY); built: (1,1) (1,2) (1,3) (2,2) (2,3) (2,5)");
#undef L0
    STT_HIP(hipGetLastError());
    return 0;
}

extern "C" int sttode_mlp_block1(const float* A1y, const float* chunks_y, const float* b2y, const float* w3y, const float* b3y,
                                 const float* z, const float* state1, const float* ybuf, const float* cur, const float* orig,
                                 float* pred, int ncols, int K, int Tf, int NOY, void* stream) {
    STT_REQUIRE(A1y && chunks_y && b2y && w3y && b3y && z && state1 && ybuf && cur && orig && pred, "sttode_mlp_block1: null pointer");
    STT_REQUIRE(ncols > 0 && K > 0 && Tf > 0 && 2 * Tf <= 16 * NOY, "sttode_mlp_block1: bad ncols/K/Tf/NOY");
    const int ngroups = (ncols + 63) / 64;
    int grid = 2 * num_cus();
    if (grid > ngroups) grid = ngroups;
    hipStream_t s = (hipStream_t)stream;
    MlpDesc my = mk(A1y, chunks_y, b2y, w3y, b3y);
#define L1(NY)                                                                                                              \
    do {                                                                                                                    \
        STT_HIP(hipFuncSetAttribute((const void*)mlp_block1_kernel<NY>, hipFuncAttributeMaxDynamicSharedMemorySize, MLP1_LDS)); \
        hipLaunchKernelGGL((mlp_block1_kernel<NY>), dim3(grid), dim3(256), MLP1_LDS, s, my, z, state1, ybuf, cur, orig, pred, ncols, K, 2 * Tf); \
    } while (0)
    switch (NOY) {
        case 1: L1(1); break;
        case 2: L1(2); break;
        case 3: L1(3); break;
        case 5: L1(5); break;
        default: STT_REQUIRE(false, "sttode_mlp_block1: unsupported NOY; built: 1 2 3 5");
    }
#undef L1
    STT_HIP(hipGetLastError());
    return 0;
}
