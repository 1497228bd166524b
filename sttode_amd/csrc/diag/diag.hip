// Diagnostics library (libsttode_diag.so) -- NOT part of the product ABI (include/sttode_hip.h) and never loaded by
// sttode_amd: probes used by profiles/*.py to separate "what the part sustains" from "what our kernels lose".
//
//   sttode_diag_mfma_kinds    bare MFMA issue loops: v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32, operands in
//                             registers or one ds_read_b128 A fragment per 4 MFMAs.  FLOP are counted PER KIND
//                             (round 1 credited the 32x32x2 kinds with half their FLOP and misread them as half rate).
//   sttode_diag_stream        the weight-stream structure of the decoder MLP kernels (double-buffered LDS chunks filled
//                             by LDS-DMA, one workgroup barrier per chunk, accumulators in registers) in both MFMA shapes:
//                               shape 0: 16x16x4, 4 waves x 16 columns per workgroup, 18 KiB chunks  (round-1 mlp_block0)
//                               shape 1: 32x32x2, 4 waves x 32 columns per workgroup, chunks of C 4-KiB tiles
//                             so a re-tiling can be priced before the real kernel is rewritten.
#include "../chain.hpp"
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));

static thread_local char g_diag_err[256] = "";
extern "C" const char* sttode_diag_last_error() { return g_diag_err; }
#define DIAG_REQUIRE(c, msg) do { if (!(c)) { snprintf(g_diag_err, sizeof g_diag_err, "%s", msg); return 1; } } while (0)
#define DIAG_HIP(e) do { hipError_t _e = (e); if (_e != hipSuccess) { snprintf(g_diag_err, sizeof g_diag_err, "%s: %s", #e, hipGetErrorString(_e)); return 2; } } while (0)

// ---------------------------------------------------------------------------------------------------
// bare issue loops
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void diag_mfma_kernel(float* __restrict__ out, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = splat4(seed * (float)(i + 1));
    f32x4 a = splat4(seed + 0.001f * (float)lane), b = splat4(1.0f - 0.002f * (float)lane);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = mfma_k16(acc[i], a, b);
        a[0] += 1e-7f;  // keep operands live / data dependent without adding real work
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) s = s + acc[i];
    if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = s[0];
}

__global__ __launch_bounds__(1024) void diag_mfma_lds_kernel(float* __restrict__ out, int iters, float seed) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* lds = reinterpret_cast<f32x4*>(smem);
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = splat4(seed + 1e-6f * (float)i);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = splat4(seed * (float)(i + 1));
    f32x4 b = splat4(1.0f - 0.002f * (float)lane);
    for (int it = 0; it < iters; ++it) {
        const f32x4* base = lds + ((it & 7) * 8) * 64 + lane;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = mfma_k16(acc[i], base[i * 64], b);
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) s = s + acc[i];
    if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = s[0];
}

__global__ __launch_bounds__(1024) void diag_mfma32_kernel(float* __restrict__ out, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = seed * (float)(i + 1);
    f32x4 a = splat4(seed + 0.001f * (float)lane), b = splat4(1.0f - 0.002f * (float)lane);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], b[r], acc[i], 0, 0, 0);
        a[0] += 1e-7f;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(1024) void diag_mfma32_lds_kernel(float* __restrict__ out, int iters, float seed) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* lds = reinterpret_cast<f32x4*>(smem);
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = splat4(seed + 1e-6f * (float)i);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = seed * (float)(i + 1);
    f32x4 b = splat4(1.0f - 0.002f * (float)lane);
    for (int it = 0; it < iters; ++it) {
        const f32x4* base = lds + ((it & 7) * 8) * 64 + lane;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 a = base[i * 64];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], b[r], acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static int time_launches(hipStream_t s, int repeats, double* ms_out, void (*go)(void*), void* ctx) {
    hipEvent_t e0, e1;
    DIAG_HIP(hipEventCreate(&e0));
    DIAG_HIP(hipEventCreate(&e1));
    go(ctx);  // warm-up
    DIAG_HIP(hipEventRecord(e0, s));
    for (int r = 0; r < repeats; ++r) go(ctx);
    DIAG_HIP(hipEventRecord(e1, s));
    DIAG_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    DIAG_HIP(hipEventElapsedTime(&ms, e0, e1));
    DIAG_HIP(hipGetLastError());
    *ms_out = ms;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return 0;
}

static int num_cus() {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 256;
    return p.multiProcessorCount;
}

// kind: 0 16x16x4 registers only | 1 16x16x4 + one ds_read_b128 per 4 MFMAs | 2 32x32x2 registers only | 3 32x32x2 + ds_read_b128
// per 4 MFMAs.  waves_per_cu in {4, 8, 12, 16}.  One iteration = 32 MFMAs of 2 048 FLOP (kinds 0, 1) or 16 MFMAs of 4 096 FLOP
// (kinds 2, 3): 65 536 FLOP per wave per iteration in every kind.
struct KindCtx { int kind, cus, waves, iters; float* scratch; hipStream_t s; };
static void kinds_go(void* p) {
    KindCtx* c = (KindCtx*)p;
    const dim3 g(c->cus), b(64 * c->waves);
    switch (c->kind) {
        case 0: hipLaunchKernelGGL(diag_mfma_kernel, g, b, 0, c->s, c->scratch, c->iters, 0.5f); break;
        case 1: hipLaunchKernelGGL(diag_mfma_lds_kernel, g, b, 65536, c->s, c->scratch, c->iters, 0.5f); break;
        case 2: hipLaunchKernelGGL(diag_mfma32_kernel, g, b, 0, c->s, c->scratch, c->iters, 0.5f); break;
        default: hipLaunchKernelGGL(diag_mfma32_lds_kernel, g, b, 65536, c->s, c->scratch, c->iters, 0.5f); break;
    }
}
extern "C" int sttode_diag_mfma_kinds(int kind, int waves_per_cu, int iters, int repeats, float* scratch, double* tflops, void* stream) {
    DIAG_REQUIRE(scratch && tflops && iters > 0 && repeats > 0 && kind >= 0 && kind <= 3, "sttode_diag_mfma_kinds: bad arguments");
    DIAG_REQUIRE(waves_per_cu == 4 || waves_per_cu == 8 || waves_per_cu == 12 || waves_per_cu == 16, "sttode_diag_mfma_kinds: waves_per_cu must be 4, 8, 12 or 16");
    KindCtx c{kind, num_cus(), waves_per_cu, iters, scratch, (hipStream_t)stream};
    double ms = 0;
    if (int rc = time_launches(c.s, repeats, &ms, kinds_go, &c)) return rc;
    const double mfma_per_iter = kind < 2 ? 32.0 : 16.0, flop_per_mfma = kind < 2 ? 2048.0 : 4096.0;
    *tflops = (double)repeats * c.cus * waves_per_cu * (double)iters * mfma_per_iter * flop_per_mfma / (ms * 1e-3) / 1e12;
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// weight-stream probes
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// shape 0: the round-1 mlp_block0 structure.  chunk = 18 fragment tiles of 1 KiB: 2 layer-1 k-tiles + 16 layer-2 row tiles.
// per chunk and wave: 8 + 64 MFMAs (16x16x4) = 147 456 FLOP.
#define S0_CHW 1152
__global__ __launch_bounds__(256, 3) void diag_stream16_kernel(const f32x4* __restrict__ blob, int total, float* __restrict__ out,
                                                               int nchunks, float seed) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* lds = reinterpret_cast<f32x4*>(smem);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    auto dma = [&](int chunk, int buf) {
        const f32x4* src = blob + (size_t)(chunk % total) * S0_CHW + lane;
        f32x4* dst = lds + buf * S0_CHW;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int idx = i * 4 + wave;
            if (idx < 18) glds16(src + idx * 64, dst + idx * 64);
        }
    };
    f32x4 acc2[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc2[i] = splat4(seed * (float)(i + 1));
    f32x4 B[2] = {splat4(1.0f - 0.002f * (float)lane), splat4(0.5f + 0.001f * (float)lane)};
    dma(blockIdx.x, 0);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        __builtin_amdgcn_sched_barrier(0);
        dma(blockIdx.x + ch + 1, (ch + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        const f32x4* buf = lds + (ch & 1) * S0_CHW;
        f32x4 h1 = splat4(0.f);
        STT_FENCE();
#pragma unroll
        for (int T = 0; T < 2; ++T) h1 = mfma_k16(h1, buf[T * 64 + lane], B[T]);
        h1 = relu4(h1);
        const f32x4* w2 = buf + 2 * 64 + lane;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            if ((it & 7) == 0) STT_FENCE();
            acc2[it] = mfma_k16(acc2[it], w2[it * 64], h1);
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
    }
    f32x4 s = acc2[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) s = s + acc2[i];
    if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = s[0];
}

// shape 1: 32x32x2.  A "tile" = the A operand of 16 MFMAs (32 rows x 32 k) = 64 lanes x 4 x f32x4 = 4 KiB, read by four
// ds_read_b128 per lane.  chunk = C tiles; hidden tile h (32 rows): tile 0 = layer-1 (K = 32), tiles 1..8 = layer-2 row tiles.
// The stream is consumed tile by tile; a barrier + DMA every C tiles.  Per tile and wave: 16 MFMAs = 65 536 FLOP.
// PRE = 1: the four fragment reads of tile t+1 are issued before the MFMAs of tile t (software pipelined).
template <int C, int PRE>
__global__ __launch_bounds__(256, 2) void diag_stream32_kernel(const f32x4* __restrict__ blob, int total_tiles, float* __restrict__ out,
                                                               int ngroups, float seed) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* lds = reinterpret_cast<f32x4*>(smem);
    constexpr int TILE = 256;  // f32x4 per tile
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    auto dma = [&](int chunk, int buf) {
        // C tiles x 4 pieces of 1 KiB, dealt round-robin over the 4 waves: C pieces per wave
        const f32x4* src = blob + ((size_t)chunk * C % total_tiles) * TILE + lane;
        f32x4* dst = lds + buf * (C * TILE);
#pragma unroll
        for (int i = 0; i < C; ++i) {
            const int idx = i * 4 + wave;
            glds16(src + idx * 64, dst + idx * 64);
        }
    };
    f32x16 acc2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[i][e] = seed * (float)(i + 1);
    f32x16 Bz;
#pragma unroll
    for (int e = 0; e < 16; ++e) Bz[e] = 1.0f - 0.002f * (float)(lane + e);
    f32x16 h1;
#pragma unroll
    for (int e = 0; e < 16; ++e) h1[e] = 0.f;
    dma(blockIdx.x, 0);
    __syncthreads();
    // 9 tiles per hidden tile (1 + 8); groups of 9*C tiles = 9 chunks keep every index static
    int chunk = 0;
    f32x4 a[4], an[4];
    auto rd = [&](f32x4 (&d)[4], const f32x4* tilep) {
#pragma unroll
        for (int g = 0; g < 4; ++g) d[g] = tilep[g * 64 + lane];
    };
    for (int grp = 0; grp < ngroups; ++grp) {
#pragma unroll
        for (int cc = 0; cc < 9; ++cc) {
            __builtin_amdgcn_sched_barrier(0);
            dma(blockIdx.x + chunk + 1, (chunk + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            const f32x4* buf = lds + (chunk & 1) * (C * TILE);
            if (PRE) rd(a, buf);
#pragma unroll
            for (int t = 0; t < C; ++t) {
                const int ti = (cc * C + t) % 9;  // 0: layer 1, 1..8: layer 2 row tile ti-1
                if (PRE) {
                    if (t + 1 < C) rd(an, buf + (t + 1) * TILE);
                } else {
                    rd(a, buf + t * TILE);
                }
                if (ti == 0) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) h1[e] = 0.f;
#pragma unroll
                    for (int j = 0; j < 16; ++j) h1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j >> 2][j & 3], Bz[j], h1, 0, 0, 0);
#pragma unroll
                    for (int e = 0; e < 16; ++e) h1[e] = fmaxf(h1[e], 0.f);
                } else {
#pragma unroll
                    for (int j = 0; j < 16; ++j)
                        acc2[ti - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j >> 2][j & 3], h1[j], acc2[ti - 1], 0, 0, 0);
                }
                if (PRE && t + 1 < C) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) a[g] = an[g];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
            ++chunk;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc2[i][e];
    if (s == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// shape 5: as shape 3 (36 KiB chunks = one hidden tile: 1 layer-1 tile + 8 layer-2 row tiles), but the layer-2 MFMAs of two row tiles
// ALTERNATE (acc2[2p], acc2[2p+1]): two independent accumulator chains instead of 16 back-to-back dependent MFMAs on one.
__global__ __launch_bounds__(256, 2) void diag_stream32i_kernel(const f32x4* __restrict__ blob, int total_tiles, float* __restrict__ out,
                                                                int ngroups, float seed) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* lds = reinterpret_cast<f32x4*>(smem);
    constexpr int TILE = 256, C = 9;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    auto dma = [&](int chunk, int buf) {
        const f32x4* src = blob + ((size_t)chunk * C % total_tiles) * TILE + lane;
        f32x4* dst = lds + buf * (C * TILE);
#pragma unroll
        for (int i = 0; i < C; ++i) {
            const int idx = i * 4 + wave;
            glds16(src + idx * 64, dst + idx * 64);
        }
    };
    f32x16 acc2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[i][e] = seed * (float)(i + 1);
    f32x16 Bz;
#pragma unroll
    for (int e = 0; e < 16; ++e) Bz[e] = 1.0f - 0.002f * (float)(lane + e);
    f32x16 h1;
    dma(blockIdx.x, 0);
    __syncthreads();
    int chunk = 0;
    for (int grp = 0; grp < ngroups * 9; ++grp) {
        __builtin_amdgcn_sched_barrier(0);
        dma(blockIdx.x + chunk + 1, (chunk + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        const f32x4* buf = lds + (chunk & 1) * (C * TILE) + lane;
        f32x4 a[4], b[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) a[g] = buf[g * 64];
#pragma unroll
        for (int e = 0; e < 16; ++e) h1[e] = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) h1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j >> 2][j & 3], Bz[j], h1, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 16; ++e) h1[e] = fmaxf(h1[e], 0.f);
#pragma unroll
        for (int p2 = 0; p2 < 4; ++p2) {
#pragma unroll
            for (int g = 0; g < 4; ++g) { a[g] = buf[(1 + 2 * p2) * TILE + g * 64]; b[g] = buf[(2 + 2 * p2) * TILE + g * 64]; }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                acc2[2 * p2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j >> 2][j & 3], h1[j], acc2[2 * p2], 0, 0, 0);
                acc2[2 * p2 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j >> 2][j & 3], h1[j], acc2[2 * p2 + 1], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        ++chunk;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc2[i][e];
    if (s == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}


// ---------------------------------------------------------------------------------------------------
// shape 6 / 7 (round 3, exploratory): the same weight-stream structure with the fp32 product replaced by a THREE-WAY bf16 SPLIT on the
// bf16 matrix cores: x = hi + mid + lo (8 mantissa bits each), W likewise (split on the host), W x ~ hi.hi + hi.mid + mid.hi + hi.lo +
// lo.hi + mid.mid (the dropped terms are <= 2^-24 relative): six v_mfma_f32_32x32x16_bf16 per 16-deep k block, fp32 accumulate.
// A "tile" (32 rows x 32 k) = 2 k blocks x 3 planes x 1 KiB = 6 KiB, read by six ds_read_b128 per lane; per tile and wave 12 MFMAs
// of 32 cycles = 384 matrix-pipe cycles against 1 024 for the sixteen 32x32x2 fp32 MFMAs of shape 1.  The accumulator layout of a
// layer still IS the B layout of the next: k slot s of lane half h <-> accumulator register 8 kb + s (feature 16 kb + 8 (s / 4) + 4 h + s % 4),
// a permutation inside the 16-block that the host applies to the weight columns.  Per hidden tile: tile 0 = layer 1 (B = z planes),
// relu, SPLIT of the 32 x 32 activation tile into planes (11 VALU instructions per pair of values), tiles 1..8 = layer-2 row tiles.
// FLOP are counted as fp32-EQUIVALENT (65 536 per tile and wave, as for shape 1), so the figure compares directly with shapes 1..5.
// WPS = waves per SIMD the kernel is compiled for: 2 (256 VGPRs, two workgroups per CU) or 1 (512 VGPRs, one workgroup per CU).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
struct B3 { bf16x8 p[3][2]; };   // [plane][k block]
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ void split3(const f32x16& x, B3& o) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        unsigned hi[4], mi[4], lo[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float x0 = x[8 * kb + 2 * p], x1 = x[8 * kb + 2 * p + 1];
            f32x2v v = {x0, x1};
            const bf16x2 h = __builtin_convertvector(v, bf16x2);
            hi[p] = __builtin_bit_cast(unsigned, h);
            const float r0 = x0 - bf_lo(hi[p]), r1 = x1 - bf_hi(hi[p]);
            f32x2v rv = {r0, r1};
            const bf16x2 m = __builtin_convertvector(rv, bf16x2);
            mi[p] = __builtin_bit_cast(unsigned, m);
            f32x2v qv = {r0 - bf_lo(mi[p]), r1 - bf_hi(mi[p])};
            const bf16x2 l = __builtin_convertvector(qv, bf16x2);
            lo[p] = __builtin_bit_cast(unsigned, l);
        }
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 H = {hi[0], hi[1], hi[2], hi[3]}, M = {mi[0], mi[1], mi[2], mi[3]}, Lo = {lo[0], lo[1], lo[2], lo[3]};
        o.p[0][kb] = __builtin_bit_cast(bf16x8, H);
        o.p[1][kb] = __builtin_bit_cast(bf16x8, M);
        o.p[2][kb] = __builtin_bit_cast(bf16x8, Lo);
    }
}
// acc += tile (6 fragments at t: [kb][plane][64 lanes]) x B
__device__ __forceinline__ void tile_mma_b3(f32x16& acc, const f32x4* __restrict__ t, const B3& B) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const bf16x8 ah = __builtin_bit_cast(bf16x8, t[(kb * 3 + 0) * 64]);
        const bf16x8 am = __builtin_bit_cast(bf16x8, t[(kb * 3 + 1) * 64]);
        const bf16x8 al = __builtin_bit_cast(bf16x8, t[(kb * 3 + 2) * 64]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, B.p[0][kb], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, B.p[2][kb], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, B.p[1][kb], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, B.p[0][kb], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, B.p[1][kb], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, B.p[0][kb], acc, 0, 0, 0);
    }
}
template <int C, int WPS>
__global__ __launch_bounds__(256, WPS) void diag_stream_b3_kernel(const f32x4* __restrict__ blob, int total_tiles, float* __restrict__ out,
                                                                  int ngroups, float seed) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* lds = reinterpret_cast<f32x4*>(smem);
    constexpr int TILE = 384;  // f32x4 per tile (6 KiB)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    auto dma = [&](int chunk, int buf) {
        const f32x4* src = blob + ((size_t)chunk * C % total_tiles) * TILE + lane;
        f32x4* dst = lds + buf * (C * TILE);
#pragma unroll
        for (int i = 0; i < (6 * C + 3) / 4; ++i) {
            const int idx = i * 4 + wave;
            if (idx < 6 * C) glds16(src + idx * 64, dst + idx * 64);
        }
    };
    f32x16 acc2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[i][e] = seed * (float)(i + 1);
    f32x16 zf;
#pragma unroll
    for (int e = 0; e < 16; ++e) zf[e] = 1.0f - 0.002f * (float)(lane + e);
    B3 Bz, Bh;
    split3(zf, Bz);
    Bh = Bz;
    f32x16 h1;
    dma(blockIdx.x, 0);
    __syncthreads();
    int chunk = 0;
    for (int grp = 0; grp < ngroups; ++grp) {
#pragma unroll
        for (int cc = 0; cc < 9; ++cc) {
            __builtin_amdgcn_sched_barrier(0);
            dma(blockIdx.x + chunk + 1, (chunk + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            const f32x4* buf = lds + (chunk & 1) * (C * TILE) + lane;
#pragma unroll
            for (int t = 0; t < C; ++t) {
                const int ti = (cc * C + t) % 9;  // 0: layer 1, 1..8: layer 2 row tile ti-1
                if (ti == 0) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) h1[e] = 0.f;
                    tile_mma_b3(h1, buf + t * TILE, Bz);
#pragma unroll
                    for (int e = 0; e < 16; ++e) h1[e] = fmaxf(h1[e], 0.f);
                    split3(h1, Bh);
                } else {
                    tile_mma_b3(acc2[ti - 1], buf + t * TILE, Bh);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
            ++chunk;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc2[i][e];
    if (s == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

struct StreamCtx { int shape, grid, lds, n, total; const f32x4* blob; float* scratch; hipStream_t s; };
static void stream_go(void* p) {
    StreamCtx* c = (StreamCtx*)p;
    const dim3 g(c->grid), b(256);
    switch (c->shape) {
        case 0: hipLaunchKernelGGL(diag_stream16_kernel, g, b, c->lds, c->s, c->blob, c->total, c->scratch, c->n, 0.5f); break;
        case 1: hipLaunchKernelGGL((diag_stream32_kernel<3, 0>), g, b, c->lds, c->s, c->blob, c->total, c->scratch, c->n, 0.5f); break;
        case 2: hipLaunchKernelGGL((diag_stream32_kernel<3, 1>), g, b, c->lds, c->s, c->blob, c->total, c->scratch, c->n, 0.5f); break;
        case 3: hipLaunchKernelGGL((diag_stream32_kernel<9, 0>), g, b, c->lds, c->s, c->blob, c->total, c->scratch, c->n, 0.5f); break;
        case 4: hipLaunchKernelGGL((diag_stream32_kernel<9, 1>), g, b, c->lds, c->s, c->blob, c->total, c->scratch, c->n, 0.5f); break;
        case 5: hipLaunchKernelGGL(diag_stream32i_kernel, g, b, c->lds, c->s, c->blob, c->total, c->scratch, c->n, 0.5f); break;
        case 6: hipLaunchKernelGGL((diag_stream_b3_kernel<3, 2>), g, b, c->lds, c->s, c->blob, c->total, c->scratch, c->n, 0.5f); break;
        default: hipLaunchKernelGGL((diag_stream_b3_kernel<3, 1>), g, b, c->lds, c->s, c->blob, c->total, c->scratch, c->n, 0.5f); break;
    }
}

// shape: 0 16x16x4 stream (3 WG/CU) | 1 32x32x2 C=3 | 2 32x32x2 C=3 prefetched | 3 32x32x2 C=9 | 4 32x32x2 C=9 prefetched | 5 C=9, two
// interleaved accumulator chains
// (2 WG/CU).  wgs_per_cu <= 3 (shape 0) / 2 (others); blob: >= 2 MiB of random weights (L2 resident); n: chunks (shape 0) or
// 9-chunk groups (others) per workgroup.
extern "C" int sttode_diag_stream(int shape, int wgs_per_cu, int n, int repeats, const float* blob, long blob_floats, float* scratch,
                                  double* tflops, void* stream) {
    DIAG_REQUIRE(blob && scratch && tflops && n > 0 && repeats > 0 && shape >= 0 && shape <= 7, "sttode_diag_stream: bad arguments");
    DIAG_REQUIRE(wgs_per_cu >= 1 && wgs_per_cu <= (shape == 0 ? 3 : 2), "sttode_diag_stream: too many workgroups per CU");
    DIAG_REQUIRE(blob_floats >= 512 * 1024, "sttode_diag_stream: blob must hold >= 2 MiB");
    StreamCtx c;
    c.shape = shape; c.grid = wgs_per_cu * num_cus(); c.n = n; c.blob = (const f32x4*)blob; c.scratch = scratch; c.s = (hipStream_t)stream;
    double flop_per_wg;
    if (shape == 0) {
        c.lds = 2 * S0_CHW * 16; c.total = (int)(blob_floats / 4 / S0_CHW);
        flop_per_wg = 4.0 * n * 72.0 * 2048.0;
        DIAG_HIP(hipFuncSetAttribute((const void*)diag_stream16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, c.lds));
    } else if (shape >= 6) {   // bf16 three-way split: 6 KiB tiles, 18 KiB chunks; FLOP counted as fp32-equivalent (65 536 per tile and wave)
        const int C = 3;
        c.lds = 2 * C * 6144; c.total = (int)(blob_floats / 4 / 384) / C * C;
        flop_per_wg = 4.0 * n * 9.0 * C * 16.0 * 4096.0;
        const void* f = shape == 6 ? (const void*)diag_stream_b3_kernel<3, 2> : (const void*)diag_stream_b3_kernel<3, 1>;
        DIAG_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, c.lds));
    } else {
        const int C = shape <= 2 ? 3 : 9;
        c.lds = 2 * C * 4096; c.total = (int)(blob_floats / 4 / 256) / C * C;
        flop_per_wg = 4.0 * n * 9.0 * C * 16.0 * 4096.0;
        const void* f = shape == 1 ? (const void*)diag_stream32_kernel<3, 0> : shape == 2 ? (const void*)diag_stream32_kernel<3, 1>
                      : shape == 3 ? (const void*)diag_stream32_kernel<9, 0> : shape == 4 ? (const void*)diag_stream32_kernel<9, 1>
                      : (const void*)diag_stream32i_kernel;
        DIAG_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, c.lds));
    }
    double ms = 0;
    if (int rc = time_launches(c.s, repeats, &ms, stream_go, &c)) return rc;
    *tflops = (double)repeats * c.grid * flop_per_wg / (ms * 1e-3) / 1e12;
    return 0;
}
