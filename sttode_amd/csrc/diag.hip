// Diagnostics (not on the product path): sustained issue-rate ceiling of v_mfma_f32_16x16x4_f32 on this device,
// operands in registers only (no LDS / memory), to separate "clock held under load" from "non-MFMA overhead" when
// reading the roofline fractions of the real kernels.
#include "chain.hpp"
#include "api_util.hpp"

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

// Same loop, but every MFMA quad takes its A fragment from LDS with one ds_read_b128 (the operand traffic of the real
// column-chain kernels: 1 KiB per wave per 4 MFMAs), fragments spread over a 64 KiB image.
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

// v_mfma_f32_32x32x2_f32 variants: same FLOP rate on paper (4096 FLOP in 16 passes), but half the MFMA instructions and half
// the A-operand bytes per FLOP (one ds_read_b128 feeds 4 MFMAs = 16 384 FLOP instead of 8 192) -- measured to decide whether a
// 32-column re-tiling of the column chain would buy sustained throughput (power-limited clocks) on this part.
typedef float f32x16 __attribute__((ext_vector_type(16)));

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
        for (int rep = 0; rep < 2; ++rep)
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
        for (int rep = 0; rep < 2; ++rep)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 a = base[(rep * 4 + i) * 64];
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

// kind: 0 16x16x4 registers only | 1 16x16x4 + one ds_read_b128 per 4 MFMAs | 2 32x32x2 registers only | 3 32x32x2 + ds_read_b128
// per 4 MFMAs.  waves_per_cu in {4, 8, 12, 16}.  Same FLOP per iteration (65 536 per wave) in all four.
extern "C" int sttode_diag_mfma_kinds(int kind, int waves_per_cu, int iters, int repeats, float* scratch, double* tflops, void* stream) {
    STT_REQUIRE(scratch && tflops && iters > 0 && repeats > 0 && kind >= 0 && kind <= 3, "sttode_diag_mfma_kinds: bad arguments");
    STT_REQUIRE(waves_per_cu == 4 || waves_per_cu == 8 || waves_per_cu == 12 || waves_per_cu == 16, "sttode_diag_mfma_kinds: waves_per_cu must be 4, 8, 12 or 16");
    hipStream_t s = (hipStream_t)stream;
    int dev = 0;
    hipDeviceProp_t p;
    STT_HIP(hipGetDevice(&dev));
    STT_HIP(hipGetDeviceProperties(&p, dev));
    const int cus = p.multiProcessorCount;
    hipEvent_t e0, e1;
    STT_HIP(hipEventCreate(&e0));
    STT_HIP(hipEventCreate(&e1));
    auto go = [&]() {
        const dim3 g(cus), b(64 * waves_per_cu);
        switch (kind) {
            case 0: hipLaunchKernelGGL(diag_mfma_kernel, g, b, 0, s, scratch, iters, 0.5f); break;
            case 1: hipLaunchKernelGGL(diag_mfma_lds_kernel, g, b, 65536, s, scratch, iters, 0.5f); break;
            case 2: hipLaunchKernelGGL(diag_mfma32_kernel, g, b, 0, s, scratch, iters, 0.5f); break;
            default: hipLaunchKernelGGL(diag_mfma32_lds_kernel, g, b, 65536, s, scratch, iters, 0.5f); break;
        }
    };
    go();
    STT_HIP(hipEventRecord(e0, s));
    for (int r = 0; r < repeats; ++r) go();
    STT_HIP(hipEventRecord(e1, s));
    STT_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    STT_HIP(hipEventElapsedTime(&ms, e0, e1));
    *tflops = (double)repeats * cus * waves_per_cu * (double)iters * 65536.0 / (ms * 1e-3) / 1e12;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return 0;
}

// Runs the register-only MFMA loop on every CU with `waves_per_cu` waves and returns the achieved TFLOP/s.
extern "C" int sttode_diag_mfma_peak(int waves_per_cu, int iters, int repeats, float* scratch, double* tflops, void* stream) {
    const bool with_lds = waves_per_cu < 0;
    if (with_lds) waves_per_cu = -waves_per_cu;
    STT_REQUIRE(scratch && tflops && iters > 0 && repeats > 0, "sttode_diag_mfma_peak: bad arguments");
    STT_REQUIRE(waves_per_cu == 4 || waves_per_cu == 8 || waves_per_cu == 16, "sttode_diag_mfma_peak: waves_per_cu must be 4, 8 or 16");
    hipStream_t s = (hipStream_t)stream;
    int dev = 0;
    hipDeviceProp_t p;
    STT_HIP(hipGetDevice(&dev));
    STT_HIP(hipGetDeviceProperties(&p, dev));
    const int cus = p.multiProcessorCount;
    hipEvent_t e0, e1;
    STT_HIP(hipEventCreate(&e0));
    STT_HIP(hipEventCreate(&e1));
    auto go = [&]() {
        if (with_lds) hipLaunchKernelGGL(diag_mfma_lds_kernel, dim3(cus), dim3(64 * waves_per_cu), 65536, s, scratch, iters, 0.5f);
        else hipLaunchKernelGGL(diag_mfma_kernel, dim3(cus), dim3(64 * waves_per_cu), 0, s, scratch, iters, 0.5f);
    };
    go();  // warm-up
    STT_HIP(hipEventRecord(e0, s));
    for (int r = 0; r < repeats; ++r) go();
    STT_HIP(hipEventRecord(e1, s));
    STT_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    STT_HIP(hipEventElapsedTime(&ms, e0, e1));
    const double flop = (double)repeats * cus * waves_per_cu * (double)iters * 32.0 * 2048.0;  // 32 MFMAs/iter, 2048 FLOP each
    *tflops = flop / (ms * 1e-3) / 1e12;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return 0;
}
