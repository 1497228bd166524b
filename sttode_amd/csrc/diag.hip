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
