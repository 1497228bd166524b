// Stage-2 latent sampler support kernels.
//   reference: sampler.py:47-54 (z = A*eps + b, logvar = log(A^2 + 1e-8)),
//              utils/dist.py:22-30 (KL(q || p) of two diagonal normals),
//              samplerloss.py:4-20 (per-agent KL sum; diversity = mean over the K(K-1)/2 sample pairs of exp(-|a-b|^2 / scale)).
// The Q-net itself (linear 128->64, tanh MLP, q_A / q_b / q_c) runs on sttode_linear_cols (decoder.hip).
// All of this is O(n*K*nz) elementwise / O(n*K^2*Tf) pairwise work: HBM/VALU bound, one wave per agent for the losses.
#include "api_util.hpp"

__global__ void sampler_latent_kernel(const float* __restrict__ A, const float* __restrict__ b, const float* __restrict__ eps,
                                      int eps_mode, float* __restrict__ z, float* __restrict__ logvar, long total, int nz, int K) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float a = A[i], bb = b[i];
    float zz = bb;
    if (eps_mode != 0) {
        const int d = (int)(i % nz);
        const long row = i / nz;  // agent*K + k
        const float e = eps_mode == 1 ? eps[d] : eps[(row / K) * nz + d];
        zz = a * e + bb;
    }
    z[i] = zz;
    logvar[i] = logf(a * a + 1e-8f);
}

static __device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// one wave per agent; motion [n][K][D] staged in LDS (K*D <= 4096 floats)
__global__ __launch_bounds__(64) void sampler_loss_kernel(const float* __restrict__ mu, const float* __restrict__ logvar,
                                                          const float* __restrict__ pmu, const float* __restrict__ plogvar,
                                                          const float* __restrict__ motion, int n, int K, int nz, int D,
                                                          float scale, float* __restrict__ kld, float* __restrict__ div) {
    __shared__ float sM[4096];
    const int a = blockIdx.x, lane = threadIdx.x;
    // KL(q || p), utils/dist.py:26-29
    float kl = 0.f;
    const size_t base = (size_t)a * K * nz;
    for (int i = lane; i < K * nz; i += 64) {
        const float m = mu[base + i], sg = expf(0.5f * logvar[base + i]);
        const float pm = pmu ? pmu[base + i] : 0.f;
        const float ps = (plogvar ? expf(0.5f * plogvar[base + i]) : 1.f) + 1e-8f;
        const float t1 = (m - pm) / ps, t2 = sg / ps;
        kl += 0.5f * (t1 * t1 + t2 * t2) - 0.5f - logf(t2);
    }
    kl = wave_sum(kl);
    for (int i = lane; i < K * D; i += 64) sM[i] = motion[(size_t)a * K * D + i];
    __syncthreads();
    // pairs (i < j) in F.pdist order; the order does not matter for the mean
    const int np = K * (K - 1) / 2;
    float acc = 0.f;
    for (int p = lane; p < np; p += 64) {
        int i = 0, r = p;
        while (r >= K - 1 - i) { r -= K - 1 - i; ++i; }
        const int j = i + 1 + r;
        float s = 0.f;
        for (int d = 0; d < D; ++d) {
            const float t = sM[i * D + d] - sM[j * D + d];
            s += t * t;
        }
        const float dist = sqrtf(s);  // F.pdist(p=2) ... ** 2 (samplerloss.py:17)
        acc += expf(-(dist * dist) / scale);
    }
    acc = wave_sum(acc);
    if (lane == 0) {
        kld[a] = kl;
        div[a] = acc / (float)np;
    }
}

// backward of sampler_loss_kernel: upstream gradients g_kld[a], g_div[a] of the per-agent outputs
__global__ __launch_bounds__(64) void sampler_loss_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ logvar,
                                                              const float* __restrict__ pmu, const float* __restrict__ plogvar,
                                                              const float* __restrict__ motion, const float* __restrict__ g_kld,
                                                              const float* __restrict__ g_div, int n, int K, int nz, int D, float scale,
                                                              float* __restrict__ dmu, float* __restrict__ dlogvar,
                                                              float* __restrict__ dmotion) {
    __shared__ float sM[4096];
    const int a = blockIdx.x, lane = threadIdx.x;
    const size_t base = (size_t)a * K * nz;
    const float gk = g_kld[a], gd = g_div[a];
    for (int i = lane; i < K * nz; i += 64) {
        const float ps = (plogvar ? expf(0.5f * plogvar[base + i]) : 1.f) + 1e-8f;
        const float t1 = (mu[base + i] - (pmu ? pmu[base + i] : 0.f)) / ps, t2 = expf(0.5f * logvar[base + i]) / ps;
        dmu[base + i] = gk * t1 / ps;
        dlogvar[base + i] = gk * 0.5f * (t2 * t2 - 1.0f);
    }
    for (int i = lane; i < K * D; i += 64) sM[i] = motion[(size_t)a * K * D + i];
    __syncthreads();
    const float coef = gd * (-2.0f / scale) / (float)(K * (K - 1) / 2);
    // thread = (sample i, coordinate d): d/dm_i = coef * sum_{j != i} exp(-|m_i - m_j|^2 / scale) * (m_i - m_j)
    for (int e = lane; e < K * D; e += 64) {
        const int i = e / D, d = e % D;
        float acc = 0.f;
        for (int j = 0; j < K; ++j) {
            if (j == i) continue;
            float s = 0.f;
            for (int t = 0; t < D; ++t) {
                const float df = sM[i * D + t] - sM[j * D + t];
                s += df * df;
            }
            if (s > 0.f) acc += expf(-s / scale) * (sM[i * D + d] - sM[j * D + d]);   // F.pdist backward is 0 at zero distance
        }
        dmotion[(size_t)a * K * D + e] = coef * acc;
    }
}

extern "C" int sttode_sampler_latent(const float* A, const float* b, const float* eps, int eps_mode, float* z, float* logvar,
                                     int n, int K, int nz, void* stream) {
    STT_REQUIRE(A && b && z && logvar, "sttode_sampler_latent: null pointer");
    STT_REQUIRE(eps_mode >= 0 && eps_mode <= 2 && (eps_mode == 0 || eps), "sttode_sampler_latent: eps_mode 0 (mean) | 1 (shared [nz]) | 2 (per agent [n,nz]); eps required unless 0");
    STT_REQUIRE(n > 0 && K > 0 && nz > 0, "sttode_sampler_latent: n, K, nz must be positive");
    const long total = (long)n * K * nz;
    hipLaunchKernelGGL(sampler_latent_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, A, b, eps,
                       eps_mode, z, logvar, total, nz, K);
    STT_HIP(hipGetLastError());
    return 0;
}

extern "C" int sttode_sampler_loss(const float* mu, const float* logvar, const float* pmu, const float* plogvar,
                                   const float* motion, int n, int K, int nz, int D, float scale, float* kld, float* div,
                                   void* stream) {
    STT_REQUIRE(mu && logvar && motion && kld && div, "sttode_sampler_loss: null pointer");
    STT_REQUIRE(n > 0 && K > 1 && nz > 0 && D > 0, "sttode_sampler_loss: n, nz, D must be positive and K > 1");
    STT_REQUIRE((long)K * D <= 4096, "sttode_sampler_loss: K * D must be <= 4096");
    STT_REQUIRE(scale > 0.f, "sttode_sampler_loss: scale must be positive");
    hipLaunchKernelGGL(sampler_loss_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, mu, logvar, pmu, plogvar, motion, n, K, nz,
                       D, scale, kld, div);
    STT_HIP(hipGetLastError());
    return 0;
}

extern "C" int sttode_sampler_loss_bwd(const float* mu, const float* logvar, const float* pmu, const float* plogvar,
                                       const float* motion, const float* g_kld, const float* g_div, int n, int K, int nz, int D,
                                       float scale, float* dmu, float* dlogvar, float* dmotion, void* stream) {
    STT_REQUIRE(mu && logvar && motion && g_kld && g_div && dmu && dlogvar && dmotion, "sttode_sampler_loss_bwd: null pointer");
    STT_REQUIRE(n > 0 && K > 1 && nz > 0 && D > 0 && (long)K * D <= 4096 && scale > 0.f, "sttode_sampler_loss_bwd: bad shape");
    hipLaunchKernelGGL(sampler_loss_bwd_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, mu, logvar, pmu, plogvar, motion, g_kld,
                       g_div, n, K, nz, D, scale, dmu, dlogvar, dmotion);
    STT_HIP(hipGetLastError());
    return 0;
}
