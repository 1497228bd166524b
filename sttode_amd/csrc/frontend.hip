// Batched scene front-end (set_data / set_data_nba equivalents) and best-of-K metrics.
//   reference: model/STTODE.py:397-461 (scene normalisation, velocities, cur_location),
//              model/STTODE.py:463-486,578-596 (NBA branch / inference() input assembly),
//              utils/metrics.py:7-26 (min-over-K ADE / FDE).
// HBM-bound byte shuffling: one thread per scene / agent, coalesced over agents.
#include "api_util.hpp"
#include "frontend_body.hpp"
#include "../../include/sttode_hip.h"
#include <string>

static thread_local std::string g_err;
void stt_set_error(const char* msg) { g_err = msg ? msg : ""; }
extern "C" const char* sttode_last_error() { return g_err.c_str(); }
// bumped whenever an exported signature or an ABI enum (SttodeWeight / SttodeBuffer / SttodeStage) changes; capi.py checks it at load
extern "C" int sttode_abi_version() { return STTODE_ABI_VERSION; }

__global__ void scene_orig_kernel(const float* __restrict__ past, const int* __restrict__ scene_ptr, int S, int Tp,
                                  float* __restrict__ scene_orig, int* __restrict__ agent_scene) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const int a0 = scene_ptr[s], a1 = scene_ptr[s + 1];
    float sx = 0.f, sy = 0.f;
    for (int a = a0; a < a1; ++a) {
        sx += past[((size_t)a * Tp + (Tp - 1)) * 2 + 0];
        sy += past[((size_t)a * Tp + (Tp - 1)) * 2 + 1];
        agent_scene[a] = s;
    }
    const float inv = (float)(a1 - a0);
    scene_orig[2 * s] = sx / inv;
    scene_orig[2 * s + 1] = sy / inv;
}

// mode 0: ETH/UCY/SDD (normalise by scene_orig, flag last agent of each scene)
// mode 1: NBA (no normalisation, flag slot N-1)
__device__ __forceinline__ void agent_inputs_one(int a, const float* __restrict__ seq, int n, int T, int TPX, int mode, int vel_from_norm,
                                                 const float* __restrict__ prev_last,  // optional [n][2]: frame preceding seq (future encoder), world coords
                                                 const float* __restrict__ scene_orig, const int* __restrict__ agent_scene,
                                                 const int* __restrict__ scene_ptr, int nba_N, float* __restrict__ xpad,
                                                 float* __restrict__ enc_in, float* __restrict__ cur, float* __restrict__ orig,
                                                 int* __restrict__ last_flag) {
    float ox = 0.f, oy = 0.f;
    int last;
    if (mode == 0) {
        const int s = agent_scene[a];
        ox = scene_orig[2 * s];
        oy = scene_orig[2 * s + 1];
        last = (a == scene_ptr[s + 1] - 1);
    } else {
        last = (a % nba_N == nba_N - 1);
    }
    agent_inputs_core<false>(a, seq, T, TPX, vel_from_norm, ox, oy, last, prev_last, xpad, enc_in, cur, orig, last_flag);
}
__global__ void agent_inputs_kernel(const float* __restrict__ seq, int n, int T, int TPX, int mode, int vel_from_norm,
                                    const float* __restrict__ prev_last, const float* __restrict__ scene_orig,
                                    const int* __restrict__ agent_scene, const int* __restrict__ scene_ptr, int nba_N,
                                    float* __restrict__ xpad, float* __restrict__ enc_in, float* __restrict__ cur,
                                    float* __restrict__ orig, int* __restrict__ last_flag) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n) return;
    agent_inputs_one(a, seq, n, T, TPX, mode, vel_from_norm, prev_last, scene_orig, agent_scene, scene_ptr, nba_N, xpad, enc_in, cur, orig, last_flag);
}
// Few scenes (the per-scene loop of test.py:171-188): both steps in ONE single-workgroup launch -- thread s sums scene s in agent order
// exactly as scene_orig_kernel does, a barrier, then the threads stride the agents.
__global__ __launch_bounds__(256) void frontend_small_kernel(const float* __restrict__ past, const int* __restrict__ scene_ptr, int n, int S,
                                                             int Tp, int TPX, int vel_from_norm, float* __restrict__ scene_orig,
                                                             int* __restrict__ agent_scene, float* __restrict__ xpad,
                                                             float* __restrict__ enc_in, float* __restrict__ cur, float* __restrict__ orig,
                                                             int* __restrict__ last_flag) {
    for (int s = threadIdx.x; s < S; s += 256) {
        const int a0 = scene_ptr[s], a1 = scene_ptr[s + 1];
        float sx = 0.f, sy = 0.f;
        for (int a = a0; a < a1; ++a) {
            sx += past[((size_t)a * Tp + (Tp - 1)) * 2 + 0];
            sy += past[((size_t)a * Tp + (Tp - 1)) * 2 + 1];
            agent_scene[a] = s;
        }
        const float inv = (float)(a1 - a0);
        scene_orig[2 * s] = sx / inv;
        scene_orig[2 * s + 1] = sy / inv;
    }
    __syncthreads();   // scene_orig / agent_scene written above are visible to the workgroup
    for (int a = threadIdx.x; a < n; a += 256)
        agent_inputs_one(a, past, n, Tp, TPX, 0, vel_from_norm, nullptr, scene_orig, agent_scene, scene_ptr, 1, xpad, enc_in, cur, orig, last_flag);
}

// one wave per agent: lanes stride the K*Tf displacement norms (coalesced 8-byte reads), per-sample sums by a
// segmented pass through LDS-free shuffles is overkill here: K*Tf <= a few hundred, so each lane owns whole samples
// k = lane, lane+64, ... only when K > 64; otherwise lanes split (k, t) pairs and reduce with xor-shuffles per sample.
__global__ __launch_bounds__(256) void best_of_k_kernel(const float* __restrict__ pred, const float* __restrict__ gt, int n, int K,
                                                        int Tf, float scale, float* __restrict__ ade, float* __restrict__ fde) {
    __shared__ float sd[4][1024];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int a = blockIdx.x * 4 + w;
    if (a >= n) return;
    const float2* g = reinterpret_cast<const float2*>(gt + (size_t)a * Tf * 2);
    const float2* p = reinterpret_cast<const float2*>(pred + (size_t)a * K * Tf * 2);
    const int tot = K * Tf;
    float best_a = INFINITY, best_f = INFINITY;
    if (tot <= 1024) {
        // stage all K*Tf distances in LDS (coalesced global reads), then lanes reduce whole samples
        for (int i = lane; i < tot; i += 64) {
            const float2 v = p[i], r = g[i % Tf];
            sd[w][i] = bok_dist(v.x, v.y, r.x, r.y, scale);
        }
        __builtin_amdgcn_wave_barrier();
        for (int k = lane; k < K; k += 64) {
            float sum = 0.f;
            for (int t = 0; t < Tf; ++t) sum += sd[w][k * Tf + t];
            best_a = fminf(best_a, sum / (float)Tf);
            best_f = fminf(best_f, sd[w][k * Tf + Tf - 1]);
        }
    } else {
        for (int k = lane; k < K; k += 64) {
            float sum = 0.f, dl = 0.f;
            for (int t = 0; t < Tf; ++t) {
                const float2 v = p[k * Tf + t], r = g[t];
                dl = bok_dist(v.x, v.y, r.x, r.y, scale);
                sum += dl;
            }
            best_a = fminf(best_a, sum / (float)Tf);
            best_f = fminf(best_f, dl);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        best_a = fminf(best_a, __shfl_xor(best_a, o, 64));
        best_f = fminf(best_f, __shfl_xor(best_f, o, 64));
    }
    if (lane == 0) { ade[a] = best_a; fde[a] = best_f; }
}

// NBA evaluation metric (test.py:530-551): for every horizon h = 1 .. Tf, min over the K samples of the mean displacement over the first h
// frames and of the displacement of frame h.  One wave per agent: the K Tf displacement norms are staged in LDS (coalesced 8-byte reads,
// best_of_k_kernel's bok_dist), lane k walks sample k's frames with a running fp32 sum in frame order (sum_h / h: the NumPy restatement
// tests/helpers.py horizon_metrics_np sums in the same order), and the wave takes the minimum per frame by xor shuffles.
// out [n][Tf][2] = (avg_h, dest_h).  K <= 64, K Tf <= 2048.
__global__ __launch_bounds__(256) void horizon_metrics_kernel(const float* __restrict__ pred, const float* __restrict__ gt, int n, int K,
                                                              int Tf, float scale, float* __restrict__ out) {
    __shared__ float sd[4][2048];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int a = blockIdx.x * 4 + w;
    if (a >= n) return;
    const float2* g = reinterpret_cast<const float2*>(gt + (size_t)a * Tf * 2);
    const float2* p = reinterpret_cast<const float2*>(pred + (size_t)a * K * Tf * 2);
    const int tot = K * Tf;
    for (int i = lane; i < tot; i += 64) {
        const float2 v = p[i], r = g[i % Tf];
        sd[w][i] = bok_dist(v.x, v.y, r.x, r.y, scale);
    }
    __builtin_amdgcn_wave_barrier();
    float sum = 0.f;
    for (int t = 0; t < Tf; ++t) {
        float va = INFINITY, vd = INFINITY;
        if (lane < K) {
            vd = sd[w][lane * Tf + t];
            sum += vd;
            va = sum / (float)(t + 1);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            va = fminf(va, __shfl_xor(va, o, 64));
            vd = fminf(vd, __shfl_xor(vd, o, 64));
        }
        if (lane == 0) {
            out[((size_t)a * Tf + t) * 2] = va;
            out[((size_t)a * Tf + t) * 2 + 1] = vd;
        }
    }
}

// Train-mode augmentation of set_data (model/STTODE.py:417-426): the scene's tracks rotated IN PLACE about scene_orig = mean over the agents
// of the last observed position, x' = R (x - orig) + orig, R = [[c, -s], [s, c]] (rotation_2d_torch, :6-14).  One workgroup: every thread
// forms the mean itself, in agent order, before any position is overwritten.
__global__ __launch_bounds__(256) void rotate_scene_kernel(float* __restrict__ past, float* __restrict__ fut, int n, int Tp, int Tf, float c, float s) {
    float sx = 0.f, sy = 0.f;
    for (int a = 0; a < n; ++a) {
        sx += past[((size_t)a * Tp + (Tp - 1)) * 2 + 0];
        sy += past[((size_t)a * Tp + (Tp - 1)) * 2 + 1];
    }
    const float ox = sx / (float)n, oy = sy / (float)n;
    __syncthreads();
    for (int i = threadIdx.x; i < n * (Tp + Tf); i += 256) {
        float* p = i < n * Tp ? past + (size_t)i * 2 : fut + (size_t)(i - n * Tp) * 2;
        const float dx = p[0] - ox, dy = p[1] - oy;
        p[0] = (dx * c + dy * (-s)) + ox;
        p[1] = (dx * s + dy * c) + oy;
    }
}
extern "C" int sttode_rotate_scene(float* past, float* fut, int n, int Tp, int Tf, float c, float s, void* stream) {
    STT_REQUIRE(past && n > 0 && Tp >= 1 && Tf >= 0 && (fut || Tf == 0), "sttode_rotate_scene: bad argument");
    hipLaunchKernelGGL(rotate_scene_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, past, fut, n, Tp, fut ? Tf : 0, c, s);
    STT_HIP(hipGetLastError());
    return 0;
}

extern "C" int sttode_frontend_scenes(const float* past, const int* scene_ptr, int n, int S, int Tp, int TPX, int vel_from_norm,
                                      float* scene_orig, int* agent_scene, float* xpad, float* enc_in, float* cur, float* orig,
                                      int* last_flag, void* stream) {
    STT_REQUIRE(past && scene_ptr && scene_orig && agent_scene && xpad && enc_in && cur && orig && last_flag, "sttode_frontend_scenes: null pointer");
    STT_REQUIRE(n > 0 && S > 0 && Tp >= 2 && 2 * Tp <= 16 * TPX, "sttode_frontend_scenes: need n,S > 0, Tp >= 2, 2*Tp <= 16*TPX");
    hipStream_t s = (hipStream_t)stream;
    if (S <= 16 && n <= 1024) {   // a scene or a handful: one single-workgroup launch
        hipLaunchKernelGGL(frontend_small_kernel, dim3(1), dim3(256), 0, s, past, scene_ptr, n, S, Tp, TPX, vel_from_norm, scene_orig,
                           agent_scene, xpad, enc_in, cur, orig, last_flag);
        STT_HIP(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(scene_orig_kernel, dim3((S + 127) / 128), dim3(128), 0, s, past, scene_ptr, S, Tp, scene_orig, agent_scene);
    hipLaunchKernelGGL(agent_inputs_kernel, dim3((n + 127) / 128), dim3(128), 0, s, past, n, Tp, TPX, 0, vel_from_norm,
                       (const float*)nullptr, scene_orig, agent_scene, scene_ptr, 1, xpad, enc_in, cur, orig, last_flag);
    STT_HIP(hipGetLastError());
    return 0;
}

extern "C" int sttode_frontend_nba(const float* past, int n, int N, int Tp, int TPX, float* xpad, float* enc_in, float* cur,
                                   float* orig, int* last_flag, void* stream) {
    STT_REQUIRE(past && xpad && enc_in && cur && orig && last_flag, "sttode_frontend_nba: null pointer");
    STT_REQUIRE(n > 0 && N > 0 && n % N == 0 && Tp >= 2 && 2 * Tp <= 16 * TPX, "sttode_frontend_nba: need n % N == 0, Tp >= 2, 2*Tp <= 16*TPX");
    hipLaunchKernelGGL(agent_inputs_kernel, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, past, n, Tp, TPX, 1, 1,
                       (const float*)nullptr, (const float*)nullptr, (const int*)nullptr, (const int*)nullptr, N, xpad, enc_in, cur,
                       orig, last_flag);
    STT_HIP(hipGetLastError());
    return 0;
}

// inputs of the posterior (future) encoder: normalised future + velocity w.r.t. the preceding frame
// (model/STTODE.py:434,457 ; :477,481).  mode as above; writes enc_in [n][Tf][4] only.
extern "C" int sttode_frontend_future(const float* future, const float* past_last, int n, int Tf, int mode, int nba_N,
                                      const float* scene_orig, const int* agent_scene, const int* scene_ptr, float* enc_in,
                                      void* stream) {
    STT_REQUIRE(future && past_last && enc_in, "sttode_frontend_future: null pointer");
    STT_REQUIRE(n > 0 && Tf >= 1 && (mode == 1 || (scene_orig && agent_scene && scene_ptr)), "sttode_frontend_future: bad arguments");
    hipLaunchKernelGGL(agent_inputs_kernel, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, future, n, Tf, 1, mode, 0,
                       past_last, scene_orig, agent_scene, scene_ptr, nba_N > 0 ? nba_N : 1, (float*)nullptr, enc_in,
                       (float*)nullptr, (float*)nullptr, (int*)nullptr);
    STT_HIP(hipGetLastError());
    return 0;
}

extern "C" int sttode_best_of_k(const float* pred, const float* gt, int n, int K, int Tf, float scale, float* ade, float* fde,
                                void* stream) {
    STT_REQUIRE(pred && gt && ade && fde, "sttode_best_of_k: null pointer");
    STT_REQUIRE(n > 0 && K > 0 && Tf > 0, "sttode_best_of_k: n, K, Tf must be positive");
    hipLaunchKernelGGL(best_of_k_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, pred, gt, n, K, Tf, scale, ade, fde);
    STT_HIP(hipGetLastError());
    return 0;
}

extern "C" int sttode_horizon_metrics(const float* pred, const float* gt, int n, int K, int Tf, float scale, float* out, void* stream) {
    STT_REQUIRE(pred && gt && out, "sttode_horizon_metrics: null pointer");
    STT_REQUIRE(n > 0 && K > 0 && K <= 64 && Tf > 0 && K * Tf <= 2048, "sttode_horizon_metrics: need 0 < K <= 64 and K * Tf <= 2048");
    hipLaunchKernelGGL(horizon_metrics_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, pred, gt, n, K, Tf, scale, out);
    STT_HIP(hipGetLastError());
    return 0;
}

// Shader clock right now: one lane counts shader cycles (s_memtime) over ~20 us of the constant 100 MHz clock (s_memrealtime).  bench.py
// reads it on both sides of a timed region: after an idle gap the clock needs ~25-40 ms of load to climb from ~2.1 to 2.4 GHz
// (profiles/r04/clock_ramp.txt), which a 20-step region feels and an 80-step one hardly does.
__global__ void clock_probe_kernel(long long* out) {
    if (threadIdx.x == 0) {
        const long long t0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
        long long t1 = t0;
        while (t1 - t0 < 2000) { __builtin_amdgcn_s_sleep(8); t1 = __builtin_amdgcn_s_memrealtime(); }
        out[0] = __builtin_amdgcn_s_memtime() - c0;
        out[1] = t1 - t0;
    }
}
extern "C" int sttode_clock_probe(long long* out, void* stream) {
    STT_REQUIRE(out, "sttode_clock_probe: null pointer");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
    STT_HIP(hipGetLastError());
    return 0;
}

// Device -> pinned host copy by a FEW workgroups (a metric path that wants every future on the host, test.py:194,526, beside a full chip):
// hipMemcpyAsync of a 17-MB prediction tensor costs the pipelined step its whole duration (62.9 against 76.0 M trajectories/s,
// profiles/r04/final_bench.json -- whatever executes it holds the chip's workgroup slots while it waits for the bus); `wgs` persistent
// workgroups (default 8: 1.5 % of the slots) move 16 bytes per lane per trip instead, and the rest of the chip keeps computing.  dst must
// be device-accessible host memory (hipHostMalloc / torch pin_memory), both pointers 16-byte aligned.
typedef float hc_f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void host_copy_kernel(hc_f32x4* __restrict__ dst, const hc_f32x4* __restrict__ src, long n16) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) {
        const hc_f32x4 v = __builtin_nontemporal_load(src + i);
        __builtin_nontemporal_store(v, dst + i);
    }
}
extern "C" int sttode_copy_to_host(void* dst, const void* src, long bytes, int wgs, void* stream) {
    STT_REQUIRE(dst && src && bytes > 0 && bytes % 16 == 0, "sttode_copy_to_host: null pointer or size not a multiple of 16 bytes");
    STT_REQUIRE(((size_t)dst | (size_t)src) % 16 == 0, "sttode_copy_to_host: pointers must be 16-byte aligned");
    if (wgs <= 0) wgs = 8;
    if (wgs > 256) wgs = 256;
    hipLaunchKernelGGL(host_copy_kernel, dim3(wgs), dim3(256), 0, (hipStream_t)stream, (hc_f32x4*)dst, (const hc_f32x4*)src, bytes / 16);
    STT_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// Host staging of ONE scene for the reference's one-scene-per-call loop (test.py:171-188 -> STTODENet.set_data, model/STTODE.py:397-404):
// the loader's pageable [N][2][T] tracks are transposed into a pinned ring slot as [N][T][2] (past, then future) and travel to the device
// in ONE asynchronous copy on the caller's stream.  A ring of four slots per device, each guarded by an event: the slot is reused only
// once the copy that last read it has completed (normally long ago).  Replaces a dozen torch calls (~35 us of host time per scene).
// ---------------------------------------------------------------------------------------------------
#include <cstring>
#include <mutex>
struct StageSlot { float* host = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; };
static StageSlot g_stage[STT_ATTR_DEVICES][4];
static int g_stage_k[STT_ATTR_DEVICES];
static std::mutex g_stage_mu[STT_ATTR_DEVICES];   // one per device: staging threads of different devices do not serialise on each other

// the next slot of device d's ring with room for `need` floats, its last copy complete (d: the device that owns the destination; must be current)
static int stage_slot_take(const void* dev, size_t need, StageSlot** out, const char* who) {
    int d = 0;
    hipPointerAttribute_t pa;
    if (hipPointerGetAttributes(&pa, dev) == hipSuccess) d = pa.device;
    else { (void)hipGetLastError(); STT_HIP(hipGetDevice(&d)); }
    STT_REQUIRE(d >= 0 && d < STT_ATTR_DEVICES, "host staging: device index beyond the staging table");
    // the ring's event and pinned buffer belong to device d and the event is recorded on the caller's stream: all three must be device d's
    // (round-4 advice: created on the current device they mixed devices for a model on another one)
    int cur_dev = 0;
    STT_HIP(hipGetDevice(&cur_dev));
    if (cur_dev != d) { stt_set_error(who); return 1; }
    StageSlot& s = g_stage[d][g_stage_k[d] = (g_stage_k[d] + 1) & 3];
    if (s.ev) STT_HIP(hipEventSynchronize(s.ev));   // the copy that last read this slot is done
    else STT_HIP(hipEventCreateWithFlags(&s.ev, hipEventDisableTiming));
    if (s.cap < need) {
        if (s.host) STT_HIP(hipHostFree(s.host));
        s.host = nullptr;
        s.cap = 0;                                   // (a failed allocation below must not leave a capacity behind a null pointer)
        const size_t want = need < 4096 ? 4096 : 2 * need;
        STT_HIP(hipHostMalloc((void**)&s.host, want * sizeof(float), hipHostMallocDefault));
        s.cap = want;
    }
    *out = &s;
    return 0;
}
static inline int stage_device_of(const void* dev) {
    hipPointerAttribute_t pa;
    if (hipPointerGetAttributes(&pa, dev) == hipSuccess && pa.device >= 0 && pa.device < STT_ATTR_DEVICES) return pa.device;
    (void)hipGetLastError();
    return 0;
}

extern "C" int sttode_stage_scene(const float* pre, const float* fut, int N, int Tp, int Tf, float* dev, void* stream) {
    STT_REQUIRE(pre && dev, "sttode_stage_scene: null pointer");
    STT_REQUIRE(N > 0 && Tp > 0 && Tf >= 0 && (fut || Tf == 0), "sttode_stage_scene: bad N/Tp/Tf");
    // the ring of the device that OWNS `dev` (a model on a non-current device must not be staged on the current device's ring and stream)
    std::lock_guard<std::mutex> lock(g_stage_mu[stage_device_of(dev)]);
    const size_t need = (size_t)N * (Tp + Tf) * 2;
    StageSlot* sp = nullptr;
    if (int rc = stage_slot_take(dev, need, &sp, "sttode_stage_scene: the current device must be the one that owns `dev` (hipSetDevice / torch.cuda.device first)")) return rc;
    StageSlot& s = *sp;
    float* h = s.host;
    for (int a = 0; a < N; ++a)
        for (int t = 0; t < Tp; ++t) {
            h[((size_t)a * Tp + t) * 2] = pre[((size_t)a * 2) * Tp + t];
            h[((size_t)a * Tp + t) * 2 + 1] = pre[((size_t)a * 2 + 1) * Tp + t];
        }
    float* hf = h + (size_t)N * Tp * 2;
    for (int a = 0; a < N; ++a)
        for (int t = 0; t < Tf; ++t) {
            hf[((size_t)a * Tf + t) * 2] = fut[((size_t)a * 2) * Tf + t];
            hf[((size_t)a * Tf + t) * 2 + 1] = fut[((size_t)a * 2 + 1) * Tf + t];
        }
    STT_HIP(hipMemcpyAsync(dev, h, need * sizeof(float), hipMemcpyHostToDevice, (hipStream_t)stream));
    STT_HIP(hipEventRecord(s.ev, (hipStream_t)stream));
    return 0;
}
// ... and of a batch whose host layout is already the device layout (set_data_nba, model/STTODE.py:463-486: past_traj [B,N,Tp,2],
// future_traj [B,N,Tf,2]): a [na] and b [nb] (HOST, pageable; b may be NULL with nb = 0) go through one ring slot into dev [na4 + nb]
// (na4 = na rounded up to a multiple of 4: a at 0, b at na4) with
// ONE asynchronous copy -- `.to(device)` of a pageable tensor waits for everything queued on the stream first, which in a training loop
// is the previous step's whole backward pass.
extern "C" int sttode_stage_rows(const float* a, long na, const float* b, long nb, float* dev, void* stream) {
    STT_REQUIRE(a && dev && na > 0 && nb >= 0 && (b || nb == 0), "sttode_stage_rows: null pointer or bad counts");
    std::lock_guard<std::mutex> lock(g_stage_mu[stage_device_of(dev)]);
    StageSlot* sp = nullptr;
    const long nap = (na + 3) / 4 * 4;               // b starts on a 16-byte boundary of dev
    if (int rc = stage_slot_take(dev, (size_t)(nap + nb), &sp, "sttode_stage_rows: the current device must be the one that owns `dev` (hipSetDevice / torch.cuda.device first)")) return rc;
    memcpy(sp->host, a, (size_t)na * sizeof(float));
    for (long i = na; i < nap; ++i) sp->host[i] = 0.f;
    if (nb) memcpy(sp->host + nap, b, (size_t)nb * sizeof(float));
    STT_HIP(hipMemcpyAsync(dev, sp->host, (size_t)(nap + nb) * sizeof(float), hipMemcpyHostToDevice, (hipStream_t)stream));
    // (measured and not adopted: a small kernel reading the pinned slot instead of the DMA copy, to keep the copy in the compute queue of a
    // busy stream -- 0.788 against 0.787 ms per free-running one-scene step, profiles/r05/train_stress.txt)
    STT_HIP(hipEventRecord(sp->ev, (hipStream_t)stream));
    return 0;
}
