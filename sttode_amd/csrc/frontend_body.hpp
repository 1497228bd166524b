// Scene front-end per agent (set_data / set_data_nba equivalents: model/STTODE.py:397-461, 463-486, 578-596), shared by the front-end
// kernels (frontend.hip) and the per-agent role of the fused chain launch (chain32.hip).
#pragma once
#include <hip/hip_runtime.h>

// SC1: the value is read by ANOTHER workgroup of the same launch -> agent-scope (write-through) store (cdna_hip_programming.md §6 G16 R1);
// scalar sc1 stores are slow per byte, which is irrelevant for the ~20 floats per agent written this way
template <bool SC1> __device__ __forceinline__ void fe_store(float* p, float v) {
    if (SC1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// One agent: normalised track (xpad, flattened (t, c), zero padded to 16*TPX), encoder inputs [T][4] = (normalised position, velocity with
// the first one duplicated: model/STTODE.py:432-433,582-583), cur_location, scene origin per agent, last-agent flag.
// (ox, oy): the agent's scene origin (0 for the NBA branch); `last`: the add_category flag (model/STTODE.py:199-210).
// TMAX > 0 (T <= TMAX): the track is read into registers first and the frame loop is unrolled -- with stores between the loads (atomic ones
// when SC1) every frame otherwise costs a memory round trip, 4-5 us for 8 frames on the critical path of a one-scene call.
template <bool SC1, int TMAX = 0>
__device__ __forceinline__ void agent_inputs_core(int a, const float* __restrict__ seq, int T, int TPX, int vel_from_norm, float ox, float oy,
                                                  int last, const float* __restrict__ prev_last,  // optional [n][2]: frame preceding seq, world coords
                                                  float* __restrict__ xpad, float* __restrict__ enc_in, float* __restrict__ cur,
                                                  float* __restrict__ orig, int* __restrict__ last_flag,
                                                  const float2* __restrict__ preloaded = nullptr) {   // TMAX > 0: the track, read by the caller ahead of time
    const float* p = seq + (size_t)a * T * 2;
    float* xp = xpad ? xpad + (size_t)a * 16 * TPX : nullptr;
    float pnx = 0.f, pny = 0.f, pwx = 0.f, pwy = 0.f;  // previous frame: normalised / world
    const bool have_prev = prev_last != nullptr;
    if (have_prev) {
        pwx = prev_last[2 * a];
        pwy = prev_last[2 * a + 1];
        pnx = pwx - ox;
        pny = pwy - oy;
    }
    float2 wb[TMAX > 0 ? TMAX : 1];
    if (TMAX > 0) {
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
            if (t < T) wb[t] = preloaded ? preloaded[t] : reinterpret_cast<const float2*>(p)[t];
    }
#pragma unroll
    for (int t = 0; t < (TMAX > 0 ? TMAX : T); ++t) {
        // (a guard, not a break: with a break the loop is not fully unrolled, wb[] is indexed dynamically and lives in SCRATCH --
        // 128 B per lane of every wave of whatever launch contains this code)
        if (TMAX > 0 && t >= T) continue;
        const float wx = TMAX > 0 ? wb[t].x : p[2 * t], wy = TMAX > 0 ? wb[t].y : p[2 * t + 1];
        const float nx = wx - ox, ny = wy - oy;
        float vx, vy;
        if (t == 0 && !have_prev) {
            // first velocity duplicates the second one (model/STTODE.py:432-433,582-583)
            const float w1x = TMAX > 0 ? wb[1 < TMAX ? 1 : 0].x : p[2], w1y = TMAX > 0 ? wb[1 < TMAX ? 1 : 0].y : p[3];
            if (vel_from_norm) { vx = (w1x - ox) - nx; vy = (w1y - oy) - ny; }
            else { vx = w1x - wx; vy = w1y - wy; }
        } else {
            if (vel_from_norm) { vx = nx - pnx; vy = ny - pny; }
            else { vx = wx - pwx; vy = wy - pwy; }
        }
        if (enc_in) {
            float* e = enc_in + ((size_t)a * T + t) * 4;
            e[0] = nx; e[1] = ny; e[2] = vx; e[3] = vy;
        }
        if (xp) { fe_store<SC1>(xp + 2 * t, nx); fe_store<SC1>(xp + 2 * t + 1, ny); }
        pnx = nx; pny = ny; pwx = wx; pwy = wy;
    }
    if (xp)
        for (int k = 2 * T; k < 16 * TPX; ++k) fe_store<SC1>(xp + k, 0.f);
    if (cur) { fe_store<SC1>(cur + 2 * a, pnx); fe_store<SC1>(cur + 2 * a + 1, pny); }
    if (orig) { fe_store<SC1>(orig + 2 * a, ox); fe_store<SC1>(orig + 2 * a + 1, oy); }
    if (last_flag) last_flag[a] = last;
}

// One displacement norm of compute_ADE / compute_FDE (utils/metrics.py:7-26), with the contraction spelled out so that every kernel that
// evaluates it (best_of_k_kernel, the trajectory groups' fused metrics) produces the same bits from the same inputs.
__device__ __forceinline__ float bok_dist(float px, float py, float gx, float gy, float scale) {
    const float dx = (px - gx) * scale, dy = (py - gy) * scale;
    return sqrtf(__fmaf_rn(dx, dx, dy * dy));
}
