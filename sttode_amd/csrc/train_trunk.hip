// Training step, forward of one encoder trunk WITH its tape in one launch (attention length 1: scene batches).
//
// Reference: PastEncoder / FutureEncoder trunk, model/STTODE.py:214-236 / :276-300 (input_fc -> PositionalAgentEncoding :150-176 ->
// input_fc2 -> add_category :199-210 -> input_fc3), hypertransformer.py:134-153 (encoder layer), hyptransformerlib.py:191-300 (attention: a
// softmax over ONE key is 1, so the attention output is the value projection), ode_demo.py:188,228 (one Euler step of size T + relu).
//
// The layer-by-layer form (training.Engine.trunk_fwd) is ~21 launches of a few microseconds each per trunk, and a one-scene training step is
// bound by the NUMBER of launches.  Here one workgroup (4 waves) owns a 16-agent tile, wave w computes output row tile w of
// every layer from the row-major nn.Parameter storage (an A fragment of v_mfma_f32_16x16x4_f32 is a float4 of a weight row), tiles are
// exchanged through LDS, and every tensor the backward pass needs is written on the way -- the same tape trunk_bwd reads.
#include <mutex>
#include "chain.hpp"
#include "api_util.hpp"
#include "../../include/sttode_hip.h"

struct TrunkArgs {
    const float* p[STT_TT_COUNT];
    int n, T;
    long ld_feat;
    float ode_time;
    int phase;   // 0: the whole trunk (attention length 1: the attention output is v); attention over the forward-call batch (the NBA branch,
                 // round 5): 1 = up to the in-projection (writes qkv), the attention kernel runs between, 2 = from the attention output
                 // p[STT_TT_ATTN] [n,64] on -- two launches per trunk instead of twelve layer launches
};

// A fragment (row tile it, k tile T) of a row-major weight W [I, ld]: lane (i, q) holds W[16 it + i][16 T + 4 q + 0..3]
template <bool ALIGNED>
__device__ __forceinline__ f32x4 wfrag(const float* __restrict__ W, long ld, int it, int T, int lane) {
    const float* p = W + (long)(16 * it + (lane & 15)) * ld + 16 * T + 4 * (lane >> 4);
    if (ALIGNED) return ld4(p);
    f32x4 r = {p[0], p[1], p[2], p[3]};
    return r;
}

static __device__ __forceinline__ void ttrunk_fwd_body(const TrunkArgs& a, int tile, char* smem_t) {
    f32x4* sPt = reinterpret_cast<f32x4*>(smem_t);          // [T][4][64]  post-dropout positional features of every frame
    f32x4* sX = sPt + (size_t)a.T * 256;                    // [4][4][64]  exchange slots
    const float* const* P = a.p;
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = a.n, T = a.T;
    const int col = tile * 16 + c;
    const int colc = col < n ? col : n - 1;
    const bool live = col < n;
    f32x4 x[4], wof[4], wif[4], wgf[4];
    if (a.phase != 2) {   // (uniform) ---------------------------------------------------------------- up to the in-projection
    // ---- frames: input_fc (K = 4: one MFMA per row tile) -> cat(., pe[t]) -> pos fc -> dropout
    f32x4 b1[4];
    float f1w[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        b1[it] = ld4(P[STT_TT_FC1_B] + 16 * it + 4 * q);
        f1w[it] = P[STT_TT_FC1_W][(16 * it + c) * 4 + q];
    }
    f32x4 wp[8];
#pragma unroll
    for (int Tk = 0; Tk < 8; ++Tk) wp[Tk] = wfrag<true>(P[STT_TT_POS_W], 128, w, Tk, lane);
    const f32x4 bp = ld4(P[STT_TT_POS_B] + 16 * w + 4 * q);
    float* posin = const_cast<float*>(P[STT_TT_POSIN]);
    float* tp = const_cast<float*>(P[STT_TT_TP]);
    const float* drop = P[STT_TT_DROP];
    // the next frame's inputs travel while this frame feeds the MFMAs
    float xin_n = P[STT_TT_ENC_IN][((long)colc * T) * 4 + q];
    f32x4 pe_n[4], m_n = splat4(1.f);
#pragma unroll
    for (int it = 0; it < 4; ++it) pe_n[it] = ld4(P[STT_TT_PE] + 16 * it + 4 * q);
    if (drop) m_n = ld4(drop + ((long)colc * T) * 64 + 16 * w + 4 * q);
    for (int t = 0; t < T; ++t) {
        const long row = (long)colc * T + t;
        const float xin = xin_n;
        f32x4 pe[4];
        const f32x4 m = m_n;
#pragma unroll
        for (int it = 0; it < 4; ++it) pe[it] = pe_n[it];
        if (t + 1 < T) {
            xin_n = P[STT_TT_ENC_IN][(row + 1) * 4 + q];
#pragma unroll
            for (int it = 0; it < 4; ++it) pe_n[it] = ld4(P[STT_TT_PE] + (long)(t + 1) * 64 + 16 * it + 4 * q);
            if (drop) m_n = ld4(drop + (row + 1) * 64 + 16 * w + 4 * q);
        }
        f32x4 acc = bp;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const f32x4 xt = __builtin_amdgcn_mfma_f32_16x16x4f32(f1w[it], xin, b1[it], 0, 0, 0);
            if (it == w && live) st4(posin + row * 128 + 16 * it + 4 * q, xt);
            acc = mfma_k16(acc, wp[it], xt);
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            if (it == w && live) st4(posin + row * 128 + 64 + 16 * it + 4 * q, pe[it]);
            acc = mfma_k16(acc, wp[4 + it], pe[it]);
        }
        if (drop) acc = acc * m;
        if (live) st4(tp + row * 64 + 16 * w + 4 * q, acc);
        sPt[(t * 4 + w) * 64 + lane] = acc;
    }
    // fragments of the small layers behind the barriers travel now (a load cannot be hoisted across a workgroup barrier by the compiler)
    f32x4 w3f[4], win[3][4];
#pragma unroll
    for (int Tk = 0; Tk < 4; ++Tk) {
        w3f[Tk] = wfrag<false>(P[STT_TT_FC3_W], 67, w, Tk, lane);
        wof[Tk] = wfrag<true>(P[STT_TT_OUT_W], 64, w, Tk, lane);
        wif[Tk] = wfrag<true>(P[STT_TT_INFO_W], 64, w, Tk, lane);
        wgf[Tk] = wfrag<true>(P[STT_TT_GATE_W], 64, w, Tk, lane);
#pragma unroll
        for (int i = 0; i < 3; ++i) win[i][Tk] = wfrag<true>(P[STT_TT_INPROJ_W], 64, w + 4 * i, Tk, lane);
    }
    __syncthreads();
    // ---- input_fc2 over all frames, row tile w
    f32x4 f = ld4(P[STT_TT_FC2_B] + 16 * w + 4 * q);
    {
        const long ld2 = 64L * T;
        f32x4 wn[4];
#pragma unroll
        for (int Tk = 0; Tk < 4; ++Tk) wn[Tk] = wfrag<true>(P[STT_TT_FC2_W], ld2, w, Tk, lane);
        for (int t = 0; t < T; ++t) {
            f32x4 wc[4];
#pragma unroll
            for (int Tk = 0; Tk < 4; ++Tk) wc[Tk] = wn[Tk];
            const int tn = t + 1 < T ? t + 1 : t;
#pragma unroll
            for (int Tk = 0; Tk < 4; ++Tk) wn[Tk] = wfrag<true>(P[STT_TT_FC2_W], ld2, w, 4 * tn + Tk, lane);
#pragma unroll
            for (int Tk = 0; Tk < 4; ++Tk) f = mfma_k16(f, wc[Tk], sPt[(t * 4 + Tk) * 64 + lane]);
        }
    }
    const float lastf = P[STT_TT_LAST] ? (reinterpret_cast<const int*>(P[STT_TT_LAST])[colc] ? 1.0f : 0.0f) : 0.0f;
    {
        float* h3in = const_cast<float*>(P[STT_TT_H3IN]);
        if (live) {
#pragma unroll
            for (int r = 0; r < 4; ++r) h3in[(long)col * 68 + 16 * w + 4 * q + r] = f[r];
            if (w == 0 && q == 0) {                       // add_category: [0, 0, 1] for the last agent of a scene (+ one pad column)
                h3in[(long)col * 68 + 64] = 0.f; h3in[(long)col * 68 + 65] = 0.f; h3in[(long)col * 68 + 66] = lastf; h3in[(long)col * 68 + 67] = 0.f;
            }
        }
    }
    sX[(0 * 4 + w) * 64 + lane] = f;
    __syncthreads();
    // ---- input_fc3 (67 inputs: 64 features + category), row tile w  ->  x = ftraj_input
    {
        f32x4 acc = ld4(P[STT_TT_FC3_B] + 16 * w + 4 * q);
#pragma unroll
        for (int Tk = 0; Tk < 4; ++Tk) acc = mfma_k16(acc, w3f[Tk], sX[(0 * 4 + Tk) * 64 + lane]);
        f32x4 wl;
#pragma unroll
        for (int r = 0; r < 4; ++r) wl[r] = P[STT_TT_FC3_W][(long)(16 * w + 4 * q + r) * 67 + 66];
        acc = acc + wl * lastf;
        if (live) {
            st4(const_cast<float*>(P[STT_TT_FEAT]) + (long)col * a.ld_feat + 16 * w + 4 * q, acc);
            st4(const_cast<float*>(P[STT_TT_XC]) + (long)col * 64 + 16 * w + 4 * q, acc);
        }
        sX[(1 * 4 + w) * 64 + lane] = acc;
    }
    __syncthreads();
#pragma unroll
    for (int Tk = 0; Tk < 4; ++Tk) x[Tk] = sX[(1 * 4 + Tk) * 64 + lane];
    // ---- in-projection: row tiles w, w + 4, w + 8 (q | k | v); attention length 1: the attention output is v
    {
        float* qkv = const_cast<float*>(P[STT_TT_QKV]);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int it = w + 4 * i;
            f32x4 acc = ld4(P[STT_TT_INPROJ_B] + 16 * it + 4 * q);
#pragma unroll
            for (int Tk = 0; Tk < 4; ++Tk) acc = mfma_k16(acc, win[i][Tk], x[Tk]);
            if (live) st4(qkv + (long)col * 192 + 16 * it + 4 * q, acc);
            if (i == 2) sX[(2 * 4 + w) * 64 + lane] = acc;
        }
    }
    if (a.phase == 1) return;   // (uniform) the attention over the forward-call batch runs as its own launch; phase 2 continues behind it
    } else {              // (uniform) ---------------------------------------------------------------- phase 2: from the attention output on
#pragma unroll
        for (int Tk = 0; Tk < 4; ++Tk) {
            wof[Tk] = wfrag<true>(P[STT_TT_OUT_W], 64, w, Tk, lane);
            wif[Tk] = wfrag<true>(P[STT_TT_INFO_W], 64, w, Tk, lane);
            wgf[Tk] = wfrag<true>(P[STT_TT_GATE_W], 64, w, Tk, lane);
            x[Tk] = ld4(P[STT_TT_XC] + (long)colc * 64 + 16 * Tk + 4 * q);
        }
        sX[(2 * 4 + w) * 64 + lane] = ld4(P[STT_TT_ATTN] + (long)colc * 64 + 16 * w + 4 * q);   // slot 2: the attention output (before out_proj)
    }
    __syncthreads();
    // ---- out_proj(v), row tile w
    {
        f32x4 acc = ld4(P[STT_TT_OUT_B] + 16 * w + 4 * q);
#pragma unroll
        for (int Tk = 0; Tk < 4; ++Tk) acc = mfma_k16(acc, wof[Tk], sX[(2 * 4 + Tk) * 64 + lane]);
        if (live) st4(const_cast<float*>(P[STT_TT_AO]) + (long)col * 64 + 16 * w + 4 * q, acc);
        sX[(3 * 4 + w) * 64 + lane] = acc;
    }
    __syncthreads();
    // ---- tanh(info) * sigmoid(gate), + x, row tile w
    {
        f32x4 vi = ld4(P[STT_TT_INFO_B] + 16 * w + 4 * q), vg = ld4(P[STT_TT_GATE_B] + 16 * w + 4 * q);
#pragma unroll
        for (int Tk = 0; Tk < 4; ++Tk) {
            const f32x4 o = sX[(3 * 4 + Tk) * 64 + lane];
            vi = mfma_k16(vi, wif[Tk], o);
            vg = mfma_k16(vg, wgf[Tk], o);
        }
        f32x4 tt, ss, s1;
        const f32x4 xw = w == 0 ? x[0] : w == 1 ? x[1] : w == 2 ? x[2] : x[3];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            tt[r] = tanhf(vi[r]);
            ss[r] = 1.0f / (1.0f + expf(-vg[r]));
            s1[r] = xw[r] + tt[r] * ss[r];
        }
        if (live) {
            st4(const_cast<float*>(P[STT_TT_TT]) + (long)col * 64 + 16 * w + 4 * q, tt);
            st4(const_cast<float*>(P[STT_TT_SS]) + (long)col * 64 + 16 * w + 4 * q, ss);
        }
        sX[(0 * 4 + w) * 64 + lane] = s1;                  // slot 0's readers (input_fc3) passed two barriers ago
    }
    __syncthreads();
    // ---- LayerNorm 1 on the full 64 features (every wave), tape: normalised value + 1/std
    auto ln = [&](f32x4 (&v)[4], const float* gamma, const float* beta, float* xhat, float* rstd) {
        float s = 0.f;
#pragma unroll
        for (int Tk = 0; Tk < 4; ++Tk) s += (v[Tk][0] + v[Tk][1]) + (v[Tk][2] + v[Tk][3]);
        const float mean = colsum_q(s) * (1.0f / 64.0f);
        float var = 0.f;
#pragma unroll
        for (int Tk = 0; Tk < 4; ++Tk)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = v[Tk][r] - mean; var += d * d; }
        const float rs = 1.0f / sqrtf(colsum_q(var) * (1.0f / 64.0f) + 1e-5f);
#pragma unroll
        for (int Tk = 0; Tk < 4; ++Tk) {
            const f32x4 g = ld4(gamma + 16 * Tk + 4 * q), b = ld4(beta + 16 * Tk + 4 * q);
            f32x4 xh;
#pragma unroll
            for (int r = 0; r < 4; ++r) { xh[r] = (v[Tk][r] - mean) * rs; v[Tk][r] = xh[r] * g[r] + b[r]; }
            if (Tk == w && live) st4(xhat + (long)col * 64 + 16 * Tk + 4 * q, xh);
        }
        if (w == 0 && q == 0 && live) rstd[col] = rs;
    };
    f32x4 h[4];
#pragma unroll
    for (int Tk = 0; Tk < 4; ++Tk) h[Tk] = sX[(0 * 4 + Tk) * 64 + lane];
    ln(h, P[STT_TT_LN1_W], P[STT_TT_LN1_B], const_cast<float*>(P[STT_TT_XH1]), const_cast<float*>(P[STT_TT_RS1]));
    if (live) {
        const f32x4 hw = w == 0 ? h[0] : w == 1 ? h[1] : w == 2 ? h[2] : h[3];
        st4(const_cast<float*>(P[STT_TT_H]) + (long)col * 64 + 16 * w + 4 * q, hw);
    }
    // ---- FFN 64 -> 1024 -> 64: wave w owns hidden tiles w, w + 4, ...; the hidden activation is part of the tape
    f32x4 ff[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) ff[it] = splat4(0.f);
    {
        float* f1 = const_cast<float*>(P[STT_TT_F1]);
        // fragments of the next hidden tile travel while this one feeds the MFMAs (a row-major fragment is 16 x 64 B of L2 traffic)
        f32x4 n1[4], n2[4], nb;
#pragma unroll
        for (int Tk = 0; Tk < 4; ++Tk) { n1[Tk] = wfrag<true>(P[STT_TT_L1_W], 64, w, Tk, lane); n2[Tk] = wfrag<true>(P[STT_TT_L2_W], 1024, Tk, w, lane); }
        nb = ld4(P[STT_TT_L1_B] + 16 * w + 4 * q);
#pragma unroll 1
        for (int i = 0; i < 16; ++i) {
            const int hn = 4 * i + w;
            f32x4 c1[4], c2[4];
#pragma unroll
            for (int Tk = 0; Tk < 4; ++Tk) { c1[Tk] = n1[Tk]; c2[Tk] = n2[Tk]; }
            f32x4 hid = nb;
            {
                const int hx = 4 * (i + 1 < 16 ? i + 1 : i) + w;
#pragma unroll
                for (int Tk = 0; Tk < 4; ++Tk) { n1[Tk] = wfrag<true>(P[STT_TT_L1_W], 64, hx, Tk, lane); n2[Tk] = wfrag<true>(P[STT_TT_L2_W], 1024, Tk, hx, lane); }
                nb = ld4(P[STT_TT_L1_B] + 16 * hx + 4 * q);
            }
#pragma unroll
            for (int Tk = 0; Tk < 4; ++Tk) hid = mfma_k16(hid, c1[Tk], h[Tk]);
            hid = relu4(hid);
            if (live) st4(f1 + (long)col * 1024 + 16 * hn + 4 * q, hid);
#pragma unroll
            for (int it = 0; it < 4; ++it) ff[it] = mfma_k16(ff[it], c2[it], hid);
        }
    }
    __syncthreads();                                     // slot readers above are done in every wave
#pragma unroll
    for (int it = 0; it < 4; ++it) sX[(w * 4 + it) * 64 + lane] = ff[it];
    __syncthreads();
    f32x4 y[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const f32x4 tsum = ((sX[(0 * 4 + it) * 64 + lane] + sX[(1 * 4 + it) * 64 + lane]) + sX[(2 * 4 + it) * 64 + lane]) + sX[(3 * 4 + it) * 64 + lane];
        y[it] = h[it] + (tsum + ld4(P[STT_TT_L2_B] + 16 * it + 4 * q));
    }
    ln(y, P[STT_TT_LN2_W], P[STT_TT_LN2_B], const_cast<float*>(P[STT_TT_XH2]), const_cast<float*>(P[STT_TT_RS2]));
    // ---- one Euler step of size ode_time + relu (ode_demo.py:188,228), row tile w
    if (live) {
        const f32x4 xw = w == 0 ? x[0] : w == 1 ? x[1] : w == 2 ? x[2] : x[3];
        const f32x4 yw = w == 0 ? y[0] : w == 1 ? y[1] : w == 2 ? y[2] : y[3];
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = fmaxf(xw[r] + a.ode_time * yw[r], 0.f);
        st4(const_cast<float*>(P[STT_TT_ODE]) + (long)col * 64 + 16 * w + 4 * q, o);
        st4(const_cast<float*>(P[STT_TT_FEAT]) + (long)col * a.ld_feat + 64 + 16 * w + 4 * q, o);
    }
}

__global__ __launch_bounds__(256) void ttrunk_fwd_kernel(TrunkArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_t[];
    ttrunk_fwd_body(a, blockIdx.x, smem_t);
}
// Two trunks (the past and the future encoder: independent of each other) in ONE launch: the first n0 workgroups walk trunk 0's tiles, the
// rest trunk 1's.  A trunk's forward is one or two workgroups busy for ~40 us; side by side the two cost one of them (sttode_tgemm_group).
__global__ __launch_bounds__(256) void ttrunk_fwd2_kernel(TrunkArgs a0, TrunkArgs a1, int n0) {
    extern __shared__ __attribute__((aligned(16))) char smem_t[];
    if ((int)blockIdx.x < n0) ttrunk_fwd_body(a0, blockIdx.x, smem_t);
    else ttrunk_fwd_body(a1, (int)blockIdx.x - n0, smem_t);
}

// group mode (sttode_tgemm_group -> stt_trunk_group): a trunk launch waits for a second one of the same group
static thread_local struct { TrunkArgs a; size_t lds; void* stream; bool have, on; } g_tq = {};   // (per host thread, like the group state of train.hip)
static std::mutex g_tq_mu;
static void tq_flush_locked() {
    if (!g_tq.have) return;
    g_tq.have = false;
    hipLaunchKernelGGL(ttrunk_fwd_kernel, dim3((g_tq.a.n + 15) / 16), dim3(256), g_tq.lds, (hipStream_t)g_tq.stream, g_tq.a);
}
int stt_trunk_group(int on) {
    std::lock_guard<std::mutex> lk(g_tq_mu);
    if (on < 0) g_tq.have = false;
    tq_flush_locked();
    g_tq.on = on > 0;
    STT_HIP(hipGetLastError());
    return 0;
}

extern "C" int sttode_ttrunk_fwd(const void* const* ptrs, int count, int n, int T, long ld_feat, float ode_time, int phase, void* stream) {
    STT_REQUIRE(ptrs && count == STT_TT_COUNT, "sttode_ttrunk_fwd: pointer table must have STT_TT_COUNT entries");
    STT_REQUIRE(n > 0 && T >= 1 && ld_feat >= 128 && (ld_feat % 4) == 0 && phase >= 0 && phase <= 2, "sttode_ttrunk_fwd: bad n / T / ld_feat / phase");
    const size_t lds = ((size_t)T * 256 + 1024) * 16;
    STT_REQUIRE(lds <= 64 * 1024, "sttode_ttrunk_fwd: T too large for the fused form (T <= 12): use the layer-by-layer path");
    TrunkArgs a;
    for (int i = 0; i < STT_TT_COUNT; ++i) {
        a.p[i] = (const float*)ptrs[i];
        STT_REQUIRE(a.p[i] || i == STT_TT_DROP || i == STT_TT_LAST || (i == STT_TT_ATTN && phase != 2), "sttode_ttrunk_fwd: null pointer in the table");
    }
    a.n = n; a.T = T; a.ld_feat = ld_feat; a.ode_time = ode_time; a.phase = phase;
    std::lock_guard<std::mutex> lk(g_tq_mu);
    if (g_tq.on && g_tq.have && g_tq.stream == stream) {   // the group's second trunk: both in one launch
        g_tq.have = false;
        const int n0 = (g_tq.a.n + 15) / 16;
        const size_t l = lds > g_tq.lds ? lds : g_tq.lds;
        hipLaunchKernelGGL(ttrunk_fwd2_kernel, dim3(n0 + (n + 15) / 16), dim3(256), l, (hipStream_t)stream, g_tq.a, a, n0);
        STT_HIP(hipGetLastError());
        return 0;
    }
    tq_flush_locked();
    if (g_tq.on) { g_tq.a = a; g_tq.lds = lds; g_tq.stream = stream; g_tq.have = true; return 0; }
    hipLaunchKernelGGL(ttrunk_fwd_kernel, dim3((n + 15) / 16), dim3(256), lds, (hipStream_t)stream, a);
    STT_HIP(hipGetLastError());
    return 0;
}
