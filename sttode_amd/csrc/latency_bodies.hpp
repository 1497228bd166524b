// Bodies of the latency-form kernels that more than one translation unit launches (decoder.hip: sttode_gru_cols; encoder.hip: the
// stand-alone encoder kernels and the fused per-agent stage; chain32.hip: the per-agent ROLE of the fused chain launch).  See decoder.hip
// and encoder.hip for the design notes.
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


// 16-byte agent-scope (sc1, write-through) store: the payload form of an in-launch hand-off to ANOTHER workgroup -- no release fence is
// needed behind it (cdna_hip_programming.md §6 Guideline 16, R1).  hipcc does not count an asm store: every storing wave runs
// `s_waitcnt vmcnt(0)` itself before the workgroup's flag is published.  (`s_nop 1`: §5.7 item 1 -- the data registers may otherwise be
// overwritten before the store has read them.)
__device__ __forceinline__ void st4_sc1(float* p, const f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}

// ---------------------------------------------------------------------------------------------------
// encoder (PastEncoder trunk): model/STTODE.py:214-236, hypertransformer.py:55-89,134-153, ode_demo.py:186-190,223-231
// ---------------------------------------------------------------------------------------------------
struct EmbedW {
    const float* fc1P;    // [4 row tiles][64 lanes]  lane(i,q) -> W_fc[16it+i][q]
    const float* fc1b;    // [64]
    const f32x4* posP;    // PK16 of pos fc weight[:, :64]   [4][4][64]
    const float* peb;     // [Tlen][64]  pos fc weight[:, 64:] @ pe[t] + pos fc bias
    const f32x4* fc2P;    // PK16 of input_fc2 [64 x 64*Tlen] -> [4][4*Tlen][64]
    const float* fc2b;    // [64]
    const f32x4* fc3P;    // PK16 of input_fc3[:, :64]  [4][4][64]
    const float* fc3b;    // [64]
    const float* fc3last; // [64] = input_fc3.weight[:, 66]  (category [0,0,1] of the last agent)
    const f32x4* inP;     // PK16 of in_proj_weight [192 x 64] -> [12][4][64]
    const float* inb;     // [192]
};


// Latency form of embed_qkv: ONE workgroup (4 waves) per 16-agent tile, the waves split every layer by output row tile instead of
// each owning a tile (whose serial chain is ~1300 fp32 MFMAs = ~20 us however few tiles exist):
//   1  wave w: pos-enc fc row tile w of EVERY frame (input_fc recomputed: 4 MFMAs per frame) -> LDS;         one barrier
//   2  wave w: input_fc2 row tile w accumulated over frames and k-tiles in the throughput kernel's order -> LDS; one barrier
//   3  wave w: input_fc3 row tile w (+ category column) -> g, LDS;                                              one barrier
//   4  wave w: in-projection row tiles w, w+4, w+8 -> qkv.
// ~350 MFMAs per wave.  Every output element is summed in the order of embed_qkv_kernel: identical bits.
// smem: (Tlen * 256 + 512) f32x4 of LDS; `tile` = the workgroup's 16-agent tile; waves 0..3 of the workgroup.
__device__ __forceinline__ void embed_lat_body(const EmbedW& w, const float* __restrict__ enc_in, const int* __restrict__ last_flag,
                                               float* __restrict__ g, float* __restrict__ qkv, int n, int Tlen, int tile, f32x4* smem) {
    f32x4* sPt = smem;                               // [Tlen][4][64]
    f32x4* sF = sPt + (size_t)Tlen * 256;            // [4][64]
    f32x4* sG = sF + 256;                            // [4][64]
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = tile * 16 + c;
    const int colc = col < n ? col : n - 1;
    f32x4 pw[4], b1[4];
    float f1[4];
#pragma unroll
    for (int T = 0; T < 4; ++T) {
        pw[T] = w.posP[(wv * 4 + T) * 64 + lane];
        f1[T] = w.fc1P[T * 64 + lane];
        b1[T] = ld4(w.fc1b + 16 * T + 4 * q);
    }
    // first fc2 fragments of this wave's row tile travel during phase 1
    f32x4 w2n[4];
#pragma unroll
    for (int T = 0; T < 4; ++T) w2n[T] = w.fc2P[((size_t)wv * 4 * Tlen + T) * 64 + lane];
    float xn = enc_in[((size_t)colc * Tlen) * 4 + q];
    for (int t = 0; t < Tlen; ++t) {
        const float xin = xn;
        if (t + 1 < Tlen) xn = enc_in[((size_t)colc * Tlen + t + 1) * 4 + q];
        f32x4 a = ld4(w.peb + (size_t)t * 64 + 16 * wv + 4 * q);
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            const f32x4 xt = __builtin_amdgcn_mfma_f32_16x16x4f32(f1[T], xin, b1[T], 0, 0, 0);
            a = mfma_k16(a, pw[T], xt);
        }
        sPt[(t * 4 + wv) * 64 + lane] = a;
    }
    __syncthreads();
    f32x4 f = ld4(w.fc2b + 16 * wv + 4 * q);
    for (int t = 0; t < Tlen; ++t) {
        f32x4 w2c[4];
#pragma unroll
        for (int T = 0; T < 4; ++T) w2c[T] = w2n[T];
        const int tn = t + 1 < Tlen ? t + 1 : t;
#pragma unroll
        for (int T = 0; T < 4; ++T) w2n[T] = w.fc2P[((size_t)wv * 4 * Tlen + 4 * tn + T) * 64 + lane];
#pragma unroll
        for (int T = 0; T < 4; ++T) f = mfma_k16(f, w2c[T], sPt[(t * 4 + T) * 64 + lane]);
    }
    f32x4 w3[4], wi[3][4];
#pragma unroll
    for (int T = 0; T < 4; ++T) w3[T] = w.fc3P[(wv * 4 + T) * 64 + lane];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int T = 0; T < 4; ++T) wi[i][T] = w.inP[((wv + 4 * i) * 4 + T) * 64 + lane];
    sF[wv * 64 + lane] = f;
    __syncthreads();
    const float lastf = last_flag[colc] ? 1.0f : 0.0f;
    f32x4 gg;
    {
        f32x4 a = ld4(w.fc3b + 16 * wv + 4 * q);
#pragma unroll
        for (int T = 0; T < 4; ++T) a = mfma_k16(a, w3[T], sF[T * 64 + lane]);
        const f32x4 wl = ld4(w.fc3last + 16 * wv + 4 * q);
        gg = a + wl * lastf;
    }
    if (col < n) st4(g + (size_t)col * 64 + 16 * wv + 4 * q, gg);
    sG[wv * 64 + lane] = gg;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int it = wv + 4 * i;
        f32x4 a = ld4(w.inb + 16 * it + 4 * q);
#pragma unroll
        for (int T = 0; T < 4; ++T) a = mfma_k16(a, wi[i][T], sG[T * 64 + lane]);
        if (col < n) st4(qkv + (size_t)col * 192 + 16 * it + 4 * q, a);
    }
}


struct PostW {
    const f32x4* outP;  const float* outb;    // out_proj       PK16 [4][4][64], [64]
    const f32x4* infoP; const float* infob;   // temporal_info
    const f32x4* gateP; const float* gateb;   // temporal_gate
    const float* ln1w;  const float* ln1b;
    const f32x4* l1P;   const float* l1b;     // linear1 [1024 x 64]  PK16 [64][4][64], [1024]
    const f32x4* l2P;   const float* l2b;     // linear2 [64 x 1024]  PK16 [4][64][64], [64]
    const float* ln2w;  const float* ln2b;
};

// One workgroup (4 waves) per 16-agent tile; the waves split every layer instead of each owning a tile:
//   out_proj / info / gate : wave w computes output row tile w (16 of the 64 features), tiles are exchanged through LDS;
//   FFN                    : wave w owns hidden tiles w, w+4, ... (16 of 64) and accumulates a PARTIAL 64-wide output,
//                            the four partials are summed through LDS;
//   LayerNorms / Euler     : recomputed by every wave on the full 64 features (cheap VALU), wave w stores tile w.
// The serial MFMA chain per wave drops from ~2200 to ~560 instructions (this kernel is latency-bound: 541 tiles only).
// Right-hand side of the tensor ODE for 16 columns held by this workgroup: f(y) = LN2(h + FFN(h)), h = LN1(y + gate(out_proj(a)))
// (hypertransformer.py:134-153, :81-83), `a` = attention output for state y.  Every wave enters with the full a[4], y[4] tiles and
// leaves with the full result in x[4]; sX is the 16 KiB exchange buffer.  Ends with a barrier-protected sX, so calls can be chained.
__device__ __forceinline__ void ode_rhs(const PostW& w, f32x4 (*sX)[4][64], const f32x4 (&a)[4], const f32x4 (&y)[4], f32x4 (&x)[4],
                                        int lane, int q, int wv) {
    // FFN fragments of this wave's first hidden tile: issue early
    f32x4 wn1[4], wn2[4];
#pragma unroll
    for (int T = 0; T < 4; ++T) { wn1[T] = w.l1P[(wv * 4 + T) * 64 + lane]; wn2[T] = w.l2P[(T * 64 + wv) * 64 + lane]; }
    f32x4 hbn = ld4(w.l1b + 16 * wv + 4 * q);
    // out_proj, row tile wv
    {
        f32x4 v = ld4(w.outb + 16 * wv + 4 * q);
#pragma unroll
        for (int T = 0; T < 4; ++T) v = mfma_k16(v, w.outP[(wv * 4 + T) * 64 + lane], a[T]);
        sX[0][wv][lane] = v;
    }
    __syncthreads();
    f32x4 o[4];
#pragma unroll
    for (int T = 0; T < 4; ++T) o[T] = sX[0][T][lane];
    // info / gate, row tile wv  ->  x = y + tanh(info) * sigmoid(gate)
    {
        f32x4 vi = ld4(w.infob + 16 * wv + 4 * q), vg = ld4(w.gateb + 16 * wv + 4 * q);
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            vi = mfma_k16(vi, w.infoP[(wv * 4 + T) * 64 + lane], o[T]);
            vg = mfma_k16(vg, w.gateP[(wv * 4 + T) * 64 + lane], o[T]);
        }
        f32x4 xr;
        const f32x4 gw = wv == 0 ? y[0] : wv == 1 ? y[1] : wv == 2 ? y[2] : y[3];
#pragma unroll
        for (int r = 0; r < 4; ++r) xr[r] = gw[r] + tanhf(vi[r]) * sigmoidf_(vg[r]);
        sX[1][wv][lane] = xr;
    }
    __syncthreads();
#pragma unroll
    for (int T = 0; T < 4; ++T) x[T] = sX[1][T][lane];
    layernorm64(x, w.ln1w, w.ln1b, q);
    // FFN: hidden tiles wv, wv+4, ... ; partial output in ff
    f32x4 ff[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) ff[it] = splat4(0.f);
#pragma unroll 1
    for (int i = 0; i < 16; ++i) {
        f32x4 wc1[4], wc2[4];
#pragma unroll
        for (int T = 0; T < 4; ++T) { wc1[T] = wn1[T]; wc2[T] = wn2[T]; }
        f32x4 hid = hbn;
        {
            const int hn = (i + 1 < 16 ? i + 1 : i) * 4 + wv;
#pragma unroll
            for (int T = 0; T < 4; ++T) { wn1[T] = w.l1P[(hn * 4 + T) * 64 + lane]; wn2[T] = w.l2P[(T * 64 + hn) * 64 + lane]; }
            hbn = ld4(w.l1b + 16 * hn + 4 * q);
        }
#pragma unroll
        for (int T = 0; T < 4; ++T) hid = mfma_k16(hid, wc1[T], x[T]);
        hid = relu4(hid);
#pragma unroll
        for (int it = 0; it < 4; ++it) ff[it] = mfma_k16(ff[it], wc2[it], hid);
    }
    // sum the four partial FFN outputs through LDS (fixed order 0+1+2+3: deterministic)
    __syncthreads();  // sX[0..3] reads above are complete in every wave before they are overwritten
#pragma unroll
    for (int it = 0; it < 4; ++it) sX[wv][it][lane] = ff[it];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const f32x4 t = ((sX[0][it][lane] + sX[1][it][lane]) + sX[2][it][lane]) + sX[3][it][lane];
        x[it] = x[it] + (t + ld4(w.l2b + 16 * it + 4 * q));
    }
    layernorm64(x, w.ln2w, w.ln2b, q);
}

// ODE = false: the reference's integrator -- ONE explicit Euler step of size ode_time (ode_demo.py:186-190) with the attention output
// given (any attention length).  ODE = true: `steps` steps of `method` (0 Euler, 1 torchdiffeq's fixed-grid rk4 = 3/8 rule,
// 2 classical RK4) over [0, ode_time]; every stage needs the attention output of ITS state, which for attention length 1 (the
// ETH/UCY/SDD path: softmax over one element) is just v(y) = W_v y + b_v and is computed here (vP, vb = value rows of the packed
// in-projection).  With attention length > 1 a stage needs a pass over the whole group: op level (hypertransformer.ODEG_Encoder).
// sX: [4][4][64] f32x4 exchange buffer (16 KiB of LDS); `tile` = the workgroup's 16-agent tile; waves 0..3 of the workgroup.
template <bool ODE, bool SC1 = false>   // SC1: pf is read by OTHER workgroups of the same launch (write-through stores)
__device__ __forceinline__ void post_attn_body(const PostW& w, const float* __restrict__ g,  // [n][64]
                                               const float* __restrict__ attn, int ld_attn,  // [n][ld] attention output (pre out_proj)
                                               float* __restrict__ pf,                       // [n][128]
                                               int n, float ode_time, int method, int steps, const f32x4* __restrict__ vP,
                                               const float* __restrict__ vb, int tile, f32x4 (*sX)[4][64]) {
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int wv = threadIdx.x >> 6;
    const int col = tile * 16 + c;
    const int colc = col < n ? col : n - 1;
    f32x4 a[4], gg[4], x[4];
#pragma unroll
    for (int T = 0; T < 4; ++T) gg[T] = ld4(g + (size_t)colc * 64 + 16 * T + 4 * q);
    f32x4 yo[4];  // integrated state
    if (!ODE) {
#pragma unroll
        for (int T = 0; T < 4; ++T) a[T] = ld4(attn + (size_t)colc * ld_attn + 16 * T + 4 * q);
        ode_rhs(w, sX, a, gg, x, lane, q, wv);
        // torchdiffeq fixed-grid euler on t=[0,T]: y1 = y0 + T*f(y0) (ode_demo.py:188)
#pragma unroll
        for (int T = 0; T < 4; ++T) yo[T] = gg[T] + x[T] * ode_time;
    } else {
        auto F = [&](const f32x4 (&y)[4], f32x4 (&k)[4]) {
            __syncthreads();  // previous stage's sX reads are done
            {   // attention output for state y at attention length 1: v(y), row tile wv, exchanged through sX[2]
                f32x4 v = ld4(vb + 16 * wv + 4 * q);
#pragma unroll
                for (int T = 0; T < 4; ++T) v = mfma_k16(v, vP[(wv * 4 + T) * 64 + lane], y[T]);
                sX[2][wv][lane] = v;
            }
            __syncthreads();
#pragma unroll
            for (int T = 0; T < 4; ++T) a[T] = sX[2][T][lane];
            __syncthreads();
            ode_rhs(w, sX, a, y, k, lane, q, wv);
        };
        const float hstep = ode_time / (float)steps;
#pragma unroll
        for (int T = 0; T < 4; ++T) yo[T] = gg[T];
        for (int s = 0; s < steps; ++s) {
            f32x4 k1[4], k2[4], k3[4], k4[4], t[4];
            F(yo, k1);
            if (method == 0) {
#pragma unroll
                for (int T = 0; T < 4; ++T) yo[T] = yo[T] + k1[T] * hstep;
            } else if (method == 1) {   // 3/8 rule (torchdiffeq rk4_alt_step_func)
#pragma unroll
                for (int T = 0; T < 4; ++T) t[T] = yo[T] + k1[T] * (hstep / 3.f);
                F(t, k2);
#pragma unroll
                for (int T = 0; T < 4; ++T) t[T] = yo[T] + (k2[T] - k1[T] * (1.f / 3.f)) * hstep;
                F(t, k3);
#pragma unroll
                for (int T = 0; T < 4; ++T) t[T] = yo[T] + (k1[T] - k2[T] + k3[T]) * hstep;
                F(t, k4);
#pragma unroll
                for (int T = 0; T < 4; ++T) yo[T] = yo[T] + (k1[T] + (k2[T] + k3[T]) * 3.f + k4[T]) * (hstep / 8.f);
            } else {                    // classical RK4
#pragma unroll
                for (int T = 0; T < 4; ++T) t[T] = yo[T] + k1[T] * (hstep / 2.f);
                F(t, k2);
#pragma unroll
                for (int T = 0; T < 4; ++T) t[T] = yo[T] + k2[T] * (hstep / 2.f);
                F(t, k3);
#pragma unroll
                for (int T = 0; T < 4; ++T) t[T] = yo[T] + k3[T] * hstep;
                F(t, k4);
#pragma unroll
                for (int T = 0; T < 4; ++T) yo[T] = yo[T] + (k1[T] + k2[T] * 2.f + k3[T] * 2.f + k4[T]) * (hstep / 6.f);
            }
        }
    }
    if (col < n) {
        // pf = cat(ftraj_input, relu(ODE state at t = ode_time)) (ode_demo.py:231, model/STTODE.py:233-235); wave wv stores tile wv
        const f32x4 go = wv == 0 ? gg[0] : wv == 1 ? gg[1] : wv == 2 ? gg[2] : gg[3];
        const f32x4 xo = wv == 0 ? yo[0] : wv == 1 ? yo[1] : wv == 2 ? yo[2] : yo[3];
        if (SC1) {
            st4_sc1(pf + (size_t)col * 128 + 16 * wv + 4 * q, go);
            st4_sc1(pf + (size_t)col * 128 + 64 + 16 * wv + 4 * q, relu4(xo));
        } else {
            st4(pf + (size_t)col * 128 + 16 * wv + 4 * q, go);
            st4(pf + (size_t)col * 128 + 64 + 16 * wv + 4 * q, relu4(xo));
        }
    }
}



// ---------------------------------------------------------------------------------------------------
// Bodies used only by the per-agent ROLE of the fused chain launch (chain32.hip): a 256-thread workgroup with <= 256 VGPRs per wave
// ---------------------------------------------------------------------------------------------------
// Four-wave form of gru_lat_body (block-0 conv + GRU of ONE 16-agent tile): wave w keeps the 24 weight fragments of hidden tile w in
// registers for all steps; hidden tiles 4 and 5 are computed by waves 0 and 1 as well, from fragments staged once in LDS
// (sW45: [2][3 gates][8 k-tiles][64] f32x4 = 48 KiB, by LDS-DMA, + 8 KiB behind it for the raw gate sums handed to waves 2, 3 = 56 KiB).  A split over output rows: every element of h is the k-ordered chain
// of gru_lat_body / gru_cols_kernel, so the bits are those of the other forms.  Returns the index of the sH buffer that holds the final
// hidden state (all six tiles, B-operand fragment layout), which the caller may feed straight into the next layer.
struct GruNoPre { __device__ __forceinline__ void operator()() const {} };
// PRE: called once every weight load of the prologue is in flight and before the tile's input is read (a workgroup whose input is made by
// its own front-end runs it there, under the weights' latency)
template <int TPX, bool DLDS = false, bool SC1 = false, class PRE = GruNoPre>   // SC1: state is read by OTHER workgroups of the same launch
__device__ __forceinline__ int gru_lat4_body(const float* __restrict__ xin, const f32x4* __restrict__ convP, const float* __restrict__ convB,
                                             const f32x4* __restrict__ wihP, const f32x4* __restrict__ whhP, const float* __restrict__ gbias,
                                             float* __restrict__ state, int ncols, int Tp, int tile, f32x4 (*sH)[6][64], f32x4* sW45,
                                             const f32x4* d_lds = nullptr, PRE pre = PRE()) {
    // DLDS: the tile's input comes from d_lds (B-operand fragments [TPX][64] in LDS) instead of xin, and the final state stays in sH only
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = tile * 16 + c;
    const int colc = col < ncols ? col : ncols - 1;
    {   // 48 fragments of 1 KiB (tiles 4, 5): 12 wave-copies per wave, all in flight at once
        const unsigned base = lds_addr(sW45);
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int f = w * 12 + i, j = 4 + f / 24, r = f % 24, g = r >> 3, T = r & 7;
            const f32x4* src = T < 2 ? wihP + ((g * 6 + j) * 2 + T) * 64 + lane : whhP + ((g * 6 + j) * 6 + (T - 2)) * 64 + lane;
            glds16_asm(src, __builtin_amdgcn_readfirstlane(base + (unsigned)f * 1024u));
        }
    }
    f32x4 wi[3][2], wh[3][6], b0[4], b1[4];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
#pragma unroll
        for (int T = 0; T < 2; ++T) wi[g][T] = wihP[((g * 6 + w) * 2 + T) * 64 + lane];
#pragma unroll
        for (int T = 0; T < 6; ++T) wh[g][T] = whhP[((g * 6 + w) * 6 + T) * 64 + lane];
    }
    const int w2 = w < 2 ? 4 + w : 4;   // second hidden tile of waves 0 and 1 (its gate functions run on waves 2 and 3)
#pragma unroll
    for (int g = 0; g < 4; ++g) { b0[g] = ld4(gbias + g * 96 + 16 * w + 4 * q); b1[g] = ld4(gbias + g * 96 + 16 * w2 + 4 * q); }
    const f32x4 cb0 = ld4(convB + 4 * q), cb1 = ld4(convB + 16 + 4 * q);
    f32x4 cw[2][TPX];
#pragma unroll
    for (int io = 0; io < 2; ++io)
#pragma unroll
        for (int T = 0; T < TPX; ++T) cw[io][T] = convP[(io * TPX + T) * 64 + lane];
    pre();
    f32x4 d[TPX];
#pragma unroll
    for (int T = 0; T < TPX; ++T) d[T] = DLDS ? d_lds[T * 64 + lane] : ld4(xin + (size_t)colc * (16 * TPX) + 16 * T + 4 * q);
    f32x4 hn = splat4(0.f), hn2 = splat4(0.f);
    sH[0][w][lane] = hn;
    if (w < 2) sH[0][4 + w][lane] = hn2;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA pieces have landed; the barrier publishes everybody's
    __syncthreads();
    const f32x4* wB = sW45 + (size_t)(w & 1) * (24 * 64) + lane;
    f32x4* sG = sW45 + 2 * 24 * 64;                    // [2 tiles][4][64]: raw gate sums of hidden tiles 4, 5 (8 KiB behind the image)
    int cur = 0;
    // One step = two halves around an LDS hand-off.  The matrix work of hidden tiles 4, 5 sits on waves 0, 1 (from the LDS image), but their
    // gate functions (12 exp / rcp evaluations per lane, ~560 cycles with the matrix pipe idle) move to waves 2, 3, which have nothing
    // else left by then; waves 2, 3 likewise keep their own tile's gate functions for the second half.  First half: every wave runs 104
    // MFMAs; second half: waves 0, 1 their own tile (96 MFMAs + gates), waves 2, 3 gates only.  Same sums, same functions: same bits.
    auto gates = [&](const f32x4& ar, const f32x4& az, const f32x4& ai, const f32x4& ah, const f32x4& hp) {
        f32x4 o;
#ifdef LAT_DIAG_NO_GATES    // (diagnostic builds only)
        return ar * 1e-3f + az * 1e-3f + ai * 1e-3f + ah * 1e-3f + hp * 0.5f;
#endif
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float rg = sigmoid_prescaled(ar[r]);
            const float zg = sigmoid_prescaled(az[r]);
            const float ng = tanh_prescaled(fmaf(rg, ah[r], ai[r]));
            o[r] = fmaf(zg, hp[r] - ng, ng);  // (1-z) n + z h
        }
        return o;
    };
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
        f32x4 ar, az, ai, ah;
        auto own_sums = [&]() {                        // hidden tile w from the register fragments
            ar = b0[0]; az = b0[1]; ai = b0[2]; ah = b0[3];
#pragma unroll
            for (int T = 0; T < 2; ++T) {
                ar = mfma_k16(ar, wi[0][T], e[T]);
                az = mfma_k16(az, wi[1][T], e[T]);
                ai = mfma_k16(ai, wi[2][T], e[T]);
            }
#pragma unroll
            for (int T = 0; T < 6; ++T) {
                const f32x4 hb = sH[cur][T][lane];
                ar = mfma_k16(ar, wh[0][T], hb);
                az = mfma_k16(az, wh[1][T], hb);
                ah = mfma_k16(ah, wh[2][T], hb);
            }
        };
        if (w < 2) {   // (wave-uniform) first half: hidden tile 4 + w from the LDS image, raw sums handed to wave 2 + w
            f32x4 br = b1[0], bz = b1[1], bi = b1[2], bh = b1[3];
#ifndef LAT_DIAG_NO_TILEB   // (diagnostic builds only: timing without the second tile's matrix work; results are garbage)
#pragma unroll
            for (int T = 0; T < 2; ++T) {
                br = mfma_k16(br, wB[(0 * 8 + T) * 64], e[T]);
                bz = mfma_k16(bz, wB[(1 * 8 + T) * 64], e[T]);
                bi = mfma_k16(bi, wB[(2 * 8 + T) * 64], e[T]);
            }
#pragma unroll
            for (int T = 0; T < 6; ++T) {
                const f32x4 hb = sH[cur][T][lane];
                br = mfma_k16(br, wB[(0 * 8 + 2 + T) * 64], hb);
                bz = mfma_k16(bz, wB[(1 * 8 + 2 + T) * 64], hb);
                bh = mfma_k16(bh, wB[(2 * 8 + 2 + T) * 64], hb);
            }
#endif
            f32x4* g = sG + (size_t)w * 256 + lane;
            g[0] = br; g[64] = bz; g[128] = bi; g[192] = bh;
        } else {
            own_sums();
        }
        lds_barrier();
        if (w < 2) {   // second half
            own_sums();
            hn = gates(ar, az, ai, ah, sH[cur][w][lane]);
        } else {
            hn = gates(ar, az, ai, ah, sH[cur][w][lane]);
            const f32x4* g = sG + (size_t)(w - 2) * 256 + lane;
            hn2 = gates(g[0], g[64], g[128], g[192], sH[cur][2 + w][lane]);
            sH[cur ^ 1][2 + w][lane] = hn2;
        }
        sH[cur ^ 1][w][lane] = hn;
        lds_barrier();
        cur ^= 1;
    }
    if (!DLDS && col < ncols) {
        if (SC1) {
            st4_sc1(state + (size_t)col * 96 + 16 * w + 4 * q, hn);
            if (w >= 2) st4_sc1(state + (size_t)col * 96 + 16 * (2 + w) + 4 * q, hn2);
        } else {
            st4(state + (size_t)col * 96 + 16 * w + 4 * q, hn);
            if (w >= 2) st4(state + (size_t)col * 96 + 16 * (2 + w) + 4 * q, hn2);
        }
    }
    return cur;
}

// ---------------------------------------------------------------------------------------------------
// BALANCED four-wave form of the same conv + GRU, for a workgroup that has the CU to itself (one wave per SIMD, 512 VGPRs): the one-launch
// scene form (scene_lat.hip), where two of these 8-step recurrences are more than half of a call's critical path.
// A step is 576 fp32 16x16x4 MFMAs of gate sums + the gate functions.  Split by hidden tile (gru_lat4_body) two waves carry 200 MFMAs and
// two carry 104; here the unit is a GATE SUM of one hidden tile -- r_j, z_j: 8 k-tiles (32 MFMAs) each; the candidate's two halves
// ni_j = b + W_in e (2 k-tiles) and nh_j = b + W_hn h (6 k-tiles), which the GRU keeps apart anyway -- and every wave gets exactly 144:
//     wave pair p = 0, 1 owns hidden tiles A = 2p, B = 2p + 1, C = 4 + p
//     even wave:  r_A, z_A, r_C, nh_A, nh_C                      odd wave:  r_B, z_B, z_C, nh_B, ni_A, ni_B, ni_C
// One LDS exchange per step hands the five sums a partner needs across (ni_A, z_C, ni_C one way; r_C, nh_C the other), then the gate
// functions run balanced too: even wave tile A and registers 0, 1 of tile C, odd wave tile B and registers 2, 3 of tile C.  All 36
// weight fragments of a wave stay in registers for all steps (no LDS image).  Every sum is bias, then W_i k-tiles, then W_h k-tiles in
// order, every gate function the same expression: the bits of gru_cols_kernel / gru_lat_body / gru_lat4_body.
// sH: [2][6][64] f32x4 (h as B-operand fragments), sX: [2 pairs][5][64] f32x4 exchange.  Returns the sH buffer holding the final state.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float gru_gate(float ar, float az, float ai, float ah, float hp) {
    const float rg = sigmoid_prescaled(ar);
    const float zg = sigmoid_prescaled(az);
    const float ng = tanh_prescaled(fmaf(rg, ah, ai));
    return fmaf(zg, hp - ng, ng);  // (1-z) n + z h
}
template <int TPX, bool DLDS, bool ODD, class PRE>
__device__ __forceinline__ int gru_bal_half(const float* __restrict__ xin, const f32x4* __restrict__ convP, const float* __restrict__ convB,
                                            const f32x4* __restrict__ wihP, const f32x4* __restrict__ whhP, const float* __restrict__ gbias,
                                            int ncols, int Tp, int tile, f32x4 (*sH)[6][64], f32x4* sX, const f32x4* d_lds, PRE& pre) {
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int p = __builtin_amdgcn_readfirstlane(threadIdx.x >> 7);
    const int col = tile * 16 + c;
    const int colc = col < ncols ? col : ncols - 1;
    const int tA = 2 * p, tB = 2 * p + 1, tC = 4 + p;
    const int tO = ODD ? tB : tA;                          // the tile whose gate functions this wave runs in full
    // full units (bias, 2 W_i k-tiles, 6 W_h k-tiles): even (r, A) (z, A) (r, C);  odd (r, B) (z, B) (z, C)
    const int fg[3] = {0, 1, ODD ? 1 : 0};
    const int fj[3] = {tO, tO, tC};
    f32x4 fw[3][8], fb[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
#pragma unroll
        for (int T = 0; T < 2; ++T) fw[u][T] = wihP[((fg[u] * 6 + fj[u]) * 2 + T) * 64 + lane];
#pragma unroll
        for (int T = 0; T < 6; ++T) fw[u][2 + T] = whhP[((fg[u] * 6 + fj[u]) * 6 + T) * 64 + lane];
        fb[u] = ld4(gbias + fg[u] * 96 + 16 * fj[u] + 4 * q);
    }
    // candidate halves: nh (6 W_h k-tiles): even A and C, odd B;  ni (2 W_i k-tiles): odd A, B, C
    constexpr int NH = ODD ? 1 : 2, NI = ODD ? 3 : 1;
    const int hj[2] = {tO, tC};
    const int ij[3] = {tA, tB, tC};
    f32x4 hw[NH][6], hbias[NH], iw[NI][2], ibias[NI];
#pragma unroll
    for (int u = 0; u < NH; ++u) {
#pragma unroll
        for (int T = 0; T < 6; ++T) hw[u][T] = whhP[((2 * 6 + hj[u]) * 6 + T) * 64 + lane];
        hbias[u] = ld4(gbias + 3 * 96 + 16 * hj[u] + 4 * q);
    }
    if (ODD) {
#pragma unroll
        for (int u = 0; u < NI; ++u) {
#pragma unroll
            for (int T = 0; T < 2; ++T) iw[u][T] = wihP[((2 * 6 + ij[u]) * 2 + T) * 64 + lane];
            ibias[u] = ld4(gbias + 2 * 96 + 16 * ij[u] + 4 * q);
        }
    }
    const f32x4 cb0 = ld4(convB + 4 * q), cb1 = ld4(convB + 16 + 4 * q);
    f32x4 cw[2][TPX];
#pragma unroll
    for (int io = 0; io < 2; ++io)
#pragma unroll
        for (int T = 0; T < TPX; ++T) cw[io][T] = convP[(io * TPX + T) * 64 + lane];
    pre();
    f32x4 d[TPX];
#pragma unroll
    for (int T = 0; T < TPX; ++T) d[T] = DLDS ? d_lds[T * 64 + lane] : ld4(xin + (size_t)colc * (16 * TPX) + 16 * T + 4 * q);
    sH[0][tO][lane] = splat4(0.f);
    if (!ODD) sH[0][tC][lane] = splat4(0.f);
    lds_barrier();
    f32x4* X = sX + (size_t)p * 5 * 64 + lane;            // [0] r_C  [1] nh_C  (from even)   [2] ni_A  [3] z_C  [4] ni_C  (from odd)
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
        f32x4 a[3] = {fb[0], fb[1], fb[2]}, nh[NH], ni[NI];
#pragma unroll
        for (int u = 0; u < NH; ++u) nh[u] = hbias[u];
        if (ODD) {
#pragma unroll
            for (int u = 0; u < NI; ++u) ni[u] = ibias[u];
        }
#pragma unroll
        for (int T = 0; T < 2; ++T) {
#pragma unroll
            for (int u = 0; u < 3; ++u) a[u] = mfma_k16(a[u], fw[u][T], e[T]);
            if (ODD) {
#pragma unroll
                for (int u = 0; u < NI; ++u) ni[u] = mfma_k16(ni[u], iw[u][T], e[T]);
            }
        }
#pragma unroll
        for (int T = 0; T < 6; ++T) {
            const f32x4 hb = sH[cur][T][lane];
#pragma unroll
            for (int u = 0; u < 3; ++u) a[u] = mfma_k16(a[u], fw[u][2 + T], hb);
#pragma unroll
            for (int u = 0; u < NH; ++u) nh[u] = mfma_k16(nh[u], hw[u][T], hb);
        }
        if (ODD) { X[2 * 64] = ni[0]; X[3 * 64] = a[2]; X[4 * 64] = ni[2]; }
        else { X[0 * 64] = a[2]; X[1 * 64] = nh[1]; }
        lds_barrier();
        const f32x4 hpO = sH[cur][tO][lane], hpC = sH[cur][tC][lane];
        f32x4 hn;
        float* hc = reinterpret_cast<float*>(&sH[cur ^ 1][tC][lane]);
        if (ODD) {   // tile B in full (all four sums are this wave's); registers 2, 3 of tile C
#pragma unroll
            for (int r = 0; r < 4; ++r) hn[r] = gru_gate(a[0][r], a[1][r], ni[1][r], nh[0][r], hpO[r]);
            const f32x4 rC = X[0 * 64], nhC = X[1 * 64];
            hc[2] = gru_gate(rC[2], a[2][2], ni[2][2], nhC[2], hpC[2]);
            hc[3] = gru_gate(rC[3], a[2][3], ni[2][3], nhC[3], hpC[3]);
        } else {     // tile A in full (ni_A from the partner); registers 0, 1 of tile C
            const f32x4 niA = X[2 * 64], zC = X[3 * 64], niC = X[4 * 64];
#pragma unroll
            for (int r = 0; r < 4; ++r) hn[r] = gru_gate(a[0][r], a[1][r], niA[r], nh[0][r], hpO[r]);
            hc[0] = gru_gate(a[2][0], zC[0], niC[0], nh[1][0], hpC[0]);
            hc[1] = gru_gate(a[2][1], zC[1], niC[1], nh[1][1], hpC[1]);
        }
        sH[cur ^ 1][tO][lane] = hn;
        lds_barrier();
        cur ^= 1;
    }
    return cur;
}
template <int TPX, bool DLDS = false, bool SC1 = false, class PRE = GruNoPre>
__device__ __forceinline__ int gru_bal_body(const float* __restrict__ xin, const f32x4* __restrict__ convP, const float* __restrict__ convB,
                                            const f32x4* __restrict__ wihP, const f32x4* __restrict__ whhP, const float* __restrict__ gbias,
                                            float* __restrict__ state, int ncols, int Tp, int tile, f32x4 (*sH)[6][64], f32x4* sX,
                                            const f32x4* d_lds = nullptr, PRE pre = PRE()) {
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cur = (w & 1) ? gru_bal_half<TPX, DLDS, true, PRE>(xin, convP, convB, wihP, whhP, gbias, ncols, Tp, tile, sH, sX, d_lds, pre)
                            : gru_bal_half<TPX, DLDS, false, PRE>(xin, convP, convB, wihP, whhP, gbias, ncols, Tp, tile, sH, sX, d_lds, pre);
    if (!DLDS) {   // final state to memory: wave w stores hidden tile w, waves 0 and 1 also tiles 4 and 5 (the last barrier published them)
        const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
        const int col = tile * 16 + c;
        if (col < ncols) {
            float* p0 = state + (size_t)col * 96 + 16 * w + 4 * q;
            float* p1 = state + (size_t)col * 96 + 16 * (4 + (w & 1)) + 4 * q;
            if (SC1) { st4_sc1(p0, sH[cur][w][lane]); if (w < 2) st4_sc1(p1, sH[cur][4 + (w & 1)][lane]); }
            else { st4(p0, sH[cur][w][lane]); if (w < 2) st4(p1, sH[cur][4 + (w & 1)][lane]); }
        }
    }
    return cur;
}

// Per-agent layer-1 pre-activations of the decoder for ONE 16-agent tile -- the rows sttode_agent_preact computes with linear_cols_kernel
// (Decoder.forward's cat(past_feature, z) / DecomposeBlock's cat(hidden, state) hoisted to per-agent work: model/STTODE.py:323-328,71-75):
//     out[col][0:512] = W [B_0 .. B_{KT-1}] + b        KT = 14: [pf | state0] (A0x, A0y);  KT = 8: pf only (A1y)
// Wave w computes row tiles w, w+4, .., w+28 in two groups of four; the fragments of the next two k-tiles are in flight while one feeds the
// MFMAs (the weights come straight from L2: every fragment is used by exactly one wave).  Each output element is linear_cols_kernel's
// k-ordered chain: identical bits.  SC1: store the rows write-through for a consumer in ANOTHER workgroup of the same launch.
// out_lds != nullptr: the rows go to LDS as fragments [row tile][lane] instead (the one-launch scene path: the COLUMNS are then a
// trajectory tile's agents and the table feeds mlp_lat_run's layer 1 of the same workgroup; wave w reads back exactly what it wrote).
template <int KT, bool SC1>
__device__ __forceinline__ void preact_rows(const f32x4* __restrict__ WP, const float* __restrict__ bias, float* __restrict__ out,
                                            const f32x4 (&B)[14], int col, bool live, int lane, int q, int wv, f32x4* out_lds = nullptr) {
    constexpr int D = 4;   // k-tiles of weight fragments in flight per row tile (L2 latency under load ~2-3 us vs 0.25 us of MFMAs per k-tile)
#pragma unroll 1
    for (int rg = 0; rg < 2; ++rg) {
        f32x4 acc[4], w[D][4];
        const f32x4* wp[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rt = wv + 4 * (4 * rg + i);
            wp[i] = WP + (size_t)rt * KT * 64 + lane;
            acc[i] = ld4(bias + 16 * rt + 4 * q);
#pragma unroll
            for (int dd = 0; dd < D; ++dd) w[dd][i] = wp[i][dd * 64];
        }
#pragma unroll
        for (int T = 0; T < KT; ++T) {
            f32x4 wc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) wc[i] = w[T % D][i];
            if (T + D < KT) {
#pragma unroll
                for (int i = 0; i < 4; ++i) w[T % D][i] = wp[i][(T + D) * 64];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = mfma_k16(acc[i], wc[i], B[T]);
        }
        if (out_lds) {
#pragma unroll
            for (int i = 0; i < 4; ++i) out_lds[(wv + 4 * (4 * rg + i)) * 64 + lane] = acc[i];
        } else if (live) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float* p = out + (size_t)col * 512 + 16 * (wv + 4 * (4 * rg + i)) + 4 * q;
                if (SC1) st4_sc1(p, acc[i]);
                else st4(p, acc[i]);
            }
        }
    }
}

// The same rows for a TRAJECTORY tile's own columns (the one-launch scene form; two passes there: pf arrives before state0): k-tiles
// [T0, T1) of all 32 row tiles (wave w: row tiles
// w, w+4, ..; acc[8], initialised with the bias by the caller) as ONE pipeline of 2 (T1 - T0) steps -- step s = (row-tile quad s / N,
// k-tile T0 + s % N) -- with D steps of weight fragments in flight; preact_prime issues the first D steps (weights only: callable BEFORE
// the data the pass multiplies has arrived).  Each element is still bias, then k-tiles in order: preact_rows' bits.
template <int KT, int T0, int T1, int D>
__device__ __forceinline__ void preact_load(const f32x4* __restrict__ WP, f32x4 (&w)[4], int s, int lane, int wv) {
    constexpr int N = T1 - T0;
    const int rg = s / N, T = s % N;
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = WP[((size_t)(wv + 4 * (4 * rg + i)) * KT + T0 + T) * 64 + lane];
}
template <int KT, int T0, int T1, int D>
__device__ __forceinline__ void preact_prime(const f32x4* __restrict__ WP, f32x4 (&w)[D][4], int lane, int wv) {
#pragma unroll
    for (int s = 0; s < D; ++s) preact_load<KT, T0, T1, D>(WP, w[s], s, lane, wv);
}
template <int KT, int T0, int T1, int D>
__device__ __forceinline__ void preact_run(const f32x4* __restrict__ WP, f32x4 (&w)[D][4], f32x4 (&acc)[8], const f32x4 (&B)[14], int lane, int wv) {
    constexpr int N = T1 - T0, S = 2 * N;
    static_assert(D <= S, "prefetch depth beyond the pass");
#pragma unroll
    for (int s = 0; s < S; ++s) {
        f32x4 wc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) wc[i] = w[s % D][i];
        if (s + D < S) preact_load<KT, T0, T1, D>(WP, w[s % D], s + D, lane, wv);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[4 * (s / N) + i] = mfma_k16(acc[4 * (s / N) + i], wc[i], B[T0 + s % N]);
    }
}

// ---------------------------------------------------------------------------------------------------
// Latency form of the decoder MLPs (see the comment at the kernels in decoder.hip)
// ---------------------------------------------------------------------------------------------------
struct MlpLatArgs {
    const float* A0; const f32x4* blob; const float* z; const float* state; const float* xpad; const float* ybuf; const float* cur;
    const float* orig; float* out; int ncols, K, Tf2;
    // LDSIO forms (the one-launch scene path keeps a tile's d and GRU state in LDS, as B-operand fragments [k-tile][lane]):
    f32x4* out_lds;            // MODE 0: d = x_true - x_hat0 goes here instead of `out`
    const f32x4* state_lds;    // MODE 2: the block-1 GRU state comes from here instead of `state`
    const f32x4* a0_lds;       // != nullptr: layer-1 pre-activations of THIS tile's columns as fragments [32 row tiles][64] instead of A0[agent]
};
template <int KTV>
struct MlpLatFrag {            // one group's operands of one wave
    f32x4 a0;                  // A0[agent][16 (4g + w) + 4q ..]
    f32x4 w1[KTV];             // W1v tiles of chunk 4g + w
    f32x4 w2[4][4];            // [chunk of the group][own row tile]
};
// DEEP: weight fragments travel TWO groups ahead (three register sets; a workgroup that has the CU to itself, 512 VGPRs per wave): one group
// of MFMAs (~1 us) does not cover an L2 round trip plus the group's 72-100 KiB at the CU's 64 B/clk, two do.  Same arithmetic, same order.
template <int KTV, int NO, int MODE, bool SC1 = false, bool LDSIO = false, bool DEEP = false>
__device__ __forceinline__ void mlp_lat_run(const MlpLatArgs& a, f32x4* sH1, f32x4* sH2, int tile) {
    constexpr int CHW = (KTV + 16) * 64;            // f32x4 per chunk
    constexpr int NR = (NO + 3) / 4;                 // output tiles this wave finishes: o = wave, wave + 4
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = tile * 16 + c;
    const int colc = col < a.ncols ? col : a.ncols - 1;
    const int agent = colc / a.K;
    const float* arow = a.A0 + (size_t)agent * 512 + 4 * q;
    const f32x4* wl = a.blob + lane;
    auto fetch = [&](MlpLatFrag<KTV>& f, int g) {
        f.a0 = (LDSIO && a.a0_lds) ? a.a0_lds[(4 * g + wave) * 64 + lane] : ld4(arow + 16 * (4 * g + wave));
        const f32x4* own = wl + (size_t)(4 * g + wave) * CHW;
#pragma unroll
        for (int T = 0; T < KTV; ++T) f.w1[T] = own[T * 64];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
#pragma unroll
            for (int i = 0; i < 4; ++i) f.w2[cc][i] = wl[(size_t)(4 * g + cc) * CHW + (KTV + 4 * wave + i) * 64];
    };
    f32x4 B[KTV];
    B[0] = ld4(a.z + (size_t)colc * 32 + 4 * q);
    B[1] = ld4(a.z + (size_t)colc * 32 + 16 + 4 * q);
    if (KTV == 8) {
#pragma unroll
        for (int T = 0; T < 6; ++T)
            B[2 + (T < KTV - 2 ? T : 0)] = LDSIO ? a.state_lds[T * 64 + lane] : ld4(a.state + (size_t)colc * 96 + 16 * T + 4 * q);
    }
    f32x4 acc2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc2[i] = splat4(0.f);
    auto group = [&](const MlpLatFrag<KTV>& f, int g) {
        f32x4 h1 = f.a0;
#pragma unroll
        for (int T = 0; T < KTV; ++T) h1 = mfma_k16(h1, f.w1[T], B[T]);
        f32x4* hb = sH1 + (g & 1) * 256;
        hb[wave * 64 + lane] = relu4(h1);
        lds_barrier();                                // (not __syncthreads: the next groups' weight loads stay in flight)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const f32x4 hv = hb[cc * 64 + lane];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc2[i] = mfma_k16(acc2[i], f.w2[cc][i], hv);
        }
    };
    const f32x4* l3 = a.blob + (size_t)32 * CHW;
    f32x4 w3[NR][16], b3v[NR], b2v[4];
    // layer 3 operands travel during the last group(s): b2 sits behind the first layer-3 chunk's 16 tiles and its b3 (packing.mlp_stream)
    auto fetch3 = [&]() {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int o = wave + 4 * i < NO ? wave + 4 * i : NO - 1;
#pragma unroll
            for (int T = 0; T < 16; ++T) w3[i][T] = l3[(size_t)o * CHW + T * 64 + lane];
            b3v[i] = ld4(reinterpret_cast<const float*>(l3 + (size_t)o * CHW + 16 * 64) + 4 * q);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) b2v[i] = ld4(reinterpret_cast<const float*>(l3 + 16 * 64) + 16 + 16 * (4 * wave + i) + 4 * q);
    };
    if (DEEP) {
        MlpLatFrag<KTV> f0, f1, f2;
        fetch(f0, 0); fetch(f1, 1); STT_FENCE();
        fetch(f2, 2); STT_FENCE(); group(f0, 0); STT_FENCE();
        fetch(f0, 3); STT_FENCE(); group(f1, 1); STT_FENCE();
        fetch(f1, 4); STT_FENCE(); group(f2, 2); STT_FENCE();
        fetch(f2, 5); STT_FENCE(); group(f0, 3); STT_FENCE();
        fetch(f0, 6); STT_FENCE(); group(f1, 4); STT_FENCE();
        fetch(f1, 7); STT_FENCE(); group(f2, 5); STT_FENCE();
        fetch3();     STT_FENCE(); group(f0, 6); STT_FENCE();
        group(f1, 7);
    } else {
        MlpLatFrag<KTV> fa, fb;
        fetch(fa, 0);
#pragma unroll 1
        for (int g = 0; g < 6; g += 2) {
            fetch(fb, g + 1);
            group(fa, g);
            fetch(fa, g + 2);
            group(fb, g + 1);
        }
        fetch(fb, 7);
        group(fa, 6);
        fetch3();
        group(fb, 7);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) sH2[(4 * wave + i) * 64 + lane] = relu4(acc2[i] + b2v[i]);
    lds_barrier();                                   // the whole 256-wide layer-2 activation as B-operand fragments
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int o = wave + 4 * i;
        if (o >= NO) continue;
        f32x4 v = b3v[i];
#pragma unroll
        for (int T = 0; T < 16; ++T) v = mfma_k16(v, w3[i][T], sH2[T * 64 + lane]);
        if (MODE == 0 && LDSIO) {                    // (columns past the end carry the last column's values: never stored anywhere)
            const f32x4 dv = ld4(a.xpad + (size_t)agent * (16 * NO) + 16 * o + 4 * q) - v;
            a.out_lds[o * 64 + lane] = dv;
            if (a.out && col < a.ncols) st4(a.out + (size_t)col * (16 * NO) + 16 * o + 4 * q, dv);   // the workspace copy (diagnostic views)
            continue;
        }
        if (col >= a.ncols) continue;
        if (MODE == 0) {
            const f32x4 xt = ld4(a.xpad + (size_t)agent * (16 * NO) + 16 * o + 4 * q);
            st4(a.out + (size_t)col * (16 * NO) + 16 * o + 4 * q, xt - v);
        } else if (MODE == 1) {
            if (SC1) st4_sc1(a.out + (size_t)col * (16 * NO) + 16 * o + 4 * q, v);
            else st4(a.out + (size_t)col * (16 * NO) + 16 * o + 4 * q, v);
        } else {
            const int row0 = 16 * o + 4 * q;
            if (row0 < a.Tf2) {
                const float cx = a.cur[2 * agent], cy = a.cur[2 * agent + 1];
                const float ox = a.orig[2 * agent], oy = a.orig[2 * agent + 1];
                const f32x4 y0 = ld4(a.ybuf + (size_t)col * (16 * NO) + row0);
                f32x4 r;
                r[0] = ((y0[0] + v[0]) + cx) + ox;
                r[1] = ((y0[1] + v[1]) + cy) + oy;
                r[2] = ((y0[2] + v[2]) + cx) + ox;
                r[3] = ((y0[3] + v[3]) + cy) + oy;
                float* pp = a.out + (size_t)col * a.Tf2 + row0;
                if (row0 + 3 < a.Tf2 && (a.Tf2 & 3) == 0) {
                    st4(pp, r);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (row0 + e < a.Tf2) pp[e] = r[e];
                }
            }
        }
    }
}

