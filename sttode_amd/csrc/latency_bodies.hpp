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
template <bool ODE>
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
        st4(pf + (size_t)col * 128 + 16 * wv + 4 * q, go);
        st4(pf + (size_t)col * 128 + 64 + 16 * wv + 4 * q, relu4(xo));
    }
}


// ---------------------------------------------------------------------------------------------------
// Bodies used only by the per-agent ROLE of the fused chain launch (chain32.hip): a 256-thread workgroup with <= 256 VGPRs per wave
// ---------------------------------------------------------------------------------------------------
// 16-byte agent-scope (sc1, write-through) store: the payload form of an in-launch hand-off to ANOTHER workgroup -- no release fence is
// needed behind it (cdna_hip_programming.md §6 Guideline 16, R1).  hipcc does not count an asm store: every storing wave runs
// `s_waitcnt vmcnt(0)` itself before the workgroup's flag is published.  (`s_nop 1`: §5.7 item 1 -- the data registers may otherwise be
// overwritten before the store has read them.)
__device__ __forceinline__ void st4_sc1(float* p, const f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}

// Four-wave form of gru_lat_body (block-0 conv + GRU of ONE 16-agent tile): wave w keeps the 24 weight fragments of hidden tile w in
// registers for all steps; hidden tiles 4 and 5 are computed by waves 0 and 1 as well, from fragments staged once in LDS
// (sW45: [2][3 gates][8 k-tiles][64] f32x4 = 48 KiB, by LDS-DMA).  A split over output rows: every element of h is the k-ordered chain
// of gru_lat_body / gru_cols_kernel, so the bits are those of the other forms.  Returns the index of the sH buffer that holds the final
// hidden state (all six tiles, B-operand fragment layout), which the caller may feed straight into the next layer.
template <int TPX>
__device__ __forceinline__ int gru_lat4_body(const float* __restrict__ xin, const f32x4* __restrict__ convP, const float* __restrict__ convB,
                                             const f32x4* __restrict__ wihP, const f32x4* __restrict__ whhP, const float* __restrict__ gbias,
                                             float* __restrict__ state, int ncols, int Tp, int tile, f32x4 (*sH)[6][64], f32x4* sW45) {
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
    const int w2 = w < 2 ? 4 + w : 4;   // second hidden tile of waves 0 and 1
#pragma unroll
    for (int g = 0; g < 4; ++g) { b0[g] = ld4(gbias + g * 96 + 16 * w + 4 * q); b1[g] = ld4(gbias + g * 96 + 16 * w2 + 4 * q); }
    const f32x4 cb0 = ld4(convB + 4 * q), cb1 = ld4(convB + 16 + 4 * q);
    f32x4 d[TPX];
#pragma unroll
    for (int T = 0; T < TPX; ++T) d[T] = ld4(xin + (size_t)colc * (16 * TPX) + 16 * T + 4 * q);
    f32x4 hn = splat4(0.f), hn2 = splat4(0.f);
    sH[0][w][lane] = hn;
    if (w < 2) sH[0][4 + w][lane] = hn2;
    f32x4 cw[2][TPX];
#pragma unroll
    for (int io = 0; io < 2; ++io)
#pragma unroll
        for (int T = 0; T < TPX; ++T) cw[io][T] = convP[(io * TPX + T) * 64 + lane];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA pieces have landed; the barrier publishes everybody's
    __syncthreads();
    const f32x4* wB = sW45 + (size_t)(w & 1) * (24 * 64) + lane;
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
        {
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
        }
        if (w < 2) {   // wave-uniform: hidden tile 4 + w from the LDS image
            STT_FENCE();
            f32x4 ar = b1[0], az = b1[1], ai = b1[2], ah = b1[3];
#pragma unroll
            for (int T = 0; T < 2; ++T) {
                ar = mfma_k16(ar, wB[(0 * 8 + T) * 64], e[T]);
                az = mfma_k16(az, wB[(1 * 8 + T) * 64], e[T]);
                ai = mfma_k16(ai, wB[(2 * 8 + T) * 64], e[T]);
            }
            const f32x4 hp = sH[cur][4 + w][lane];
#pragma unroll
            for (int T = 0; T < 6; ++T) {
                const f32x4 hb = sH[cur][T][lane];
                ar = mfma_k16(ar, wB[(0 * 8 + 2 + T) * 64], hb);
                az = mfma_k16(az, wB[(1 * 8 + 2 + T) * 64], hb);
                ah = mfma_k16(ah, wB[(2 * 8 + 2 + T) * 64], hb);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float rg = sigmoid_prescaled(ar[r]);
                const float zg = sigmoid_prescaled(az[r]);
                const float ng = tanh_prescaled(fmaf(rg, ah[r], ai[r]));
                hn2[r] = fmaf(zg, hp[r] - ng, ng);
            }
            sH[cur ^ 1][4 + w][lane] = hn2;
        }
        sH[cur ^ 1][w][lane] = hn;
        __syncthreads();
        cur ^= 1;
    }
    if (col < ncols) {
        st4(state + (size_t)col * 96 + 16 * w + 4 * q, hn);
        if (w < 2) st4(state + (size_t)col * 96 + 16 * (4 + w) + 4 * q, hn2);
    }
    return cur;
}

// Per-agent layer-1 pre-activations of the decoder for ONE 16-agent tile -- the rows sttode_agent_preact computes with linear_cols_kernel
// (Decoder.forward's cat(past_feature, z) / DecomposeBlock's cat(hidden, state) hoisted to per-agent work: model/STTODE.py:323-328,71-75):
//     out[col][0:512] = W [B_0 .. B_{KT-1}] + b        KT = 14: [pf | state0] (A0x, A0y);  KT = 8: pf only (A1y)
// Wave w computes row tiles w, w+4, .., w+28 in two groups of four; the fragments of the next two k-tiles are in flight while one feeds the
// MFMAs (the weights come straight from L2: every fragment is used by exactly one wave).  Each output element is linear_cols_kernel's
// k-ordered chain: identical bits.  SC1: store the rows write-through for a consumer in ANOTHER workgroup of the same launch.
template <int KT, bool SC1>
__device__ __forceinline__ void preact_rows(const f32x4* __restrict__ WP, const float* __restrict__ bias, float* __restrict__ out,
                                            const f32x4 (&B)[14], int col, bool live, int lane, int q, int wv) {
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
        if (live) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float* p = out + (size_t)col * 512 + 16 * (wv + 4 * (4 * rg + i)) + 4 * q;
                if (SC1) st4_sc1(p, acc[i]);
                else st4(p, acc[i]);
            }
        }
    }
}
