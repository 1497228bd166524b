// THROUGHPUT form of the per-agent stage (round 4).  Included by chain32.hip behind ChainStreamT / tile_mma / gru32_steps.
//
// The round-3 roles (role_body.hpp) are latency forms: one 16-agent tile per workgroup, rows split over the waves, an LDS exchange and a
// barrier per layer, weights straight from L2 -- 150-200 us per tile beside a pipe-saturating trajectory group, 9-12 % of the chip's
// workgroup slot-time for 5 % of the FLOP, and the groups of the SAME launch wait for them (small launches are bound by exactly that
// latency).  Here a workgroup takes 128 agents -- a wave's 32 agents on the 32 MFMA columns, features in registers, every weight streamed
// L2 -> LDS as PK32 tiles by the trajectory chain's own machinery (ChainStreamT: LDS-DMA chunks of <= 3 tiles, double buffered, one barrier
// per chunk) -- and runs, with no exchange between waves at all:
//     block-0 conv + GRU (gru32_steps)  ->  E: g = Wc x + bc, info / gate  ->  LN1  ->  FFN 64 -> 1024 -> 64  ->  LN2  ->  Euler + relu -> pf
//     ->  the three layer-1 pre-activation tables  A0x, A0y (k = [pf | state0])  and  A1y (k = pf)
// (model/STTODE.py:214-236 PastEncoder.forward, ode_demo.py:186-190,217-231, hypertransformer.py:55-89,134-153, model/STTODE.py:62-69 and the
// per-agent part of :71-75).  722 tiles of 16 MFMAs per wave at ETH shapes -- 0.9 of a trajectory group -- for 128 agents: the per-agent
// stage of 512 scenes is 68 such workgroups instead of 541 latency roles.  Its latency (one group's) is hidden by LAG: the launch of call k
// carries the roles of call k and the trajectory groups of the call made `streams` calls EARLIER on the same stream (pipeline.hip), so
// nothing in a launch waits for anything else in it: no flags, no spinning, no ordering assumption on the dispatcher.
// The embedding is folded on the host (packing.role_fold): input_fc -> positional fc -> input_fc2 -> input_fc3 is affine in eval mode, and
// with attention length 1 so is everything up to the gate nonlinearity.  Attention groups > 1 (NBA): embed_qkv and mhgsa_attn run as
// launches in front; g and the attention output are read from the workspace.
#pragma once

struct R32C {   // == packing.R32_CONSTS
    static constexpr int bc = 0, wlast = 64, bi = 128, bg = 192, ln1w = 256, ln1b = 320, l1b = 384, l2b = 1408, ln2w = 1472, ln2b = 1536,
                         gb = 1600, cb = 1984, b1x = 2016, b1y = 2528, b11 = 3040, total = 3552;
};

// (struct Role32Args: chain32.hip, beside ChainArgs)
// The role reads its arguments straight from the kernel-argument segment (constant address space: scalar loads where they are needed).
// Taken from the by-value ChainArgs parameter instead, hipcc loads 64 bytes of them at the top of the kernel -- in front of the branch
// that separates roles from groups -- and, with every VGPR taken by the groups' code, parks them in SCRATCH: 128 B per lane of every wave.
typedef const __attribute__((address_space(4))) Role32Args KRole32Args;

// The stream as a flat sequence of tiles: mma() multiplies the next tile of the program into an accumulator and steps over chunk boundaries
// (prefetch of the following chunk, barrier) wherever they fall -- a layer need not be a whole number of chunks.  Fragment look-ahead as in
// the trajectory chain (tile_mma2): the fragments of the next tile OF THE SAME CHUNK are requested halfway through the current tile's
// MFMAs; the first tile of a chunk is read before the chunk's DMA pieces are issued.  (Without it a role alone on its SIMD ran at 71 % of
// its MFMA time, and beside a trajectory group it outlasted the group: 0.8-1.0 ms against 0.73 -- the long pole of every small launch.)
// Uniform control flow: every wave of the workgroup asks for the same tiles in the same order.
template <class ST> struct TileFeed {
    ST& st; int t, cnt; bool pre;
    Frag cur, nxt;
    __device__ __forceinline__ explicit TileFeed(ST& s) : st(s), t(0), cnt(0), pre(false) {}
    __device__ __forceinline__ void open() {        // the current chunk has landed and passed its barrier; begin() was not yet called for it
        cnt = __builtin_amdgcn_readfirstlane(st.lprog[st.p].y);
        t = 0;
        ldfrag(cur, st.cur());                      // first tile of the chunk: its read latency overlaps the DMA issue below
        st.begin();
        pre = false;
    }
    __device__ __forceinline__ void mma(f32x16& acc, const f32x16& B) {
        if (t == cnt) { st.end(); open(); }
        else if (pre) cur = nxt;                    // (16 register moves per 16 MFMAs; the arrays must not be indexed by a run-time value)
        const f32x4* tn = t + 1 < cnt ? st.cur() + (t + 1) * C32_TILE : nullptr;
        tile_mma2(acc, cur, B, nxt, tn);
        pre = tn != nullptr;
        ++t;
    }
    __device__ __forceinline__ void close() { st.end(); }   // drains the last prefetch (nobody reads it) before the workgroup leaves
};

__device__ __forceinline__ float halfsum32(float v) { return v + __shfl_xor(v, 32, 64); }   // lanes c and c + 32 hold the two halves of a column

// LayerNorm over the 64 features of a column held as two 32-feature tiles in accumulator layout (chain.hpp layernorm64 for this layout)
__device__ __forceinline__ void layernorm64_c32(f32x16 (&x)[2], const float* gamma, const float* beta, int h) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; r += 4) s += (x[j][r] + x[j][r + 1]) + (x[j][r + 2] + x[j][r + 3]);
    const float mean = halfsum32(s) * (1.0f / 64.0f);
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float dd = x[j][r] - mean; v += dd * dd; }
    const float rstd = 1.0f / sqrtf(halfsum32(v) * (1.0f / 64.0f) + 1e-5f);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const f32x16 g = ldrows(gamma + 32 * j, h), b = ldrows(beta + 32 * j, h);
#pragma unroll
        for (int r = 0; r < 16; ++r) x[j][r] = (x[j][r] - mean) * rstd * g[r] + b[r];
    }
}

// 32 consecutive features of this lane's column -> global memory (the inverse of ldrows)
__device__ __forceinline__ void strows(float* p, const f32x16& v, int h) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const f32x4 w = {v[4 * a], v[4 * a + 1], v[4 * a + 2], v[4 * a + 3]};
        st4(p + 8 * a + 4 * h, w);
    }
}

// one layer-1 table: out[col][32 rt ..] = bias + W[rt] B  for the 16 row tiles of the 512 pre-activations, KT k-tiles each
template <int KT, class FD>
__device__ __forceinline__ void table32(FD& fd, const float* bias, float* out, const f32x16 (&B)[7], int col, bool live, int h) {
#pragma unroll 1
    for (int rt = 0; rt < 16; ++rt) {
        f32x16 acc = ldrows(bias + 32 * rt, h);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) fd.mma(acc, B[kt]);
        if (live) strows(out + (size_t)col * 512 + 32 * rt, acc, h);
    }
}

// Latents of the role's call drawn by the launch itself: z [n K][32] ~ N(0, I), the prior samples of Decoder.forward (model/STTODE.py:609-616;
// Normal.rsample :89-93 is torch.randn_like there).  Philox4x32-10 (Salmon et al., SC'11; the counter-based generator torch and cuRAND use),
// key = 64 bits the host draws from torch's generator per call, counter = index of the float4 in z; two Box-Muller pairs per block.  As a
// separate torch.randn kernel the draw waited ~0.7-1 ms per call for workgroup slots on a chip the chain launches keep full and held the
// CU halves it got all that time (profiles/r04/cadence_*.txt); here it is ~10 k VALU instructions per role lane, issued beside the other
// wave's MFMAs.  Workgroup wg writes the rows of its own 128 agents; the trajectory groups read them two launches later.
__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0], p1 = (unsigned long long)0xCD9E8D57u * c[2];
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1;
        c[0] = n0; c[1] = (unsigned)p1; c[2] = n2; c[3] = (unsigned)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ void latents32(float* z, unsigned k0, unsigned k1, int wg, int n, int K) {
    const int a0 = wg * 128, na = n - a0 < 128 ? n - a0 : 128;
    const long first4 = (long)a0 * K * 8;                       // float4 index of the first row of this workgroup's agents (32 floats per row)
    const int count4 = na * K * 8;
    for (int i = threadIdx.x; i < count4; i += blockDim.x) {
        const unsigned long long g4 = (unsigned long long)(first4 + i);
        unsigned c[4] = {(unsigned)g4, (unsigned)(g4 >> 32), 0u, 0u};
        philox4x32_10(c, k0, k1);
        f32x4 v;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const float u1 = (float)(c[2 * p] >> 8) * 5.9604644775390625e-8f + 2.98023223876953125e-8f;      // (0, 1): (x >> 8) 2^-24 + 2^-25
            const float u2 = (float)(c[2 * p + 1] >> 8) * 5.9604644775390625e-8f + 2.98023223876953125e-8f;
            const float rad = sqrtf(-2.0f * __logf(u1));
            float sn, cs;
            __sincosf(6.283185307179586f * u2, &sn, &cs);
            v[2 * p] = rad * cs; v[2 * p + 1] = rad * sn;
        }
        st4(z + (first4 + i) * 4, v);
    }
}

// STTODENet.set_data for the workgroup's 128 agents (model/STTODE.py:397-461), one thread per agent: the agent's scene by binary search in
// the CSR, the scene origin = mean of the scene's last observed positions summed in agent order (as scene_orig_kernel does: identical bits),
// then the normalised track, velocities, cur_location and the last-agent flag (agent_inputs_core, frontend_body.hpp) into the workspace
// rows the other phases of this workgroup and -- two launches later -- the trajectory groups read.
__device__ __forceinline__ void frontend32(KRole32Args& R, int wg) {
    const int a = wg * 128 + (int)threadIdx.x;
    if (threadIdx.x < 128 && a < R.n) {
        int lo = 0, hi = R.S - 1;                     // largest s with scene_ptr[s] <= a
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (R.scene_ptr[mid] <= a) lo = mid; else hi = mid - 1;
        }
        const int a0 = R.scene_ptr[lo], a1 = R.scene_ptr[lo + 1];
        float sx = 0.f, sy = 0.f;
        for (int aa = a0; aa < a1; ++aa) {
            sx += R.past[((size_t)aa * R.Tp + (R.Tp - 1)) * 2 + 0];
            sy += R.past[((size_t)aa * R.Tp + (R.Tp - 1)) * 2 + 1];
        }
        const float inv = (float)(a1 - a0);
        const float ox = sx / inv, oy = sy / inv;
        if (a == a0) { R.scene_orig[2 * lo] = ox; R.scene_orig[2 * lo + 1] = oy; }
        R.agent_scene[a] = lo;
        // (no preloaded track handed over by pointer: an array whose address is taken lives in scratch -- 128 B per lane of EVERY wave of the
        // launch for a value 128 lanes use once)
        agent_inputs_core<false, 16>(a, R.past, R.Tp, R.ldx / 16, 1, ox, oy, a == a1 - 1, nullptr, R.xpad_w, R.enc_in_w, R.cur_w, R.orig_w, R.last_w);
    }
}

// The whole per-agent stage of 128 agents: workgroup `wg` of R.nwg, 4 waves x 32 columns.  smem: ring (24 KiB) | consts | program.
__device__ __forceinline__ void role32_body(KRole32Args& R, int wg, char* smem) {
    f32x4* ring = reinterpret_cast<f32x4*>(smem);
    float* cst = reinterpret_cast<float*>(ring + C32_RING);
    int2* lprog = reinterpret_cast<int2*>(cst + R32C::total);
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (R.prio > 0) __builtin_amdgcn_s_setprio(3);     // (uniform) the role waves win their SIMD's issue arbitration: they are the long pole of a small launch
    if (wg == 0 && threadIdx.x == 0) *R.counter = 0;   // the work queue of this call's trajectory groups (read by a LATER launch of this stream)
    if (R.m_ade && threadIdx.x < 128 && wg * 128 + (int)threadIdx.x < R.n) {   // fused metrics of this call: the minimum starts at +inf
        R.m_ade[wg * 128 + threadIdx.x] = INFINITY;
        R.m_fde[wg * 128 + threadIdx.x] = INFINITY;
    }
    if (R.past) frontend32(R, wg);                                   // (uniform) scene batches: set_data for this workgroup's agents
    if (R.zgen) latents32(R.zgen, R.zkey0, R.zkey1, wg, R.n, R.K);   // (uniform)
    if (R.past) {   // the rows written above are read by other threads of this workgroup below
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    for (int i = threadIdx.x; i < R32C::total; i += blockDim.x) cst[i] = R.consts[i];
    for (int i = threadIdx.x; i < R.prog_len; i += blockDim.x) lprog[i] = R.prog[i];
    __syncthreads();
    ChainStream st;
    st.init(R.pool, lprog, R.prog_len, ring);
    const int col = wg * 128 + wave * 32 + c;
    const bool live = col < R.n;
    const int colc = live ? col : R.n - 1;
    f32x16 d;
    {
        const float* xp = R.xpad + (size_t)colc * R.ldx;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            f32x4 v = splat4(0.f);
            if (8 * a + 4 * h < R.ldx) v = ld4(xp + 8 * a + 4 * h);
#pragma unroll
            for (int b = 0; b < 4; ++b) d[4 * a + b] = v[b];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                            // chunk 0 (the first conv tile) has landed
    f32x16 hs[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) hs[j] = splat16(0.f);
    gru32_steps(st, cst + R32C::gb, cst + R32C::cb, d, hs, R.Tp, h);   // block-0 conv + GRU (model/STTODE.py:62-69, x_hat = 0)
    if (live) {
#pragma unroll
        for (int j = 0; j < 3; ++j) strows(R.state0 + (size_t)col * 96 + 32 * j, hs[j], h);
    }
    STT_FENCE();
    TileFeed<ChainStream> fd(st);
    fd.open();
    f32x16 G[2], S[2];   // g = ftraj_input; S = what info / gate read (scenes: g itself, the foldings absorb v and out_proj; NBA: the attention output)
    if (R.attn == nullptr) {   // (uniform)
        f32x16 X[2];
        const float* ep = R.enc_in + (size_t)colc * (4 * R.Tp);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                f32x4 v = splat4(0.f);
                if (32 * kt + 8 * a + 4 * h < 4 * R.Tp) v = ld4(ep + 32 * kt + 8 * a + 4 * h);   // (4 Tp is a multiple of 4: whole f32x4 or nothing)
#pragma unroll
                for (int b = 0; b < 4; ++b) X[kt][4 * a + b] = v[b];
            }
        const float lastf = R.last[colc] ? 1.0f : 0.0f;      // category one-hot of the scene's last agent (model/STTODE.py:199-210)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const f32x16 b = ldrows(cst + R32C::bc + 32 * j, h), wl = ldrows(cst + R32C::wlast + 32 * j, h);
#pragma unroll
            for (int r = 0; r < 16; ++r) G[j][r] = fmaf(wl[r], lastf, b[r]);
            fd.mma(G[j], X[0]);
            if (R.kte > 1) fd.mma(G[j], X[1]);
        }
        S[0] = G[0]; S[1] = G[1];
    } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            G[j] = ldrows(R.g_in + (size_t)colc * 64 + 32 * j, h);
            S[j] = ldrows(R.attn + (size_t)colc * R.ld_attn + 32 * j, h);
        }
    }
    f32x16 XR[2];        // h = LN1(y + tanh(info) * sigmoid(gate))  (hypertransformer.py:81-83,148)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        f32x16 vi = ldrows(cst + R32C::bi + 32 * j, h);
        fd.mma(vi, S[0]);
        fd.mma(vi, S[1]);
        f32x16 vg = ldrows(cst + R32C::bg + 32 * j, h);
        fd.mma(vg, S[0]);
        fd.mma(vg, S[1]);
#pragma unroll
        for (int r = 0; r < 16; ++r) XR[j][r] = G[j][r] + tanhf(vi[r]) * sigmoidf_(vg[r]);
    }
    layernorm64_c32(XR, cst + R32C::ln1w, cst + R32C::ln1b, h);
    {   // FFN 64 -> 1024 relu -> 64 (hypertransformer.py:149-151): per 32-row hidden tile 2 + 2 tiles
        f32x16 FF[2];
        FF[0] = splat16(0.f); FF[1] = splat16(0.f);
#pragma unroll 1
        for (int ht = 0; ht < 32; ++ht) {
            f32x16 hid = ldrows(cst + R32C::l1b + 32 * ht, h);
            fd.mma(hid, XR[0]);
            fd.mma(hid, XR[1]);
            hid = relu16(hid);
            fd.mma(FF[0], hid);
            fd.mma(FF[1], hid);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const f32x16 b = ldrows(cst + R32C::l2b + 32 * j, h);
            XR[j] = XR[j] + (FF[j] + b);
        }
    }
    layernorm64_c32(XR, cst + R32C::ln2w, cst + R32C::ln2b, h);
    f32x16 B7[7];        // [g | relu(g + T f(g))] = pf, then state0: the B operand of the tables
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        B7[j] = G[j];
        B7[2 + j] = relu16(G[j] + XR[j] * R.ode_time);          // one explicit Euler step over [0, T] (ode_demo.py:186-190), relu (:231)
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) B7[4 + j] = hs[j];
    if (live) {
#pragma unroll
        for (int j = 0; j < 4; ++j) strows(R.pf + (size_t)col * 128 + 32 * j, B7[j], h);
    }
    table32<7>(fd, cst + R32C::b1x, R.A0x, B7, col, live, h);
    table32<7>(fd, cst + R32C::b1y, R.A0y, B7, col, live, h);
    table32<4>(fd, cst + R32C::b11, R.A1y, B7, col, live, h);
    fd.close();
}
