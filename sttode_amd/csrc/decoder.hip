// Decoder hot path: the residual-decomposition decoder of model/STTODE.py:16-77,302-347 (reference),
// re-designed for CDNA4 as three kernel families over "columns" (agents or trajectories):
//
//   gru_cols      conv1d(k=3) + relu + GRU(32->96) over Tp steps, recurrent weights RESIDENT IN LDS
//                 (W_ih 36 KiB + W_hh 108 KiB PK16-packed, loaded once per persistent workgroup), hidden
//                 state in registers, every gate GEMM on v_mfma_f32_16x16x4_f32.
//                 block 0 runs once per AGENT (x_hat = 0 => the K samples of an agent share it),
//                 block 1 runs once per TRAJECTORY.
//   mlp_cols      the 256->512->256->{2Tp,2Tf} relu MLPs.  Layer 1 is split algebraically:
//                     W1 * [pf | z | state] = (W1[:, pf] * pf (+ W1[:, state] * state0) + b1)   <- per AGENT (linear_cols)
//                                           +  W1[:, z] * z (+ W1[:, state] * state1)           <- per TRAJECTORY
//                 so the per-trajectory K of layer 1 drops from 256 to 32 (block 0) / 128 (block 1).
//                 Layer 2's 512 KiB weight streams through LDS in fragment-ordered chunks shared by the
//                 workgroup's waves; the 512-wide hidden activation never exists outside registers.
//   linear_cols   generic  Y^T = W * [X1 | X2]^T + b  (per-agent pre-activations).
//
// All arithmetic fp32 (MFMA fp32-in is exact fmaf chaining).  Launch geometry is chosen by the host
// wrappers at the bottom (persistent grids sized to the 256 CUs, interleaved tile assignment for tail balance).
#include "chain.hpp"
#include "latency_bodies.hpp"
#include "api_util.hpp"
#include <cstdlib>

// ---------------------------------------------------------------------------------------------------
// conv + GRU over columns
// ---------------------------------------------------------------------------------------------------
#define GRU_LDS_BYTES ((18 * 2 * 64 + 18 * 6 * 64) * 16 + 4 * 96 * 4)

#ifndef GRU_THREADS
#define GRU_THREADS 1024
#endif
// LDS image: per gate-tile j (6 of them) the 24 fragments it needs are contiguous:
//     sW[j][g][T]  g = 0,1,2 (r,z,n rows)  T = 0,1 (W_ih k-tiles) then 2..7 (W_hh k-tiles)        -> 6 x 24 KiB
// so one base address per j plus immediate offsets covers every ds_read (no per-read address arithmetic).
// Gate rows arrive PRE-SCALED from packing.pack_block (r,z rows by -log2 e, n rows by 2 log2 e; biases alike).
#define GRU_W_F4 (6 * 3 * 8 * 64)
template <int TPX>
__device__ __forceinline__ void gru_step(const f32x4* __restrict__ sW, const float* __restrict__ sB, const f32x4 (&e)[2],
                                         const f32x4 (&h)[6], f32x4 (&hn)[6], int lane, int q) {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        STT_FENCE();
        // one opaque LDS base per gate tile: its 24 fragment reads then use immediate offsets (< 24 KiB) instead of one
        // v_add_u32 per read for everything beyond the 64 KiB immediate range (VALU work does not hide behind fp32 MFMA)
        typedef const __attribute__((address_space(3))) f32x4* lds_f4p;
        lds_f4p wj = (lds_f4p)sW + ((j * 24) * 64 + lane);
        asm volatile("" : "+v"(wj));
        f32x4 ar = ld4(sB + 0 * 96 + 16 * j + 4 * q);
        f32x4 az = ld4(sB + 1 * 96 + 16 * j + 4 * q);
        f32x4 ai = ld4(sB + 2 * 96 + 16 * j + 4 * q);
        f32x4 ah = ld4(sB + 3 * 96 + 16 * j + 4 * q);
#pragma unroll
        for (int T = 0; T < 2; ++T) {
            ar = mfma_k16(ar, wj[(0 * 8 + T) * 64], e[T]);
            az = mfma_k16(az, wj[(1 * 8 + T) * 64], e[T]);
            ai = mfma_k16(ai, wj[(2 * 8 + T) * 64], e[T]);
        }
#pragma unroll
        for (int T = 0; T < 6; ++T) {
            if (T == 3) STT_FENCE();
            ar = mfma_k16(ar, wj[(0 * 8 + 2 + T) * 64], h[T]);
            az = mfma_k16(az, wj[(1 * 8 + 2 + T) * 64], h[T]);
            ah = mfma_k16(ah, wj[(2 * 8 + 2 + T) * 64], h[T]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#ifdef STT_DIAG_NOGATES
            hn[j][r] = (ar[r] + az[r]) * 1e-3f + (ai[r] + ah[r]) * 1e-3f + 0.5f * h[j][r];
#else
            const float rg = sigmoid_prescaled(ar[r]);
            const float zg = sigmoid_prescaled(az[r]);
            const float ng = tanh_prescaled(fmaf(rg, ah[r], ai[r]));
            hn[j][r] = fmaf(zg, h[j][r] - ng, ng);  // (1-z) n + z h
#endif
        }
    }
}

template <int TPX>
__global__ __launch_bounds__(GRU_THREADS) void gru_cols_kernel(
    const float* __restrict__ xin,    // [ncols][16*TPX]  flattened (t,c) input sequence, zero padded
    const f32x4* __restrict__ convP,  // PK16 Toeplitz conv  [2*Tp row tiles][TPX][64]
    const float* __restrict__ convB,  // [32]
    const f32x4* __restrict__ wihP,   // PK16 [18][2][64]   (gate rows pre-scaled)
    const f32x4* __restrict__ whhP,   // PK16 [18][6][64]   (gate rows pre-scaled)
    const float* __restrict__ gbias,  // [4][96] : b_ir+b_hr, b_iz+b_hz, b_in, b_hn   (pre-scaled like their rows)
    float* __restrict__ state,        // [ncols][96]
    int ncols, int Tp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* sW = reinterpret_cast<f32x4*>(smem);
    float* sB = reinterpret_cast<float*>(sW + GRU_W_F4);
    // fill by LDS-DMA: every wave-instruction moves one 1 KiB fragment L2 -> LDS with no VGPR round trip, so all of a wave's
    // copies are in flight at once (the register-staged loop left the matrix pipe idle for ~15 us at every launch)
    for (int i = threadIdx.x; i < GRU_W_F4; i += blockDim.x) {
        const int l = i & 63, T = (i >> 6) & 7, g = (i >> 9) % 3, j = i / (64 * 8 * 3);
        const f32x4* src = T < 2 ? wihP + ((g * 6 + j) * 2 + T) * 64 + l : whhP + ((g * 6 + j) * 6 + (T - 2)) * 64 + l;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(sW + (i - l)), 16, 0, 0);
    }
    for (int i = threadIdx.x; i < 4 * 96; i += blockDim.x) sB[i] = gbias[i];
    __syncthreads();

    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int ntiles = (ncols + 15) >> 4;
    const f32x4 cb0 = ld4(convB + 4 * q), cb1 = ld4(convB + 16 + 4 * q);
    // conv fragments of step t are fetched one step ahead (step 0's are tile independent, so the prefetch wraps)
    // (TPX == 2, Tp > 8: the four prefetched fragments + the second input tile do not fit the 128-VGPR budget of a 16-wave workgroup --
    // that instantiation spilled 44 B per lane -- so it reads each conv fragment where it is used: one L2 round trip per step on a
    // shape whose model path takes the streaming / chain forms anyway)
    constexpr int CWN = TPX == 1 ? 1 : 0;
    f32x4 cw[2][TPX];
    if (CWN) {
#pragma unroll
        for (int io = 0; io < 2; ++io)
#pragma unroll
            for (int T = 0; T < TPX; ++T) cw[io][T] = convP[(io * TPX + T) * 64 + lane];
    }
    auto conv = [&](int t, f32x4 (&e)[2], const f32x4 (&d)[TPX]) {
        e[0] = cb0;
        e[1] = cb1;
#pragma unroll
        for (int io = 0; io < 2; ++io) {
#pragma unroll
            for (int T = 0; T < TPX; ++T) e[io] = mfma_k16(e[io], CWN ? cw[io][T] : convP[((2 * t + io) * TPX + T) * 64 + lane], d[T]);
            e[io] = relu4(e[io]);
        }
        if (CWN) {
            const int tn = (t + 1 < Tp) ? t + 1 : 0;
#pragma unroll
            for (int io = 0; io < 2; ++io)
#pragma unroll
                for (int T = 0; T < TPX; ++T) cw[io][T] = convP[((2 * tn + io) * TPX + T) * 64 + lane];
        }
    };
    // interleaved assignment: consecutive tiles go to different CUs first, then to different waves
    for (int tile = blockIdx.x + gridDim.x * wave; tile < ntiles; tile += gridDim.x * nw) {
        const int col = tile * 16 + c;
        const int colc = col < ncols ? col : ncols - 1;
        f32x4 d[TPX];
#pragma unroll
        for (int T = 0; T < TPX; ++T) d[T] = ld4(xin + (size_t)colc * (16 * TPX) + 16 * T + 4 * q);
        f32x4 h[6], h2[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) h[j] = splat4(0.f);
#pragma unroll 1
        for (int t = 0; t < Tp; ++t) {
            f32x4 e[2];
            conv(t, e, d);
            gru_step<TPX>(sW, sB, e, h, h2, lane, q);
#pragma unroll
            for (int j = 0; j < 6; ++j) h[j] = h2[j];
        }
        if (col < ncols) {
#pragma unroll
            for (int j = 0; j < 6; ++j) st4(state + (size_t)col * 96 + 16 * j + 4 * q, h[j]);
        }
    }
}

// Latency form of the same conv + GRU for FEW columns: body in latency_bodies.hpp (gru_lat_body)
template <int TPX>
__global__ __launch_bounds__(384) void gru_cols_lat_kernel(const float* __restrict__ xin, const f32x4* __restrict__ convP,
                                                           const float* __restrict__ convB, const f32x4* __restrict__ wihP,
                                                           const f32x4* __restrict__ whhP, const float* __restrict__ gbias,
                                                           float* __restrict__ state, int ncols, int Tp) {
    __shared__ f32x4 sH[2][6][64];
    gru_lat_body<TPX>(xin, convP, convB, wihP, whhP, gbias, state, ncols, Tp, blockIdx.x, sH);
}

// Balanced latency form (round 3): four waves, every wave 144 of a step's 576 gate-sum MFMAs and a quarter of the gate functions
// (latency_bodies.hpp: gru_bal_body): 22 instead of 29 us for 8 steps.
template <int TPX>
__global__ __launch_bounds__(256) void gru_cols_bal_kernel(const float* __restrict__ xin, const f32x4* __restrict__ convP,
                                                           const float* __restrict__ convB, const f32x4* __restrict__ wihP,
                                                           const f32x4* __restrict__ whhP, const float* __restrict__ gbias,
                                                           float* __restrict__ state, int ncols, int Tp) {
    __shared__ f32x4 sH[2][6][64], sX[2 * 5 * 64];
    gru_bal_body<TPX>(xin, convP, convB, wihP, whhP, gbias, state, ncols, Tp, blockIdx.x, sH, sX);
}

// ---------------------------------------------------------------------------------------------------
// generic per-column linear:  out[col][0:N] = act( W * [X1 | X2] + b ), up to 3 independent jobs per launch
// (blockIdx.z).  Each wave: 16 columns x RT row tiles; all B tiles are loaded up front and the A fragments of
// k-tile T+1 are prefetched while k-tile T feeds the MFMAs (weights come straight from L2: few columns per launch).
// ---------------------------------------------------------------------------------------------------
struct LinJob {
    const float* X1; const float* X2; const f32x4* WP; const float* bias; float* out;
    int ld1, KT1, ld2, KT2, ldo, NT, act;  // act: 0 none, 1 relu, 2 tanh
};
struct LinJobs { LinJob j[3]; };

template <int RT, int CT>
__global__ __launch_bounds__(256) void linear_cols_kernel(LinJobs jobs, int ncols) {
    // wave: CT column tiles x RT row tiles; every A fragment feeds CT MFMA quads (weight traffic from L2 / CT), every
    // B tile RT quads; the operands of k-tile T+1 are in flight while k-tile T feeds the MFMAs.
    const LinJob J = jobs.j[blockIdx.z];
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int wave = threadIdx.x >> 6;
    const int col0 = blockIdx.x * (16 * CT);
    const int rt0 = (blockIdx.y * 4 + wave) * RT;
    if (col0 >= ncols || rt0 >= J.NT) return;
    const int KT = J.KT1 + J.KT2;
    int colc[CT];
#pragma unroll
    for (int j = 0; j < CT; ++j) { const int col = col0 + 16 * j + c; colc[j] = col < ncols ? col : ncols - 1; }
    auto ldB = [&](int j, int T) {
        return T < J.KT1 ? ld4(J.X1 + (size_t)colc[j] * J.ld1 + 16 * T + 4 * q) : ld4(J.X2 + (size_t)colc[j] * J.ld2 + 16 * (T - J.KT1) + 4 * q);
    };
    int rts[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) rts[i] = rt0 + i < J.NT ? rt0 + i : J.NT - 1;
    f32x4 acc[RT][CT], wn[RT], bn[CT];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const f32x4 bv = J.bias ? ld4(J.bias + 16 * rts[i] + 4 * q) : splat4(0.f);
#pragma unroll
        for (int j = 0; j < CT; ++j) acc[i][j] = bv;
        wn[i] = J.WP[((size_t)rts[i] * KT) * 64 + lane];
    }
#pragma unroll
    for (int j = 0; j < CT; ++j) bn[j] = ldB(j, 0);
#pragma unroll 1
    for (int T = 0; T < KT; ++T) {
        f32x4 wc[RT], bc[CT];
#pragma unroll
        for (int i = 0; i < RT; ++i) wc[i] = wn[i];
#pragma unroll
        for (int j = 0; j < CT; ++j) bc[j] = bn[j];
        const int Tn = T + 1 < KT ? T + 1 : T;
#pragma unroll
        for (int i = 0; i < RT; ++i) wn[i] = J.WP[((size_t)rts[i] * KT + Tn) * 64 + lane];
#pragma unroll
        for (int j = 0; j < CT; ++j) bn[j] = ldB(j, Tn);
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < CT; ++j) acc[i][j] = mfma_k16(acc[i][j], wc[i], bc[j]);
    }
#pragma unroll
    for (int j = 0; j < CT; ++j) {
        const int col = col0 + 16 * j + c;
        if (col < ncols) {
#pragma unroll
            for (int i = 0; i < RT; ++i)
                if (rt0 + i < J.NT) {
                    f32x4 v = acc[i][j];
                    if (J.act == 1) v = relu4(v);
                    else if (J.act == 2) { v[0] = tanhf(v[0]); v[1] = tanhf(v[1]); v[2] = tanhf(v[2]); v[3] = tanhf(v[3]); }
                    st4(J.out + (size_t)col * J.ldo + 16 * (rt0 + i) + 4 * q, v);
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// MLP over columns:  out = W3 * relu( W2 * relu( A0[agent] + W1v * B ) + b2 ) + b3
//
// Weight stream: per workgroup pass the weights of all MLPs of the block form ONE flat sequence of equally
// sized chunks (fragment order, packed by sttode_amd/packing.py):
//     per MLP:  NCH "L12" chunks { W1v tiles [CHT][KTV] , W2 tiles [CHT][16] }   then  N3 "L3" chunks { W3 tiles [TP3][16] }
// double-buffered in LDS; chunk p+1 is fetched (global -> registers) while chunk p feeds the MFMAs and is
// committed (registers -> LDS, one barrier) afterwards.  The position counter never resets, so the stream wraps
// from the last chunk of one column group to the first chunk of the next without a bubble.
// ---------------------------------------------------------------------------------------------------
// LDS-DMA weight stream: chunk p+1 is copied L2 -> LDS by global_load_lds_dwordx4 (no VGPR round trip, no ds_write) while
// chunk p feeds the MFMAs from the other LDS buffer.  The fragment-ordered chunk is lane-linear, which is exactly the shape
// the DMA writes (wave-uniform LDS base + lane*16 B).  The per-agent layer-1 pre-activations (a 16-byte gather per lane and
// chunk) travel the same way into a 1 KiB per-wave slot: with a DMA in flight hipcc drains vmcnt(0) before the first use of
// ANY register-destination global load, so the loop must not contain one.  A step is
//     a0 = slot[lane] (ds_read, the value DMA'd during the previous step)
//     begin(): DMA chunk p+1 -> buffer (p+1)&1 (its last readers passed the barrier of step p-1); DMA next a0 -> slot
//     ... MFMAs on cur() ...
//     end():   __syncthreads() -- the barrier's fence waits vmcnt(0) first, which is exactly the wait the DMAs need.
// The layer-2/3 biases ride in the spare space of the layer-3 chunks (packing.mlp_stream), so they also arrive by DMA.
template <int CHW>
struct WStream {
    const f32x4* blob;
    f32x4* lds;
    f32x4* slot;  // this wave's 64 x 16 B a0 slot
    int total, pos;
    static __device__ __forceinline__ void glds16(const void* g, void* l) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
    }
    __device__ __forceinline__ void dma(int chunk, int buf) {
        // CHW/64 wave-instructions of 1 KiB each, dealt round-robin to the 4 waves (CHW need not be a multiple of 256)
        constexpr int N16 = CHW / 64;
        // readfirstlane makes the wave id provably uniform: scalar branch below, M0 (the LDS base) computed on the SALU
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
        const f32x4* src = blob + (size_t)(chunk % total) * CHW + lane;
        f32x4* dst = lds + buf * CHW;  // wave-uniform base; the hardware adds lane * 16 B
#pragma unroll
        for (int i = 0; i < (N16 + 3) / 4; ++i) {
            const int idx = i * 4 + wave;
            if (idx < N16) glds16(src + idx * 64, dst + idx * 64);
        }
    }
    __device__ __forceinline__ void dma_a0(const float* lane_ptr) { glds16(lane_ptr, slot); }
    __device__ __forceinline__ void init(const f32x4* b, f32x4* l, f32x4* a0slots, int tot, const float* a0_first) {
        blob = b; lds = l; total = tot; pos = 0;
        slot = a0slots + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * 64;
        dma(0, 0);
        dma_a0(a0_first);
    }
    __device__ __forceinline__ const f32x4* cur() const { return lds + (pos & 1) * CHW; }
    __device__ __forceinline__ void begin(const float* a0_next) {
        __builtin_amdgcn_sched_barrier(0);
        dma(pos + 1, (pos + 1) & 1);
        dma_a0(a0_next);
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ void begin_nogather() {
        __builtin_amdgcn_sched_barrier(0);
        dma(pos + 1, (pos + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ void end() {
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        ++pos;
        __builtin_amdgcn_sched_barrier(0);
    }
};

// One MLP for this wave's 16 columns (CHT = 1: one 16-row tile of the hidden layer per chunk).  a0 points at this lane's
// A0 row (+4q); a0_next at the row the NEXT phase starts with.
template <int KTV, int NO, int CHW>
__device__ __forceinline__ void mlp_phase(WStream<CHW>& st, const f32x4 (&B)[KTV], const float* __restrict__ a0,
                                          const float* __restrict__ a0_next, f32x4 (&out)[NO], int lane, int q) {
    constexpr int NCH = 32;
    constexpr int TP3 = CHW / (16 * 64);
    constexpr int N3 = (NO + TP3 - 1) / TP3;
    static_assert(TP3 >= 1, "chunk too small for a layer-3 tile");
    f32x4 acc2[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) acc2[it] = splat4(0.f);
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
        f32x4 h1 = st.slot[lane];  // A0[agent][16 ch + 4q ..], DMA'd during the previous step
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slot is re-filled by the DMA issued next
        st.begin((ch + 1 < NCH) ? a0 + (ch + 1) * 16 : a0_next);
        const f32x4* buf = st.cur();
        STT_FENCE();
#pragma unroll
        for (int T = 0; T < KTV; ++T) h1 = mfma_k16(h1, buf[T * 64 + lane], B[T]);
        h1 = relu4(h1);
        const f32x4* w2 = buf + KTV * 64 + lane;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            if ((it & 7) == 0) STT_FENCE();
            acc2[it] = mfma_k16(acc2[it], w2[it * 64], h1);
        }
        st.end();
    }
    {
        // the first layer-3 chunk is already resident (landed at the last barrier): b2 sits behind its tiles and b3
        const float* b2 = reinterpret_cast<const float*>(st.cur() + TP3 * 16 * 64) + 16 * TP3;
#pragma unroll
        for (int it = 0; it < 16; ++it) acc2[it] = relu4(acc2[it] + ld4(b2 + 16 * it + 4 * q));
    }
#pragma unroll
    for (int c3 = 0; c3 < N3; ++c3) {
        st.begin_nogather();
        const f32x4* buf = st.cur();
        const float* b3 = reinterpret_cast<const float*>(buf + TP3 * 16 * 64);
#pragma unroll
        for (int oo = 0; oo < TP3; ++oo) {
            const int o = c3 * TP3 + oo;
            if (o < NO) {
                f32x4 a = ld4(b3 + 16 * oo + 4 * q);
#pragma unroll
                for (int T = 0; T < 16; ++T) {
                    if ((T & 7) == 0) STT_FENCE();
                    a = mfma_k16(a, buf[(oo * 16 + T) * 64 + lane], acc2[T]);
                }
                out[o] = a;
            }
        }
        st.end();
    }
}

#define MLP0_CHW 1152                 // CHT = 1, KTV = 2 : 18 fragment tiles = 18 KiB per chunk
#define MLP1_CHW (1 * (8 + 16) * 64)  // CHT = 1, KTV = 8 : 1536 float4 = 24 KiB per chunk

// block 0: decoder_x and decoder_y MLPs per trajectory as two WORKGROUP ROLES (even blockIdx: x, odd: y): a work item is
// (64-column group, one MLP), half the size of "both MLPs", which halves the grid tail; each role streams only its own
// MLP's chunks.   x role: d = x_true - x_hat0 -> dbuf ;  y role: y_hat0 -> ybuf
template <int NO, bool IS_X>
__device__ __forceinline__ void mlp0_role(const float* __restrict__ A0, const f32x4* __restrict__ blob, int nchunks,
                                          const float* __restrict__ z,
                                          const float* __restrict__ xpad, float* __restrict__ obuf, int ncols, int K, f32x4* lds) {
    f32x4* a0slots = lds + 2 * MLP0_CHW;
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4, wave = threadIdx.x >> 6;
    const int ngroups = (ncols + 63) >> 6;  // 4 waves x 16 columns per workgroup step
    const int gstride = (int)gridDim.x >> 1;
    int g = (int)blockIdx.x >> 1;
    auto agent_of = [&](int gg) {
        int col = gg * 64 + wave * 16 + c;
        col = col < ncols ? col : ncols - 1;
        return col / K;
    };
    WStream<MLP0_CHW> st;
    st.init(blob, lds, a0slots, nchunks, A0 + (size_t)agent_of(g < ngroups ? g : 0) * 512 + 4 * q);
    __syncthreads();
    for (; g < ngroups; g += gstride) {
        const int col = g * 64 + wave * 16 + c;
        const int colc = col < ncols ? col : ncols - 1;
        const int agent = colc / K;
        const int gn = g + gstride;
        const int agent_nx = agent_of(gn < ngroups ? gn : g);
        f32x4 B[2];
        B[0] = ld4(z + (size_t)colc * 32 + 4 * q);
        B[1] = ld4(z + (size_t)colc * 32 + 16 + 4 * q);
        f32x4 o[NO];
        mlp_phase<2, NO, MLP0_CHW>(st, B, A0 + (size_t)agent * 512 + 4 * q, A0 + (size_t)agent_nx * 512 + 4 * q, o, lane, q);
        if (col < ncols) {
#pragma unroll
            for (int t = 0; t < NO; ++t) {
                if (IS_X) {
                    const f32x4 xt = ld4(xpad + (size_t)agent * (16 * NO) + 16 * t + 4 * q);
                    st4(obuf + (size_t)col * (16 * NO) + 16 * t + 4 * q, xt - o[t]);
                } else {
                    st4(obuf + (size_t)col * (16 * NO) + 16 * t + 4 * q, o[t]);
                }
            }
        }
    }
}

#define MLP0_WGS 3
template <int TPX, int NOY>
__global__ __launch_bounds__(256, MLP0_WGS) void mlp_block0_kernel(
    const float* __restrict__ A0x, const float* __restrict__ A0y,  // [nagents][512]
    const f32x4* __restrict__ blob, int total_chunks,              // weight stream (x chunks, then y chunks; biases inside)
    const float* __restrict__ z,                                   // [ncols][32]
    const float* __restrict__ xpad,                                // [nagents][16*TPX] normalised past (t,c), zero padded
    float* __restrict__ dbuf,                                      // [ncols][16*TPX]
    float* __restrict__ ybuf,                                      // [ncols][16*NOY]
    int ncols, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* lds = reinterpret_cast<f32x4*>(smem);
    constexpr int NCX = 32 + TPX;  // layer-1/2 chunks + one layer-3 chunk per output tile (TP3 = 1 at this chunk size)
    if ((blockIdx.x & 1) == 0)
        mlp0_role<TPX, true>(A0x, blob, NCX, z, xpad, dbuf, ncols, K, lds);
    else
        mlp0_role<NOY, false>(A0y, blob + (size_t)NCX * MLP0_CHW, total_chunks - NCX, z, xpad, ybuf, ncols, K, lds);
}

// block 1: y MLP per trajectory with the per-trajectory GRU state; final epilogue
//   pred[col][t][c] = ((y_hat0 + y_hat1) + cur[agent][c]) + orig[agent][c]      (model/STTODE.py:338,344,622)
template <int NOY>
__global__ __launch_bounds__(256, 3) void mlp_block1_kernel(
    const float* __restrict__ A1y, const f32x4* __restrict__ blob, int total_chunks,  // weight stream (biases inside)
    const float* __restrict__ z,              // [ncols][32]
    const float* __restrict__ state1,         // [ncols][96]
    const float* __restrict__ ybuf,           // [ncols][16*NOY]  y_hat0
    const float* __restrict__ cur,            // [nagents][2]
    const float* __restrict__ orig,           // [nagents][2]
    float* __restrict__ pred,                 // [ncols][2*Tf]
    int ncols, int K, int Tf2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* lds = reinterpret_cast<f32x4*>(smem);
    f32x4* a0slots = lds + 2 * MLP1_CHW;
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4, wave = threadIdx.x >> 6;
    const int ngroups = (ncols + 63) >> 6;
    int g = blockIdx.x;
    auto agent_of = [&](int gg) {
        int col = gg * 64 + wave * 16 + c;
        col = col < ncols ? col : ncols - 1;
        return col / K;
    };
    WStream<MLP1_CHW> st;
    st.init(blob, lds, a0slots, total_chunks, A1y + (size_t)agent_of(g < ngroups ? g : 0) * 512 + 4 * q);
    __syncthreads();
    for (; g < ngroups; g += gridDim.x) {
        const int col = g * 64 + wave * 16 + c;
        const int colc = col < ncols ? col : ncols - 1;
        const int agent = colc / K;
        const int gn = g + (int)gridDim.x;
        const int agent_nx = agent_of(gn < ngroups ? gn : g);
        f32x4 B[8];
        B[0] = ld4(z + (size_t)colc * 32 + 4 * q);
        B[1] = ld4(z + (size_t)colc * 32 + 16 + 4 * q);
#pragma unroll
        for (int T = 0; T < 6; ++T) B[2 + T] = ld4(state1 + (size_t)colc * 96 + 16 * T + 4 * q);
        f32x4 yo[NOY];
        mlp_phase<8, NOY, MLP1_CHW>(st, B, A1y + (size_t)agent * 512 + 4 * q, A1y + (size_t)agent_nx * 512 + 4 * q, yo, lane, q);
        if (col < ncols) {
            const float cx = cur[2 * agent], cy = cur[2 * agent + 1];
            const float ox = orig[2 * agent], oy = orig[2 * agent + 1];
#pragma unroll
            for (int o = 0; o < NOY; ++o) {
                const int row0 = 16 * o + 4 * q;
                if (row0 < Tf2) {
                    const f32x4 y0 = ld4(ybuf + (size_t)col * (16 * NOY) + row0);
                    f32x4 v;
                    v[0] = ((y0[0] + yo[o][0]) + cx) + ox;
                    v[1] = ((y0[1] + yo[o][1]) + cy) + oy;
                    v[2] = ((y0[2] + yo[o][2]) + cx) + ox;
                    v[3] = ((y0[3] + yo[o][3]) + cy) + oy;
                    float* p = pred + (size_t)col * Tf2 + row0;
                    if (row0 + 3 < Tf2 && (Tf2 & 3) == 0) {
                        st4(p, v);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (row0 + r < Tf2) p[r] = v[r];
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Latency form of the decoder MLPs for FEW columns (a single scene).  The throughput kernels give every wave a 16-column tile and
// run its 72-96 MFMAs per chunk serially (~35 chunks x ~1.2 us however few tiles exist: an fp32 16x16x4 MFMA holds a SIMD's matrix
// pipe for 32 cycles).  Here the FOUR waves of a workgroup share ONE 16-column tile and work through the hidden layer in groups of
// four 16-row chunks:
//   phase A   wave w computes layer 1 of chunk 4g+w (KTV k-tiles, in the throughput kernels' order) and publishes relu(h1) as a
//             B-operand fragment in LDS (double buffered: one barrier per group);
//   phase B   every wave adds the four chunks, in chunk order, into ITS 4 of the 16 layer-2 row tiles (4 independent MFMA chains).
// 2 + 16 (block 0) / 8 + 16 (block 1) k16 quads (of four MFMAs) per group and wave instead of 4 x (KTV + 16) = 72 / 96.  Every weight fragment is
// used by exactly one wave, so the weights skip LDS: plain coalesced 1 KiB loads (same fragment-ordered stream as the throughput
// kernels), fetched one group ahead into registers.  Layer 3: the 256-wide activation is exchanged through LDS and wave o mod 4 runs
// the 16-quad chain of output tile o.  Same summation order as the throughput kernels everywhere: identical bits.
// MODE 0: block-0 decoder_x (dbuf = x_true - x_hat0) | 1: block-0 decoder_y (ybuf) | 2: block-1 decoder_y + epilogue (pred)
// ---------------------------------------------------------------------------------------------------
// MlpLatArgs / mlp_lat_run: latency_bodies.hpp (shared with the one-launch scene path, scene_lat.hip)
// block 0: blockIdx.y = role (0: decoder_x, 1: decoder_y); block 1: one role
template <int TPX, int NOY>
__global__ __launch_bounds__(256) void mlp0_lat_kernel(MlpLatArgs ax, MlpLatArgs ay) {
    __shared__ f32x4 sH1[2 * 4 * 64], sH2[16 * 64];
    if (blockIdx.y == 0) mlp_lat_run<2, TPX, 0>(ax, sH1, sH2, blockIdx.x);
    else mlp_lat_run<2, NOY, 1>(ay, sH1, sH2, blockIdx.x);
}
template <int NOY>
__global__ __launch_bounds__(256) void mlp1_lat_kernel(MlpLatArgs a) {
    __shared__ f32x4 sH1[2 * 4 * 64], sH2[16 * 64];
    mlp_lat_run<8, NOY, 2>(a, sH1, sH2, blockIdx.x);
}
// STTODE_MLP_LAT_TILES: largest 16-column tile count served by the latency form of the MLP kernels
static int g_mlp_lat_tiles = -1, g_gru_lat_tiles = -1;
static int mlp_lat_tiles() {
    if (g_mlp_lat_tiles < 0) { const char* e = getenv("STTODE_MLP_LAT_TILES"); g_mlp_lat_tiles = e ? atoi(e) : 1024; }
    return g_mlp_lat_tiles;
}

// generic single MLP over columns (training-forward needs decoder_x of the LAST block too: recover_traj sums the x_hat of
// every block, model/STTODE.py:339-341).  KTV = 8: B = [z | state];  output raw tiles [ncols][16*NO].
template <int NO>
__global__ __launch_bounds__(256, 2) void mlp_cols_kernel(const float* __restrict__ A0, const f32x4* __restrict__ blob, int nchunks,
                                                          const float* __restrict__ z,
                                                          const float* __restrict__ state, float* __restrict__ out, int ncols, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* lds = reinterpret_cast<f32x4*>(smem);
    f32x4* a0slots = lds + 2 * MLP1_CHW;
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4, wave = threadIdx.x >> 6;
    const int ngroups = (ncols + 63) >> 6;
    int g = blockIdx.x;
    auto agent_of = [&](int gg) {
        int col = gg * 64 + wave * 16 + c;
        col = col < ncols ? col : ncols - 1;
        return col / K;
    };
    WStream<MLP1_CHW> st;
    st.init(blob, lds, a0slots, nchunks, A0 + (size_t)agent_of(g < ngroups ? g : 0) * 512 + 4 * q);
    __syncthreads();
    for (; g < ngroups; g += gridDim.x) {
        const int col = g * 64 + wave * 16 + c;
        const int colc = col < ncols ? col : ncols - 1;
        const int agent = colc / K;
        const int gn = g + (int)gridDim.x;
        const int agent_nx = agent_of(gn < ngroups ? gn : g);
        f32x4 B[8];
        B[0] = ld4(z + (size_t)colc * 32 + 4 * q);
        B[1] = ld4(z + (size_t)colc * 32 + 16 + 4 * q);
#pragma unroll
        for (int T = 0; T < 6; ++T) B[2 + T] = ld4(state + (size_t)colc * 96 + 16 * T + 4 * q);
        f32x4 o[NO];
        mlp_phase<8, NO, MLP1_CHW>(st, B, A0 + (size_t)agent * 512 + 4 * q, A0 + (size_t)agent_nx * 512 + 4 * q, o, lane, q);
        if (col < ncols) {
#pragma unroll
            for (int t = 0; t < NO; ++t) st4(out + (size_t)col * (16 * NO) + 16 * t + 4 * q, o[t]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------
// STTODE_NONPERSISTENT=1 (experiment): one work item per workgroup instead of persistent grids
static int nonpersistent() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("STTODE_NONPERSISTENT"); v = (e && e[0] == '1') ? 1 : 0; }
    return v;
}
static int g_num_cu = 0;
static int num_cus() {
    if (!g_num_cu) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) g_num_cu = p.multiProcessorCount;
        if (g_num_cu <= 0) g_num_cu = 256;
    }
    return g_num_cu;
}

// STTODE_GRU_LAT_TILES: largest tile count served by the latency form (default: measured crossover)
static int gru_lat_tiles() {
    if (g_gru_lat_tiles < 0) { const char* e = getenv("STTODE_GRU_LAT_TILES"); g_gru_lat_tiles = e ? atoi(e) : 512; }
    return g_gru_lat_tiles;
}
int stt_gru_lat_tiles() { return gru_lat_tiles(); }
static int g_enc_lat_tiles = -1;
int stt_enc_lat_tiles() {
    if (g_enc_lat_tiles < 0) { const char* e = getenv("STTODE_ENC_LAT_TILES"); g_enc_lat_tiles = e ? atoi(e) : 1024; }
    return g_enc_lat_tiles;
}
extern "C" int sttode_set_latency_tiles(int gru_tiles, int mlp_tiles, int enc_tiles) {
    if (gru_tiles >= 0) g_gru_lat_tiles = gru_tiles;
    if (mlp_tiles >= 0) g_mlp_lat_tiles = mlp_tiles;
    if (enc_tiles >= 0) g_enc_lat_tiles = enc_tiles;
    return 0;
}

extern "C" int sttode_gru_cols(const float* xin, const float* convP, const float* convB, const float* wihP, const float* whhP,
                               const float* gbias, float* state, int ncols, int Tp, int TPX, void* stream) {
    return stt_gru_cols_form(xin, convP, convB, wihP, whhP, gbias, state, ncols, Tp, TPX, 0, stream);
}
// lat_max_tiles > 0: the caller's own crossover (the pipelined per-agent stage: the latency form has no LDS-resident weights, so its
// workgroups run BESIDE a chain workgroup of another stream instead of waiting for a chain-free CU); 0: sttode_set_latency_tiles
int stt_gru_cols_form(const float* xin, const float* convP, const float* convB, const float* wihP, const float* whhP, const float* gbias,
                      float* state, int ncols, int Tp, int TPX, int lat_max_tiles, void* stream) {
    STT_REQUIRE(xin && convP && convB && wihP && whhP && gbias && state, "sttode_gru_cols: null pointer");
    STT_REQUIRE(ncols > 0 && Tp > 0 && (TPX == 1 || TPX == 2) && 2 * Tp <= 16 * TPX, "sttode_gru_cols: bad ncols/Tp/TPX");
    hipStream_t s = (hipStream_t)stream;
    const int ntiles = (ncols + 15) / 16;
    if (ntiles <= (lat_max_tiles > 0 ? lat_max_tiles : gru_lat_tiles())) {   // few columns: one tile per workgroup, gate sums split evenly over four waves (latency form)
        if (lat_max_tiles > 0) {   // beside a chain workgroup of another stream: the six-wave form (<= 128 VGPRs per wave); the balanced one needs 270-290
            if (TPX == 1)
                hipLaunchKernelGGL(gru_cols_lat_kernel<1>, dim3(ntiles), dim3(384), 0, s, xin, (const f32x4*)convP, convB, (const f32x4*)wihP,
                                   (const f32x4*)whhP, gbias, state, ncols, Tp);
            else
                hipLaunchKernelGGL(gru_cols_lat_kernel<2>, dim3(ntiles), dim3(384), 0, s, xin, (const f32x4*)convP, convB, (const f32x4*)wihP,
                                   (const f32x4*)whhP, gbias, state, ncols, Tp);
        } else if (TPX == 1)
            hipLaunchKernelGGL(gru_cols_bal_kernel<1>, dim3(ntiles), dim3(256), 0, s, xin, (const f32x4*)convP, convB, (const f32x4*)wihP,
                               (const f32x4*)whhP, gbias, state, ncols, Tp);
        else
            hipLaunchKernelGGL(gru_cols_bal_kernel<2>, dim3(ntiles), dim3(256), 0, s, xin, (const f32x4*)convP, convB, (const f32x4*)wihP,
                               (const f32x4*)whhP, gbias, state, ncols, Tp);
        STT_HIP(hipGetLastError());
        return 0;
    }
    // Workgroup shape follows the amount of work: a full launch uses 16 waves (4 per SIMD) on every CU; a small one
    // (per-agent block 0: a few hundred tiles) uses just enough waves per workgroup to cover the tiles, so it occupies
    // only as many CUs as it needs and the concurrently running encoder kernels get the rest of the chip.
    int nw = (ntiles + num_cus() - 1) / num_cus();
    if (nw < 4) nw = 4;
    if (nw > GRU_THREADS / 64) nw = GRU_THREADS / 64;
    const int threads = 64 * nw;
    int grid = (ntiles + nw - 1) / nw;
    if (grid > num_cus()) grid = num_cus();
    if (TPX == 1) {
        STT_SET_LDS_ONCE(gru_cols_kernel<1>, GRU_LDS_BYTES);
        hipLaunchKernelGGL(gru_cols_kernel<1>, dim3(grid), dim3(threads), GRU_LDS_BYTES, s, xin, (const f32x4*)convP, convB,
                           (const f32x4*)wihP, (const f32x4*)whhP, gbias, state, ncols, Tp);
    } else {
        STT_SET_LDS_ONCE(gru_cols_kernel<2>, GRU_LDS_BYTES);
        hipLaunchKernelGGL(gru_cols_kernel<2>, dim3(grid), dim3(threads), GRU_LDS_BYTES, s, xin, (const f32x4*)convP, convB,
                           (const f32x4*)wihP, (const f32x4*)whhP, gbias, state, ncols, Tp);
    }
    STT_HIP(hipGetLastError());
    return 0;
}

static int lin_check(const float* X1, int ld1, int K1, const float* X2, int ld2, int K2, const float* WP, float* out, int ldo, int N) {
    STT_REQUIRE(X1 && WP && out, "sttode_linear_cols: null pointer");
    STT_REQUIRE(N > 0 && N % 16 == 0 && K1 > 0 && K1 % 16 == 0 && K2 >= 0 && K2 % 16 == 0, "sttode_linear_cols: N, K1, K2 must be multiples of 16");
    STT_REQUIRE(K1 + K2 <= 4096, "sttode_linear_cols: K1 + K2 must be <= 4096");
    STT_REQUIRE(ld1 % 4 == 0 && ld2 % 4 == 0 && ldo % 4 == 0 && (K2 == 0 || X2), "sttode_linear_cols: leading dims must be multiples of 4");
    return 0;
}
static LinJob mkjob(const float* X1, int ld1, int K1, const float* X2, int ld2, int K2, const float* WP, const float* bias, float* out,
                    int ldo, int N, int act) {
    LinJob j;
    j.X1 = X1; j.X2 = X2; j.WP = (const f32x4*)WP; j.bias = bias; j.out = out;
    j.ld1 = ld1; j.KT1 = K1 / 16; j.ld2 = ld2; j.KT2 = K2 / 16; j.ldo = ldo; j.NT = N / 16; j.act = act;
    return j;
}
static int lin_launch(const LinJobs& jobs, int njobs, int ncols, int maxNT, int maxKT, hipStream_t s) {
    (void)maxKT;
    if (ncols > 64) {
        dim3 grid((ncols + 63) / 64, (maxNT + 15) / 16, njobs);  // wave = 64 columns x 4 row tiles, 4 waves = 16 row tiles
        hipLaunchKernelGGL((linear_cols_kernel<4, 4>), grid, dim3(256), 0, s, jobs, ncols);
    } else {
        dim3 grid((ncols + 15) / 16, (maxNT + 15) / 16, njobs);  // few columns: one column tile per wave
        hipLaunchKernelGGL((linear_cols_kernel<4, 1>), grid, dim3(256), 0, s, jobs, ncols);
    }
    STT_HIP(hipGetLastError());
    return 0;
}

extern "C" int sttode_linear_cols(const float* X1, int ld1, int K1, const float* X2, int ld2, int K2, const float* WP,
                                  const float* bias, float* out, int ldo, int ncols, int N, int act, void* stream) {
    if (int rc = lin_check(X1, ld1, K1, X2, ld2, K2, WP, out, ldo, N)) return rc;
    STT_REQUIRE(ncols > 0, "sttode_linear_cols: ncols must be positive");
    STT_REQUIRE(act >= 0 && act <= 2, "sttode_linear_cols: act must be 0 (none), 1 (relu) or 2 (tanh)");
    LinJobs jobs;
    jobs.j[0] = jobs.j[1] = jobs.j[2] = mkjob(X1, ld1, K1, X2, ld2, K2, WP, bias, out, ldo, N, act);
    return lin_launch(jobs, 1, ncols, N / 16, (K1 + K2) / 16, (hipStream_t)stream);
}

// The three per-agent layer-1 pre-activations of the decoder in ONE launch:
//   A0x = W1x[:, pf|state] [pf | state0] + b1x ; A0y likewise (block 0) ; A1y = W1y'[:, pf] pf + b1y' (block 1)
extern "C" int sttode_agent_preact(const float* pf, const float* state0, const float* WAx, const float* b1x, const float* WAy,
                                   const float* b1y, const float* WA1, const float* b11, float* A0x, float* A0y, float* A1y,
                                   int n, void* stream) {
    STT_REQUIRE(pf && state0 && WAx && b1x && WAy && b1y && WA1 && b11 && A0x && A0y && A1y, "sttode_agent_preact: null pointer");
    STT_REQUIRE(n > 0, "sttode_agent_preact: n must be positive");
    LinJobs jobs;
    jobs.j[0] = mkjob(pf, 128, 128, state0, 96, 96, WAx, b1x, A0x, 512, 512, 0);
    jobs.j[1] = mkjob(pf, 128, 128, state0, 96, 96, WAy, b1y, A0y, 512, 512, 0);
    jobs.j[2] = mkjob(pf, 128, 128, nullptr, 0, 0, WA1, b11, A1y, 512, 512, 0);
    return lin_launch(jobs, 3, n, 32, 14, (hipStream_t)stream);
}

#define MLP0_LDS(TX, NY) (2 * MLP0_CHW * 16 + 4096)
#define MLP1_LDS(NY) (2 * MLP1_CHW * 16 + 4096)

extern "C" int sttode_mlp_block0(const float* A0x, const float* A0y, const float* stream, int total_chunks,
                                 const float* z, const float* xpad, float* dbuf, float* ybuf, int ncols, int K, int TPX, int NOY,
                                 void* stream_) {
    STT_REQUIRE(A0x && A0y && stream && z && xpad && dbuf && ybuf, "sttode_mlp_block0: null pointer");
    STT_REQUIRE(ncols > 0 && K > 0, "sttode_mlp_block0: ncols and K must be positive");
    STT_REQUIRE(total_chunks == 64 + TPX + NOY, "sttode_mlp_block0: weight stream must hold (32+TPX) + (32+NOY) chunks");
    if ((ncols + 15) / 16 <= mlp_lat_tiles()) {   // few columns: latency form (four waves share one 16-column tile)
        MlpLatArgs ax, ay;
        ax.A0 = A0x; ax.blob = (const f32x4*)stream; ax.z = z; ax.state = nullptr; ax.xpad = xpad; ax.ybuf = nullptr; ax.cur = nullptr;
        ax.orig = nullptr; ax.out = dbuf; ax.ncols = ncols; ax.K = K; ax.Tf2 = 0; ax.out_lds = nullptr; ax.state_lds = nullptr; ax.a0_lds = nullptr;
        ay = ax;
        ay.A0 = A0y; ay.blob = (const f32x4*)stream + (size_t)(32 + TPX) * MLP0_CHW; ay.out = ybuf;
        const dim3 g((ncols + 15) / 16, 2);
        hipStream_t sl = (hipStream_t)stream_;
#define L0L(TX, NY)                                                                                      \
    do {                                                                                                 \
        hipLaunchKernelGGL((mlp0_lat_kernel<TX, NY>), g, dim3(256), 0, sl, ax, ay);                      \
    } while (0)
        // every (TPX, NOY) with 2 Tp <= 32, 2 Tf <= 96 (round 5: the reference's --past_length / --future_length are free CLI flags,
        // train.py:25-26; rounds 1-4 built (1,1) (1,2) (1,3) (2,2) (2,3) (2,5) only)
#define L0L_ROW(TX)                                                                                              \
        switch (NOY) {                                                                                           \
            case 1: L0L(TX, 1); break; case 2: L0L(TX, 2); break; case 3: L0L(TX, 3); break;                     \
            case 4: L0L(TX, 4); break; case 5: L0L(TX, 5); break; case 6: L0L(TX, 6); break;                     \
            default: STT_REQUIRE(false, "sttode_mlp_block0: future length beyond the built instantiations (2*Tf <= 96)"); \
        }
        if (TPX == 1) { L0L_ROW(1) } else if (TPX == 2) { L0L_ROW(2) }
        else STT_REQUIRE(false, "sttode_mlp_block0: past length beyond the built instantiations (2*Tp <= 32)");
#undef L0L_ROW
#undef L0L
        STT_HIP(hipGetLastError());
        return 0;
    }
    const int ngroups = (ncols + 63) / 64;
    int grid = MLP0_WGS * num_cus();   // workgroups per CU; even blockIdx = x role, odd = y role
    if (grid > 2 * ngroups || nonpersistent()) grid = 2 * ngroups;
    grid &= ~1;
    if (grid < 2) grid = 2;
    hipStream_t s = (hipStream_t)stream_;
#define L0(TX, NY)                                                                                                              \
    do {                                                                                                                        \
        STT_SET_LDS_ONCE((mlp_block0_kernel<TX, NY>), MLP0_LDS(TX, NY));                                                        \
        hipLaunchKernelGGL((mlp_block0_kernel<TX, NY>), dim3(grid), dim3(256), MLP0_LDS(TX, NY), s, A0x, A0y, (const f32x4*)stream, \
                           total_chunks, z, xpad, dbuf, ybuf, ncols, K);                                                \
    } while (0)
#define L0_ROW(TX)                                                                                               \
    switch (NOY) {                                                                                               \
        case 1: L0(TX, 1); break; case 2: L0(TX, 2); break; case 3: L0(TX, 3); break;                            \
        case 4: L0(TX, 4); break; case 5: L0(TX, 5); break; case 6: L0(TX, 6); break;                            \
        default: STT_REQUIRE(false, "sttode_mlp_block0: future length beyond the built instantiations (2*Tf <= 96)"); \
    }
    if (TPX == 1) { L0_ROW(1) } else if (TPX == 2) { L0_ROW(2) }
    else STT_REQUIRE(false, "sttode_mlp_block0: past length beyond the built instantiations (2*Tp <= 32)");
#undef L0_ROW
#undef L0
    STT_HIP(hipGetLastError());
    return 0;
}

extern "C" int sttode_mlp_block1(const float* A1y, const float* stream, int total_chunks, const float* z,
                                 const float* state1, const float* ybuf, const float* cur, const float* orig, float* pred,
                                 int ncols, int K, int Tf, int NOY, void* stream_) {
    STT_REQUIRE(A1y && stream && z && state1 && ybuf && cur && orig && pred, "sttode_mlp_block1: null pointer");
    STT_REQUIRE(ncols > 0 && K > 0 && Tf > 0 && 2 * Tf <= 16 * NOY, "sttode_mlp_block1: bad ncols/K/Tf/NOY");
    STT_REQUIRE(total_chunks == 32 + NOY, "sttode_mlp_block1: weight stream must hold 32+NOY chunks");
    if ((ncols + 15) / 16 <= mlp_lat_tiles()) {   // few columns: latency form
        MlpLatArgs a;
        a.A0 = A1y; a.blob = (const f32x4*)stream; a.z = z; a.state = state1; a.xpad = nullptr; a.ybuf = ybuf; a.cur = cur; a.orig = orig;
        a.out = pred; a.ncols = ncols; a.K = K; a.Tf2 = 2 * Tf; a.out_lds = nullptr; a.state_lds = nullptr; a.a0_lds = nullptr;
        const dim3 g((ncols + 15) / 16);
        hipStream_t sl = (hipStream_t)stream_;
#define L1L(NY)                                                                                   \
    do {                                                                                          \
        hipLaunchKernelGGL(mlp1_lat_kernel<NY>, g, dim3(256), 0, sl, a);                              \
    } while (0)
        switch (NOY) {
            case 1: L1L(1); break;
            case 2: L1L(2); break;
            case 3: L1L(3); break;
            case 4: L1L(4); break;
            case 5: L1L(5); break;
            case 6: L1L(6); break;
            default: STT_REQUIRE(false, "sttode_mlp_block1: future length beyond the built instantiations (2*Tf <= 96)");
        }
#undef L1L
        STT_HIP(hipGetLastError());
        return 0;
    }
    const int ngroups = (ncols + 63) / 64;
    int grid = 3 * num_cus();
    if (grid > ngroups || nonpersistent()) grid = ngroups;
    hipStream_t s = (hipStream_t)stream_;
#define L1(NY)                                                                                                              \
    do {                                                                                                                    \
        STT_SET_LDS_ONCE(mlp_block1_kernel<NY>, MLP1_LDS(NY));                                                              \
        hipLaunchKernelGGL((mlp_block1_kernel<NY>), dim3(grid), dim3(256), MLP1_LDS(NY), s, A1y, (const f32x4*)stream, total_chunks, \
                           z, state1, ybuf, cur, orig, pred, ncols, K, 2 * Tf);                                      \
    } while (0)
    switch (NOY) {
        case 1: L1(1); break;
        case 2: L1(2); break;
        case 3: L1(3); break;
        case 4: L1(4); break;
        case 5: L1(5); break;
        case 6: L1(6); break;
        default: STT_REQUIRE(false, "sttode_mlp_block1: future length beyond the built instantiations (2*Tf <= 96)");
    }
#undef L1
    STT_HIP(hipGetLastError());
    return 0;
}

// One MLP with B = [z | state] per column (decoder_x / decoder_y of a non-first DecomposeBlock, model/STTODE.py:71-75):
// out [ncols, 16*NO] raw output tiles.  stream: packing.mlp_stream(W1[:, 128:], W2, W3 padded, CHT = 1), 32 + NO chunks.
extern "C" int sttode_mlp_cols(const float* A0, const float* stream, int total_chunks, const float* z,
                               const float* state, float* out, int ncols, int K, int NO, void* stream_) {
    STT_REQUIRE(A0 && stream && z && state && out, "sttode_mlp_cols: null pointer");
    STT_REQUIRE(ncols > 0 && K > 0 && total_chunks == 32 + NO, "sttode_mlp_cols: bad ncols/K/chunk count");
    const int ngroups = (ncols + 63) / 64;
    int grid = 2 * num_cus();
    if (grid > ngroups) grid = ngroups;
    hipStream_t s = (hipStream_t)stream_;
#define LC(NY)                                                                                                              \
    do {                                                                                                                    \
        STT_SET_LDS_ONCE(mlp_cols_kernel<NY>, MLP1_LDS(NY));                                                                \
        hipLaunchKernelGGL((mlp_cols_kernel<NY>), dim3(grid), dim3(256), MLP1_LDS(NY), s, A0, (const f32x4*)stream, total_chunks, z, \
                           state, out, ncols, K);                                                                           \
    } while (0)
    switch (NO) {
        case 1: LC(1); break;
        case 2: LC(2); break;
        case 3: LC(3); break;
        case 4: LC(4); break;
        case 5: LC(5); break;
        case 6: LC(6); break;
        default: STT_REQUIRE(false, "sttode_mlp_cols: output width beyond the built instantiations (<= 96)");
    }
#undef LC
    STT_HIP(hipGetLastError());
    return 0;
}
