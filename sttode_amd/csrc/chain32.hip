// Fused per-trajectory decoder chain (round 2): ONE persistent kernel runs, for a group of 128 trajectories,
//     decoder_x MLP (block 0) -> d = x_true - x_hat0 -> decoder_y MLP (block 0) -> conv + GRU (block 1)
//     -> decoder_y MLP (block 1) -> pred = ((y_hat0 + y_hat1) + cur) + orig
// (model/STTODE.py:51-77 DecomposeBlock.forward x2, :320-347 Decoder.forward, :621-622), with every intermediate
// (d, state, the 512/256-wide hidden activations) in REGISTERS.  Round 1 ran this as three kernels (mlp_block0 ->
// gru_cols -> mlp_block1, csrc/decoder.hip) that exchanged dbuf / ybuf / state1 through HBM and each paid a grid tail.
//
// CDNA4 design:
//   * v_mfma_f32_32x32x2_f32: a wave owns 32 trajectories ("columns") on the 32 MFMA columns, features in registers.
//     One 32-feature x 32-column tile is a f32x16 per lane:  lane l: column l & 31, half h = l >> 5;
//     register j  <->  feature 8*(j/4) + 4*h + (j%4).  That accumulator layout IS the B-operand layout of the next
//     layer (MFMA step j consumes k-pair (8*(j/4) + j%4, +4)), so the whole chain needs no transpose and no LDS round trip.
//     Versus the 16x16x4 chain of round 1: half the MFMA instructions and half the A-operand LDS bytes per FLOP, no
//     dependent-accumulator stall (64-cycle issue = 64-cycle latency); measured 143.6 vs 135.2 TFLOP/s in the
//     weight-stream probe (profiles/r02/diag_probe.json).
//   * ALL weights (both MLPs of block 0, conv + GRU of block 1, the MLP of block 1: 3.2 MB per group) stream L2 -> LDS
//     by LDS-DMA (global_load_lds_dwordx4) in chunks of <= 3 "PK32" tiles of 4 KiB (tile = A operand of 16 MFMAs:
//     32 rows x 32 k), double buffered, one workgroup barrier per chunk; the chunk order is a host-built PROGRAM
//     (packing.chain_stream), so the kernel only consumes tiles in order.  The GRU's recurrent weights are streamed per
//     step like everything else (66 FLOP per streamed byte, the same intensity as the MLP layers), which frees the
//     144 KiB of LDS round 1 pinned for them: LDS per workgroup is 24 KiB ring + 32 KiB gather / z slots + 8 KiB biases + program.
//   * workgroup = 4 waves x 32 columns = 128 trajectories, 2 workgroups per CU (<= 256 VGPRs per wave); groups are
//     handed out by an atomic work counter (one tail for the whole chain instead of three, and a later launch on
//     another stream fills it: the kernel holds no chip-wide resource).
#include "chain.hpp"
#include "latency_bodies.hpp"
#include "frontend_body.hpp"
#include "api_util.hpp"
#include "../../include/sttode_hip.h"
#include <cstdlib>
#include <cstddef>

// C32_DIAG_TRACE (diagnostic build, profiles/exp_r03_trace.py): every workgroup appends one record {launch tag, block, kind, start, end (100 MHz
// s_memrealtime), HW_ID, XCC_ID} to the debug buffer -- who ran where and when, across overlapping launches
#ifdef C32_DIAG_TRACE
__shared__ long long g_tr_ph[4];   // role phase stamps (thread 0)
#define C32_TRACE_BEGIN() long long _tr_t0 = 0, _tr_c0 = 0; if (threadIdx.x == 0 && A.dbg) { _tr_t0 = __builtin_amdgcn_s_memrealtime(); _tr_c0 = __builtin_amdgcn_s_memtime(); }
#define C32_TRACE_END(kind) do { if (threadIdx.x == 0 && A.dbg) { \
        const long long _t1 = __builtin_amdgcn_s_memrealtime(); unsigned _hw, _xcc; \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(_hw)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(_xcc)); \
        const unsigned long long _r = atomicAdd((unsigned long long*)A.dbg, 1ull); if ((long long)_r >= A.dbg[1]) break; /* dbg[1] = capacity in records */ \
        long long* _p = A.dbg + 8 + _r * 12; _p[0] = A.trace_tag; _p[1] = blockIdx.x; _p[2] = (kind); _p[3] = _tr_t0; _p[4] = _t1; _p[5] = _hw; _p[6] = _xcc; \
        _p[8] = g_tr_ph[0]; _p[9] = g_tr_ph[1]; _p[10] = g_tr_ph[2]; _p[7] = __builtin_amdgcn_s_memtime() - _tr_c0; _p[11] = g_tr_ph[3]; } } while (0)
#define C32_TRACE_PHASE(i) do { if (threadIdx.x == 0) g_tr_ph[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define C32_TRACE_PHASE(i) do { } while (0)
#define C32_TRACE_BEGIN() do { } while (0)
#define C32_TRACE_END(kind) do { } while (0)
#endif

#include "role_body.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// diagnostic builds (profiles/exp_chain_variants.sh; never shipped): -DC32_DIAG_NODMA / _NOGATHER / _NOBARRIER / _NOGATES
#ifdef C32_DIAG_NOGATES
#define C32_SIG(x) ((x) * 1e-3f)
#define C32_TANH(x) ((x) * 1e-3f)
#else
#define C32_SIG(x) sigmoid_prescaled(x)
#define C32_TANH(x) tanh_prescaled(x)
#endif

#define C32_TILE 256                        // f32x4 per tile (4 KiB)
#define C32_CMAX 3                          // tiles per chunk
#define C32_RING (2 * C32_CMAX * C32_TILE)  // f32x4 in the double buffer (24 KiB)
#define C32_SLOT 256                        // f32x4 per wave gather slot (4 KiB)

struct Role32Args {   // throughput-form per-agent roles (role32.hpp)
    const f32x4* pool; const int2* prog; int prog_len; const float* consts;   // packing.role_stream (the host picks the variant)
    const float* enc_in; const int* last;                                     // [n][4 Tp] encoder inputs, [n] last-agent flag (scenes variant)
    const float* g_in; const float* attn; int ld_attn;                        // NBA variant: g [n][64], attention output before out_proj; attn == nullptr: scenes
    const float* xpad; int ldx;                                               // [n][ldx] normalised track, flattened (t, c), zero padded
    float* pf; float* state0; float* A0x; float* A0y; float* A1y;             // [n][128], [n][96], [n][512] x 3
    int n, Tp, kte, nwg;                                                      // agents, observed frames, k-tiles of x, role workgroups in the grid
    float ode_time;
    int prio;       // > 0: the role waves run at s_setprio 3 (default; STTODE_ROLE_PRIO=0: A/B)
    int* counter;   // work queue of THIS call's trajectory groups (a later launch of the same stream): zeroed by role workgroup 0
    float* zgen; unsigned zkey0, zkey1; int K;   // zgen != nullptr: the roles draw this call's latents z [n K][32] ~ N(0, I) themselves (role32.hpp latents32)
    // past != nullptr (scene batches): the roles run STTODENet.set_data for their own 128 agents first (role32.hpp frontend32) -- no front-end
    // launch.  past / scene_ptr may be device memory or pinned host memory (the reads then ARE the H2D transfer of the call's inputs).
    const float* past; const int* scene_ptr; int S; float* scene_orig; int* agent_scene;
    float* enc_in_w; float* xpad_w; float* cur_w; float* orig_w; int* last_w;
    float* m_ade; float* m_fde;   // fused metrics of this call (ChainArgs::m_*): set to +inf here, atomicMin'ed by its groups two launches later
};

struct ChainArgs {
    const float* A0x; const float* A0y; const float* A1y;  // [nagents][512] per-agent layer-1 pre-activations (b1 included)
    const f32x4* pool;                                      // PK32 tile pool
    const int2* prog; int prog_len;                         // chunk program of ONE group: (first tile, tiles <= 3)
    const float* consts;                                    // biases, layout below
    const float* z;                                         // [ncols][32]
    const float* xpad; int ldx;                             // [nagents][ldx] normalised past, flattened (t,c), zero padded
    const float* cur; const float* orig;                    // [nagents][2]
    float* pred;                                            // [ncols][Tf2]; WRITE-ONLY for the kernel when `park` is another buffer (may then be pinned host memory)
    float* park;                                            // [ncols][Tf2] where block 0's y_hat0 waits for the epilogue (== pred: the round-2/3 layout)
    int* counter;                                           // work queue (zeroed before the launch)
    int ncols, K, Tp, Tf2;
    int persistent;  // 1: workgroups pull groups from the work queue until it is empty; 0: one group per workgroup (grid = groups)
    // fused metrics (lagged launch, optional): min-over-K ADE / FDE of the predictions against gt [nagents][Tf][2] (utils/metrics.py:7-26) by
    // the groups themselves -- each column's two values in best_of_k_kernel's summation order, then atomicMin on the float bits (>= 0)
    // into ade / fde [nagents], which the roles of the same call set to +inf two launches earlier
    const float* m_gt; float* m_ade; float* m_fde; float m_scale;
    int nworkers;    // lagged launch with workers: worker w starts with group w, further groups are tickets nworkers + counter++
    long long* dbg;  // diagnostic builds only (C32_DIAG_STAMPS / C32_DIAG_TRACE): per-workgroup stamps
    int trace_tag;   // diagnostic builds only: launch number
    int xcd_map;     // fused launch, roles in front: group blocks b, b + 8, b + 16, .. (dispatched to ONE XCD) take consecutive group ids
    RoleArgs R;      // fused launch only (traj_chain_kernel<NY, 1>)
    Role32Args R32;  // lagged launch only (traj_chain_kernel<NY, 2>): throughput-form roles of ANOTHER call in front of this call's groups
};

// consts layout (floats): b2x[256] b3x[32] | b2y[256] b3y[32*NY] | gbias[4][96] convb[32] | b2m[256] b3m[32*NY]
template <int NY> struct C32Const {
    static constexpr int b2x = 0, b3x = 256, b2y = 288, b3y = 544, gb = 544 + 32 * NY, cb = gb + 384, b2m = cb + 32,
                         b3m = b2m + 256, total = b3m + 32 * NY;
};

__device__ __forceinline__ f32x16 splat16(float v) {
    f32x16 r;
#pragma unroll
    for (int e = 0; e < 16; ++e) r[e] = v;
    return r;
}
__device__ __forceinline__ f32x16 relu16(f32x16 v) {
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = fmaxf(v[e], 0.f);
    return v;
}
// 32 consecutive features starting at p (LDS or global), in accumulator layout: register 4a+b <- p[8a + 4h + b]
__device__ __forceinline__ f32x16 ldrows(const float* p, int h) {
    f32x16 r;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const f32x4 v = ld4(p + 8 * a + 4 * h);
        r[4 * a + 0] = v[0]; r[4 * a + 1] = v[1]; r[4 * a + 2] = v[2]; r[4 * a + 3] = v[3];
    }
    return r;
}

// BUF: f32x4 per ring buffer (= the largest chunk).  GEN = false: program entries are (first PK32 tile, tiles <= 3) of 4-KiB tiles;
// GEN = true (exploratory bf16-split mode: tiles of 6 and 4 KiB in one pool): (offset in f32x4 units, number of 1-KiB pieces).
template <int BUF, bool GEN>
struct ChainStreamT {
    static constexpr int kBuf = BUF;
    const f32x4* pool; const int2* lprog; f32x4* ring;  // lprog: the chunk program, copied to LDS at kernel start
    unsigned ring_addr;
    int len, p, par, lane, wave;
    int2 nxt_v;  // program entry of the chunk after the one in flight: read from LDS one step early, consumed at the next begin()
    __device__ __forceinline__ void dma(int2 ev, int buf) {
        // a chunk = `pieces` pieces of 1 KiB; wave w moves pieces w, w+4, w+8, ...
        const int off = __builtin_amdgcn_readfirstlane(ev.x), cnt = __builtin_amdgcn_readfirstlane(ev.y);
        const f32x4* src = pool + (GEN ? (size_t)off : (size_t)off * C32_TILE) + lane;
        const int pieces = GEN ? cnt : 4 * cnt;
        const unsigned dst = __builtin_amdgcn_readfirstlane(ring_addr + (unsigned)buf * (BUF * 16) + (unsigned)wave * 1024);
#ifndef C32_DIAG_NODMA
#pragma unroll
        for (int i = 0; i < (BUF / 64 + 3) / 4; ++i)
            if (4 * i + wave < pieces) glds16_asm(src + (4 * i + wave) * 64, dst + i * 4096);
#endif
    }
    __device__ __forceinline__ void init(const f32x4* pool_, const int2* lprog_, int len_, f32x4* ring_) {
        pool = pool_; lprog = lprog_; len = len_; ring = ring_;
        ring_addr = lds_addr(ring_);
        lane = threadIdx.x & 63;
        wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        p = 0; par = 0;
        dma(lprog[0], 0);
        nxt_v = lprog[1 % len];
    }
    // start of a chunk step: prefetch the next chunk into the other buffer (its readers passed the previous barrier)
    __device__ __forceinline__ void begin() {
        __builtin_amdgcn_sched_barrier(0);
        dma(nxt_v, par ^ 1);
        int q = p + 2;
        q = q >= len ? q - len : q;
        nxt_v = lprog[q];
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ const f32x4* cur() const { return ring + par * BUF + lane; }
    // end of a chunk step: this wave's DMA pieces have landed (vmcnt), then everybody's (barrier)
    __device__ __forceinline__ void end() {
        __builtin_amdgcn_sched_barrier(0);
#ifndef C32_DIAG_NOBARRIER
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#endif
        p = p + 1 >= len ? 0 : p + 1;
        par ^= 1;
        __builtin_amdgcn_sched_barrier(0);
    }
};
typedef ChainStreamT<C32_CMAX * C32_TILE, false> ChainStream;

// acc += Atile (32 rows x 32 k, PK32) * B (32 k x 32 columns in accumulator layout)
__device__ __forceinline__ void tile_mma(f32x16& acc, const f32x4* __restrict__ t, const f32x16& B) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 a = t[g * 64];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], B[4 * g + r], acc, 0, 0, 0);
    }
}

// Software-pipelined form.  A lone wave per SIMD issues in order: with the fragment reads placed right in front of their MFMAs
// (what hipcc does with tile_mma) every pair of reads costs ~40 idle matrix-pipe cycles (LDS latency ~100 vs the 64-cycle shadow of
// the previous MFMA), and the first reads of a chunk sit behind the DMA issue.  Here the fragments of a tile are registers handed
// in (`cur`), the next tile's fragments of the same chunk are requested halfway through this tile, and the first tile of a chunk is
// read BEFORE the chunk's DMA pieces are issued (ChainStream::begin), so its latency overlaps that issue.
struct Frag { f32x4 a[4]; };
__device__ __forceinline__ void ldfrag(Frag& f, const f32x4* __restrict__ t) {
#pragma unroll
    for (int g = 0; g < 4; ++g) f.a[g] = t[g * 64];
}
__device__ __forceinline__ void tile_mma2(f32x16& acc, const Frag& cur, const f32x16& B, Frag& nxt, const f32x4* __restrict__ tn) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (g == 2 && tn) {
            ldfrag(nxt, tn);
            __builtin_amdgcn_sched_barrier(0);   // hipcc otherwise sinks these reads back in front of their first use
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[g][r], B[4 * g + r], acc, 0, 0, 0);
    }
}

// Layers 1+2 of one MLP for this wave's 32 columns: acc2 (256 rows = 8 tiles) += W2 relu(A0[agent] + W1v [z | Bh]).
// Per 32-row hidden tile: 1 + KH layer-1 tiles then 8 layer-2 tiles, a chunk boundary every 3 tiles ((1 + KH + 8) % 3 == 0).
// z (the first layer-1 k-tile's B operand) is read from the wave's LDS z slot every hidden tile (it would cost 16 VGPRs for the
// whole group otherwise; the kernel has 256 and must not spill: a kernel with ANY scratch pays a scratch set-up per dispatch).
// The per-agent pre-activation rows arrive through the wave's gather slot (4 x 16 B per lane per hidden tile, LDS-DMA):
// a0 points at this lane's A0 row; a0_next at the row the NEXT phase starts with.
template <int KH, class ST>
__device__ __forceinline__ void mlp_l12(ST& st, const f32x4* slot, const f32x4* zslot, const f32x16* Bh,
                                        const float* __restrict__ a0, const float* __restrict__ a0_next, f32x16 (&acc2)[8], int lane, int h) {
    const unsigned slot_addr = __builtin_amdgcn_readfirstlane(lds_addr(slot));
    constexpr int KT1 = 1 + KH;
    static_assert((KT1 + 8) % 3 == 0, "hidden tile must be a whole number of chunks");
#pragma unroll
    for (int R = 0; R < 8; ++R) acc2[R] = splat16(0.f);
#pragma unroll 1
    for (int ht = 0; ht < 16; ++ht) {
        f32x16 h1, zb;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const f32x4 v = slot[a * 64 + lane];
            h1[4 * a + 0] = v[0]; h1[4 * a + 1] = v[1]; h1[4 * a + 2] = v[2]; h1[4 * a + 3] = v[3];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const f32x4 v = zslot[a * 64 + lane];
            zb[4 * a + 0] = v[0]; zb[4 * a + 1] = v[1]; zb[4 * a + 2] = v[2]; zb[4 * a + 3] = v[3];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slot is re-filled by the gather issued next
        __builtin_amdgcn_sched_barrier(0);
        const float* nx = (ht + 1 < 16) ? a0 + 32 * (ht + 1) : a0_next;
        Frag fr[2];
#pragma unroll
        for (int i = 0; i < KT1 + 8; ++i) {
            if (i % 3 == 0) {
                if (i > 0) st.end();
                ldfrag(fr[i & 1], st.cur());       // first tile of the chunk: its read latency overlaps the DMA issue below
                st.begin();
                if (i == 0) {
#ifndef C32_DIAG_NOGATHER
#pragma unroll
                    for (int a = 0; a < 4; ++a) glds16_asm(nx + 8 * a + 4 * h, slot_addr + a * 1024);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            const f32x4* tn = (i % 3 < 2) ? st.cur() + (i % 3 + 1) * C32_TILE : nullptr;
            if (i == 0) {
                tile_mma2(h1, fr[i & 1], zb, fr[(i + 1) & 1], tn);
            } else if (i < KT1) {
                tile_mma2(h1, fr[i & 1], Bh[i - 1], fr[(i + 1) & 1], tn);
            } else {
                if (i == KT1) h1 = relu16(h1);
                tile_mma2(acc2[i - KT1], fr[i & 1], h1, fr[(i + 1) & 1], tn);
            }
        }
        st.end();
    }
}

// Layer 3: out[o] = b3 + W3[o] relu(acc2 + b2), NO output tiles of 32 rows; 8 k-tiles per output tile, chunks of 3 tiles.
template <int NO, class ST>
__device__ __forceinline__ void mlp_l3(ST& st, f32x16 (&acc2)[8], const float* __restrict__ b2, const float* __restrict__ b3,
                                       f32x16 (&out)[NO], int h) {
#pragma unroll
    for (int R = 0; R < 8; ++R) {
        STT_FENCE();  // keep the bias reads next to their use (hoisted as a block they spill)
        const f32x16 b = ldrows(b2 + 32 * R, h);
        acc2[R] = relu16(acc2[R] + b);
        asm volatile("" : "+v"(acc2[R]));  // ... and the add right behind its read (8 bias tiles in flight = 128 VGPRs otherwise)
    }
    STT_FENCE();
#pragma unroll
    for (int o = 0; o < NO; ++o) out[o] = ldrows(b3 + 32 * o, h);
    Frag fr[2];
#pragma unroll
    for (int i = 0; i < 8 * NO; ++i) {
        if (i % 3 == 0) {
            if (i > 0) st.end();
            ldfrag(fr[i & 1], st.cur());
            st.begin();
        }
        const f32x4* tn = (i % 3 < 2 && i + 1 < 8 * NO) ? st.cur() + (i % 3 + 1) * C32_TILE : nullptr;
        tile_mma2(out[i / 8], fr[i & 1], acc2[i % 8], fr[(i + 1) & 1], tn);
    }
    st.end();
}

// ---------------------------------------------------------------------------------------------------
// EXPLORATORY (round 3, opt-in, never the default): the two block-0 MLPs on the bf16 matrix cores with fp32-class accuracy.
// x = hi + mid + lo (three bf16 values, 8 mantissa bits each: 24 bits, what fp32 holds), W likewise (split on the host, packing.pk32b_tile);
// W x ~ lo.hi + hi.lo + mid.mid + mid.hi + hi.mid + hi.hi (the three dropped products are <= 2^-24 relative), fp32 accumulate:
// six v_mfma_f32_32x32x16_bf16 (32 cycles each) per 16-deep k block = 384 matrix-pipe cycles per 32 x 32 x 32 tile against 1 024 for
// sixteen 32x32x2 fp32 MFMAs.  The fp32 accumulator layout of a layer is still the B layout of the next: k slot s of lane half h of
// k block kb <-> accumulator register 8 kb + s.  Costs: the split of every activation tile (11 VALU instructions per pair of values),
// 1.5x the A-operand bytes.  Measured in the chain's weight-stream structure: 252 vs 143 fp32-equivalent TFLOP/s
// (profiles/r03/bf16x3_probe.json, csrc/diag/diag.hip shape 6).
// ---------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
#define C32_TILE_B3 384                     // f32x4 per bf16-split tile (6 KiB)
#define C32_BUF_B3 (C32_CMAX * C32_TILE_B3) // f32x4 per ring buffer in the exploratory mode (18 KiB)
struct B3 { bf16x8 p[3][2]; };              // [plane: hi, mid, lo][k block]
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ void split3(const f32x16& x, B3& o) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        unsigned hi[4], mi[4], lo[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float x0 = x[8 * kb + 2 * q], x1 = x[8 * kb + 2 * q + 1];
            const f32x2v v = {x0, x1};
            hi[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
            const float r0 = x0 - bf_lo(hi[q]), r1 = x1 - bf_hi(hi[q]);
            const f32x2v rv = {r0, r1};
            mi[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(rv, bf16x2));
            const f32x2v qv = {r0 - bf_lo(mi[q]), r1 - bf_hi(mi[q])};
            lo[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(qv, bf16x2));
        }
        const u32x4v H = {hi[0], hi[1], hi[2], hi[3]}, M = {mi[0], mi[1], mi[2], mi[3]}, L = {lo[0], lo[1], lo[2], lo[3]};
        o.p[0][kb] = __builtin_bit_cast(bf16x8, H);
        o.p[1][kb] = __builtin_bit_cast(bf16x8, M);
        o.p[2][kb] = __builtin_bit_cast(bf16x8, L);
    }
}
struct Frag6 { f32x4 a[6]; };               // [k block * 3 + plane]
__device__ __forceinline__ void ldfrag6(Frag6& f, const f32x4* __restrict__ t) {
#pragma unroll
    for (int i = 0; i < 6; ++i) f.a[i] = t[i * 64];
}
// acc += tile x B; the next tile's six fragments are requested halfway (tn != nullptr), like tile_mma2.  Small products first.
__device__ __forceinline__ void tile_mma_b3(f32x16& acc, const Frag6& cur, const B3& B, Frag6& nxt, const f32x4* __restrict__ tn) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        if (kb == 1 && tn) {
            ldfrag6(nxt, tn);
            __builtin_amdgcn_sched_barrier(0);
        }
        const bf16x8 ah = __builtin_bit_cast(bf16x8, cur.a[kb * 3 + 0]), am = __builtin_bit_cast(bf16x8, cur.a[kb * 3 + 1]),
                     al = __builtin_bit_cast(bf16x8, cur.a[kb * 3 + 2]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, B.p[0][kb], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, B.p[2][kb], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, B.p[1][kb], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, B.p[0][kb], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, B.p[1][kb], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, B.p[0][kb], acc, 0, 0, 0);
    }
}
// acc += tile x B without look-ahead: the tile's six fragments are read here (the register-tight block-1 form; the SIMD's other wave
// covers the LDS latency)
__device__ __forceinline__ void tile_mma_b3n(f32x16& acc, const f32x4* __restrict__ t, const B3& B) {
    Frag6 f, dummy;
    ldfrag6(f, t);
    tile_mma_b3(acc, f, B, dummy, nullptr);
}
// mlp_l12<KH> on split tiles: per hidden tile 1 + KH layer-1 tiles (B = z, then the KH state tiles Bh: each split HERE from its fp32
// registers -- keeping three more tiles as planes would cost 72 VGPRs the kernel does not have) + 8 layer-2 row tiles.
// PRE: fragment look-ahead across tiles (KH = 0); without it 24 fewer VGPRs (KH = 3).
template <int KH, bool PRE, class ST>
__device__ __forceinline__ void mlp_l12_b3(ST& st, const f32x4* slot, const f32x4* zslot, const f32x16* Bh, const float* __restrict__ a0,
                                           const float* __restrict__ a0_next, f32x16 (&acc2)[8], int lane, int h) {
    const unsigned slot_addr = __builtin_amdgcn_readfirstlane(lds_addr(slot));
    constexpr int KT1 = 1 + KH;
    static_assert((KT1 + 8) % 3 == 0, "hidden tile must be a whole number of chunks");
#pragma unroll
    for (int R = 0; R < 8; ++R) acc2[R] = splat16(0.f);
    // KH == 0 (block 0): z's planes are made ONCE per phase and stay in registers (24 VGPRs this form can afford: 16 splits and 64 LDS
    // reads fewer per phase); KH == 3 (block 1) has no registers to spare and re-splits z from its fp32 LDS slot per hidden tile
    B3 Bz;
    if (KH == 0) {
        f32x16 zb;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const f32x4 v = zslot[a * 64 + lane];
            zb[4 * a + 0] = v[0]; zb[4 * a + 1] = v[1]; zb[4 * a + 2] = v[2]; zb[4 * a + 3] = v[3];
        }
        split3(zb, Bz);
    }
#pragma unroll 1
    for (int ht = 0; ht < 16; ++ht) {
        f32x16 h1;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const f32x4 v = slot[a * 64 + lane];
            h1[4 * a + 0] = v[0]; h1[4 * a + 1] = v[1]; h1[4 * a + 2] = v[2]; h1[4 * a + 3] = v[3];
        }
        B3 Bt;
        if (KH == 0) {
            Bt = Bz;
        } else {
            f32x16 zb;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const f32x4 v = zslot[a * 64 + lane];
                zb[4 * a + 0] = v[0]; zb[4 * a + 1] = v[1]; zb[4 * a + 2] = v[2]; zb[4 * a + 3] = v[3];
            }
            split3(zb, Bt);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slot is re-filled by the gather issued next
        __builtin_amdgcn_sched_barrier(0);
        const float* nx = (ht + 1 < 16) ? a0 + 32 * (ht + 1) : a0_next;
        Frag6 fr[2];
#pragma unroll
        for (int i = 0; i < KT1 + 8; ++i) {
            if (i % 3 == 0) {
                if (i > 0) st.end();
                if (PRE) ldfrag6(fr[i & 1], st.cur());
                st.begin();
                if (i == 0) {
#ifndef C32_DIAG_NOGATHER
#pragma unroll
                    for (int a = 0; a < 4; ++a) glds16_asm(nx + 8 * a + 4 * h, slot_addr + a * 1024);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            const f32x4* tc = st.cur() + (i % 3) * C32_TILE_B3;
            const f32x4* tn = (i % 3 < 2) ? tc + C32_TILE_B3 : nullptr;
            if (i > 0 && i < KT1) {
                STT_FENCE();
                split3(Bh[i - 1], Bt);                // state tile i-1 of this wave's columns
            }
            f32x16& acc = i < KT1 ? h1 : acc2[i < KT1 ? 0 : i - KT1];
            if (PRE) tile_mma_b3(acc, fr[i & 1], Bt, fr[(i + 1) & 1], tn);
            else tile_mma_b3n(acc, tc, Bt);
            if (i == KT1 - 1) {
                h1 = relu16(h1);
                split3(h1, Bt);                       // the hidden tile's planes take the registers of the layer-1 operand's
            }
        }
        st.end();
    }
}
// mlp_l3 on split tiles, k-tile major: activation tile T is split once and feeds all NO output tiles.
template <int NO, class ST>
__device__ __forceinline__ void mlp_l3_b3(ST& st, f32x16 (&acc2)[8], const float* __restrict__ b2, const float* __restrict__ b3,
                                          f32x16 (&out)[NO], int h) {
#pragma unroll
    for (int o = 0; o < NO; ++o) out[o] = ldrows(b3 + 32 * o, h);
    Frag6 fr[2];
    B3 Bt;
#pragma unroll
    for (int i = 0; i < 8 * NO; ++i) {
        if (i % NO == 0) {
            STT_FENCE();
            const f32x16 b = ldrows(b2 + 32 * (i / NO), h);
            const f32x16 a = relu16(acc2[i / NO] + b);
            split3(a, Bt);
        }
        if (i % 3 == 0) {
            if (i > 0) st.end();
            ldfrag6(fr[i & 1], st.cur());
            st.begin();
        }
        const f32x4* tn = (i % 3 < 2 && i + 1 < 8 * NO) ? st.cur() + (i % 3 + 1) * C32_TILE_B3 : nullptr;
        tile_mma_b3(out[i % NO], fr[i & 1], Bt, fr[(i + 1) & 1], tn);
    }
    st.end();
}

// planes -> fp32 (hi + mid + lo; within 2^-25 relative of the value that was split): the previous state for the GRU's blend
__device__ __forceinline__ f32x16 b3_to_f32(const B3& b) {
    f32x16 r;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const u32x4v H = __builtin_bit_cast(u32x4v, b.p[0][kb]), M = __builtin_bit_cast(u32x4v, b.p[1][kb]), L = __builtin_bit_cast(u32x4v, b.p[2][kb]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            r[8 * kb + 2 * q] = (bf_lo(H[q]) + bf_lo(M[q])) + bf_lo(L[q]);
            r[8 * kb + 2 * q + 1] = (bf_hi(H[q]) + bf_hi(M[q])) + bf_hi(L[q]);
        }
    }
    return r;
}
// gru32_steps on split tiles (exploratory mode): the step's B operands -- conv input d (constant over the steps), conv output e, the
// three state tiles -- are split ONCE per step (each feeds 9 gate tiles); the new state is kept fp32 until the step ends, then split.
// Same tile order as the fp32 stream: per step 1 conv tile (a chunk of its own), then per j: r:[e h0 h1 h2] z:[e h0 h1 h2] n_h:[h0 h1 h2]
// n_i:[e] = 4 chunks of 3 tiles.  No fragment look-ahead (registers: 96 + 24 of planes, 48 of new state, up to 48 of gates).
template <class ST>
__device__ __forceinline__ void gru32_steps_b3(ST& st, const float* gb, const float* cb, const f32x16& d, f32x16 (&hs)[3], int Tp, int h) {
    B3 dB, hB[3];
    split3(d, dB);
#pragma unroll
    for (int j = 0; j < 3; ++j) split3(hs[j], hB[j]);
    constexpr int T = C32_TILE_B3;
#pragma unroll 1
    for (int t = 0; t < Tp; ++t) {
        B3 eB;
        {
            f32x16 e = ldrows(cb, h);
            st.begin();
            tile_mma_b3n(e, st.cur(), dB);
            e = relu16(e);
            st.end();
            split3(e, eB);
        }
        f32x16 hn[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            STT_FENCE();
            f32x16 ar = ldrows(gb + 0 * 96 + 32 * j, h);
            st.begin();
            tile_mma_b3n(ar, st.cur(), eB);
            tile_mma_b3n(ar, st.cur() + T, hB[0]);
            tile_mma_b3n(ar, st.cur() + 2 * T, hB[1]);
            st.end(); st.begin();
            tile_mma_b3n(ar, st.cur(), hB[2]);
#pragma unroll
            for (int r = 0; r < 16; ++r) ar[r] = C32_SIG(ar[r]);          // r gate
            STT_FENCE();
            f32x16 az = ldrows(gb + 1 * 96 + 32 * j, h);
            tile_mma_b3n(az, st.cur() + T, eB);
            tile_mma_b3n(az, st.cur() + 2 * T, hB[0]);
            st.end(); st.begin();
            tile_mma_b3n(az, st.cur(), hB[1]);
            tile_mma_b3n(az, st.cur() + T, hB[2]);
#pragma unroll
            for (int r = 0; r < 16; ++r) az[r] = C32_SIG(az[r]);          // z gate
            STT_FENCE();
            f32x16 an = ldrows(gb + 3 * 96 + 32 * j, h);
            tile_mma_b3n(an, st.cur() + 2 * T, hB[0]);
            st.end(); st.begin();
            tile_mma_b3n(an, st.cur(), hB[1]);
            tile_mma_b3n(an, st.cur() + T, hB[2]);
            {
                const f32x16 bi = ldrows(gb + 2 * 96 + 32 * j, h);
#pragma unroll
                for (int r = 0; r < 16; ++r) an[r] = fmaf(ar[r], an[r], bi[r]);     // b_in + r * (W_hn h + b_hn)
            }
            STT_FENCE();
            tile_mma_b3n(an, st.cur() + 2 * T, eB);                                 // + W_in e
            st.end();
            const f32x16 hp = b3_to_f32(hB[j]);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float ng = C32_TANH(an[r]);
                hn[j][r] = fmaf(az[r], hp[r] - ng, ng);  // (1-z) n + z h
            }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            hs[j] = hn[j];
            split3(hn[j], hB[j]);
        }
    }
}

// conv1d(k=3) + relu + GRU(32 -> 96) over Tp steps for this wave's 32 columns, every weight tile streamed per step
// (model/STTODE.py:62-69; gate rows pre-scaled, chain.hpp): d = the flattened (t, c) input rows in accumulator layout, hs = h (in/out),
// gb = gate biases [4][96] (r, z, b_in, b_hn), cb = conv bias [32], both in LDS.  Per step: 1 conv tile (a chunk of its own) and 36
// gate tiles = 12 chunks.  Shared by the fused chain (block 1, per trajectory) and gru32_kernel (block 0, per agent).
template <class ST>
__device__ __forceinline__ void gru32_steps(ST& st, const float* gb, const float* cb, const f32x16& d, f32x16 (&hs)[3], int Tp, int h) {
#pragma unroll 1
    for (int t = 0; t < Tp; ++t) {
        f32x16 e = ldrows(cb, h);
        Frag fa, fb;
        ldfrag(fa, st.cur());
        st.begin();
        tile_mma2(e, fa, d, fb, nullptr);
        e = relu16(e);
        st.end();
        f32x16 hn[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            // 12 tiles: r:[e h0 h1 h2] z:[e h0 h1 h2] n_h:[h0 h1 h2] n_i:[e]; chunk boundary every 3 tiles.
            // One gate accumulator is live at a time (finished gates shrink to their 16 outputs).
            STT_FENCE();
            f32x16 ar = ldrows(gb + 0 * 96 + 32 * j, h);
            ldfrag(fa, st.cur()); st.begin();
            tile_mma2(ar, fa, e, fb, st.cur() + 1 * C32_TILE);
            tile_mma2(ar, fb, hs[0], fa, st.cur() + 2 * C32_TILE);
            tile_mma2(ar, fa, hs[1], fb, nullptr);
            st.end(); ldfrag(fa, st.cur()); st.begin();
            tile_mma2(ar, fa, hs[2], fb, st.cur() + 1 * C32_TILE);
#pragma unroll
            for (int r = 0; r < 16; ++r) ar[r] = C32_SIG(ar[r]);          // r gate
            STT_FENCE();
            f32x16 az = ldrows(gb + 1 * 96 + 32 * j, h);
            tile_mma2(az, fb, e, fa, st.cur() + 2 * C32_TILE);
            tile_mma2(az, fa, hs[0], fb, nullptr);
            st.end(); ldfrag(fa, st.cur()); st.begin();
            tile_mma2(az, fa, hs[1], fb, st.cur() + 1 * C32_TILE);
            tile_mma2(az, fb, hs[2], fa, st.cur() + 2 * C32_TILE);
#pragma unroll
            for (int r = 0; r < 16; ++r) az[r] = C32_SIG(az[r]);          // z gate
            STT_FENCE();
            f32x16 an = ldrows(gb + 3 * 96 + 32 * j, h);
            tile_mma2(an, fa, hs[0], fb, nullptr);
            st.end(); ldfrag(fa, st.cur()); st.begin();
            tile_mma2(an, fa, hs[1], fb, st.cur() + 1 * C32_TILE);
            tile_mma2(an, fb, hs[2], fa, st.cur() + 2 * C32_TILE);
            {
                const f32x16 bi = ldrows(gb + 2 * 96 + 32 * j, h);
#pragma unroll
                for (int r = 0; r < 16; ++r) an[r] = fmaf(ar[r], an[r], bi[r]);     // b_in + r * (W_hn h + b_hn)
            }
            STT_FENCE();
            tile_mma2(an, fa, e, fb, nullptr);                                       // + W_in e
            st.end();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float ng = C32_TANH(an[r]);
                hn[j][r] = fmaf(az[r], hs[j][r] - ng, ng);  // (1-z) n + z h
            }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) hs[j] = hn[j];
    }

}

#include "role32.hpp"

// makes a per-lane integer opaque to the optimiser: address arithmetic derived from it is redone where it is used instead of
// being computed once at the top of the group and kept live (64-bit pointers held across phases were what spilled)
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }

#ifdef C32_DIAG_STAMPS
#define C32_STAMP(k) do { if (threadIdx.x == 0 && A.dbg && gi < 4) { A.dbg[((size_t)blockIdx.x * 4 + gi) * 16 + 2 * (k)] = __builtin_amdgcn_s_memtime(); \
                                                                    A.dbg[((size_t)blockIdx.x * 4 + gi) * 16 + 2 * (k) + 1] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define C32_STAMP(k) do { } while (0)
#endif

// Grid order of the fused launch.  Roles and groups are interleaved: the role of tile t is placed `lead` groups ahead of the first group
// that reads its tables, g_first(t) = floor(t K / 8) (a tile = 16 agents = 16 K trajectories, a group = 128), so that in a running
// pipeline a role has finished by the time its consumers are dispatched (nobody spins), and a single serial launch starts its
// trajectory groups at once instead of behind ALL roles (the per-agent stage then costs its matrix work, not a prologue).  Sort key:
// role t -> max(0, g_first(t) - lead), group g -> g, roles first on ties; every role a group needs has g_first <= g, hence a smaller
// block index: workgroups are dispatched in index order PER XCD (each XCD takes block % 8), and a role never waits, so within ONE launch
// every producer finishes whatever its consumers do (no deadlock; the bounded spin of wait_tiles is the backstop).  Several such launches
// in flight on different queues have no such guarantee across XCDs: the pipelined path uses the lagged form (FUSE = 2), which has no hand-off.  Returns the group index, or -1 - tile for a role.  Scalar code: ~10 binary-search steps.
__host__ __device__ __forceinline__ int fused_block_of(long b, long T, long G, long K, long lead) {
    auto roles_upto = [&](long x) { const long c = (8 * (x + lead + 1) + K - 1) / K; return c < T ? c : T; };   // #roles with key <= x
    long lo = -1, hi = G - 1;                       // largest g with position g + roles_upto(g) <= b
    while (lo < hi) {
        const long mid = (lo + hi + 1) >> 1;
        if (mid + roles_upto(mid) <= b) lo = mid; else hi = mid - 1;
    }
    if (lo >= 0 && lo + roles_upto(lo) == b) return (int)lo;
    return -1 - (int)(b - (lo + 1));
}

// XCD-aware group order.  Workgroups are dealt to the 8 XCDs round-robin in dispatch order, and each XCD has its own L2.  A 16-agent tile's
// table rows (3 x 2 KiB per agent, written through to memory by its role) are read by the 2-3 trajectory groups of its 320 trajectories; in
// plain block order those are consecutive blocks = different XCDs, and every one of them fetches the rows from the fabric again.  With
// group id = (blocks of one XCD get a contiguous range) the sharers run behind the same L2.  A permutation of which block computes which
// group: results unchanged; producers (roles) still precede every group.
__host__ __device__ __forceinline__ int xcd_group(int b, int G) {
    const int x = b & 7, i = b >> 3, q = G >> 3, r = G & 7;
    return x * q + (x < r ? x : r) + i;
}

// FUSE: 0 = trajectory groups only; 1 = round-3 fused launch (latency-form roles of THIS call in front, tile flags); 2 = lagged launch
// (round 4: throughput-form roles of a LATER call in front, no dependency inside the launch; either part may be empty)
template <int NY, int FUSE, bool B3M>
__global__ __launch_bounds__(256, 2) void traj_chain_kernel(ChainArgs A) {
    typedef ChainStreamT<B3M ? C32_BUF_B3 : C32_CMAX * C32_TILE, B3M> Stream;   // B3M: exploratory bf16-split mode (block-0 MLPs)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* ring = reinterpret_cast<f32x4*>(smem);
    f32x4* slots = ring + 2 * Stream::kBuf;
    f32x4* zslots = slots + 4 * C32_SLOT;
    float* cst = reinterpret_cast<float*>(zslots + 4 * C32_SLOT);
    int2* lprog = reinterpret_cast<int2*>(cst + C32Const<NY>::total);
    int* sq = reinterpret_cast<int*>(lprog + A.prog_len);  // [2]: this workgroup's first group / the group after the current one
    typedef C32Const<NY> CO;

    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const f32x4* slot = slots + wave * C32_SLOT;
    const f32x4* zslot = zslots + wave * C32_SLOT;
    const unsigned slot_addr = __builtin_amdgcn_readfirstlane(lds_addr(slot));
    const unsigned zslot_addr = __builtin_amdgcn_readfirstlane(lds_addr(zslot));
    const int ngroups = (A.ncols + 127) >> 7;

    C32_TRACE_BEGIN();
    if (FUSE == 2 && (int)blockIdx.x < A.R32.nwg) {   // (uniform) a throughput-form role workgroup: 128 agents of the roles' call
        role32_body(*(KRole32Args*)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(ChainArgs, R32)),
                    blockIdx.x, smem);
        C32_TRACE_END(1);
        return;
    }
    if (FUSE == 1 && A.R.split && (int)blockIdx.x < 5 * A.R.ntiles) {   // (uniform) split per-agent roles, all in front of the groups
        split_role(A.R, (A.ncols + A.K - 1) / A.K, A.Tp, A.ldx, A.xpad, blockIdx.x, smem);
        C32_TRACE_END(1);
        return;
    }
    int fb = FUSE == 0 ? (int)blockIdx.x : FUSE == 2 ? (int)blockIdx.x - A.R32.nwg : A.R.split ? (int)blockIdx.x - 5 * A.R.ntiles
                                                : fused_block_of(blockIdx.x, A.R.ntiles, ngroups, A.K, A.R.lead);
    if (FUSE && A.xcd_map && fb >= 0) fb = xcd_group(fb, ngroups);   // (roles in front: fb was the group block's position in dispatch order)
    if (FUSE == 1 && fb < 0) {   // (uniform) a per-agent role
        agent_role(A.R, (A.ncols + A.K - 1) / A.K, A.Tp, A.ldx, A.xpad, -1 - fb, smem);
        C32_TRACE_END(1);
        return;
    }
    for (int i = threadIdx.x; i < CO::total; i += blockDim.x) cst[i] = A.consts[i];
    for (int i = threadIdx.x; i < A.prog_len; i += blockDim.x) lprog[i] = A.prog[i];
    // A.persistent (FUSE 0 / 2): the workgroup is a WORKER that pulls groups from the call's work queue until it is empty.  Lagged launch:
    // worker w starts with group w and draws tickets only for its further groups, and not at all when every group has a worker --
    // same-address atomics of a whole grid starting together serialise at ~1 us each (measured: 213 workers, +0.2 ms per launch)
    if (threadIdx.x == 0) sq[0] = FUSE == 2 ? fb : (FUSE == 0 && A.persistent) ? atomicAdd(A.counter, 1) : FUSE ? fb : (int)blockIdx.x;
    if (FUSE == 1) {   // this group's per-agent tables come from role workgroups of THIS launch: wait for their tiles (one wave polls)
        const int g0 = fb;
        const int c_lo = g0 * 128, c_hi = (c_lo + 127 < A.ncols ? c_lo + 127 : A.ncols - 1);
        const int t_lo = (c_lo / A.K) >> 4, t_hi = (c_hi / A.K) >> 4;
        if (wave == 0) {
            const bool ok = A.R.split ? wait_tiles(A.R.pflags, 3 * t_lo, 3 * t_hi + 2, A.R.flags + A.R.ntiles, lane, A.R.tmo_host)   // the three tables of every tile
                                      : wait_tiles(A.R.flags, t_lo, t_hi, A.R.flags + A.R.ntiles, lane, A.R.tmo_host);
            if (!ok && lane == 0) sq[0] = -1;
        }
    }
    __syncthreads();
    C32_TRACE_PHASE(3);   // (groups) flags seen
    int g = sq[0];
    if (FUSE == 1 && g < 0) {   // (uniform) time-out: poison the group's predictions, never hang
        for (int i = threadIdx.x; i < 128 * A.Tf2; i += blockDim.x) {
            const size_t o = (size_t)fb * 128 * A.Tf2 + i;
            if (o < (size_t)A.ncols * A.Tf2) A.pred[o] = __builtin_nanf("");
        }
        return;
    }
    if (g >= ngroups) return;  // (uniform) cannot happen with grid <= ngroups; nothing is in flight yet
    Stream st;
    st.init(A.pool, lprog, A.prog_len, ring);

    auto col_of = [&](int gg) { int col = gg * 128 + wave * 32 + c; return col < A.ncols ? col : A.ncols - 1; };
    {   // first gathers: A0x rows of hidden tile 0, and z of this group
        const int c0 = col_of(g);
        const float* a0 = A.A0x + (size_t)(c0 / A.K) * 512;
        const float* zp = A.z + (size_t)c0 * 32;
#pragma unroll
        for (int a = 0; a < 4; ++a) glds16_asm(a0 + 8 * a + 4 * h, slot_addr + a * 1024);
#pragma unroll
        for (int a = 0; a < 4; ++a) glds16_asm(zp + 8 * a + 4 * h, zslot_addr + a * 1024);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int gi = 0;
    (void)gi;
    while (true) {
        C32_STAMP(0);
        const int col = g * 128 + wave * 32 + c;
        int colc = col < A.ncols ? col : A.ncols - 1;
        int agent = colc / A.K;
        const bool live = col < A.ncols;
        // the NEXT group is requested now (one ticket of look-ahead: the last MLP prefetches its first gather); every wave reads
        // it after the many barriers of this group
        __syncthreads();  // every wave has read sq[1] of the previous hand-over before it is overwritten
        if (threadIdx.x == 0)
            sq[1] = (FUSE == 2 && A.persistent && ngroups > A.nworkers) ? A.nworkers + atomicAdd(A.counter, 1)
                    : (FUSE == 0 && A.persistent) ? atomicAdd(A.counter, 1) : ngroups;

        f32x16 acc2[8];
        f32x16 d;
        {   // ---- block 0, decoder_x: x_hat0, d = x_true - x_hat0
            agent = opaque(agent);
            f32x16 xo[1];
            if (B3M) {
                mlp_l12_b3<0, true>(st, slot, zslot, nullptr, A.A0x + (size_t)agent * 512, A.A0y + (size_t)agent * 512, acc2, lane, h);
                mlp_l3_b3<1>(st, acc2, cst + CO::b2x, cst + CO::b3x, xo, h);
            } else {
                mlp_l12<0>(st, slot, zslot, nullptr, A.A0x + (size_t)agent * 512, A.A0y + (size_t)agent * 512, acc2, lane, h);
                mlp_l3<1>(st, acc2, cst + CO::b2x, cst + CO::b3x, xo, h);
            }
            agent = opaque(agent);
            const float* xp = A.xpad + (size_t)agent * A.ldx;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                f32x4 v = splat4(0.f);
                if (8 * a + 4 * h < A.ldx) v = ld4(xp + 8 * a + 4 * h);
#pragma unroll
                for (int b = 0; b < 4; ++b) d[4 * a + b] = v[b] - xo[0][4 * a + b];
            }
        }
        STT_FENCE();
        C32_STAMP(1);
        {   // ---- block 0, decoder_y: y_hat0 parked in pred (re-read by the epilogue)
            agent = opaque(agent);
            f32x16 yo[NY];
            if (B3M) {
                mlp_l12_b3<0, true>(st, slot, zslot, nullptr, A.A0y + (size_t)agent * 512, A.A1y + (size_t)agent * 512, acc2, lane, h);
                mlp_l3_b3<NY>(st, acc2, cst + CO::b2y, cst + CO::b3y, yo, h);
            } else {
                mlp_l12<0>(st, slot, zslot, nullptr, A.A0y + (size_t)agent * 512, A.A1y + (size_t)agent * 512, acc2, lane, h);
                mlp_l3<NY>(st, acc2, cst + CO::b2y, cst + CO::b3y, yo, h);
            }
            if (live) {
                float* prow = A.park + (size_t)opaque(col) * A.Tf2;
#pragma unroll
                for (int o = 0; o < NY; ++o)
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const int row0 = 32 * o + 8 * a + 4 * h;
                        float* p = prow + row0;
                        if (row0 + 3 < A.Tf2 && (A.Tf2 & 3) == 0) {
                            f32x4 v = {yo[o][4 * a], yo[o][4 * a + 1], yo[o][4 * a + 2], yo[o][4 * a + 3]};
                            st4(p, v);
                        } else {
#pragma unroll
                            for (int b = 0; b < 4; ++b)
                                if (row0 + b < A.Tf2) p[b] = yo[o][4 * a + b];
                        }
                    }
            }
        }
        STT_FENCE();
        C32_STAMP(2);
        f32x16 hs[3];
        {   // ---- block 1: conv1d + relu + GRU over Tp steps, weights streamed per step
#pragma unroll
            for (int j = 0; j < 3; ++j) hs[j] = splat16(0.f);
            if (B3M) gru32_steps_b3(st, cst + CO::gb, cst + CO::cb, d, hs, A.Tp, h);
            else gru32_steps(st, cst + CO::gb, cst + CO::cb, d, hs, A.Tp, h);
        }
        STT_FENCE();
        C32_STAMP(3);
        const int gnext = sq[1];  // written at the top of this group
        {   // ---- block 1, decoder_y + epilogue
            agent = opaque(agent);
            const int cnx = col_of(gnext < ngroups ? gnext : g);
            if (B3M) mlp_l12_b3<3, false>(st, slot, zslot, hs, A.A1y + (size_t)agent * 512, A.A0x + (size_t)(cnx / A.K) * 512, acc2, lane, h);
            else mlp_l12<3>(st, slot, zslot, hs, A.A1y + (size_t)agent * 512, A.A0x + (size_t)(cnx / A.K) * 512, acc2, lane, h);
            {   // z of the NEXT group into the z slot (this group's last read of it is behind us); lands during layer 3
                const float* zp = A.z + (size_t)opaque(cnx) * 32;
#pragma unroll
                for (int a = 0; a < 4; ++a) glds16_asm(zp + 8 * a + 4 * h, zslot_addr + a * 1024);
            }
            f32x16 yo[NY];
            if (B3M) mlp_l3_b3<NY>(st, acc2, cst + CO::b2m, cst + CO::b3m, yo, h);
            else mlp_l3<NY>(st, acc2, cst + CO::b2m, cst + CO::b3m, yo, h);
            {
                agent = opaque(agent);
                const float cx = A.cur[2 * agent], cy = A.cur[2 * agent + 1];
                const float ox = A.orig[2 * agent], oy = A.orig[2 * agent + 1];
                const bool vec = (A.Tf2 & 3) == 0;
                const float* yrow = A.park + (size_t)opaque(colc) * A.Tf2;     // y_hat0 parked by this lane (dead lanes: a live column's, unused)
                float* prow = A.pred + (size_t)opaque(col) * A.Tf2;
                const float* grow = FUSE == 2 && A.m_gt ? A.m_gt + (size_t)agent * A.Tf2 : nullptr;
                float dd[NY][4][2];                                           // fused metrics: this lane's displacement norms (steps t0, t0 + 1)
#pragma unroll
                for (int o = 0; o < NY; ++o)
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const int row0 = 32 * o + 8 * a + 4 * h;
                        const bool whole = vec && row0 + 3 < A.Tf2;
                        f32x4 y0 = splat4(0.f);
                        if (whole) y0 = ld4(yrow + row0);
                        else {
#pragma unroll
                            for (int b = 0; b < 4; ++b)
                                if (row0 + b < A.Tf2) y0[b] = yrow[row0 + b];
                        }
                        f32x4 v;
                        v[0] = ((y0[0] + yo[o][4 * a + 0]) + cx) + ox;
                        v[1] = ((y0[1] + yo[o][4 * a + 1]) + cy) + oy;
                        v[2] = ((y0[2] + yo[o][4 * a + 2]) + cx) + ox;
                        v[3] = ((y0[3] + yo[o][4 * a + 3]) + cy) + oy;
                        if (live) {
                            if (whole) st4(prow + row0, v);
                            else {
#pragma unroll
                                for (int b = 0; b < 4; ++b)
                                    if (row0 + b < A.Tf2) prow[row0 + b] = v[b];
                            }
                        }
                        dd[o][a][0] = dd[o][a][1] = 0.f;
                        if (FUSE == 2 && grow) {   // (uniform)
                            if (row0 < A.Tf2) { const float2 g = *reinterpret_cast<const float2*>(grow + row0); dd[o][a][0] = bok_dist(v[0], v[1], g.x, g.y, A.m_scale); }
                            if (row0 + 2 < A.Tf2) { const float2 g = *reinterpret_cast<const float2*>(grow + row0 + 2); dd[o][a][1] = bok_dist(v[2], v[3], g.x, g.y, A.m_scale); }
                        }
                    }
                if (FUSE == 2 && grow) {   // (uniform) sum over t in order t = 0, 1, ..: lane (c, h = 0) holds steps 4a, 4a + 1 of tile row group a, lane (c, 1) steps 4a + 2, 4a + 3
                    const int Tf = A.Tf2 >> 1;
                    float acc = 0.f, fdl = -1.f;
#pragma unroll
                    for (int o = 0; o < NY; ++o)
#pragma unroll
                        for (int a = 0; a < 4; ++a) {
                            const int t0 = (32 * o + 8 * a + 4 * h) >> 1;
                            float mine = acc;
                            if (h == 0) { if (t0 < Tf) mine += dd[o][a][0]; if (t0 + 1 < Tf) mine += dd[o][a][1]; }
                            float mine1 = __shfl(mine, c, 64);
                            if (h == 1) { if (t0 < Tf) mine1 += dd[o][a][0]; if (t0 + 1 < Tf) mine1 += dd[o][a][1]; }
                            acc = __shfl(mine1, c + 32, 64);
                            if (t0 == Tf - 1) fdl = dd[o][a][0];
                            if (t0 + 1 == Tf - 1) fdl = dd[o][a][1];
                        }
                    fdl = fmaxf(fdl, __shfl_xor(fdl, 32, 64));
                    const float adev = acc / (float)Tf;
                    if (live && h == 0) {
                        atomicMin(reinterpret_cast<unsigned*>(A.m_ade) + agent, __float_as_uint(adev));
                        atomicMin(reinterpret_cast<unsigned*>(A.m_fde) + agent, __float_as_uint(fdl));
                    }
                }
            }
        }
        C32_STAMP(4);
        ++gi;
        g = gnext;
        if (g >= ngroups) break;  // uniform: every wave read the same sq word
    }
    C32_TRACE_END(0);
}

// Stand-alone streaming GRU over columns (block 0: one column per AGENT): the same 32-column MFMA tiles and weight stream as the
// fused chain, 24 KiB of LDS and <= 256 VGPRs per wave, so its workgroups co-reside with a chain workgroup of another stream.  The
// resident-weights gru_cols kernel (144 KiB of LDS) cannot: in the pipelined form it had to wait for the running chain's tail.
struct Gru32Args {
    const float* xin; int ldx;                    // [ncols][ldx] flattened (t, c) input rows, zero padded
    const f32x4* pool; const int2* prog; int prog_len;
    const float* consts;                          // gbias[4][96] convb[32]
    float* state;                                 // [ncols][96]
    int ncols, Tp;
};
__global__ __launch_bounds__(256, 2) void gru32_kernel(Gru32Args A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* ring = reinterpret_cast<f32x4*>(smem);
    float* cst = reinterpret_cast<float*>(ring + C32_RING);
    int2* lprog = reinterpret_cast<int2*>(cst + 416);
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 416; i += blockDim.x) cst[i] = A.consts[i];
    for (int i = threadIdx.x; i < A.prog_len; i += blockDim.x) lprog[i] = A.prog[i];
    __syncthreads();
    ChainStream st;
    st.init(A.pool, lprog, A.prog_len, ring);
    const int col = blockIdx.x * 128 + wave * 32 + c;
    const int colc = col < A.ncols ? col : A.ncols - 1;
    f32x16 d;
    {
        const float* xp = A.xin + (size_t)colc * A.ldx;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            f32x4 v = splat4(0.f);
            if (8 * a + 4 * h < A.ldx) v = ld4(xp + 8 * a + 4 * h);
#pragma unroll
            for (int b = 0; b < 4; ++b) d[4 * a + b] = v[b];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x16 hs[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) hs[j] = splat16(0.f);
    gru32_steps(st, cst, cst + 384, d, hs, A.Tp, h);
    if (col < A.ncols) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const f32x4 v = {hs[j][4 * a], hs[j][4 * a + 1], hs[j][4 * a + 2], hs[j][4 * a + 3]};
                st4(A.state + (size_t)col * 96 + 32 * j + 8 * a + 4 * h, v);
            }
    }
}

// ---------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------
static int chain_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}
static int chain_lds(int NY, int prog_len, bool b3 = false) { return ((b3 ? 2 * C32_BUF_B3 : C32_RING) + 8 * C32_SLOT) * 16 + (1216 + 64 * NY) * 4 + prog_len * 8 + 16; }

static int role_lds(int Tp) {   // agent_role's phases: embed (Tp*256 + 512 f32x4), GRU (h tiles 12 KiB + image of hidden tiles 4, 5: 48 KiB)
    const int e = (Tp * 256 + 512) * 16, g = (2 * 6 * 64 + 2 * 24 * 64 + 2 * 4 * 64) * 16;   // h tiles 12 KiB + image 48 KiB + gate hand-off 8 KiB
    return e > g ? e : g;
}

static int role32_lds(int prog_len) { return C32_RING * 16 + R32C::total * 4 + prog_len * 8 + 16; }

template <int NY, int FUSE, bool B3M = false> static int chain_launch(ChainArgs a, int wgs_per_cu, hipStream_t s) {
    STT_SET_LDS_ONCE((traj_chain_kernel<NY, FUSE, B3M>), 96 * 1024);   // once per (instantiation, device)
    const int ngroups = (a.ncols + 127) / 128;
    // STTODE_CHAIN_RESERVE=r leaves r of the chip's 2-per-CU workgroup slots to concurrently running kernels (the per-agent stage
    // of the next call in the pipelined form); the work queue makes the remaining workgroups absorb the groups
    static int reserve = -1;
    if (reserve < 0) { const char* e = getenv("STTODE_CHAIN_RESERVE"); reserve = e ? atoi(e) : 0; if (reserve < 0 || reserve > chain_cus()) reserve = 0; }
    int grid = 2 * chain_cus() - reserve;
    if (grid > ngroups || !a.persistent) grid = ngroups;
    if (FUSE == 1) grid = (a.R.split ? 5 : 1) * a.R.ntiles + ngroups;   // roles ahead of their consumers, one group per workgroup
    else if (FUSE == 2) {   // another call's throughput-form roles, then this call's groups: one workgroup each, or (persistent) workers
        int workers = ngroups;
        const int cap = a.persistent > 1 ? a.persistent : 2 * chain_cus();   // workers: the chip's workgroup slots (or STTODE_LAG_WORKERS=count)
        if (a.persistent && workers > cap) workers = cap;
        grid = a.R32.nwg + (a.ncols > 0 ? workers : 0);
        a.nworkers = workers;
        if (workers < ngroups) a.xcd_map = 0;   // (the XCD-aware order is a permutation of one-workgroup-per-group grids)
    }
    else if (a.persistent) STT_HIP(hipMemsetAsync(a.counter, 0, sizeof(int), s));   // the work queue of the persistent form
    // wgs_per_cu == 1: ask for more than half of the CU's LDS so that only ONE chain workgroup is resident per CU.  A lone workgroup
    // keeps the matrix pipe about as busy as two do (469 vs 2 x 397 us per group), and the other half of the register file plus ~76 KiB
    // of LDS stay free for kernels of OTHER streams (the separate per-agent launches of the unfused pipeline).  The fused launch needs
    // no co-residency and runs two per CU everywhere.
    static int wgs_env = -1;   // STTODE_CHAIN_WGS=1|2 overrides the caller's choice (experiments)
    if (wgs_env < 0) { const char* e = getenv("STTODE_CHAIN_WGS"); wgs_env = e ? atoi(e) : 0; }
    const int wgs = wgs_env > 0 ? wgs_env : wgs_per_cu;
    int lds = chain_lds(NY, a.prog_len, B3M);
    if (FUSE == 1 && lds < role_lds(a.Tp)) lds = role_lds(a.Tp);
    if (FUSE == 2 && a.R32.nwg > 0 && lds < role32_lds(a.R32.prog_len)) lds = role32_lds(a.R32.prog_len);
    if (wgs == 1 && lds < 84 * 1024) lds = 84 * 1024;
    STT_REQUIRE(lds <= 96 * 1024, "sttode_traj_chain: dynamic LDS beyond the 96 KiB the kernel is registered for");
    if (FUSE == 1) STT_HIP(hipMemsetAsync(a.R.flags, 0, (((size_t)(a.R.split ? 5 : 1) * a.R.ntiles + 1) * 4 + 15) / 16 * 16, s));   // tile flags + time-out word
    hipLaunchKernelGGL((traj_chain_kernel<NY, FUSE, B3M>), dim3(grid), dim3(256), lds, s, a);
    STT_HIP(hipGetLastError());
    return 0;
}

// DecomposeBlock front half over columns, streaming form (same function as sttode_gru_cols: model/STTODE.py:62-69).  pool / prog /
// consts: packing.gru32_stream (36 gate tiles + Tp conv tiles; 13*Tp chunk entries; gate biases [4][96] + conv bias [32]).
extern "C" int sttode_gru_cols32(const float* xin, int ldx, const float* pool, const int* prog, int prog_len, const float* consts,
                                 float* state, int ncols, int Tp, void* stream) {
    STT_REQUIRE(xin && pool && prog && consts && state, "sttode_gru_cols32: null pointer");
    STT_REQUIRE(ncols > 0 && Tp >= 1 && (ldx == 16 || ldx == 32) && 2 * Tp <= ldx, "sttode_gru_cols32: bad ncols/Tp/ldx");
    STT_REQUIRE(prog_len == 13 * Tp, "sttode_gru_cols32: chunk program must hold 13 entries per step");
    Gru32Args a;
    a.xin = xin; a.ldx = ldx; a.pool = (const f32x4*)pool; a.prog = (const int2*)prog; a.prog_len = prog_len; a.consts = consts;
    a.state = state; a.ncols = ncols; a.Tp = Tp;
    const int lds = C32_RING * 16 + 416 * 4 + prog_len * 8 + 16;
    hipLaunchKernelGGL(gru32_kernel, dim3((ncols + 127) / 128), dim3(256), lds, (hipStream_t)stream, a);
    STT_HIP(hipGetLastError());
    return 0;
}

#if defined(C32_DIAG_STAMPS) || defined(C32_DIAG_TRACE)
static long long* g_chain_dbg = nullptr;
static int g_trace_tag = 0;
extern "C" int sttode_chain_debug_buffer(void* p) { g_chain_dbg = (long long*)p; return 0; }  // >= grid * 4 * 16 int64, zeroed
#endif

// Host view of the fused launch's grid order (tests: every block is exactly one role or group; producers precede consumers).
extern "C" int sttode_fused_block_of(long block, long tiles, long groups, long K, long lead) {
    return fused_block_of(block, tiles, groups, K, lead);
}

extern "C" int sttode_chain_prog_len(int Tp, int Tf) {
    const int NY = (2 * Tf + 31) / 32;
    const int l3y = (8 * NY + 2) / 3;
    return (48 + 3) + (48 + l3y) + 13 * Tp + (64 + l3y);
}

// The chain without roles; b3: the exploratory bf16-split stream (pool / prog = packing.chain_stream_b3's) instead of the fp32 one.
static int traj_chain_impl(const float* A0x, const float* A0y, const float* A1y, const float* pool, const int* prog, int prog_len,
                           const float* consts, const float* z, const float* xpad, int ldx, const float* cur, const float* orig,
                           float* pred, int* counter, int ncols, int K, int Tp, int Tf, int wgs_per_cu, bool b3, void* stream) {
    STT_REQUIRE(A0x && A0y && A1y && pool && prog && consts && z && xpad && cur && orig && pred && counter, "sttode_traj_chain: null pointer");
    STT_REQUIRE(ncols > 0 && K > 0 && Tp >= 1 && 2 * Tp <= 32 && Tf >= 1, "sttode_traj_chain: bad ncols/K/Tp/Tf");
    STT_REQUIRE(ldx == 16 || ldx == 32, "sttode_traj_chain: ldx must be 16 or 32");
    STT_REQUIRE(wgs_per_cu == 1 || wgs_per_cu == 2, "sttode_traj_chain: wgs_per_cu must be 1 or 2");
    STT_REQUIRE(2 * Tp <= ldx, "sttode_traj_chain: xpad rows shorter than 2*Tp");
    STT_REQUIRE(prog_len == sttode_chain_prog_len(Tp, Tf), "sttode_traj_chain: chunk program length does not match (Tp, Tf)");
    ChainArgs a;
    a.A0x = A0x; a.A0y = A0y; a.A1y = A1y; a.pool = (const f32x4*)pool; a.prog = (const int2*)prog; a.prog_len = prog_len;
    a.consts = consts; a.z = z; a.xpad = xpad; a.ldx = ldx; a.cur = cur; a.orig = orig; a.pred = pred; a.park = pred; a.counter = counter;
    a.ncols = ncols; a.K = K; a.Tp = Tp; a.Tf2 = 2 * Tf;
    a.dbg = nullptr; a.trace_tag = 0; a.xcd_map = 0; a.nworkers = 0; a.m_gt = nullptr; a.m_ade = a.m_fde = nullptr; a.m_scale = 1.0f;
    a.R = RoleArgs();   // unused by the unfused instantiation
    {
        static int pers = -1;   // default 0: one group per workgroup (slots free up continuously, so kernels of other streams -- the next
                                // call's per-agent stage, the next chain -- interleave at group granularity; measured 1.38 vs 1.87 ms per
                                // 256-scene step); STTODE_CHAIN_PERSISTENT=1: workgroups pull groups from the atomic work queue
        if (pers < 0) { const char* e = getenv("STTODE_CHAIN_PERSISTENT"); pers = e ? (atoi(e) != 0) : 0; }
        a.persistent = pers;
    }
#if defined(C32_DIAG_STAMPS) || defined(C32_DIAG_TRACE)
    a.dbg = g_chain_dbg;
    a.trace_tag = g_trace_tag++;
#endif
    const int NY = (2 * Tf + 31) / 32;
    hipStream_t s = (hipStream_t)stream;
    if (b3) {
        a.persistent = 0;
        switch (NY) {
            case 1: return chain_launch<1, 0, true>(a, wgs_per_cu, s);
            case 2: return chain_launch<2, 0, true>(a, wgs_per_cu, s);
            case 3: return chain_launch<3, 0, true>(a, wgs_per_cu, s);
            default: STT_REQUIRE(false, "sttode_traj_chain: future length beyond the built instantiations (2*Tf <= 96)");
        }
    }
    switch (NY) {
        case 1: return chain_launch<1, 0>(a, wgs_per_cu, s);
        case 2: return chain_launch<2, 0>(a, wgs_per_cu, s);
        case 3: return chain_launch<3, 0>(a, wgs_per_cu, s);
        default: STT_REQUIRE(false, "sttode_traj_chain: future length beyond the built instantiations (2*Tf <= 96)");
    }
    return 0;
}
// Fused per-trajectory chain of Decoder.forward (model/STTODE.py:320-347) for K samples per agent; see the file header.
extern "C" int sttode_traj_chain(const float* A0x, const float* A0y, const float* A1y, const float* pool, const int* prog, int prog_len,
                                 const float* consts, const float* z, const float* xpad, int ldx, const float* cur, const float* orig,
                                 float* pred, int* counter, int ncols, int K, int Tp, int Tf, int wgs_per_cu, void* stream) {
    return traj_chain_impl(A0x, A0y, A1y, pool, prog, prog_len, consts, z, xpad, ldx, cur, orig, pred, counter, ncols, K, Tp, Tf, wgs_per_cu, false, stream);
}
// Internal (pipeline.hip): the same on the exploratory bf16-split stream.
int stt_traj_chain_b3(const float* A0x, const float* A0y, const float* A1y, const float* pool, const int* prog, int prog_len,
                      const float* consts, const float* z, const float* xpad, int ldx, const float* cur, const float* orig,
                      float* pred, int* counter, int ncols, int K, int Tp, int Tf, int wgs_per_cu, void* stream) {
    return traj_chain_impl(A0x, A0y, A1y, pool, prog, prog_len, consts, z, xpad, ldx, cur, orig, pred, counter, ncols, K, Tp, Tf, wgs_per_cu, true, stream);
}

// Internal (csrc/pipeline.hip): the fused launch -- per-agent roles + trajectory groups in ONE grid (see RoleArgs).  W = the model's
// weight table (enum SttodeWeight), ws / off = the caller's workspace and its layout; the front-end (xpad, enc_in, cur, orig, last) has
// run on `stream` before -- unless `past` / `scene_ptr` are given (scene batches): the roles then run it for their tiles themselves and the
// call is ONE launch.  attn == nullptr: attention length 1 (scene batches), the roles run the embedding too; attn != nullptr (the NBA
// branch: attention groups > 1): embed_qkv and mhgsa_attn have run on `stream` before, the roles start at the post-attention layer.
// The reference's one Euler step only.
bool stt_chain_fused_covers(int Tp) { return Tp >= 2 && 2 * Tp <= 32 && role_lds(Tp) <= 80 * 1024; }
int stt_chain_fused(const float* const* W, float* ws, const long* off, int n, int K, int Tp, int Tf, int prog_len, const float* z, float* pred,
                    float ode_time, const float* attn, int ld_attn, const float* past, const int* scene_ptr, int S, int wgs_per_cu, int b3,
                    int lead, int drop_tile, unsigned* tmo_host, void* stream) {
    STT_REQUIRE(W && ws && off && z && pred, "stt_chain_fused: null pointer");
    STT_REQUIRE(n > 0 && K > 0 && stt_chain_fused_covers(Tp) && Tf >= 1, "stt_chain_fused: shape outside the fused launch");
    STT_REQUIRE(!attn || (ld_attn >= 64 && ld_attn % 4 == 0), "stt_chain_fused: bad attention leading dimension");
    STT_REQUIRE(prog_len == sttode_chain_prog_len(Tp, Tf), "stt_chain_fused: chunk program length does not match (Tp, Tf)");
    STT_REQUIRE((long)n * K <= 0x7fffffffL, "stt_chain_fused: too many trajectories");
    ChainArgs a;
    a.A0x = ws + off[STT_B_A0X]; a.A0y = ws + off[STT_B_A0Y]; a.A1y = ws + off[STT_B_A1Y];
    a.pool = (const f32x4*)W[b3 ? STT_W_CHAINB3_POOL : STT_W_CHAIN_POOL]; a.prog = (const int2*)W[b3 ? STT_W_CHAINB3_PROG : STT_W_CHAIN_PROG];
    a.prog_len = prog_len;
    a.consts = W[STT_W_CHAIN_CONSTS]; a.z = z; a.xpad = ws + off[STT_B_XPAD]; a.ldx = 2 * Tp <= 16 ? 16 : 32; a.cur = ws + off[STT_B_CUR];
    a.orig = ws + off[STT_B_ORIG]; a.pred = pred; a.park = pred; a.counter = (int*)(ws + off[STT_B_QUEUE]);
    a.ncols = n * K; a.K = K; a.Tp = Tp; a.Tf2 = 2 * Tf; a.persistent = 0; a.dbg = nullptr; a.trace_tag = 0; a.xcd_map = 0; a.nworkers = 0;
    a.m_gt = nullptr; a.m_ade = a.m_fde = nullptr; a.m_scale = 1.0f;
#if defined(C32_DIAG_STAMPS) || defined(C32_DIAG_TRACE)
    a.dbg = g_chain_dbg;
    a.trace_tag = g_trace_tag++;
#endif
    RoleArgs& r = a.R;
    role_args_fill(r, W, ws, off);
    r.attn = attn; r.ld_attn = ld_attn;
    STT_REQUIRE(!past || (scene_ptr && S > 0 && !attn), "stt_chain_fused: the in-role front-end needs scene_ptr, S > 0 and attention length 1");
    r.past = past; r.scene_ptr = scene_ptr; r.S = S;
    r.flags = (unsigned*)(ws + off[STT_B_FLAGS]); r.ntiles = (n + 15) / 16; r.ode_time = ode_time;
    r.tmo_host = tmo_host;
    // grid order: `lead` groups of head start of a role over its first consumer; < 0 (default): all roles first.  Measured on one box
    // (profiles/r03/ab_lead_frontend_depth.txt): pipelined 73.4-74.6 M trajectories/s for lead 64 / 160 / 400 / roles first alike, but a
    // SERIAL launch is 5 % slower interleaved (0.65 vs 0.69 of peak): a role beside a trajectory group runs 2x longer than beside
    // other roles, and holds its slot all the while
    r.lead = lead < 0 ? (1 << 28) : lead;
    {   // XCD-aware group order whenever all roles sit in front of the groups (STTODE_XCD_MAP=0: plain block order, for A/B)
        static int xm = -1;
        if (xm < 0) { const char* e = getenv("STTODE_XCD_MAP"); xm = e ? atoi(e) != 0 : 1; }
        a.xcd_map = xm && lead < 0;
    }
    r.split = lead == -2;   // -2: roles first, each tile's role split into E | G | three table workgroups (opt-in, sttode_set_fused mode 4); -1 (default): one workgroup per tile
    r.gflags = r.flags + r.ntiles + 1; r.pflags = r.gflags + r.ntiles;
    r.drop_tile = drop_tile;
    const int NY = (2 * Tf + 31) / 32;
    hipStream_t s = (hipStream_t)stream;
    switch (NY) {
        case 1: return b3 ? chain_launch<1, 1, true>(a, wgs_per_cu, s) : chain_launch<1, 1>(a, wgs_per_cu, s);
        case 2: return b3 ? chain_launch<2, 1, true>(a, wgs_per_cu, s) : chain_launch<2, 1>(a, wgs_per_cu, s);
        case 3: return b3 ? chain_launch<3, 1, true>(a, wgs_per_cu, s) : chain_launch<3, 1>(a, wgs_per_cu, s);
        default: STT_REQUIRE(false, "stt_chain_fused: future length beyond the built instantiations (2*Tf <= 96)");
    }
    return 0;
}

// Internal (csrc/pipeline.hip): the LAGGED launch of the pipelined path (round 4) -- ONE grid = the throughput-form per-agent roles of one
// call (role32.hpp: 128 agents per workgroup; rW / ws_r / off_r / n_r; skipped when ws_r == nullptr) followed by the trajectory groups of
// ANOTHER, earlier call whose roles ran in an earlier launch of the same stream (ws_g / off_g / n_g / z / pred; skipped when ws_g ==
// nullptr).  Nothing in the grid depends on anything else in it.  attn != nullptr (NBA: attention groups > 1): embed_qkv and mhgsa_attn of
// the roles' call have run on `stream` before, the roles read g and the attention output from ws_r.  The reference's one Euler step only.
bool stt_chain_lagged_covers(int Tp) { return Tp >= 2 && 2 * Tp <= 32; }
int stt_chain_lagged(const float* const* W, const LagRoles& lr, const LagGroups& lg, int K, int Tp, int Tf, int prog_len, int b3, void* stream) {
    float* ws_r = lr.ws; const long* off_r = lr.off; const int n_r = lr.n; const float* attn = lr.attn; const int ld_attn = lr.ld_attn;
    const float ode_time = lr.ode_time; float* zgen = lr.zgen; const unsigned long long zkey = lr.zkey; const float* past = lr.past;
    const int* scene_ptr = lr.scene_ptr; const int S = lr.S; float* r_ade = lr.ade; float* r_fde = lr.fde;
    float* ws_g = lg.ws; const long* off_g = lg.off; const int n_g = lg.n; const float* z = lg.z; float* pred = lg.pred;
    const float* g_gt = lg.gt; float* g_ade = lg.ade; float* g_fde = lg.fde; const float g_scale = lg.scale; const int lag_workers_ok = lg.workers;
    STT_REQUIRE(W && (ws_r || ws_g), "stt_chain_lagged: nothing to launch");
    STT_REQUIRE(K > 0 && stt_chain_lagged_covers(Tp) && Tf >= 1, "stt_chain_lagged: shape outside the lagged launch");
    ChainArgs a;
    a.R = RoleArgs();
    a.R32 = Role32Args();
    a.dbg = nullptr; a.trace_tag = 0; a.xcd_map = 0; a.persistent = 0; a.counter = nullptr; a.nworkers = 0;
    a.K = K; a.Tp = Tp; a.Tf2 = 2 * Tf; a.ldx = 2 * Tp <= 16 ? 16 : 32; a.ncols = 0;
    a.pool = (const f32x4*)W[b3 ? STT_W_CHAINB3_POOL : STT_W_CHAIN_POOL]; a.prog = (const int2*)W[b3 ? STT_W_CHAINB3_PROG : STT_W_CHAIN_PROG];
    a.prog_len = prog_len; a.consts = W[STT_W_CHAIN_CONSTS];
    a.A0x = a.A0y = a.A1y = nullptr; a.z = nullptr; a.xpad = nullptr; a.cur = a.orig = nullptr; a.pred = nullptr; a.park = nullptr;
    a.m_gt = nullptr; a.m_ade = a.m_fde = nullptr; a.m_scale = 1.0f;
    if (ws_g && g_gt) {
        STT_REQUIRE(g_ade && g_fde, "stt_chain_lagged: fused metrics need ade and fde");
        a.m_gt = g_gt; a.m_ade = g_ade; a.m_fde = g_fde; a.m_scale = g_scale;
    }
    if (ws_g) {
        STT_REQUIRE(off_g && z && pred && n_g > 0 && (long)n_g * K <= 0x7fffffffL, "stt_chain_lagged: bad group arguments");
        STT_REQUIRE(prog_len == sttode_chain_prog_len(Tp, Tf), "stt_chain_lagged: chunk program length does not match (Tp, Tf)");
        a.A0x = ws_g + off_g[STT_B_A0X]; a.A0y = ws_g + off_g[STT_B_A0Y]; a.A1y = ws_g + off_g[STT_B_A1Y];
        a.z = z; a.xpad = ws_g + off_g[STT_B_XPAD]; a.cur = ws_g + off_g[STT_B_CUR]; a.orig = ws_g + off_g[STT_B_ORIG]; a.pred = pred;
        a.park = ws_g + off_g[STT_B_YBUF];   // (m x 16 NOY floats >= m x 2 Tf): the kernel never reads `pred`, which may be pinned host memory
        a.ncols = n_g * K;
        // Workers (default): the launch holds at most the chip's 2-per-CU workgroup slots, and its workgroups pull groups from the call's
        // work queue.  One workgroup per group (STTODE_LAG_WORKERS=0) deals the groups to the 8 XCDs statically (block % 8), the next launch
        // on another queue starts only when this grid is fully dispatched, and the XCDs do not run at one speed: the block trace shows six
        // XCDs idle for 250-700 us at the end of every launch while the slowest still has blocks to place (profiles/r04/trace_lagged_static.txt)
        static int workers = -1;
        if (workers < 0) { const char* e = getenv("STTODE_LAG_WORKERS"); workers = e ? atoi(e) : 1; if (workers < 0) workers = 1; }
        // (calls with launches in front of the roles -- the NBA branch's front-end, embedding and attention -- keep one workgroup per group:
        // workers hold every slot until their queue is empty and those kernels would wait for the launch's end; measured -8 % there)
        a.persistent = lag_workers_ok ? workers : 0;
        a.counter = (int*)(ws_g + off_g[STT_B_QUEUE]);       // zeroed by the roles of this call (an earlier launch of this stream)
        static int xm = -1;
        if (xm < 0) { const char* e = getenv("STTODE_XCD_MAP"); xm = e ? atoi(e) != 0 : 1; }
        a.xcd_map = xm;
    }
    if (ws_r) {
        STT_REQUIRE(off_r && n_r > 0, "stt_chain_lagged: bad role arguments");
        STT_REQUIRE(!attn || (ld_attn >= 64 && ld_attn % 4 == 0), "stt_chain_lagged: bad attention leading dimension");
        Role32Args& r = a.R32;
        r.pool = (const f32x4*)W[STT_W_ROLE32_POOL];
        r.prog = (const int2*)W[attn ? STT_W_ROLE32_PROG_NBA : STT_W_ROLE32_PROG_SCENES];
        r.consts = W[attn ? STT_W_ROLE32_CONSTS_NBA : STT_W_ROLE32_CONSTS_SCENES];
        r.kte = (4 * Tp + 31) / 32;
        const int e = attn ? 8 : 2 * r.kte + 8;   // == packing.role_prog_len
        r.prog_len = attn ? 13 * Tp + (e + 2) / 3 + (128 + 288 + 2) / 3 : 13 * Tp + (e + 128 + 288 + 2) / 3;
        r.enc_in = ws_r + off_r[STT_B_ENC_IN]; r.last = (const int*)(ws_r + off_r[STT_B_LAST]);
        r.g_in = ws_r + off_r[STT_B_G]; r.attn = attn; r.ld_attn = ld_attn;
        r.xpad = ws_r + off_r[STT_B_XPAD]; r.ldx = a.ldx;
        r.pf = ws_r + off_r[STT_B_PF]; r.state0 = ws_r + off_r[STT_B_STATE0];
        r.A0x = ws_r + off_r[STT_B_A0X]; r.A0y = ws_r + off_r[STT_B_A0Y]; r.A1y = ws_r + off_r[STT_B_A1Y];
        r.n = n_r; r.Tp = Tp; r.nwg = (n_r + 127) / 128; r.ode_time = ode_time;
        static const int role_prio = getenv("STTODE_ROLE_PRIO") ? atoi(getenv("STTODE_ROLE_PRIO")) : 3;   // role waves outlast the groups of a small launch: SDD-256 / NBA-128 +1..2 %, 512 scenes the same (profiles/r04/role_prio_ab.txt)
        r.prio = role_prio;
        r.counter = (int*)(ws_r + off_r[STT_B_QUEUE]);
        r.zgen = zgen; r.zkey0 = (unsigned)zkey; r.zkey1 = (unsigned)(zkey >> 32); r.K = K;
        STT_REQUIRE(!past || (scene_ptr && S > 0 && !attn), "stt_chain_lagged: the in-role front-end needs scene_ptr, S > 0 and attention length 1");
        r.past = past; r.scene_ptr = scene_ptr; r.S = S;
        r.scene_orig = ws_r + off_r[STT_B_SCENE_ORIG]; r.agent_scene = (int*)(ws_r + off_r[STT_B_AGENT_SCENE]);
        r.enc_in_w = ws_r + off_r[STT_B_ENC_IN]; r.xpad_w = ws_r + off_r[STT_B_XPAD]; r.cur_w = ws_r + off_r[STT_B_CUR];
        r.orig_w = ws_r + off_r[STT_B_ORIG]; r.last_w = (int*)(ws_r + off_r[STT_B_LAST]);
        r.m_ade = r_ade; r.m_fde = r_fde;
    }
#if defined(C32_DIAG_STAMPS) || defined(C32_DIAG_TRACE)
    a.dbg = g_chain_dbg;
    a.trace_tag = g_trace_tag++;
#endif
    const int NY = (2 * Tf + 31) / 32;
    hipStream_t s = (hipStream_t)stream;
    switch (NY) {
        case 1: return b3 ? chain_launch<1, 2, true>(a, 2, s) : chain_launch<1, 2>(a, 2, s);
        case 2: return b3 ? chain_launch<2, 2, true>(a, 2, s) : chain_launch<2, 2>(a, 2, s);
        case 3: return b3 ? chain_launch<3, 2, true>(a, 2, s) : chain_launch<3, 2>(a, 2, s);
        default: STT_REQUIRE(false, "stt_chain_lagged: future length beyond the built instantiations (2*Tf <= 96)");
    }
    return 0;
}
