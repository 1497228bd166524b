// One-launch form of STTODENet.inference() for FEW trajectories -- the reference's evaluation loop hands the model ONE scene per call
// (test.py:171-188: set_data -> inference -> .cpu()), i.e. 2-20 agents x K = 20 samples = a handful of 16-column tiles.  Round 2 ran
// that call as six dependent launches (front-end, per-agent stage, layer-1 tables, block-0 MLPs, block-1 conv + GRU, block-1 MLP:
// 113 us of kernel time plus five launch gaps, profiles/r03/per_scene_latency_before.txt); every one of them is a short chain of
// dependent steps on a few CUs, and the call's critical path is their SUM.
//
// Here the whole call is ONE launch of 256-thread workgroups with four ROLES, ordered producers-first in the grid, cut along the
// call's dependency graph rather than along the reference's module boundaries:
//     [ E_0, G_0, E_1, G_1, ... (A = ceil(n / 16) agent tiles) ]  [ Y_0, X_0, Y_1, X_1, ... (C = ceil(n K / 16) trajectory tiles) ]
//   E role -- set_data's normalisation (encoder inputs), PastEncoder: embedding -> post-attention / ODE -> past feature pf; publishes E;
//             then the block-1 layer-1 table A1y = W1y'[:, pf] pf + b (needed last), publishes E2          (model/STTODE.py:397-461,214-236)
//   G role -- set_data's normalisation (decoder inputs), block-0 conv + GRU over the observed track -> state0; publishes G   (:62-69)
//             E and G depend only on the inputs: they run side by side (the six-launch form ran them back to back)
//   Y role -- layer-1 pre-activations of ITS 16 trajectories' agents (W1y [pf | state0] + b, to LDS), block-0 decoder_y MLP -> ybuf;
//             publishes Y                                                                                     (:71-77,323-331)
//   X role -- the same for decoder_x -> d = x_true - x_hat0 (LDS) -> block-1 conv + GRU (state in LDS) -> block-1 decoder_y MLP
//             -> pred = ((y_hat0 + y_hat1) + cur) + orig                                                      (:51-77,320-347,621-622)
// The per-agent layer-1 tables of block 0 (A0x, A0y: 2 x 512 x 224 per agent, a third of the per-agent stage's time) are not built at all:
// a trajectory tile has at most 16 distinct agents, so its X / Y workgroup computes the 512 pre-activations of its own columns as the
// first 14 k-tiles of the layer-1 chain (bias, then [pf | state0], then z: the order in which preact_rows + mlp_lat_run sum them).
// Hand-off between workgroups: write-through (sc1) stores, one flag per producer, one acquire per consumer (role_body.hpp; the fused
// chain launch of chain32.hip uses the same protocol).  Every consumer waits only for workgroups with a SMALLER block index, which
// in-order dispatch has made resident (or finished) before it starts: no deadlock whatever the grid size; the bounded spin poisons the
// tile's predictions with NaN and raises the time-out word instead of hanging.  The bodies are the separate launches' code
// (latency_bodies.hpp), every output element summed in the same order: the predictions carry the bits of the six-launch path.
#include "chain.hpp"
#include "role_body.hpp"
#include "api_util.hpp"
#include "../../include/sttode_hip.h"

struct SceneLatArgs {
    RoleArgs R;                 // weights and workspace rows of the per-agent roles (R.flags: the E flags)
    MlpLatArgs x0, y0, y1;      // block-0 decoder_x / decoder_y, block-1 decoder_y (the LDS operands are set by the kernel)
    const f32x4* convP; const float* convB; const f32x4* wihP; const f32x4* whhP; const float* gbias;   // block-1 conv + GRU
    const float* xpad; int ldx;
    unsigned *tmo, *gflags, *e2flags, *yflags;   // flag words: E [A] | time-out | G [A] | E2 [A] | Y [C] | exit counter | "initialised" word
    unsigned *done, *magic;     // exit counter of the launch; SL_MAGIC once sttode_workspace_init has zeroed the flag words
    int n, K, Tp, Tf2, ntiles_c;
    long long* dbg;             // diagnostic build only (SL_DIAG_TRACE): [block][8] phase stamps (100 MHz)
};

// LDS of a workgroup (bytes).  X role: layer-1 pre-activation fragments 32 KiB | MLP exchange 24 KiB | GRU h tiles 12 KiB | GRU gate-sum
// exchange 10 KiB | d tiles 2 KiB;  E role: embedding (Tp*256 + 512) * 16 B, then 16 KiB;  G role: h tiles + exchange from 0.
#define SL_A0 0
#define SL_MLP (32 * 1024)
#define SL_SH (56 * 1024)
#define SL_GX (68 * 1024)
#define SL_D (78 * 1024)
#define SL_TOTAL (80 * 1024)          // + 16 B: the workgroup's go / time-out word, its "last to leave" word
#define SL_MAGIC 0x5774F1A6u          // the workspace's flag words were zeroed by sttode_workspace_init

#ifdef SL_DIAG_TRACE
#define SL_STAMP(k) do { if (threadIdx.x == 0 && A.dbg) A.dbg[(size_t)b * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SL_STAMP(k) do { } while (0)
#endif

template <int TPX, int NOY>
__device__ __forceinline__ void scene_lat_body(const SceneLatArgs& A, char* smem) {
    const RoleArgs& R = A.R;
    const int A_tiles = R.ntiles;
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    SL_STAMP(0);
#ifdef SL_DIAG_TRACE
    if (threadIdx.x == 0 && A.dbg) A.dbg[(size_t)b * 8 + 7] = __builtin_amdgcn_s_memtime();   // core-clock counter at the start
#endif
    // The flag words of this workspace are ZERO when the launch starts: sttode_workspace_init zeroed them once (and left SL_MAGIC), the last
    // workgroup of every launch zeroes them again (scene_lat_kernel) -- no memset in front of the launch, and a captured launch replays
    // correctly (round-4 advice: a host-side epoch did not).  A workspace that was never initialised holds arbitrary flags: the trajectory
    // roles then poison their predictions and raise the time-out word (value 2) instead of trusting them.
    const bool inited = __hip_atomic_load(A.magic, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == SL_MAGIC;
    if (b == 0 && threadIdx.x == 0) {   // (the first workgroup of the grid; a time-out is raised milliseconds later)
        __hip_atomic_store(A.tmo, inited ? 0u : 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!inited && R.tmo_host) __hip_atomic_store(R.tmo_host, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (b < 2 * A_tiles) {   // (uniform) per-agent roles
        const int tile = b >> 1;
        if ((b & 1) == 0) {  // E: encoder
            role_frontend(R, A.n, A.Tp, A.ldx, tile, true, false);
            __syncthreads();                          // enc_in / last of this tile are visible to the workgroup
            embed_lat_body(R.ew, R.enc_in, R.last, R.g, R.qkv, A.n, A.Tp, tile, reinterpret_cast<f32x4*>(smem));
            __syncthreads();                          // g / qkv are visible; the LDS region changes hands
            SL_STAMP(1);
            post_attn_body<false, true>(R.pw, R.g, R.qkv + 128, 192, R.pf, A.n, R.ode_time, 0, 1, nullptr, nullptr, tile,
                                        reinterpret_cast<f32x4(*)[4][64]>(smem));
            role_publish(R.flags + tile, tile != R.drop_tile);
            SL_STAMP(2);
            // block-1 layer-1 table of the tile's agents: off the critical path (its readers first run block 0 and the GRU)
            const int col = tile * 16 + c;
            const int colc = col < A.n ? col : A.n - 1;
            f32x4 B[14];
#pragma unroll
            for (int T = 0; T < 8; ++T) B[T] = ld4(R.pf + (size_t)colc * 128 + 16 * T + 4 * q);
#pragma unroll
            for (int T = 8; T < 14; ++T) B[T] = B[0];
            preact_rows<8, true>(R.WA1, R.b11, R.A1y, B, col, col < A.n, lane, q, wave);
            role_publish(A.e2flags + tile, true);
            SL_STAMP(3);
        } else {             // G: block-0 conv + GRU
            f32x4 (*sH)[6][64] = reinterpret_cast<f32x4(*)[6][64]>(smem);
            f32x4* sX = reinterpret_cast<f32x4*>(smem) + 2 * 6 * 64;
            auto fe = [&]() {                         // the front-end runs under the latency of the GRU's weight loads
                role_frontend(R, A.n, A.Tp, A.ldx, tile, false, true);
                __syncthreads();                      // xpad of this tile is visible to the workgroup
                SL_STAMP(1);
            };
            gru_bal_body<TPX, false, true, decltype(fe)>(A.xpad, R.convP, R.convB, R.wihP, R.whhP, R.gbias, R.state0, A.n, A.Tp, tile, sH, sX,
                                                         nullptr, fe);
            role_publish(A.gflags + tile, true);
            SL_STAMP(2);
        }
        return;
    }
    const int tile = (b - 2 * A_tiles) >> 1;
    const bool is_x = (b - 2 * A_tiles) & 1;
    volatile int& s_ok = *reinterpret_cast<volatile int*>(smem + SL_TOTAL);   // behind the roles' regions
    // the tile's columns belong to agents [c_lo / K, c_hi / K]: wait for their 16-agent tiles (one wave polls, the barrier releases the rest)
    const int ncols = A.n * A.K;
    const int c_lo = tile * 16, c_hi = c_lo + 15 < ncols ? c_lo + 15 : ncols - 1;
    const int t_lo = (c_lo / A.K) >> 4, t_hi = (c_hi / A.K) >> 4;
    // layer-1 pre-activations of this tile's columns: bias, then [pf | state0] of each column's agent as B-operand fragments, in two passes
    // (pf arrives first; its 8 k-tiles run while the G role is in its last GRU steps); each pass's first weight fragments are requested
    // before the flags it waits for
    const f32x4* WA = is_x ? R.WAx : R.WAy;
    const float* b1 = is_x ? R.b1x : R.b1y;
    f32x4 B[14], acc[8], w1[8][4], w2[6][4];
    preact_prime<14, 0, 8, 8>(WA, w1, lane, wave);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = ld4(b1 + 16 * (wave + 4 * j) + 4 * q);
    if (threadIdx.x == 0) s_ok = inited ? 1 : 0;
    __syncthreads();
    if (wave == 0 && inited && !wait_tiles(R.flags, t_lo, t_hi, A.tmo, lane, R.tmo_host) && lane == 0) s_ok = 0;   // pf of the tile's agents
    __syncthreads();
    SL_STAMP(1);
    f32x4* sA0 = reinterpret_cast<f32x4*>(smem + SL_A0);
    f32x4* sH1 = reinterpret_cast<f32x4*>(smem + SL_MLP);
    f32x4* sH2 = sH1 + 2 * 4 * 64;
    const int col = tile * 16 + c;
    const int agent = (col < ncols ? col : ncols - 1) / A.K;
    if (s_ok) {
#pragma unroll
        for (int T = 0; T < 8; ++T) B[T] = ld4(R.pf + (size_t)agent * 128 + 16 * T + 4 * q);
    }
    preact_prime<14, 8, 14, 6>(WA, w2, lane, wave);
    if (s_ok) preact_run<14, 0, 8, 8>(WA, w1, acc, B, lane, wave);
    if (wave == 0 && inited && !wait_tiles(A.gflags, t_lo, t_hi, A.tmo, lane, R.tmo_host) && lane == 0) s_ok = 0;   // state0, xpad, cur, orig
    __syncthreads();
    if (s_ok) {
#pragma unroll
        for (int T = 0; T < 6; ++T) B[8 + T] = ld4(R.state0 + (size_t)agent * 96 + 16 * T + 4 * q);
        preact_run<14, 8, 14, 6>(WA, w2, acc, B, lane, wave);
#pragma unroll
        for (int j = 0; j < 8; ++j) sA0[(wave + 4 * j) * 64 + lane] = acc[j];
    }
    SL_STAMP(2);
    if (!is_x) {         // (uniform) Y role
        if (s_ok) {
            MlpLatArgs y0 = A.y0;
            y0.a0_lds = sA0;
            mlp_lat_run<2, NOY, 1, true, true, true>(y0, sH1, sH2, tile);
        }
        // (after a time-out the flag is still published: the X role has seen the same time-out and poisons the tile)
        role_publish(A.yflags + tile, true);
        SL_STAMP(3);
        return;
    }
    if (s_ok) {
        f32x4 (*sH)[6][64] = reinterpret_cast<f32x4(*)[6][64]>(smem + SL_SH);
        f32x4* sX = reinterpret_cast<f32x4*>(smem + SL_GX);
        f32x4* sD = reinterpret_cast<f32x4*>(smem + SL_D);
        MlpLatArgs x0 = A.x0;
        x0.out_lds = sD;
        x0.a0_lds = sA0;
        mlp_lat_run<2, TPX, 0, false, true, true>(x0, sH1, sH2, tile);
        __syncthreads();                               // d tiles are visible to the workgroup
        SL_STAMP(3);
        const int cur = gru_bal_body<TPX, true>(nullptr, A.convP, A.convB, A.wihP, A.whhP, A.gbias, nullptr, ncols, A.Tp, tile, sH, sX, sD);
        SL_STAMP(4);
        if (wave == 0) {   // A1y rows of this tile's agents and y_hat0 of this tile (both producers started long ago)
            const bool ok1 = wait_tiles(A.e2flags, t_lo, t_hi, A.tmo, lane, R.tmo_host);   // (both waits run: each ends with the acquire its data needs)
            const bool ok = wait_tiles(A.yflags, tile, tile, A.tmo, lane, R.tmo_host) && ok1;
            if (!ok && lane == 0) s_ok = 0;
        }
        __syncthreads();
        SL_STAMP(5);
        if (s_ok) {
            MlpLatArgs y1 = A.y1;
            y1.state_lds = &sH[cur][0][0];
            mlp_lat_run<8, NOY, 2, false, true, true>(y1, sH1, sH2, tile);
            SL_STAMP(6);
#ifdef SL_DIAG_TRACE
            if (threadIdx.x == 0 && A.dbg) A.dbg[(size_t)b * 8 + 7] = __builtin_amdgcn_s_memtime() - A.dbg[(size_t)b * 8 + 7];   // core clocks of this workgroup
#endif
            return;
        }
    }
    for (int i = threadIdx.x; i < 16 * A.Tf2; i += blockDim.x) {   // time-out: never hang, never return stale numbers
        const size_t o = (size_t)tile * 16 * A.Tf2 + i;
        if (o < (size_t)ncols * A.Tf2) A.y1.out[o] = __builtin_nanf("");
    }
}

// The launch: every workgroup runs its role (scene_lat_body), then leaves through the exit counter; the LAST one to leave -- nobody polls a flag
// any more -- zeroes the flag words for the next launch on this workspace (the time-out word stays: sttode_check reads it).
template <int TPX, int NOY>
__global__ __launch_bounds__(256) void scene_lat_kernel(SceneLatArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    scene_lat_body<TPX, NOY>(A, smem);
    volatile int& s_last = *reinterpret_cast<volatile int*>(smem + SL_TOTAL + 4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(A.done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u == gridDim.x;
    __syncthreads();
    if (!s_last) return;
    const int A_tiles = A.R.ntiles, total = 3 * A_tiles + 1 + A.ntiles_c;
    for (int i = threadIdx.x; i < total; i += blockDim.x)
        if (i != A_tiles) __hip_atomic_store(A.R.flags + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 0) __hip_atomic_store(A.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#ifdef SL_DIAG_TRACE
static long long* g_scene_dbg = nullptr;
extern "C" int sttode_scene_debug_buffer(void* p) { g_scene_dbg = (long long*)p; return 0; }   // >= grid * 8 int64
#endif

// sttode_workspace_init: the scene form's flag words zeroed, the "initialised" word set (once per workspace; the launches keep them zero)
int stt_scene_flags_init(float* ws, const long* off, int n, int K, void* stream) {
    const int A_tiles = (n + 15) / 16;
    const long C_tiles = ((long)n * K + 15) / 16;
    unsigned* f = (unsigned*)(ws + off[STT_B_FLAGS]) + stt_scene_flags_offset(n);
    const size_t words = (size_t)3 * A_tiles + 1 + C_tiles + 1;   // E | time-out | G | E2 | Y | exit counter
    STT_HIP(hipMemsetAsync(f, 0, words * 4, (hipStream_t)stream));
    STT_HIP(hipMemsetD32Async((hipDeviceptr_t)(f + words), (int)SL_MAGIC, 1, (hipStream_t)stream));
    return 0;
}

bool stt_scene_lat_covers(int Tp, int TPX, int NOY) {
    const bool shape = (TPX == 1 || TPX == 2) && NOY >= 1 && NOY <= 6;   // (round 5: every shape with 2 Tp <= 32, 2 Tf <= 96)
    return shape && Tp >= 2 && 2 * Tp <= 16 * TPX && (Tp * 256 + 512) * 16 <= SL_TOTAL;   // (the E role's embedding region)
}

// W: the model's packed-weight table (STT_W_*); ws / off: workspace and its layout (STT_B_*).  Scene batches with attention length 1 only.
int stt_scene_lat(const float* const* W, float* ws, const long* off, int n, int K, int Tp, int Tf, int TPX, int NOY, int n_chunks0,
                  int n_chunks1, const float* z, float* pred, float ode_time, const float* past, const int* scene_ptr, int S, int drop_tile,
                  unsigned* tmo_host, void* stream) {
    STT_REQUIRE(W && ws && off && z && pred && past && scene_ptr, "stt_scene_lat: null pointer");
    STT_REQUIRE(n > 0 && K > 0 && S > 0 && Tf >= 1 && 2 * Tf <= 16 * NOY && stt_scene_lat_covers(Tp, TPX, NOY), "stt_scene_lat: shape outside the one-launch form");
    STT_REQUIRE(n_chunks0 == 64 + TPX + NOY && n_chunks1 == 32 + NOY, "stt_scene_lat: weight streams do not match (TPX, NOY)");
    SceneLatArgs a;
    RoleArgs& r = a.R;
    role_args_fill(r, W, ws, off);
    r.attn = nullptr; r.ld_attn = 0;
    r.past = past; r.scene_ptr = scene_ptr; r.S = S;
    const int A_tiles = (n + 15) / 16;
    const long ncols = (long)n * K;
    STT_REQUIRE(ncols <= 0x3fffffffL, "stt_scene_lat: too many trajectories");
    const int C_tiles = (int)((ncols + 15) / 16);
    // the scene form's flag words sit BEHIND the fused launch's region of STT_B_FLAGS (5 tiles + 4 words): that form leaves its flags up
    r.flags = (unsigned*)(ws + off[STT_B_FLAGS]) + stt_scene_flags_offset(n);
    r.tmo_host = tmo_host;
    r.ntiles = A_tiles; r.ode_time = ode_time; r.lead = 0; r.drop_tile = drop_tile; r.split = 0; r.gflags = nullptr; r.pflags = nullptr;
    a.tmo = r.flags + A_tiles; a.gflags = a.tmo + 1; a.e2flags = a.gflags + A_tiles; a.yflags = a.e2flags + A_tiles;
    a.done = a.yflags + C_tiles; a.magic = a.done + 1;
    a.dbg = nullptr;
#ifdef SL_DIAG_TRACE
    a.dbg = g_scene_dbg;
#endif
    const float* cur = ws + off[STT_B_CUR];
    const float* orig = ws + off[STT_B_ORIG];
    const float* xpad = ws + off[STT_B_XPAD];
    float* ybuf = ws + off[STT_B_YBUF];
    MlpLatArgs& x0 = a.x0;
    x0.A0 = r.A0x; x0.blob = (const f32x4*)W[STT_W_B0_STREAM]; x0.z = z; x0.state = nullptr; x0.xpad = xpad; x0.ybuf = nullptr; x0.cur = nullptr;
    x0.orig = nullptr; x0.out = ws + off[STT_B_DBUF]; x0.ncols = (int)ncols; x0.K = K; x0.Tf2 = 0; x0.out_lds = nullptr; x0.state_lds = nullptr; x0.a0_lds = nullptr;
    a.y0 = x0;
    a.y0.A0 = r.A0y; a.y0.blob = x0.blob + (size_t)(32 + TPX) * ((2 + 16) * 64); a.y0.out = ybuf;
    a.y1 = x0;
    a.y1.A0 = r.A1y; a.y1.blob = (const f32x4*)W[STT_W_B1_STREAM]; a.y1.xpad = nullptr; a.y1.ybuf = ybuf; a.y1.cur = cur; a.y1.orig = orig;
    a.y1.out = pred; a.y1.Tf2 = 2 * Tf;
    a.convP = (const f32x4*)W[STT_W_B1_CONVP]; a.convB = W[STT_W_B1_CONVB]; a.wihP = (const f32x4*)W[STT_W_B1_WIHP];
    a.whhP = (const f32x4*)W[STT_W_B1_WHHP]; a.gbias = W[STT_W_B1_GBIAS];
    a.xpad = xpad; a.ldx = 16 * TPX; a.n = n; a.K = K; a.Tp = Tp; a.Tf2 = 2 * Tf; a.ntiles_c = C_tiles;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(2 * A_tiles + 2 * C_tiles);
#define SLK(TX, NY)                                                                                   \
    do {                                                                                              \
        STT_SET_LDS_ONCE((scene_lat_kernel<TX, NY>), SL_TOTAL + 16);                                       \
        hipLaunchKernelGGL((scene_lat_kernel<TX, NY>), grid, dim3(256), SL_TOTAL + 16, s, a);              \
    } while (0)
#define SLK_ROW(TX)                                                                                  \
    switch (NOY) {                                                                                   \
        case 1: SLK(TX, 1); break; case 2: SLK(TX, 2); break; case 3: SLK(TX, 3); break;             \
        case 4: SLK(TX, 4); break; case 5: SLK(TX, 5); break; case 6: SLK(TX, 6); break;             \
        default: STT_REQUIRE(false, "stt_scene_lat: unsupported NOY");                               \
    }
    if (TPX == 1) { SLK_ROW(1) } else if (TPX == 2) { SLK_ROW(2) }
    else STT_REQUIRE(false, "stt_scene_lat: unsupported TPX");
#undef SLK_ROW
#undef SLK
    STT_HIP(hipGetLastError());
    return 0;
}
