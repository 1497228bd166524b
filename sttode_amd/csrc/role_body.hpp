// The per-agent ROLE (latency form) shared by the launches with an in-launch hand-off (chain32.hip FUSE = 1: the SERIAL call's roles +
// trajectory groups; scene_lat.hip: roles + 16-column tiles of a single scene) and the tile-flag hand-off between workgroups of ONE launch.
// Pipelined calls use the throughput form of role32.hpp (round 4), which has no hand-off.
#pragma once
#include "latency_bodies.hpp"
#include "frontend_body.hpp"
#include "../../include/sttode_hip.h"

#ifndef C32_TRACE_PHASE
#define C32_TRACE_PHASE(i) do { } while (0)
#endif

// Per-agent ROLE of the fused launch (round 3): the first `ntiles` workgroups of the grid run, for one 16-agent tile each, the whole
// per-agent stage -- encoder (embed_lat_body -> post_attn_body), block-0 conv + GRU (gru_lat4_body) and the three layer-1 pre-activation
// tables (preact_rows) -- and publish ONE flag per tile; the trajectory groups behind them in the grid wait for the flags of the tiles
// their agents live in.  Why: as separate launches on their own stream these kernels were starved by the running chain (its queue keeps
// every freed workgroup slot until its grid is fully dispatched: 1.85 ms for a 0.1 ms stage: measured at the start of round 3, the capture was not kept), which
// forced ONE chain workgroup per CU in the pipelined path; inside the launch nothing needs a chain-free CU.
struct RoleArgs {
    EmbedW ew; PostW pw;
    const float* enc_in; const int* last; float* g; float* qkv; float* pf;
    const f32x4* convP; const float* convB; const f32x4* wihP; const f32x4* whhP; const float* gbias; float* state0;
    const f32x4* WAx; const float* b1x; const f32x4* WAy; const float* b1y; const f32x4* WA1; const float* b11;
    float* A0x; float* A0y; float* A1y;
    // scene front-end inside the role (scene batches; nullptr: the front-end ran as a launch before): set_data's normalisation for the
    // tile's 16 agents -- scene origin (mean of the scene's last observed positions, summed in agent order like scene_orig_kernel),
    // normalised track, velocities, flags -- written to the workspace rows the other phases and the trajectory groups read
    const float* past; const int* scene_ptr; int S; float* scene_orig; int* agent_scene;
    float* enc_in_w; float* xpad_w; float* cur_w; float* orig_w; int* last_w;
    const float* attn; int ld_attn;   // attention output of an EARLIER launch (attention groups > 1, the NBA branch): the role then starts
                                      // at the post-attention layer; nullptr: attention length 1, the role runs the embedding too
    unsigned* flags;     // [ntiles] tile flags + [1] time-out word, zeroed by the launcher before every launch
    unsigned* tmo_host;  // the model's time-out word in pinned host memory (sttode_timeout_word), or nullptr: set (system scope) with the workspace's
    int ntiles; float ode_time;
    int lead;            // grid order: the role of tile t sits `lead` groups ahead of the first group that needs it (fused_block_of)
    int drop_tile;       // fault injection (tests): the role of this tile never publishes its flag (-1: none) -- exercises the give-up path
    // split roles (fused chain launch, default): a tile's per-agent stage as FIVE workgroups instead of one -- E (encoder -> pf) beside
    // G (block-0 conv + GRU -> state0), then P_0 P_1 P_2 (one layer-1 table each) -- so its latency is max(E, G) + one table (~70 us)
    // instead of their sum (~150 us): calls with fewer groups than workgroup slots are bound by exactly that latency.
    int split; unsigned* gflags; unsigned* pflags;   // flags: E [T] | time-out | G [T] | P [3 T]
};

#ifndef ROLE_PRIO
#define ROLE_PRIO 3
#endif

// STTODENet.set_data for ONE 16-agent tile (model/STTODE.py:397-461), one lane per agent: scene origin (mean of the scene's last observed
// positions, summed in agent order like scene_orig_kernel: identical bits), then the agent's inputs.  enc: the encoder's inputs (enc_in,
// last flag, scene_orig, agent_scene); traj: the decoder's (xpad, cur, orig).  The caller's barrier publishes them to the workgroup.
__device__ __forceinline__ void role_frontend(const RoleArgs& R, int nag, int Tp, int ldx, int tile, bool enc, bool traj) {
    if (threadIdx.x < 16) {
        const int a = tile * 16 + (int)threadIdx.x;
        if (a < nag) {
            // (the track is NOT preloaded into a local array handed over by pointer: such an array lives in scratch, and hipcc then waits for
            // every one of its 16 loads in turn before the store -- 16 serial memory round trips at the head of every one-scene call)
            int lo = 0, hi = R.S - 1;                 // the agent's scene: largest s with scene_ptr[s] <= a
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (R.scene_ptr[mid] <= a) lo = mid; else hi = mid - 1;
            }
            const int a0 = R.scene_ptr[lo], a1 = R.scene_ptr[lo + 1];
            float sx = 0.f, sy = 0.f;
            for (int aa = a0; aa < a1; ++aa) {       // agent order, as scene_orig_kernel sums
                sx += R.past[((size_t)aa * Tp + (Tp - 1)) * 2 + 0];
                sy += R.past[((size_t)aa * Tp + (Tp - 1)) * 2 + 1];
            }
            const float inv = (float)(a1 - a0);
            const float ox = sx / inv, oy = sy / inv;
            if (enc) {
                if (a == a0) { R.scene_orig[2 * lo] = ox; R.scene_orig[2 * lo + 1] = oy; }
                R.agent_scene[a] = lo;
            }
            agent_inputs_core<true, 16>(a, R.past, Tp, ldx / 16, 1, ox, oy, a == a1 - 1, nullptr, traj ? R.xpad_w : nullptr, enc ? R.enc_in_w : nullptr,
                                    traj ? R.cur_w : nullptr, traj ? R.orig_w : nullptr, enc ? R.last_w : nullptr);
        }
    }
}

// The per-agent stage of ONE 16-agent tile on a chain workgroup's resources (4 waves, <= 256 VGPRs, the chain's dynamic LDS).  The
// bodies are the stand-alone kernels' code (latency_bodies.hpp), so g / qkv / pf / state0 / A0x / A0y / A1y carry the bits the separate
// launches produce.  LDS: [0, 40 KiB) embed, then [0, 16 KiB) post-attention exchange, then [0, 12 KiB) h tiles + [12, 60 KiB) GRU image + [60, 68 KiB) gate hand-off.
// nag: agents; Tp: observed frames; ldx: row stride of xpad (16 or 32)
__device__ __forceinline__ void agent_role(const RoleArgs& R, int nag, int Tp, int ldx, const float* __restrict__ xpad, int tile, char* smem) {
    // The role is a short chain of DEPENDENT steps (barriers, L2 round trips, 32-cycle MFMAs) sharing each SIMD with a chain wave that has
    // a 64-cycle MFMA ready every cycle it is asked: at equal priority the older chain wave wins every arbitration and the role ran 2x
    // slower than alone (283 vs 150 us, profiles/r03/trace_*), holding a workgroup slot all the while.  Raised priority lets its few
    // instructions issue first; the chain wave loses the same handful of pipe cycles either way.
    __builtin_amdgcn_s_setprio(ROLE_PRIO);
    if (R.past) {   // (uniform) STTODENet.set_data for this tile
        role_frontend(R, nag, Tp, ldx, tile, true, true);
        __syncthreads();                                  // enc_in / last / xpad of this tile are visible to the workgroup
    }
    if (R.attn == nullptr) {                      // (uniform) attention length 1: softmax over one key == 1, the attention output is v
        embed_lat_body(R.ew, R.enc_in, R.last, R.g, R.qkv, nag, Tp, tile, reinterpret_cast<f32x4*>(smem));
        __syncthreads();                          // g / qkv of this tile are visible to the workgroup; the LDS region changes hands
    }
    C32_TRACE_PHASE(0);
    post_attn_body<false>(R.pw, R.g, R.attn ? R.attn : R.qkv + 128, R.attn ? R.ld_attn : 192, R.pf, nag, R.ode_time, 0, 1, nullptr, nullptr, tile,
                          reinterpret_cast<f32x4(*)[4][64]>(smem));
    __syncthreads();                              // pf of this tile is visible to the workgroup; LDS changes hands again
    C32_TRACE_PHASE(1);
    f32x4 (*sH)[6][64] = reinterpret_cast<f32x4(*)[6][64]>(smem);
    f32x4* sW45 = reinterpret_cast<f32x4*>(smem) + 2 * 6 * 64;
    const int cur = ldx == 16 ? gru_lat4_body<1>(xpad, R.convP, R.convB, R.wihP, R.whhP, R.gbias, R.state0, nag, Tp, tile, sH, sW45)
                                : gru_lat4_body<2>(xpad, R.convP, R.convB, R.wihP, R.whhP, R.gbias, R.state0, nag, Tp, tile, sH, sW45);
    C32_TRACE_PHASE(2);
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = tile * 16 + c;
    const int colc = col < nag ? col : nag - 1;
    f32x4 B[14];                                  // [pf | state0] of this lane's agent as B-operand fragments
#pragma unroll
    for (int T = 0; T < 8; ++T) B[T] = ld4(R.pf + (size_t)colc * 128 + 16 * T + 4 * q);
#pragma unroll
    for (int T = 0; T < 6; ++T) B[8 + T] = sH[cur][T][lane];
    preact_rows<14, true>(R.WAx, R.b1x, R.A0x, B, col, col < nag, lane, q, wv);
    preact_rows<14, true>(R.WAy, R.b1y, R.A0y, B, col, col < nag, lane, q, wv);
    preact_rows<8, true>(R.WA1, R.b11, R.A1y, B, col, col < nag, lane, q, wv);
    // publish (guide §6 G16 R1): every storing wave drains its sc1 stores, the workgroup meets, ONE lane stores the flag (agent scope)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && tile != R.drop_tile) __hip_atomic_store(R.flags + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Consumer side: wave 0 polls the flags of tiles [t_lo, t_hi] (relaxed agent-scope loads, one lane per tile, s_sleep between polls), then
// ONE agent-scope acquire drops this CU's stale L1 lines; the caller's barrier releases the other waves.  The spin is bounded (~1 s): a
// producer that never arrives -- it cannot, in-order dispatch puts every producer in front of its consumers -- would poison this group's
// predictions with NaN and set the time-out word instead of hanging the device.
// A flag is up when it is non-zero: the flag words are zero when a launch starts (fused launch: zeroed by the launcher in front of every
// launch; one-launch scene form: zeroed once by sttode_workspace_init and again by the LAST workgroup of every launch, scene_lat.hip).
// tmo_host (optional): the model's pinned host word, set with system scope beside the workspace's time-out word -- the host sees a give-up
// without synchronising or copying anything (sttode_timeout_word).
__device__ __forceinline__ bool wait_tiles(unsigned* flags, int t_lo, int t_hi, unsigned* tmo, int lane, unsigned* tmo_host = nullptr) {
    bool ok = true;
    for (int t = t_lo + lane; t <= t_hi; t += 64) {
        unsigned spins = 0;
        while (true) {
            const unsigned v = __hip_atomic_load(flags + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v != 0u) break;
            __builtin_amdgcn_s_sleep(32);
            if (++spins > (1u << 20)) { ok = false; break; }
        }
    }
    ok = __all(ok);
    if (!ok && lane == 0) {
        __hip_atomic_store(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tmo_host) __hip_atomic_store(tmo_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return ok;
}

// publish (guide section 6 G16 R1): every storing wave drains its sc1 stores, the workgroup meets, ONE lane stores the flag (agent scope)
__device__ __forceinline__ void role_publish(unsigned* flag, bool really, unsigned val = 1u) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && really) __hip_atomic_store(flag, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Split roles of the fused chain launch (RoleArgs::split).  Block r < 2 T: tile r / 2, E (even) or G (odd); 2 T <= r < 5 T: table
// (r - 2 T) % 3 of tile (r - 2 T) / 3.  Same bodies, same sums as agent_role: the tables carry the same bits.
__device__ __forceinline__ void split_role(const RoleArgs& R, int nag, int Tp, int ldx, const float* __restrict__ xpad, int r, char* smem) {
    __builtin_amdgcn_s_setprio(ROLE_PRIO);
    const int T = R.ntiles;
    if (r < 2 * T) {
        const int tile = r >> 1;
        if ((r & 1) == 0) {   // E: encoder
            if (R.past) {
                role_frontend(R, nag, Tp, ldx, tile, true, false);
                __syncthreads();
            }
            if (R.attn == nullptr) {
                embed_lat_body(R.ew, R.enc_in, R.last, R.g, R.qkv, nag, Tp, tile, reinterpret_cast<f32x4*>(smem));
                __syncthreads();
            }
            C32_TRACE_PHASE(0);
            post_attn_body<false, true>(R.pw, R.g, R.attn ? R.attn : R.qkv + 128, R.attn ? R.ld_attn : 192, R.pf, nag, R.ode_time, 0, 1, nullptr,
                                        nullptr, tile, reinterpret_cast<f32x4(*)[4][64]>(smem));
            C32_TRACE_PHASE(1);
            role_publish(R.flags + tile, true);
        } else {              // G: block-0 conv + GRU
            if (R.past) {
                role_frontend(R, nag, Tp, ldx, tile, false, true);
                __syncthreads();
            }
            f32x4 (*sH)[6][64] = reinterpret_cast<f32x4(*)[6][64]>(smem);
            f32x4* sW45 = reinterpret_cast<f32x4*>(smem) + 2 * 6 * 64;
            if (ldx == 16) gru_lat4_body<1, false, true>(xpad, R.convP, R.convB, R.wihP, R.whhP, R.gbias, R.state0, nag, Tp, tile, sH, sW45);
            else gru_lat4_body<2, false, true>(xpad, R.convP, R.convB, R.wihP, R.whhP, R.gbias, R.state0, nag, Tp, tile, sH, sW45);
            C32_TRACE_PHASE(2);
            role_publish(R.gflags + tile, true);
        }
        return;
    }
    const int tile = (r - 2 * T) / 3, k = (r - 2 * T) % 3;
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int* ok = reinterpret_cast<int*>(smem);
    if (threadIdx.x == 0) *ok = 1;
    __syncthreads();
    if (wv == 0) {
        bool o = wait_tiles(R.flags, tile, tile, R.flags + T, lane);
        if (k < 2) o = wait_tiles(R.gflags, tile, tile, R.flags + T, lane) && o;
        if (!o && lane == 0) *ok = 0;
    }
    __syncthreads();
    if (!*ok) return;         // (uniform) time-out: the flag stays down, the tile's groups give up in turn
    const int col = tile * 16 + c;
    const int colc = col < nag ? col : nag - 1;
    f32x4 B[14];
#pragma unroll
    for (int Tt = 0; Tt < 8; ++Tt) B[Tt] = ld4(R.pf + (size_t)colc * 128 + 16 * Tt + 4 * q);
#pragma unroll
    for (int Tt = 0; Tt < 6; ++Tt) B[8 + Tt] = k < 2 ? ld4(R.state0 + (size_t)colc * 96 + 16 * Tt + 4 * q) : B[0];
    if (k == 0) preact_rows<14, true>(R.WAx, R.b1x, R.A0x, B, col, col < nag, lane, q, wv);
    else if (k == 1) preact_rows<14, true>(R.WAy, R.b1y, R.A0y, B, col, col < nag, lane, q, wv);
    else preact_rows<8, true>(R.WA1, R.b11, R.A1y, B, col, col < nag, lane, q, wv);
    role_publish(R.pflags + 3 * tile + k, tile != R.drop_tile);
}

// Host side: the weight fragments (model table W, STT_W_*) and workspace rows (ws + off[STT_B_*]) every launch that carries per-agent
// roles hands to them; the caller sets what differs (attention input, scene front-end inputs, flags, fault injection).
static inline void role_args_fill(RoleArgs& r, const float* const* W, float* ws, const long* off) {
    r.tmo_host = nullptr;
    r.ew.fc1P = W[STT_W_FC1P]; r.ew.fc1b = W[STT_W_FC1B]; r.ew.posP = (const f32x4*)W[STT_W_POSP]; r.ew.peb = W[STT_W_PEB];
    r.ew.fc2P = (const f32x4*)W[STT_W_FC2P]; r.ew.fc2b = W[STT_W_FC2B]; r.ew.fc3P = (const f32x4*)W[STT_W_FC3P]; r.ew.fc3b = W[STT_W_FC3B];
    r.ew.fc3last = W[STT_W_FC3LAST]; r.ew.inP = (const f32x4*)W[STT_W_INP]; r.ew.inb = W[STT_W_INB];
    r.pw.outP = (const f32x4*)W[STT_W_OUTP]; r.pw.outb = W[STT_W_OUTB]; r.pw.infoP = (const f32x4*)W[STT_W_INFOP]; r.pw.infob = W[STT_W_INFOB];
    r.pw.gateP = (const f32x4*)W[STT_W_GATEP]; r.pw.gateb = W[STT_W_GATEB]; r.pw.ln1w = W[STT_W_LN1W]; r.pw.ln1b = W[STT_W_LN1B];
    r.pw.l1P = (const f32x4*)W[STT_W_L1P]; r.pw.l1b = W[STT_W_L1B]; r.pw.l2P = (const f32x4*)W[STT_W_L2P]; r.pw.l2b = W[STT_W_L2B];
    r.pw.ln2w = W[STT_W_LN2W]; r.pw.ln2b = W[STT_W_LN2B];
    r.enc_in = ws + off[STT_B_ENC_IN]; r.last = (const int*)(ws + off[STT_B_LAST]); r.g = ws + off[STT_B_G]; r.qkv = ws + off[STT_B_QKV];
    r.pf = ws + off[STT_B_PF];
    r.convP = (const f32x4*)W[STT_W_B0_CONVP]; r.convB = W[STT_W_B0_CONVB]; r.wihP = (const f32x4*)W[STT_W_B0_WIHP];
    r.whhP = (const f32x4*)W[STT_W_B0_WHHP]; r.gbias = W[STT_W_B0_GBIAS]; r.state0 = ws + off[STT_B_STATE0];
    r.WAx = (const f32x4*)W[STT_W_B0_XWA]; r.b1x = W[STT_W_B0_XB1]; r.WAy = (const f32x4*)W[STT_W_B0_YWA]; r.b1y = W[STT_W_B0_YB1];
    r.WA1 = (const f32x4*)W[STT_W_B1_YWA]; r.b11 = W[STT_W_B1_YB1];
    r.A0x = ws + off[STT_B_A0X]; r.A0y = ws + off[STT_B_A0Y]; r.A1y = ws + off[STT_B_A1Y];
    r.scene_orig = ws + off[STT_B_SCENE_ORIG]; r.agent_scene = (int*)(ws + off[STT_B_AGENT_SCENE]);
    r.enc_in_w = ws + off[STT_B_ENC_IN]; r.xpad_w = ws + off[STT_B_XPAD]; r.cur_w = ws + off[STT_B_CUR]; r.orig_w = ws + off[STT_B_ORIG];
    r.last_w = (int*)(ws + off[STT_B_LAST]);
}
