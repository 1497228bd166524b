// Native forward pipeline: ONE C-ABI call enqueues the whole inference() hot path
// (STTODENet.inference, model/STTODE.py:574-623) on the caller's stream.
//
//   front-end -> embed_qkv -> [mhgsa_attn if attention length > 1] -> post_attn            (main stream)
//            \-> gru_cols(block 0, per agent)                                               (side stream, overlapped)
//   -> linear_cols x3 (per-agent layer-1 pre-activations) -> mlp_block0 -> gru_cols(block 1) -> mlp_block1
//
// No allocation, no synchronisation: every intermediate lives in a caller-provided workspace whose layout is
// returned by sttode_workspace_layout().  With timing enabled each stage is bracketed by hipEvents recorded on
// the stream it runs on; sttode_timing_read() returns the accumulated per-stage durations (bench.py's roofline).
#include "api_util.hpp"
#include "../../include/sttode_hip.h"
#include <vector>
#include <algorithm>
#include <utility>
#include <cstring>
#include <mutex>

#include <cstdlib>
#define STT_MAX_PARTS 8
#define STT_MAX_DEVICES 64
#define STT_MAX_SLOTS 8   // workspace / prediction slots of the cross-call pipeline (sttode_inference_*_async)
struct TimRec { int stage; hipEvent_t e0, e1; };

// LAGGED form: a call whose per-agent roles have been enqueued (in the launch it made) but whose trajectory groups have not yet
struct LagPending { bool valid; float* ws; long off[STT_B_COUNT]; int n; const float* z; float* pred; hipStream_t s; int si;
                    const float* gt; float* ade; float* fde; float scale;
                    int one_launch; };   // the call was ONE launch (scene batch, front-end in the roles): its groups may run as workers   // gt != nullptr: fused metrics of this call

struct SttodeModel {
    int Tp, Tf, TPX, NOY, K;
    const float* w[STT_W_COUNT];
    int n_chunks0, n_chunks1;
    hipStream_t side;
    hipEvent_t ev_fork, ev_join;
    // column-part pipelining of the per-trajectory kernels: part p runs mlp_block0 -> gru_cols -> mlp_block1 on its own
    // stream, so the grid tail of one part's kernel is filled by the next part's kernel (columns are independent).
    int col_parts;
    int chain_mode;  // 1 fused chain kernel, 0 three-kernel form, -1 automatic
    int fused_mode;  // 1 per-agent roles inside the chain launch wherever the shape is covered, 0 separate per-agent launches
    int role_lead;   // fused launch's grid order: groups of head start of a role over its first consumer, < 0 = all roles first (default)
    hipStream_t slot_stream[STT_MAX_SLOTS];   // the pipeline stream the slot's latest asynchronous call ran on (its follow-up work goes there)
    int scene_launch; // largest number of 16-trajectory tiles a serial scene call runs as ONE launch (scene_lat.hip); 0: never
    int drop_tile;   // fault injection (tests): the role of this 16-agent tile does not publish its flag in fused launches (-1: none)
    int b3;          // exploratory: block-0 MLPs of the fused launch as a three-way bf16 split (sttode_set_mfma_mode)
    bool fe_in_role; // fused scene batches: the roles also run the scene front-end (STTODE_FE_IN_ROLE=1; default: a launch in front)
    int ode_method, ode_steps;  // integrator of the encoder ODE (0, 1 = one Euler step = the reference)
    int prog_len;
    hipStream_t part_stream[STT_MAX_PARTS];
    hipEvent_t ev_agents, ev_part[STT_MAX_PARTS];
    // cross-call software pipeline (sttode_inference_*_async): stage A (per agent) of call i+1 runs on sA beside stage B
    // (per trajectory) of call i on sB; two workspace slots alternate.
    hipStream_t sA, sB, sB2;
    std::mutex mu;                   // serialises the asynchronous entry points of ONE model (slot tables, lag queues, stream rotation)
    unsigned* tmo_host;              // pinned host word: set by any launch of this model whose in-launch hand-off gave up (sttode_timeout_word)
    int lag_streams;                 // 0: lagged form off; 2 / 3 (default): pipeline streams the lagged calls rotate over
    long lag_calls;
    LagPending lag[STT_MAX_SLOTS];   // per slot
    int lag_q[4][STT_MAX_SLOTS]; int lag_qn[4];   // per stream: slots with outstanding groups, oldest first
    hipStream_t sX[3];   // extra streams of the fused rotation (STTODE_FUSED_STREAMS = 4..6; experiments: they share the runtime's hardware queues)
    int fused_streams;
    int b_streams;  // 1: all per-trajectory stages on sB; 2: alternate calls between sB and sB2
    long acalls;
    hipEvent_t ev_call, evA_done[STT_MAX_SLOTS], evB_done[STT_MAX_SLOTS];
    bool timing;        // brackets active for the CURRENT call
    int timing_every;   // 0 = off, n = bracket every n-th forward call
    long calls;
    std::vector<TimRec> recs;
    std::vector<hipEvent_t> pool;
};

bool stt_embed_qkv_fe_covers(int n, int Tlen);
int stt_embed_qkv_fe(const float* const* W, const float* past, int n, int N, int Tlen, int TPX, float* xpad, float* enc_in, float* cur,
                     float* orig, int* last, float* g, float* qkv, void* stream);

int stt_post_attn_stage(const float* const* W, const float* state, const float* attn, int ld_attn, const float* base, const float* kA, float cA,
                        const float* kB, float cB, const float* kC, float cC, float cN, float* kout, float* out, float* qkv, const float* g,
                        float* pf, int n, void* stream);

static std::mutex g_stream_mu;   // guards the creation of the process-wide streams (sA / sB / sB2 / side, per device)

// The models' host-visible time-out words come from ONE process-wide pinned block (allocated with the first model, never freed): a model
// may be destroyed by a garbage collector at any moment -- also while some stream of the process is being captured into a hipGraph, where
// hipHostFree (a synchronising call) aborts the process.  Words are handed out from a free list.
#define STT_TMO_WORDS 4096
static unsigned* g_tmo_pool = nullptr;
static std::vector<int> g_tmo_free;
static unsigned* tmo_word_take() {
    std::lock_guard<std::mutex> lk(g_stream_mu);
    if (!g_tmo_pool) {
        if (hipHostMalloc((void**)&g_tmo_pool, STT_TMO_WORDS * sizeof(unsigned), hipHostMallocDefault) != hipSuccess) { g_tmo_pool = nullptr; return nullptr; }
        memset(g_tmo_pool, 0, STT_TMO_WORDS * sizeof(unsigned));
        for (int i = STT_TMO_WORDS - 1; i >= 0; --i) g_tmo_free.push_back(i);
    }
    if (g_tmo_free.empty()) return nullptr;
    const int i = g_tmo_free.back();
    g_tmo_free.pop_back();
    g_tmo_pool[i] = 0u;
    return g_tmo_pool + i;
}
static void tmo_word_give(unsigned* w) {
    if (!w || !g_tmo_pool) return;
    std::lock_guard<std::mutex> lk(g_stream_mu);
    g_tmo_free.push_back((int)(w - g_tmo_pool));
}

static inline size_t al(size_t x) { return (x + 63) & ~(size_t)63; }  // 256-byte alignment in floats

extern "C" int sttode_model_create(SttodeModel** out, const void* const* weights, int count, int Tp, int Tf, int K,
                                   int n_chunks0, int n_chunks1) {
    STT_REQUIRE(out && weights, "sttode_model_create: null pointer");
    STT_REQUIRE(count == STT_W_COUNT, "sttode_model_create: weight table must have STT_W_COUNT entries");
    STT_REQUIRE(Tp >= 2 && 2 * Tp <= 32 && Tf >= 1 && K >= 1, "sttode_model_create: bad Tp/Tf/K");
    for (int i = 0; i < count; ++i)   // (the exploratory bf16-split stream may arrive later: sttode_model_set_weight)
        STT_REQUIRE(weights[i] != nullptr || i == STT_W_CHAINB3_POOL || i == STT_W_CHAINB3_PROG, "sttode_model_create: null weight pointer");
    SttodeModel* m = new SttodeModel();
    m->Tp = Tp; m->Tf = Tf; m->K = K;
    m->TPX = (2 * Tp <= 16) ? 1 : 2;
    m->NOY = (2 * Tf + 15) / 16;
    for (int i = 0; i < count; ++i) m->w[i] = (const float*)weights[i];
    m->n_chunks0 = n_chunks0; m->n_chunks1 = n_chunks1;
    m->timing = false;
    m->timing_every = 0;
    m->calls = 0;
    m->b_streams = 2;
    m->acalls = 0;
    if (const char* e = getenv("STTODE_B_STREAMS")) m->b_streams = atoi(e) == 2 ? 2 : 1;
    m->chain_mode = -1;
    m->ode_method = 0; m->ode_steps = 1;
    if (const char* e = getenv("STTODE_CHAIN")) m->chain_mode = atoi(e) > 0 ? 1 : atoi(e) == 0 ? 0 : -1;
    m->fused_mode = 1;
    m->b3 = 0;   // (STTODE_BF16X3=1 is honoured by the Python layer, which packs the bf16-split stream before switching the mode on)
    m->role_lead = getenv("STTODE_ROLE_LEAD") ? atoi(getenv("STTODE_ROLE_LEAD")) : -1;   // -1: one role workgroup per tile, all in front (default); -2: split roles
    m->drop_tile = -1;
    for (int p = 0; p < STT_MAX_SLOTS; ++p) { m->slot_stream[p] = nullptr; m->lag[p].valid = false; }
    // 3 streams: the small legs gain 3-9 % over 2, 512 scenes tie (profiles/r04/streams_2_vs_3.txt); 4: -5..-12 % everywhere -- the fourth
    // shares a hardware queue (streams_3_vs_4.txt)
    m->lag_streams = 3;
    m->lag_calls = 0;
    m->tmo_host = nullptr;
    for (int i = 0; i < 4; ++i) m->lag_qn[i] = 0;
    if (const char* e = getenv("STTODE_LAGGED")) m->lag_streams = atoi(e) >= 2 && atoi(e) <= 4 ? atoi(e) : 0;
    m->scene_launch = 128;
    if (const char* e = getenv("STTODE_SCENE_LAUNCH")) m->scene_launch = atoi(e) > 0 ? atoi(e) : 0;
    if (const char* e = getenv("STTODE_FUSED")) m->fused_mode = atoi(e) != 0;
    // default off: measured neutral to -0.6 % pipelined and -1.5 % serial at 512 scenes (the two front-end launches cost less than the
    // ~10 us they add to every role), +1 % on the 256-scene SDD leg (profiles/r03/ab_lead_frontend_depth.txt)
    m->fe_in_role = getenv("STTODE_FE_IN_ROLE") && atoi(getenv("STTODE_FE_IN_ROLE")) != 0;
    m->prog_len = sttode_chain_prog_len(Tp, Tf);
    m->col_parts = 1;  // measured on MI355X: 1 -> 62.1, 2 -> 60.6, 4 -> 55.4 M traj/s (kernels of different streams do not fill each other's tails)
    if (const char* e = getenv("STTODE_COL_PARTS")) m->col_parts = atoi(e);
    if (m->col_parts < 1) m->col_parts = 1;
    if (m->col_parts > STT_MAX_PARTS) m->col_parts = STT_MAX_PARTS;
    // All internal streams run at normal priority.  Measured on MI355X: high priority for the per-agent streams makes the
    // cross-call pipeline slower (61.6 M traj/s with sA+side raised, 65.6 M with only side raised, 66.9 M with neither).
    // STTODE_A_PRIORITY=1: the per-agent streams (sA, side) get the highest stream priority, so their small kernels are dispatched
    // first whenever a running per-trajectory kernel frees workgroup slots
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    const bool a_prio = getenv("STTODE_A_PRIORITY") && atoi(getenv("STTODE_A_PRIORITY")) > 0;
    auto mk_stream = [&](hipStream_t* st, bool high) {
        return (high ? hipStreamCreateWithPriority(st, hipStreamNonBlocking, prio_hi) : hipStreamCreateWithFlags(st, hipStreamNonBlocking)) == hipSuccess;
    };
    m->side = nullptr;   // created on the first serial call
    bool ok =              hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&m->ev_agents, hipEventDisableTiming) == hipSuccess;
    for (int p = 0; p < STT_MAX_PARTS && ok; ++p) {
        m->part_stream[p] = nullptr;
        ok = hipEventCreateWithFlags(&m->ev_part[p], hipEventDisableTiming) == hipSuccess &&
             true;   // part streams are created on first use (every stream takes a share of the 4 hardware queues)
    }
    // The pipeline's streams are PROCESS-WIDE (created with the first model, shared by every later one, never destroyed).  The runtime
    // deals its few hardware queues (4) to streams round-robin at creation: the first model's three streams get the three queues the
    // default stream does not use, but the streams of a second model -- or of one created after an earlier model was destroyed -- start
    // wherever the counter stands, and a per-agent or chain stream that shares the caller's queue serialises the pipeline (measured: the
    // same leg of the bench at 51 or 63 M trajectories/s depending on how many models had been created before it).
    static hipStream_t g_sA[STT_MAX_DEVICES] = {}, g_sB[STT_MAX_DEVICES] = {}, g_sB2[STT_MAX_DEVICES] = {};   // per device of this process
    int dev = 0;
    ok = ok && hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < STT_MAX_DEVICES;
    if (ok) {
        std::lock_guard<std::mutex> lk(g_stream_mu);   // models may be created from several host threads
        if (!g_sA[dev]) {
            ok = mk_stream(&g_sA[dev], a_prio) && hipStreamCreateWithFlags(&g_sB[dev], hipStreamNonBlocking) == hipSuccess &&
                 hipStreamCreateWithFlags(&g_sB2[dev], hipStreamNonBlocking) == hipSuccess;
            if (!ok) g_sA[dev] = nullptr;
        }
        if (ok) { m->sA = g_sA[dev]; m->sB = g_sB[dev]; m->sB2 = g_sB2[dev]; }
        m->fused_streams = 1;
        if (const char* e = getenv("STTODE_FUSED_STREAMS")) m->fused_streams = atoi(e) < 1 ? 1 : atoi(e) > 6 ? 6 : atoi(e);
        static hipStream_t g_sX[STT_MAX_DEVICES][3] = {};
        for (int i = 0; ok && i < (m->fused_streams > 4 ? m->fused_streams - 3 : 1); ++i) {   // (sX[0]: also the fourth stream of the lagged rotation)
            if (!g_sX[dev][i]) ok = hipStreamCreateWithFlags(&g_sX[dev][i], hipStreamNonBlocking) == hipSuccess;
            m->sX[i] = g_sX[dev][i];
        }
    }
    ok = ok && hipEventCreateWithFlags(&m->ev_call, hipEventDisableTiming) == hipSuccess;
    // the model's time-out word lives in pinned host memory (device-visible): a group that gives up stores to it with system scope, the
    // host reads it without any synchronisation (sttode_timeout_word)
    if (ok) { m->tmo_host = tmo_word_take(); ok = m->tmo_host != nullptr; }
    for (int p = 0; p < STT_MAX_SLOTS && ok; ++p)
        ok = hipEventCreateWithFlags(&m->evA_done[p], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&m->evB_done[p], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        tmo_word_give(m->tmo_host);
        delete m;
        stt_set_error("sttode_model_create: could not create streams / events");
        return 2;
    }
    *out = m;
    return 0;
}

// hand over (or replace) one entry of the packed-weight table after creation: the opt-in bf16-split stream is packed on first use
extern "C" int sttode_model_set_weight(SttodeModel* m, int index, const void* ptr) {
    STT_REQUIRE(m && ptr && index >= 0 && index < STT_W_COUNT, "sttode_model_set_weight: bad arguments");
    m->w[index] = (const float*)ptr;
    return 0;
}

extern "C" int sttode_model_destroy(SttodeModel* m) {
    if (!m) return 0;
    // A destructor runs whenever the host language's collector decides -- possibly while some stream of the process is being captured
    // (global capture mode: stream / event destruction from any thread is then refused, and the capture invalidated).  Relaxed mode for
    // the duration of the clean-up: these objects belong to no capture.
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    const bool exchanged = hipThreadExchangeStreamCaptureMode(&mode) == hipSuccess;
    if (!exchanged) (void)hipGetLastError();
    for (auto& r : m->recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    for (auto e : m->pool) (void)hipEventDestroy(e);
    (void)hipEventDestroy(m->ev_fork); (void)hipEventDestroy(m->ev_join); (void)hipEventDestroy(m->ev_agents);
    for (int p = 0; p < STT_MAX_PARTS; ++p) { (void)hipEventDestroy(m->ev_part[p]); if (p && m->part_stream[p]) (void)hipStreamDestroy(m->part_stream[p]); }
    // sA / sB / sB2 are process-wide (sttode_model_create)
    (void)hipEventDestroy(m->ev_call);
    for (int p = 0; p < STT_MAX_SLOTS; ++p) { (void)hipEventDestroy(m->evA_done[p]); (void)hipEventDestroy(m->evB_done[p]); }
    tmo_word_give(m->tmo_host);   // (back to the process-wide block: no hipHostFree here -- a destructor may run during a stream capture)
    delete m;
    if (exchanged) (void)hipThreadExchangeStreamCaptureMode(&mode);   // back to the caller's mode
    return 0;
}

extern "C" int sttode_workspace_layout(const SttodeModel* m, int n, int S, long* offsets, long* total_floats) {
    STT_REQUIRE(m && offsets && total_floats, "sttode_workspace_layout: null pointer");
    STT_REQUIRE(n > 0 && S >= 0, "sttode_workspace_layout: bad n/S");
    const size_t mm = (size_t)n * m->K;
    size_t o = 0;
    auto put = [&](int id, size_t cnt) { offsets[id] = (long)o; o += al(cnt); };
    put(STT_B_SCENE_ORIG, (size_t)(S > 0 ? S : 1) * 2);
    put(STT_B_AGENT_SCENE, n);
    put(STT_B_XPAD, (size_t)n * 16 * m->TPX);
    put(STT_B_ENC_IN, (size_t)n * m->Tp * 4);
    put(STT_B_CUR, (size_t)n * 2);
    put(STT_B_ORIG, (size_t)n * 2);
    put(STT_B_LAST, n);
    put(STT_B_G, (size_t)n * 64);
    put(STT_B_QKV, (size_t)n * 192);
    put(STT_B_ATTN, (size_t)n * 64);
    put(STT_B_PF, (size_t)n * 128);
    put(STT_B_STATE0, (size_t)n * 96);
    put(STT_B_A0X, (size_t)n * 512);
    put(STT_B_A0Y, (size_t)n * 512);
    put(STT_B_A1Y, (size_t)n * 512);
    put(STT_B_DBUF, mm * 16 * m->TPX);
    put(STT_B_YBUF, mm * 16 * m->NOY);
    put(STT_B_STATE1, mm * 96);
    put(STT_B_QUEUE, 64);
    // fused launches: five flags per 16-agent tile (E, G, three tables) + the time-out word, zeroed per call; behind them the one-launch scene
    // form's: three per agent tile + time-out word + one per 16-trajectory tile + exit counter + initialised word (stt_scene_flags_offset)
    // (+ the one-launch scene form's exit counter and "initialised" word: sttode_workspace_init)
    put(STT_B_FLAGS, (size_t)8 * ((n + 15) / 16) + 8 + (mm + 15) / 16);
    put(STT_B_ODE, (size_t)6 * n * 64);   // multi-stage integrator with attention groups > 1 (always laid out: sttode_set_ode may come later)
    *total_floats = (long)o;
    return 0;
}

extern "C" int sttode_set_chain(SttodeModel* m, int mode) {
    STT_REQUIRE(m, "sttode_set_chain: null model");
    STT_REQUIRE(mode >= -1 && mode <= 1, "sttode_set_chain: mode must be -1 (auto), 0 or 1");
    m->chain_mode = mode;
    return 0;
}

extern "C" int sttode_set_fused(SttodeModel* m, int mode) {
    STT_REQUIRE(m, "sttode_set_fused: null model");
    STT_REQUIRE(mode >= 0 && mode <= 4, "sttode_set_fused: mode must be 0 .. 4");
    m->fused_mode = mode ? 1 : 0;
    m->fe_in_role = mode == 2;
    m->role_lead = mode == 3 ? 160 : mode == 4 ? -2 : -1;
    return 0;
}

static int lag_flush_all(SttodeModel* m);
extern "C" int sttode_set_lagged(SttodeModel* m, int streams) {
    STT_REQUIRE(m, "sttode_set_lagged: null model");
    STT_REQUIRE(streams == 0 || (streams >= 2 && streams <= 4), "sttode_set_lagged: streams must be 0 (off), 2, 3 or 4");
    std::lock_guard<std::mutex> lk(m->mu);
    if (int rc = lag_flush_all(m)) return rc;   // outstanding groups belong to the old rotation
    m->lag_streams = streams;
    m->lag_calls = 0;
    return 0;
}

extern "C" int sttode_debug_drop_role_flag(SttodeModel* m, int tile) {
    STT_REQUIRE(m && tile >= -1, "sttode_debug_drop_role_flag: bad arguments");
    m->drop_tile = tile;
    return 0;
}

extern "C" int sttode_set_scene_launch(SttodeModel* m, int max_tiles) {
    STT_REQUIRE(m, "sttode_set_scene_launch: null model");
    STT_REQUIRE(max_tiles >= -1, "sttode_set_scene_launch: max_tiles must be -1 (default), 0 (off) or a tile count");
    m->scene_launch = max_tiles < 0 ? 128 : max_tiles;
    return 0;
}

extern "C" int sttode_set_mfma_mode(SttodeModel* m, int mode) {
    STT_REQUIRE(m, "sttode_set_mfma_mode: null model");
    STT_REQUIRE(mode == 0 || mode == 1, "sttode_set_mfma_mode: mode must be 0 (fp32) or 1 (three-way bf16 split, exploratory)");
    STT_REQUIRE(mode == 0 || (m->w[STT_W_CHAINB3_POOL] && m->w[STT_W_CHAINB3_PROG]),
                "sttode_set_mfma_mode: the bf16-split weight stream has not been handed over (sttode_model_set_weight)");
    m->b3 = mode;
    return 0;
}

extern "C" int sttode_set_ode(SttodeModel* m, int method, int steps) {
    STT_REQUIRE(m, "sttode_set_ode: null model");
    STT_REQUIRE(method >= 0 && method <= 2 && steps >= 1 && steps <= 1024, "sttode_set_ode: method in {0,1,2}, 1 <= steps <= 1024");
    m->ode_method = method; m->ode_steps = steps;
    return 0;
}

extern "C" int sttode_set_col_parts(SttodeModel* m, int parts) {
    STT_REQUIRE(m, "sttode_set_col_parts: null model");
    STT_REQUIRE(parts >= 1 && parts <= STT_MAX_PARTS, "sttode_set_col_parts: parts must be in [1, 8]");
    m->col_parts = parts;
    return 0;
}

// every = 0: off; every = n > 0: the stages of every n-th forward call are bracketed by hipEvents (each bracket costs a few
// microseconds of queue serialisation, so long runs sample instead of bracketing every call)
extern "C" int sttode_timing_enable(SttodeModel* m, int every) {
    STT_REQUIRE(m && every >= 0, "sttode_timing_enable: bad arguments");
    m->timing_every = every;
    m->timing = false;
    m->calls = 0;
    return 0;
}

static hipEvent_t get_event(SttodeModel* m) {
    if (!m->pool.empty()) { hipEvent_t e = m->pool.back(); m->pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

// Reads (and clears) the per-stage totals accumulated since the last read.  Synchronises on the recorded events.
extern "C" int sttode_timing_read(SttodeModel* m, double* total_ms, int* launches, double* busy_ms) {
    STT_REQUIRE(m && total_ms && launches, "sttode_timing_read: null pointer");
    for (int i = 0; i < STT_STAGE_COUNT; ++i) { total_ms[i] = 0.0; launches[i] = 0; if (busy_ms) busy_ms[i] = 0.0; }
    // busy time of a stage = length of the UNION of its launches' [start, end] intervals (launches of consecutive pipelined calls run
    // concurrently on two streams: their durations add up to more than the time the stage kept the device busy)
    std::vector<std::pair<double, double>> iv[STT_STAGE_COUNT];
    hipEvent_t ref = m->recs.empty() ? nullptr : m->recs.front().e0;
    for (auto& r : m->recs) {
        STT_HIP(hipEventSynchronize(r.e1));
        float ms = 0.f, t0 = 0.f;
        STT_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));
        total_ms[r.stage] += ms;
        launches[r.stage] += 1;
        if (busy_ms && r.e0 != ref) {
            STT_HIP(hipEventSynchronize(r.e0));
            STT_HIP(hipEventElapsedTime(&t0, ref, r.e0));   // may be negative: `ref` is merely the first record made on the host
        }
        iv[r.stage].push_back({(double)t0, (double)t0 + (double)ms});
    }
    if (busy_ms)
        for (int st = 0; st < STT_STAGE_COUNT; ++st) {
            std::sort(iv[st].begin(), iv[st].end());
            double cur0 = 0, cur1 = 0;
            bool open = false;
            for (auto& p : iv[st]) {
                if (!open) { cur0 = p.first; cur1 = p.second; open = true; }
                else if (p.first <= cur1) { if (p.second > cur1) cur1 = p.second; }
                else { busy_ms[st] += cur1 - cur0; cur0 = p.first; cur1 = p.second; }
            }
            if (open) busy_ms[st] += cur1 - cur0;
        }
    for (auto& r : m->recs) { m->pool.push_back(r.e0); m->pool.push_back(r.e1); }
    m->recs.clear();
    return 0;
}

struct StageTimer {
    SttodeModel* m; int stage; hipStream_t s; hipEvent_t e0 = nullptr, e1 = nullptr; bool on = false;
    StageTimer(SttodeModel* m_, int st, hipStream_t s_) : m(m_), stage(st), s(s_) {
        // the per-trajectory stages (two events per call) are bracketed on EVERY call while timing is enabled, so that overlapping launches
        // of consecutive calls are seen; the many short per-agent stages only on the sampled calls
        on = m->timing || (m->timing_every > 0 && ((stage >= STT_STAGE_MLP0 && stage <= STT_STAGE_CHAIN) || stage == STT_STAGE_FUSED));
        if (on) { e0 = get_event(m); e1 = get_event(m); (void)hipEventRecord(e0, s); }
    }
    ~StageTimer() {
        if (on) { (void)hipEventRecord(e1, s); m->recs.push_back({stage, e0, e1}); }
    }
};

#define RUN(stage, s, call)                         \
    do {                                            \
        StageTimer _t(m, stage, s);                 \
        int _rc = (call);                           \
        if (_rc) return _rc;                        \
    } while (0)

// stage A: everything per AGENT (encoder, block-0 GRU on the side stream, layer-1 pre-activations), on stream s
// self-attention of `groups` forward-call batches (attention groups; 1 = the reference's one batch per call) in one launch:
// L == S: rows = keys, columns = queries (hyptransformerlib.py:261-265 quirk); group g = agents [g attn_len attn_slots, (g + 1) attn_len attn_slots)
static int attention(float* qkv, float* attn, int groups, int attn_len, int attn_slots, hipStream_t s) {
    const long st_seq = (long)attn_slots * 192, gq = (long)attn_len * st_seq, go = (long)attn_len * attn_slots * 64;
    return sttode_mhgsa_attn_groups(qkv + 64, qkv, qkv + 128, attn, groups, gq, gq, gq, go, attn_len, attn_len, attn_slots, st_seq, 192, st_seq, 192,
                                    st_seq, 192, (long)attn_slots * 64, 64, 1.0f, 0.35355339059327373f, 8, s);
}

static int stage_agents(SttodeModel* m, float* ws, const long* off, int n, int groups, int attn_len, int attn_slots, hipStream_t s, bool use_side) {
    const float* const* W = m->w;
    const int Tp = m->Tp, TPX = m->TPX;
    float* xpad = ws + off[STT_B_XPAD];
    float* g = ws + off[STT_B_G];
    float* qkv = ws + off[STT_B_QKV];
    float* attn = ws + off[STT_B_ATTN];
    float* pf = ws + off[STT_B_PF];
    float* state0 = ws + off[STT_B_STATE0];
    float *A0x = ws + off[STT_B_A0X], *A0y = ws + off[STT_B_A0Y], *A1y = ws + off[STT_B_A1Y];

    // Scene batches (attention length 1) with the reference's integrator: encoder and block-0 GRU of every 16-agent tile as two workgroup
    // roles of ONE launch (csrc/encoder.hip agents_fused_kernel) -- no side stream, no event fork / join, three launches fewer.  Used
    // wherever the latency forms of both halves would be chosen anyway (same code, same bits); STTODE_AGENTS_FUSED=0 disables it.
    static const bool fuse_on = !(getenv("STTODE_AGENTS_FUSED") && atoi(getenv("STTODE_AGENTS_FUSED")) == 0);
    static const int gru0_lat_p = getenv("STTODE_GRU0_LAT_TILES") ? atoi(getenv("STTODE_GRU0_LAT_TILES")) : 4096;
    const int ntiles = (n + 15) / 16;
    if (fuse_on && attn_len == 1 && m->ode_method == 0 && m->ode_steps == 1 && stt_agents_fused_covers(Tp, TPX) &&
        ntiles <= (use_side ? stt_gru_lat_tiles() : gru0_lat_p) && ntiles <= stt_enc_lat_tiles()) {
        RUN(STT_STAGE_AGENTS, s,
            stt_agents_fused(W, ws + off[STT_B_ENC_IN], (const int*)(ws + off[STT_B_LAST]), g, qkv, pf, xpad, state0, n, Tp, TPX, 12.0f, s));
        RUN(STT_STAGE_LINEAR, s,
            sttode_agent_preact(pf, state0, W[STT_W_B0_XWA], W[STT_W_B0_XB1], W[STT_W_B0_YWA], W[STT_W_B0_YB1], W[STT_W_B1_YWA], W[STT_W_B1_YB1],
                                A0x, A0y, A1y, n, s));
        return 0;
    }
    // fork: block-0 conv+GRU (per agent) only needs the front-end output; it runs beside the encoder
    // (pipelined form: no side stream -- the stage already runs beside the previous calls' per-trajectory kernels, and every
    // extra stream shares one of the 4 hardware queues with the streams that must overlap)
    if (use_side && !m->side) {
        static hipStream_t g_side[STT_MAX_DEVICES] = {};     // process-wide per device, like sA / sB / sB2
        int dev = 0;
        STT_HIP(hipGetDevice(&dev));
        STT_REQUIRE(dev >= 0 && dev < STT_MAX_DEVICES, "sttode_inference_*: device index beyond STT_MAX_DEVICES");
        {
            std::lock_guard<std::mutex> lk(g_stream_mu);
            if (!g_side[dev]) STT_HIP(hipStreamCreateWithFlags(&g_side[dev], hipStreamNonBlocking));
            m->side = g_side[dev];
        }
    }
    hipStream_t gs = use_side ? m->side : s;
    if (use_side) {
        STT_HIP(hipEventRecord(m->ev_fork, s));
        STT_HIP(hipStreamWaitEvent(m->side, m->ev_fork, 0));
    }
    // STTODE_GRU0_STREAM=1 (default off): same-box A/Bs showed no gain over the resident-weights kernel once the chain keeps one
    // workgroup per CU (63.7 vs 65.7 / 64.3 M trajectories/s), so the streaming form stays an option, not the default
    static const bool gru0_stream = getenv("STTODE_GRU0_STREAM") && atoi(getenv("STTODE_GRU0_STREAM")) != 0;
    // Tp > 8 (two 16-wide input tiles): always the streaming form -- the resident-weights instantiation for that shape spills 44 B per lane
    if ((!use_side && gru0_stream && (long)n * m->K >= 16384) || TPX == 2) {
        // streaming GRU (24 KiB of LDS): co-resides with the previous calls' chain workgroups
        RUN(STT_STAGE_GRU0, gs,
            sttode_gru_cols32(xpad, 16 * TPX, W[STT_W_G0_POOL], (const int*)W[STT_W_G0_PROG], 13 * Tp, W[STT_W_G0_CONSTS], state0, n, Tp, gs));
    } else
    {
    // pipelined form: the latency form up to 4096 tiles -- its workgroups (no LDS-resident weights) co-reside with the previous calls' chain
    // workgroups, the resident-weights form (144 KiB of LDS) only gets chain-free CUs: 65 -> 68 M trajectories/s at 512 scenes (same box),
    // neutral at 128 / 256 / 1024 / 2048 scenes (STTODE_GRU0_LAT_TILES overrides)
    const int gru0_lat = gru0_lat_p;
    RUN(STT_STAGE_GRU0, gs,
        stt_gru_cols_form(xpad, W[STT_W_B0_CONVP], W[STT_W_B0_CONVB], W[STT_W_B0_WIHP], W[STT_W_B0_WHHP], W[STT_W_B0_GBIAS], state0, n,
                          Tp, TPX, use_side ? 0 : gru0_lat, gs));
    }
    if (use_side) STT_HIP(hipEventRecord(m->ev_join, m->side));

    RUN(STT_STAGE_EMBED, s,
        sttode_embed_qkv(W[STT_W_FC1P], W[STT_W_FC1B], W[STT_W_POSP], W[STT_W_PEB], W[STT_W_FC2P], W[STT_W_FC2B], W[STT_W_FC3P],
                         W[STT_W_FC3B], W[STT_W_FC3LAST], W[STT_W_INP], W[STT_W_INB], ws + off[STT_B_ENC_IN],
                         (const int*)(ws + off[STT_B_LAST]), g, qkv, n, Tp, s));
    const float* attn_src = qkv + 128;
    int ld_attn = 192;
    if (attn_len > 1) {
        RUN(STT_STAGE_ATTN, s, attention(qkv, attn, groups, attn_len, attn_slots, s));
        attn_src = attn;
        ld_attn = 64;
    }
    if ((m->ode_method != 0 || m->ode_steps != 1) && attn_len > 1) {
        // Multi-step Euler / RK4 with an attention group > 1: every stage is a pass over the whole group.  y' = f(y), f = the encoder layer
        // with the attention of state y (TransformerEncoder_ode, ode_demo.py:25-72), integrated over [0, 12] on a uniform grid with the stage
        // algebra of hypertransformer.ode_integrate (the op-level form; the Butcher rows are summed old stages first here: fp32 rounding apart).
        float* ob = ws + off[STT_B_ODE];
        const size_t nf = (size_t)n * 64;
        float *y = ob, *k1 = ob + nf, *k2 = ob + 2 * nf, *k3 = ob + 3 * nf, *t = ob + 5 * nf;
        // Round 4: every stage = ONE fused launch (right-hand side at the stage's state -> next state by the stage's Butcher row ->
        // its in-projection; the last stage of the last step writes past_feature) + the attention of the next state: 2 launches per stage,
        // 8 per RK4 step (the op-level sequence of round 3: ~30).  y0 = g; qkv / attn of y0 are the launches above.
        auto attention_of_state = [&]() -> int {
            RUN(STT_STAGE_ATTN, s, attention(qkv, attn, groups, attn_len, attn_slots, s));
            return 0;
        };
        // stage(state, base, kA cA, kB cB, kC cC, cN, kout, out, last): `last` = the integration ends here (no next attention, pf written)
        auto stage = [&](const float* state, const float* base, const float* kA, double cA, const float* kB, double cB, const float* kC, double cC,
                         double cN, float* kout, float* out, bool last) -> int {
            RUN(STT_STAGE_POST, s,
                stt_post_attn_stage(W, state, attn, 64, base, kA, (float)cA, kB, (float)cB, kC, (float)cC, (float)cN, kout, out, last ? nullptr : qkv,
                                    g, last ? pf : nullptr, n, s));
            return last ? 0 : attention_of_state();
        };
#define ODE_DO(call) do { if (int _rc = (call)) return _rc; } while (0)
        const double h = 12.0 / (double)m->ode_steps;
        for (int st = 0; st < m->ode_steps; ++st) {
            const bool fin = st + 1 == m->ode_steps;
            const float* y0 = st == 0 ? g : y;                   // (the first step reads g itself: no copy)
            if (m->ode_method == 0) { ODE_DO(stage(y0, y0, nullptr, 0, nullptr, 0, nullptr, 0, h, nullptr, y, fin)); continue; }
            if (m->ode_method == 1) {   // torchdiffeq's fixed-grid rk4 = the 3/8 rule (rk4_alt_step_func)
                ODE_DO(stage(y0, y0, nullptr, 0, nullptr, 0, nullptr, 0, h / 3, k1, t, false));
                ODE_DO(stage(t, y0, k1, -h / 3, nullptr, 0, nullptr, 0, h, k2, t, false));
                ODE_DO(stage(t, y0, k1, h, k2, -h, nullptr, 0, h, k3, t, false));
                ODE_DO(stage(t, y0, k1, h / 8, k2, 3 * h / 8, k3, 3 * h / 8, h / 8, nullptr, y, fin));
            } else {                    // classical RK4
                ODE_DO(stage(y0, y0, nullptr, 0, nullptr, 0, nullptr, 0, h / 2, k1, t, false));
                ODE_DO(stage(t, y0, nullptr, 0, nullptr, 0, nullptr, 0, h / 2, k2, t, false));
                ODE_DO(stage(t, y0, nullptr, 0, nullptr, 0, nullptr, 0, h, k3, t, false));
                ODE_DO(stage(t, y0, k1, h / 6, k2, h / 3, k3, h / 3, h / 6, nullptr, y, fin));
            }
        }
#undef ODE_DO
    } else if (m->ode_method != 0 || m->ode_steps != 1) {
        RUN(STT_STAGE_POST, s,
            sttode_post_attn_ode(W[STT_W_OUTP], W[STT_W_OUTB], W[STT_W_INFOP], W[STT_W_INFOB], W[STT_W_GATEP], W[STT_W_GATEB], W[STT_W_LN1W],
                                 W[STT_W_LN1B], W[STT_W_L1P], W[STT_W_L1B], W[STT_W_L2P], W[STT_W_L2B], W[STT_W_LN2W], W[STT_W_LN2B],
                                 W[STT_W_INP], W[STT_W_INB], g, pf, n, 12.0f, m->ode_method, m->ode_steps, s));
    } else
    RUN(STT_STAGE_POST, s,
        sttode_post_attn(W[STT_W_OUTP], W[STT_W_OUTB], W[STT_W_INFOP], W[STT_W_INFOB], W[STT_W_GATEP], W[STT_W_GATEB], W[STT_W_LN1W],
                         W[STT_W_LN1B], W[STT_W_L1P], W[STT_W_L1B], W[STT_W_L2P], W[STT_W_L2B], W[STT_W_LN2W], W[STT_W_LN2B], g,
                         attn_src, ld_attn, pf, n, 12.0f, s));
    if (use_side) STT_HIP(hipStreamWaitEvent(s, m->ev_join, 0));  // join
    RUN(STT_STAGE_LINEAR, s,
        sttode_agent_preact(pf, state0, W[STT_W_B0_XWA], W[STT_W_B0_XB1], W[STT_W_B0_YWA], W[STT_W_B0_YB1], W[STT_W_B1_YWA], W[STT_W_B1_YB1],
                            A0x, A0y, A1y, n, s));
    return 0;
}

// stage B: everything per TRAJECTORY (block-0 MLPs, block-1 GRU, block-1 MLP + epilogue), on stream s
static int stage_trajectories(SttodeModel* m, float* ws, const long* off, int n, const float* z, float* pred, hipStream_t s, bool pipelined) {
    const float* const* W = m->w;
    const int K = m->K, Tp = m->Tp, Tf = m->Tf, TPX = m->TPX, NOY = m->NOY;
    const float* xpad = ws + off[STT_B_XPAD];
    const float *A0x = ws + off[STT_B_A0X], *A0y = ws + off[STT_B_A0Y], *A1y = ws + off[STT_B_A1Y];
    float *dbuf = ws + off[STT_B_DBUF], *ybuf = ws + off[STT_B_YBUF], *state1 = ws + off[STT_B_STATE1];
    // Fused chain: one persistent kernel for the whole per-trajectory stage (csrc/chain32.hip).  Its work item is a group of
    // 128 trajectories on one workgroup, so automatic mode uses it from 128 groups on (measured: +4..5 % at 20-28 k trajectories, the SDD
    // and NBA-128 legs; smaller batches keep the 16-column kernels, whose items spread over more CUs).
    const long ncols_all = (long)n * K;
    const bool fused = m->chain_mode == 1 || (m->chain_mode < 0 && ncols_all >= 16384);
    if (fused) {
        if (m->b3) {   // exploratory mode: the same chain on the three-way bf16 split stream
            RUN(STT_STAGE_CHAIN, s,
                stt_traj_chain_b3(A0x, A0y, A1y, W[STT_W_CHAINB3_POOL], (const int*)W[STT_W_CHAINB3_PROG], m->prog_len, W[STT_W_CHAIN_CONSTS], z,
                                  xpad, 16 * TPX, ws + off[STT_B_CUR], ws + off[STT_B_ORIG], pred, (int*)(ws + off[STT_B_QUEUE]), (int)ncols_all, K,
                                  Tp, Tf, pipelined ? 1 : 2, s));
            return 0;
        }
        RUN(STT_STAGE_CHAIN, s,
            sttode_traj_chain(A0x, A0y, A1y, W[STT_W_CHAIN_POOL], (const int*)W[STT_W_CHAIN_PROG], m->prog_len, W[STT_W_CHAIN_CONSTS], z,
                              xpad, 16 * TPX, ws + off[STT_B_CUR], ws + off[STT_B_ORIG], pred, (int*)(ws + off[STT_B_QUEUE]), (int)ncols_all, K,
                              Tp, Tf, pipelined ? 1 : 2, s));
        return 0;
    }
    // optional split into column parts at agent boundaries (pointers are simply offset)
    int P = m->col_parts;
    if (P > n) P = n;
    if (P > 1) STT_HIP(hipEventRecord(m->ev_agents, s));
    const float* cur = ws + off[STT_B_CUR];
    const float* orig = ws + off[STT_B_ORIG];
    for (int p = 0; p < P; ++p) {
        const long a0 = (long)n * p / P, a1 = (long)n * (p + 1) / P;
        const int na = (int)(a1 - a0), nc = na * K;
        const long c0 = a0 * K;
        if (p > 0 && !m->part_stream[p]) STT_HIP(hipStreamCreateWithFlags(&m->part_stream[p], hipStreamNonBlocking));
        hipStream_t ps = p == 0 ? s : m->part_stream[p];
        if (p > 0) STT_HIP(hipStreamWaitEvent(ps, m->ev_agents, 0));
        RUN(STT_STAGE_MLP0, ps,
            sttode_mlp_block0(A0x + a0 * 512, A0y + a0 * 512, W[STT_W_B0_STREAM], m->n_chunks0, z + c0 * 32,
                              xpad + a0 * 16 * TPX, dbuf + c0 * 16 * TPX, ybuf + c0 * 16 * NOY, nc, K, TPX, NOY, ps));
        RUN(STT_STAGE_GRU1, ps,
            sttode_gru_cols(dbuf + c0 * 16 * TPX, W[STT_W_B1_CONVP], W[STT_W_B1_CONVB], W[STT_W_B1_WIHP], W[STT_W_B1_WHHP],
                            W[STT_W_B1_GBIAS], state1 + c0 * 96, nc, Tp, TPX, ps));
        RUN(STT_STAGE_MLP1, ps,
            sttode_mlp_block1(A1y + a0 * 512, W[STT_W_B1_STREAM], m->n_chunks1, z + c0 * 32, state1 + c0 * 96,
                              ybuf + c0 * 16 * NOY, cur + a0 * 2, orig + a0 * 2, pred + c0 * 2 * Tf, nc, K, Tf, NOY, ps));
        if (p > 0) STT_HIP(hipEventRecord(m->ev_part[p], ps));
    }
    for (int p = 1; p < P; ++p) STT_HIP(hipStreamWaitEvent(s, m->ev_part[p], 0));
    return 0;
}

static int frontend(SttodeModel* m, float* ws, const long* off, const float* past, const int* scene_ptr, int n, int S, int nbaN,
                    hipStream_t s) {
    if (scene_ptr) {
        RUN(STT_STAGE_FRONTEND, s,
            sttode_frontend_scenes(past, scene_ptr, n, S, m->Tp, m->TPX, 1, ws + off[STT_B_SCENE_ORIG],
                                   (int*)(ws + off[STT_B_AGENT_SCENE]), ws + off[STT_B_XPAD], ws + off[STT_B_ENC_IN], ws + off[STT_B_CUR],
                                   ws + off[STT_B_ORIG], (int*)(ws + off[STT_B_LAST]), s));
    } else {
        RUN(STT_STAGE_FRONTEND, s,
            sttode_frontend_nba(past, n, nbaN, m->Tp, m->TPX, ws + off[STT_B_XPAD], ws + off[STT_B_ENC_IN], ws + off[STT_B_CUR],
                                ws + off[STT_B_ORIG], (int*)(ws + off[STT_B_LAST]), s));
    }
    return 0;
}

// Fused launch (csrc/chain32.hip, RoleArgs): calls with the reference's integrator whose per-trajectory stage takes the fused chain anyway;
// the per-agent stage then needs no launch, no stream and no free compute unit of its own.  Attention groups > 1 (the NBA branch) keep
// embed_qkv and mhgsa_attn as launches in front (the attention reads every agent of the group); the roles start behind them.
static bool use_fused(const SttodeModel* m, int n) {
    const long ncols_all = (long)n * m->K;
    const bool chain = m->chain_mode == 1 || (m->chain_mode < 0 && ncols_all >= 16384);
    return m->fused_mode == 1 && chain && m->ode_method == 0 && m->ode_steps == 1 && stt_chain_fused_covers(m->Tp);
}
// Serial scene calls below the chain threshold (the reference's one-scene-per-call evaluation loop, test.py:171-188): front-end, per-agent
// stage and per-trajectory stage as roles of one launch.  Reference integrator only (the roles run one Euler step, like the fused launch).
static bool use_scene_launch(const SttodeModel* m, int n, const int* scene_ptr) {
    const long ncols_all = (long)n * m->K;
    const bool chain = m->chain_mode == 1 || (m->chain_mode < 0 && ncols_all >= 16384);
    return scene_ptr != nullptr && !chain && m->scene_launch > 0 && (ncols_all + 15) / 16 <= m->scene_launch && m->ode_method == 0 &&
           m->ode_steps == 1 && m->col_parts <= 1 && stt_scene_lat_covers(m->Tp, m->TPX, m->NOY);
}
static int stage_fused(SttodeModel* m, float* ws, const long* off, int n, int groups, int attn_len, int attn_slots, const float* z, float* pred,
                       const float* past, const int* scene_ptr, int S, hipStream_t s) {
    const float* const* W = m->w;
    const float* attn = nullptr;
    if (attn_len > 1) {
        float* qkv = ws + off[STT_B_QKV];
        RUN(STT_STAGE_EMBED, s,
            sttode_embed_qkv(W[STT_W_FC1P], W[STT_W_FC1B], W[STT_W_POSP], W[STT_W_PEB], W[STT_W_FC2P], W[STT_W_FC2B], W[STT_W_FC3P],
                             W[STT_W_FC3B], W[STT_W_FC3LAST], W[STT_W_INP], W[STT_W_INB], ws + off[STT_B_ENC_IN],
                             (const int*)(ws + off[STT_B_LAST]), ws + off[STT_B_G], qkv, n, m->Tp, s));
        RUN(STT_STAGE_ATTN, s, attention(qkv, ws + off[STT_B_ATTN], groups, attn_len, attn_slots, s));
        attn = ws + off[STT_B_ATTN];
    }
    RUN(STT_STAGE_FUSED, s, stt_chain_fused(W, ws, off, n, m->K, m->Tp, m->Tf, m->prog_len, z, pred, 12.0f, attn, 64, past, scene_ptr, S, 2, m->b3, m->role_lead, m->drop_tile,
                                            m->tmo_host, s));
    return 0;
}

// serial form: everything on the caller's stream
static inline void arm_timing(SttodeModel* m) {
    m->timing = m->timing_every > 0 && (m->calls % m->timing_every) == 0;
    ++m->calls;
}

static int run_serial(SttodeModel* m, const float* past, const int* scene_ptr, int n, int S, int G, int B, int N, const float* z, float* ws,
                      float* pred, hipStream_t s) {
    arm_timing(m);
    long off[STT_B_COUNT], tot;
    if (int rc = sttode_workspace_layout(m, n, S, off, &tot)) return rc;
    if (use_scene_launch(m, n, scene_ptr)) {   // a scene or a few: the whole call as ONE launch of cooperating workgroups (scene_lat.hip)
        RUN(STT_STAGE_FUSED, s,
            stt_scene_lat(m->w, ws, off, n, m->K, m->Tp, m->Tf, m->TPX, m->NOY, m->n_chunks0, m->n_chunks1, z, pred, 12.0f, past, scene_ptr, S,
                          m->drop_tile, m->tmo_host, s));
        return 0;
    }
    // scene batches on the fused launch: the roles run the front-end of their own tiles (the call is ONE launch); otherwise it is a launch
    const bool fe_in_role = use_fused(m, n) && scene_ptr != nullptr && m->fe_in_role;
    if (!fe_in_role)
        if (int rc = frontend(m, ws, off, past, scene_ptr, n, S, N, s)) return rc;
    if (use_fused(m, n))
        return stage_fused(m, ws, off, n, G, scene_ptr ? 1 : B, scene_ptr ? 1 : N, z, pred, fe_in_role ? past : nullptr, scene_ptr, S, s);
    if (int rc = stage_agents(m, ws, off, n, G, scene_ptr ? 1 : B, scene_ptr ? 1 : N, s, true)) return rc;
    return stage_trajectories(m, ws, off, n, z, pred, s, false);
}

// ---- LAGGED form (round 4) ---------------------------------------------------------------------------------------------------------
// Calls whose per-trajectory stage takes the chain (reference integrator) rotate over lag_streams pipeline streams.  The launch call k
// enqueues on its stream = [throughput-form roles of call k | trajectory groups of the oldest call of that stream whose groups are still
// outstanding] (csrc/role32.hpp, stt_chain_lagged): the groups read the tables an EARLIER launch of the same stream wrote, so stream order
// is the only dependency and nothing inside a launch waits.  Why: as workgroups of the launch that consumes them (round 3) the roles had
// to be latency forms, held 9-12 % of the chip's workgroup slots for 5 % of the FLOP, and bounded every small launch by role + group.
static bool use_lagged(const SttodeModel* m, int n) {
    const long ncols_all = (long)n * m->K;
    const bool chain = m->chain_mode == 1 || (m->chain_mode < 0 && ncols_all >= 16384);
    return m->lag_streams > 0 && m->fused_mode == 1 && chain && m->ode_method == 0 && m->ode_steps == 1 && stt_chain_lagged_covers(m->Tp);
}
static hipStream_t lag_stream(const SttodeModel* m, int si) { return si == 0 ? m->sB : si == 1 ? m->sB2 : si == 2 ? m->sA : m->sX[0]; }
static void lag_unqueue(SttodeModel* m, int slot) {
    const int si = m->lag[slot].si;
    int w = 0;
    for (int i = 0; i < m->lag_qn[si]; ++i)
        if (m->lag_q[si][i] != slot) m->lag_q[si][w++] = m->lag_q[si][i];
    m->lag_qn[si] = w;
}
// the outstanding groups of `slot` as a launch of their own (nobody made a later call on that stream, or the slot is wanted back)
static int lag_flush(SttodeModel* m, int slot) {
    LagPending& p = m->lag[slot];
    if (!p.valid) return 0;
    lag_unqueue(m, slot);
    p.valid = false;
    LagRoles none = {};
    LagGroups lg = {p.ws, p.off, p.n, p.z, p.pred, p.gt, p.ade, p.fde, p.scale, p.one_launch};
    RUN(STT_STAGE_FUSED, p.s, stt_chain_lagged(m->w, none, lg, m->K, m->Tp, m->Tf, m->prog_len, m->b3, p.s));
    STT_HIP(hipEventRecord(m->evB_done[slot], p.s));
    return 0;
}
static int lag_flush_all(SttodeModel* m) {
    for (int p = 0; p < STT_MAX_SLOTS; ++p)
        if (int rc = lag_flush(m, p)) return rc;
    return 0;
}
static int run_lagged(SttodeModel* m, const float* past, const int* scene_ptr, int n, int S, int G, int B, int N, const float* z, float* ws,
                      float* pred, int slot, const long* off, const SttodeAsyncOpts& o, hipStream_t s) {
    const float* const* W = m->w;
    const int si = (int)(m->lag_calls % m->lag_streams);
    hipStream_t sf = lag_stream(m, si);
    ++m->lag_calls;
    STT_HIP(hipStreamWaitEvent(sf, m->ev_call, 0));              // inputs and z of this call (a wait for itself when the caller works on sf)
    STT_HIP(hipStreamWaitEvent(sf, m->evB_done[slot], 0));       // the slot's previous user has drained (same stream in a 2 x streams rotation)
    // scene batches: set_data runs inside the roles (role32.hpp frontend32): the call is ONE launch (STTODE_LAG_FE=0: front-end launches)
    static const bool fe_in = !(getenv("STTODE_LAG_FE") && atoi(getenv("STTODE_LAG_FE")) == 0);
    const bool fe_role = fe_in && scene_ptr != nullptr;
    // NBA, attention groups > 1: set_data_nba rides in the embedding's launch (one launch fewer in front of the roles)
    const bool fe_embed = fe_in && !scene_ptr && B > 1 && stt_embed_qkv_fe_covers(n, m->Tp);
    if (!fe_role && !fe_embed)
        if (int rc = frontend(m, ws, off, past, scene_ptr, n, S, N, sf)) return rc;
    const float* attn = nullptr;
    if (!scene_ptr && B > 1) {   // attention groups > 1: embedding + attention stay launches in front (the attention reads every agent of the group)
        float* qkv = ws + off[STT_B_QKV];
        if (fe_embed) {
            RUN(STT_STAGE_EMBED, sf,
                stt_embed_qkv_fe(W, past, n, N, m->Tp, m->TPX, ws + off[STT_B_XPAD], ws + off[STT_B_ENC_IN], ws + off[STT_B_CUR], ws + off[STT_B_ORIG],
                                 (int*)(ws + off[STT_B_LAST]), ws + off[STT_B_G], qkv, sf));
        } else
        RUN(STT_STAGE_EMBED, sf,
            sttode_embed_qkv(W[STT_W_FC1P], W[STT_W_FC1B], W[STT_W_POSP], W[STT_W_PEB], W[STT_W_FC2P], W[STT_W_FC2B], W[STT_W_FC3P],
                             W[STT_W_FC3B], W[STT_W_FC3LAST], W[STT_W_INP], W[STT_W_INB], ws + off[STT_B_ENC_IN],
                             (const int*)(ws + off[STT_B_LAST]), ws + off[STT_B_G], qkv, n, m->Tp, sf));
        RUN(STT_STAGE_ATTN, sf, attention(qkv, ws + off[STT_B_ATTN], G, B, N, sf));
        attn = ws + off[STT_B_ATTN];
    }
    const int gs = m->lag_qn[si] > 0 ? m->lag_q[si][0] : -1;     // the oldest call of this stream whose groups are outstanding
    LagPending* g = gs >= 0 ? &m->lag[gs] : nullptr;
    float* zgen = o.device_latents ? const_cast<float*>(z) : nullptr;   // `z` is this call's latent BUFFER, filled by its roles
    const bool met = o.metrics_gt != nullptr;                           // this call's groups will compute its best-of-K metrics
    LagRoles lr = {ws, off, n, attn, 64, 12.0f, zgen, o.zkey, fe_role ? past : nullptr, fe_role ? scene_ptr : nullptr, fe_role ? S : 0,
                   met ? o.ade : nullptr, met ? o.fde : nullptr};
    LagGroups lg = {};
    if (g) lg = LagGroups{g->ws, g->off, g->n, g->z, g->pred, g->gt, g->ade, g->fde, g->scale, fe_role && g->one_launch};
    RUN(STT_STAGE_FUSED, sf, stt_chain_lagged(W, lr, lg, m->K, m->Tp, m->Tf, m->prog_len, m->b3, sf));
    if (g) {
        lag_unqueue(m, gs);
        g->valid = false;
        STT_HIP(hipEventRecord(m->evB_done[gs], sf));
    }
    LagPending& p = m->lag[slot];
    p.valid = true; p.ws = ws; p.n = n; p.z = z; p.pred = pred; p.s = sf; p.si = si;
    p.one_launch = fe_role;
    p.gt = met ? o.metrics_gt : nullptr; p.ade = met ? o.ade : nullptr; p.fde = met ? o.fde : nullptr; p.scale = met ? o.metrics_scale : 1.0f;
    memcpy(p.off, off, sizeof(p.off));
    m->lag_q[si][m->lag_qn[si]++] = slot;
    m->slot_stream[slot] = sf;
    return 0;
}

// pipelined form: stage A on sA, stage B on sB, two workspace slots; the caller later waits with sttode_wait(slot)
static int run_async(SttodeModel* m, const float* past, const int* scene_ptr, int n, int S, int G, int B, int N, const float* z, float* ws,
                     float* pred, int slot, const SttodeAsyncOpts* opts, hipStream_t s) {
    STT_REQUIRE(slot >= 0 && slot < STT_MAX_SLOTS, "sttode_inference_*_async: slot must be in [0, 8)");
    std::lock_guard<std::mutex> lk(m->mu);
    // Everything the options ask for is checked BEFORE anything is enqueued or any state of the model changes (round-4 advice: the
    // arm-then-call entry points left a request armed when a later check failed): a refused call has no effect at all.
    SttodeAsyncOpts o = {};
    if (opts) o = *opts;
    if (o.device_latents || o.metrics_gt) {
        STT_REQUIRE(use_lagged(m, n), "sttode_inference_*_async: device latents / fused metrics need the lagged form (sttode_async_is_lagged)");
        STT_REQUIRE(!o.metrics_gt || (o.ade && o.fde), "sttode_inference_*_async: fused metrics need the ade / fde outputs");
    }
    arm_timing(m);
    long off[STT_B_COUNT], tot;
    if (int rc = sttode_workspace_layout(m, n, S, off, &tot)) return rc;
    if (int rc = lag_flush(m, slot)) return rc;                  // the slot is wanted back: its outstanding groups (if any) go first
    STT_HIP(hipEventRecord(m->ev_call, s));                      // inputs and z of this call are ready once this fires
    if (use_lagged(m, n)) return run_lagged(m, past, scene_ptr, n, S, G, B, N, z, ws, pred, slot, off, o, s);
    if (use_fused(m, n)) {
        // ONE stream per call.  Round 5: these launches hand tables over INSIDE the launch (tile flags, bounded spin), and workgroups are
        // dispatched in index order per XCD only -- several of them in flight on different queues could wait on each other across XCDs
        // (round-3/4 advice) -- so by default they run one at a time on ONE stream (STTODE_FUSED_STREAMS > 1 restores the rotation for A/B).
        // The product's pipelined path is the lagged form above, which has no hand-off.
        const int si = (int)(m->acalls % m->fused_streams);
        hipStream_t sf = si == 0 ? m->sB : si == 1 ? m->sB2 : si == 2 ? m->sA : m->sX[si - 3];
        ++m->acalls;
        STT_HIP(hipStreamWaitEvent(sf, m->ev_call, 0));
        STT_HIP(hipStreamWaitEvent(sf, m->evB_done[slot], 0));   // the slot's previous user has drained
        const bool fe_in_role = scene_ptr != nullptr && m->fe_in_role;
        if (!fe_in_role)
            if (int rc = frontend(m, ws, off, past, scene_ptr, n, S, N, sf)) return rc;
        if (int rc = stage_fused(m, ws, off, n, G, scene_ptr ? 1 : B, scene_ptr ? 1 : N, z, pred, fe_in_role ? past : nullptr, scene_ptr, S, sf)) return rc;
        STT_HIP(hipEventRecord(m->evB_done[slot], sf));
        m->slot_stream[slot] = sf;
        return 0;
    }
    STT_HIP(hipStreamWaitEvent(m->sA, m->ev_call, 0));
    STT_HIP(hipStreamWaitEvent(m->sA, m->evB_done[slot], 0));    // the slot's previous user (call i-2) has drained
    if (int rc = frontend(m, ws, off, past, scene_ptr, n, S, N, m->sA)) return rc;
    if (int rc = stage_agents(m, ws, off, n, G, scene_ptr ? 1 : B, scene_ptr ? 1 : N, m->sA, false)) return rc;
    STT_HIP(hipEventRecord(m->evA_done[slot], m->sA));
    // per-trajectory stages of consecutive calls alternate between two streams (b_streams == 2): the persistent chain kernel of
    // call i+1 then starts on the compute units its predecessor's last workgroups leave (no chip-wide resource is held)
    hipStream_t sb = (m->b_streams == 2 && (m->acalls++ & 1)) ? m->sB2 : m->sB;
    STT_HIP(hipStreamWaitEvent(sb, m->ev_call, 0));
    STT_HIP(hipStreamWaitEvent(sb, m->evA_done[slot], 0));
    if (int rc = stage_trajectories(m, ws, off, n, z, pred, sb, true)) return rc;
    STT_HIP(hipEventRecord(m->evB_done[slot], sb));
    m->slot_stream[slot] = sb;
    return 0;
}

extern "C" int sttode_inference_scenes(SttodeModel* m, const float* past, const int* scene_ptr, int n, int S, const float* z,
                                       float* workspace, float* pred, void* stream) {
    STT_REQUIRE(m && past && scene_ptr && z && workspace && pred, "sttode_inference_scenes: null pointer");
    STT_REQUIRE(n > 0 && S > 0, "sttode_inference_scenes: n and S must be positive");
    return run_serial(m, past, scene_ptr, n, S, 1, 1, 1, z, workspace, pred, (hipStream_t)stream);
}

extern "C" int sttode_inference_nba(SttodeModel* m, const float* past, int B, int N, const float* z, float* workspace, float* pred,
                                    void* stream) {
    STT_REQUIRE(m && past && z && workspace && pred, "sttode_inference_nba: null pointer");
    STT_REQUIRE(B > 0 && N > 0, "sttode_inference_nba: B and N must be positive");
    return run_serial(m, past, nullptr, B * N, 0, 1, B, N, z, workspace, pred, (hipStream_t)stream);
}

// G forward-call batches of the NBA branch in ONE call: past [G][B][N][Tp][2]; the attention runs within each batch of B scenes (what G
// separate sttode_inference_nba calls compute, test.py:520-524), everything else is per agent / per trajectory over all G B N agents.
extern "C" int sttode_inference_nba_groups(SttodeModel* m, const float* past, int G, int B, int N, const float* z, float* workspace,
                                           float* pred, void* stream) {
    STT_REQUIRE(m && past && z && workspace && pred, "sttode_inference_nba_groups: null pointer");
    STT_REQUIRE(G > 0 && B > 0 && N > 0 && (long)G * B * N <= 0x7fffffffL / 64, "sttode_inference_nba_groups: G, B and N must be positive (and G B N small enough)");
    return run_serial(m, past, nullptr, G * B * N, 0, G, B, N, z, workspace, pred, (hipStream_t)stream);
}

extern "C" int sttode_inference_scenes_async(SttodeModel* m, const float* past, const int* scene_ptr, int n, int S, const float* z,
                                             float* workspace, float* pred, int slot, const SttodeAsyncOpts* opts, void* stream) {
    STT_REQUIRE(m && past && scene_ptr && z && workspace && pred, "sttode_inference_scenes_async: null pointer");
    STT_REQUIRE(n > 0 && S > 0, "sttode_inference_scenes_async: n and S must be positive");
    return run_async(m, past, scene_ptr, n, S, 1, 1, 1, z, workspace, pred, slot, opts, (hipStream_t)stream);
}

extern "C" int sttode_inference_nba_async(SttodeModel* m, const float* past, int B, int N, const float* z, float* workspace,
                                          float* pred, int slot, const SttodeAsyncOpts* opts, void* stream) {
    STT_REQUIRE(m && past && z && workspace && pred, "sttode_inference_nba_async: null pointer");
    const int G = opts && opts->nba_groups > 1 ? opts->nba_groups : 1;
    STT_REQUIRE(B > 0 && N > 0 && (long)G * B * N <= 0x7fffffffL / 64, "sttode_inference_nba_async: B and N must be positive (and groups x B x N small enough)");
    return run_async(m, past, nullptr, G * B * N, 0, G, B, N, z, workspace, pred, slot, opts, (hipStream_t)stream);
}

// Follow-up work of an asynchronous call ON THE CALL'S OWN pipeline stream: best-of-K metrics of its predictions (utils/metrics.py:7-26) run
// the moment its launch drains -- in stream order, no event, no workgroup slots to fight for on a chip full of other calls' chains (on the
// caller's stream that 8-us kernel sat 0.5 ms in the queue and held the next call's inputs behind it).  The slot's completion event is
// re-recorded behind it: sttode_wait(slot) and the slot's next user wait for the metrics too.
extern "C" int sttode_async_best_of_k(SttodeModel* m, int slot, const float* pred, const float* gt, int n, int K, int Tf, float scale,
                                      float* ade, float* fde) {
    STT_REQUIRE(m && slot >= 0 && slot < STT_MAX_SLOTS, "sttode_async_best_of_k: bad model / slot");
    std::lock_guard<std::mutex> lk(m->mu);
    STT_REQUIRE(m->slot_stream[slot] != nullptr, "sttode_async_best_of_k: no asynchronous call has used this slot");
    if (int rc = lag_flush(m, slot)) return rc;   // (lagged form) nobody has enqueued this call's groups yet: they go first
    if (int rc = sttode_best_of_k(pred, gt, n, K, Tf, scale, ade, fde, m->slot_stream[slot])) return rc;
    STT_HIP(hipEventRecord(m->evB_done[slot], m->slot_stream[slot]));
    return 0;
}

// The NBA evaluation's per-horizon metric (test.py:530-551) of an asynchronous call, on the call's own pipeline stream like
// sttode_async_best_of_k: out [n][Tf][2] = per agent and horizon h the min over K of (mean displacement over the first h frames, displacement
// of frame h) -- sttode_horizon_metrics on the slot's predictions.
extern "C" int sttode_async_horizon_metrics(SttodeModel* m, int slot, const float* pred, const float* gt, int n, int K, int Tf, float scale,
                                            float* out) {
    STT_REQUIRE(m && slot >= 0 && slot < STT_MAX_SLOTS, "sttode_async_horizon_metrics: bad model / slot");
    std::lock_guard<std::mutex> lk(m->mu);
    STT_REQUIRE(m->slot_stream[slot] != nullptr, "sttode_async_horizon_metrics: no asynchronous call has used this slot");
    if (int rc = lag_flush(m, slot)) return rc;
    if (int rc = sttode_horizon_metrics(pred, gt, n, K, Tf, scale, out, m->slot_stream[slot])) return rc;
    STT_HIP(hipEventRecord(m->evB_done[slot], m->slot_stream[slot]));
    return 0;
}

// The pipeline stream the NEXT asynchronous call of n agents will run on (lagged launches: one stream per call, three in rotation), or NULL
// when that call is not of the one-stream form.  A caller that prepares the call's inputs ON that stream (H2D copy, latents) and issues the
// call from it needs no cross-stream event at all: the call's wait for the caller's stream is then a wait for itself.
extern "C" int sttode_async_next_stream(SttodeModel* m, int n, void** stream) {
    STT_REQUIRE(m && stream && n > 0, "sttode_async_next_stream: bad arguments");
    std::lock_guard<std::mutex> lk(m->mu);
    *stream = nullptr;
    if (use_lagged(m, n)) { *stream = lag_stream(m, (int)(m->lag_calls % m->lag_streams)); return 0; }
    if (!use_fused(m, n)) return 0;
    const int si = (int)(m->acalls % m->fused_streams);
    *stream = si == 0 ? m->sB : si == 1 ? m->sB2 : si == 2 ? m->sA : m->sX[si - 3];
    return 0;
}

// 1 if the next sttode_inference_*_async call of n agents will take the lagged form -- the form that honours SttodeAsyncOpts' device latents
// and fused metrics, and whose trajectory groups never READ the prediction buffer (it may then be pinned host memory: the futures arrive
// on the host with the launch, no D2H copy) -- else 0.  A query, not a status.
extern "C" int sttode_async_is_lagged(SttodeModel* m, int n) { return m && n > 0 && use_lagged(m, n) ? 1 : 0; }

// the HOST waits until the async call that used `slot` has produced its predictions (outstanding groups are enqueued first): for
// predictions written straight to pinned host memory
extern "C" int sttode_wait_host(SttodeModel* m, int slot) {
    STT_REQUIRE(m && slot >= 0 && slot < STT_MAX_SLOTS, "sttode_wait_host: bad arguments");
    hipEvent_t ev;
    {
        std::lock_guard<std::mutex> lk(m->mu);
        if (int rc = lag_flush(m, slot)) return rc;
        ev = m->evB_done[slot];
    }
    STT_HIP(hipEventSynchronize(ev));
    return 0;
}

// make `stream` wait until the async call that used `slot` has produced its predictions
extern "C" int sttode_wait(SttodeModel* m, int slot, void* stream) {
    STT_REQUIRE(m && slot >= 0 && slot < STT_MAX_SLOTS, "sttode_wait: bad arguments");
    std::lock_guard<std::mutex> lk(m->mu);
    if (int rc = lag_flush(m, slot)) return rc;   // (lagged form) no later call carried this call's groups: they are enqueued now
    STT_HIP(hipStreamWaitEvent((hipStream_t)stream, m->evB_done[slot], 0));
    return 0;
}

// the outstanding trajectory-group launch of ONE slot's call is enqueued now (no-op if a later call carried it already)
extern "C" int sttode_async_enqueue(SttodeModel* m, int slot) {
    STT_REQUIRE(m && slot >= 0 && slot < STT_MAX_SLOTS, "sttode_async_enqueue: bad model / slot");
    std::lock_guard<std::mutex> lk(m->mu);
    return lag_flush(m, slot);
}

// every outstanding trajectory-group launch of the lagged form is enqueued (before buffers of pending calls are released or reused)
extern "C" int sttode_async_flush(SttodeModel* m) {
    STT_REQUIRE(m, "sttode_async_flush: null model");
    std::lock_guard<std::mutex> lk(m->mu);
    return lag_flush_all(m);
}

// once per workspace, before its first use: see include/sttode_hip.h
extern "C" int sttode_workspace_init(SttodeModel* m, float* workspace, int n, int S, void* stream) {
    STT_REQUIRE(m && workspace && n > 0 && S >= 0, "sttode_workspace_init: bad arguments");
    long off[STT_B_COUNT], tot;
    if (int rc = sttode_workspace_layout(m, n, S, off, &tot)) return rc;
    STT_HIP(hipMemsetAsync(workspace + off[STT_B_FLAGS], 0, (size_t)(off[STT_B_ODE] - off[STT_B_FLAGS]) * 4, (hipStream_t)stream));
    return stt_scene_flags_init(workspace, off, n, m->K, stream);
}

// the model's time-out word (pinned host memory): see include/sttode_hip.h
extern "C" int sttode_timeout_word(SttodeModel* m, const unsigned** word) {
    STT_REQUIRE(m && word && m->tmo_host, "sttode_timeout_word: bad arguments");
    *word = m->tmo_host;
    return 0;
}
extern "C" int sttode_timeout_clear(SttodeModel* m) {
    STT_REQUIRE(m && m->tmo_host, "sttode_timeout_clear: bad arguments");
    __atomic_store_n(m->tmo_host, 0u, __ATOMIC_RELAXED);
    return 0;
}

// time-out word of the in-launch hand-off forms (fused launch: word [tiles] of STT_B_FLAGS; one-launch scene form: word [tiles] of its own
// region behind it, stt_scene_flags_offset): see include/sttode_hip.h
extern "C" int sttode_check(SttodeModel* m, const float* workspace, int n, int S, void* stream) {
    STT_REQUIRE(m && workspace && n > 0 && S >= 0, "sttode_check: bad arguments");
    long off[STT_B_COUNT], tot;
    if (int rc = sttode_workspace_layout(m, n, S, off, &tot)) return rc;
    unsigned word[2] = {0, 0};
    const float* f = workspace + off[STT_B_FLAGS];
    STT_HIP(hipMemcpyAsync(&word[0], f + (n + 15) / 16, sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)stream));
    STT_HIP(hipMemcpyAsync(&word[1], f + stt_scene_flags_offset(n) + (n + 15) / 16, sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)stream));
    STT_HIP(hipStreamSynchronize((hipStream_t)stream));
    if (word[1] == 2u) {
        stt_set_error("sttode_check: the workspace was never initialised (sttode_workspace_init): the one-launch scene form refused its flag words; the predictions of that call are not valid");
        return 3;
    }
    if (word[0] != 0 || word[1] != 0) {
        stt_set_error("sttode_check: a trajectory group gave up waiting for its per-agent role (time-out word set): the predictions of that call are not valid");
        return 3;
    }
    return 0;
}
