// Error plumbing for the C ABI: every entry point returns 0 on success, non-zero on failure, and
// sttode_last_error() returns the message of the calling thread's last failure.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>

void stt_set_error(const char* msg);
int stt_gru_cols_form(const float* xin, const float* convP, const float* convB, const float* wihP, const float* whhP, const float* gbias,
                      float* state, int ncols, int Tp, int TPX, int lat_max_tiles, void* stream);   // decoder.hip
int stt_agents_fused(const float* const* W, const float* enc_in, const int* last, float* g, float* qkv, float* pf, const float* xpad,
                     float* state0, int n, int Tp, int TPX, float ode_time, void* stream);   // encoder.hip
bool stt_agents_fused_covers(int Tp, int TPX);                                          // encoder.hip: shapes the fused per-agent kernel is built for
int stt_chain_fused(const float* const* W, float* ws, const long* off, int n, int K, int Tp, int Tf, int prog_len, const float* z, float* pred,
                    float ode_time, const float* attn, int ld_attn, const float* past, const int* scene_ptr, int S, int wgs_per_cu, int b3, int lead, int drop_tile,
                    unsigned* tmo_host, void* stream);   // chain32.hip: per-agent roles + trajectory groups in one launch
bool stt_chain_fused_covers(int Tp);
int stt_scene_lat(const float* const* W, float* ws, const long* off, int n, int K, int Tp, int Tf, int TPX, int NOY, int n_chunks0, int n_chunks1,
                  const float* z, float* pred, float ode_time, const float* past, const int* scene_ptr, int S, int drop_tile, unsigned* tmo_host,
                  void* stream);   // scene_lat.hip: a scene call as ONE launch
// STT_B_FLAGS of a workspace for n agents, in 32-bit words: [fused launch: E [T] | time-out | G [T] | P [3 T], T = tiles of 16 agents, zeroed in
// front of every launch] [4 words] [one-launch scene form: E [T] | time-out | G [T] | E2 [T] | Y [C] | exit counter | initialised word: zeroed
// ONCE by sttode_workspace_init, kept zero by the form's own last workgroup]
static inline int stt_scene_flags_offset(int n) { return 5 * ((n + 15) / 16) + 4; }
int stt_scene_flags_init(float* ws, const long* off, int n, int K, void* stream);   // scene_lat.hip
bool stt_scene_lat_covers(int Tp, int TPX, int NOY);
int stt_traj_chain_b3(const float* A0x, const float* A0y, const float* A1y, const float* pool, const int* prog, int prog_len,
                      const float* consts, const float* z, const float* xpad, int ldx, const float* cur, const float* orig,
                      float* pred, int* counter, int ncols, int K, int Tp, int Tf, int wgs_per_cu, void* stream);   // chain32.hip                                                    // chain32.hip
int stt_gru_lat_tiles();
int stt_enc_lat_tiles();   // crossover of the encoder's latency form (decoder.hip: sttode_set_latency_tiles)

#define STT_REQUIRE(cond, msg)      \
    do {                            \
        if (!(cond)) {              \
            stt_set_error(msg);     \
            return 1;               \
        }                           \
    } while (0)

#define STT_HIP(expr)                                                                       \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            char _b[512];                                                                   \
            snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            stt_set_error(_b);                                                              \
            return 2;                                                                       \
        }                                                                                   \
    } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a driver call and applies to the function ON THE CURRENT DEVICE: made once per
// (kernel instantiation, device) -- a process that drives several GPUs (the pipeline keeps per-device stream tables) must not launch
// a > 64 KiB kernel on its second device without it.
#define STT_ATTR_DEVICES 64
#define STT_SET_LDS_ONCE(kernel, bytes)                                                                         \
    do {                                                                                                        \
        static bool _done[STT_ATTR_DEVICES] = {};                                                               \
        int _dev = 0;                                                                                           \
        STT_HIP(hipGetDevice(&_dev));                                                                           \
        if (_dev < 0 || _dev >= STT_ATTR_DEVICES || !_done[_dev]) {                                             \
            STT_HIP(hipFuncSetAttribute((const void*)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes))); \
            if (_dev >= 0 && _dev < STT_ATTR_DEVICES) _done[_dev] = true;                                       \
        }                                                                                                       \
    } while (0)

// The two halves of a LAGGED launch (chain32.hip stt_chain_lagged; pipeline.hip run_lagged / lag_flush): the call whose per-agent roles ride
// in the launch and the (earlier) call whose trajectory groups do.  ws == nullptr: that half is absent.
struct LagRoles {
    float* ws; const long* off; int n;                       // workspace, its layout, agents
    const float* attn; int ld_attn; float ode_time;          // attention output of the launches in front (NBA: attention groups > 1) or nullptr
    float* zgen; unsigned long long zkey;                    // the roles draw the call's latents into zgen (nullptr: the caller supplied z)
    const float* past; const int* scene_ptr; int S;          // scene batches: set_data inside the roles (nullptr: a front-end launch ran)
    float* ade; float* fde;                                  // fused metrics of this call start at +inf here (nullptr: none)
};
struct LagGroups {
    float* ws; const long* off; int n; const float* z; float* pred;
    const float* gt; float* ade; float* fde; float scale;    // fused metrics (gt == nullptr: none)
    int workers;                                             // the groups may run as workers on the call's work queue
};
int stt_chain_lagged(const float* const* W, const LagRoles& r, const LagGroups& g, int K, int Tp, int Tf, int prog_len, int b3, void* stream);
bool stt_chain_lagged_covers(int Tp);
int stt_trunk_group(int on);   // train_trunk.hip: the trunk-forward half of sttode_tgemm_group (train.hip)
