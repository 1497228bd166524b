// Training-step kernels: forward-with-tape and backward of STTODENet.forward()
// (model/STTODE.py:553-568) at the reference's training shapes (one scene, n <= 32 agents, K in {1, 20};
// NBA: 32 x 11 agents) -- a few hundred to a few thousand columns, so these are GENERIC kernels (any
// N / K, row-major nn.Parameter storage read in place, gradients accumulated straight into .grad
// storage), not the LDS-resident fused chains of the inference path.  The dense work still runs on
// v_mfma_f32_16x16x4_f32 in the column-chain formulation of chain.hpp:
//   tlinear      out[c, i] = epi( sum_j in[c / xdiv, j] * Wop[i, j] )   Wop = W or W^T (input gradient)
//   twgrad       dW[n, k] += sum_c dY[c, n] * X[c / xdiv, k],  db[n] += sum_c dY[c, n]   (deterministic split + reduce)
// plus the element-wise forward/backward pieces (GRU cell, conv1d k=3, LayerNorm, gate, Euler+relu,
// geodesic attention backward, reparameterisation + KL, squared-error / best-of-K losses).
#include "api_util.hpp"
#include <chrono>
#include <mutex>
#include "chain.hpp"

// ---------------------------------------------------------------------------------------------------
// tlinear
// ---------------------------------------------------------------------------------------------------
struct TLin {
    const float* X; const float* W; const float* bias; const float* mask; float* Y;
    long ldx, ldw, ldy, ldm;
    int cols, J, I, trans, act, accumulate, xdiv, xvec, wvec, yvec;
    // what `accumulate` adds: row (col / adiv) of asrc -- Y itself (asrc = Y, adiv = 1: += into the output) or a per-GROUP table broadcast over
    // adiv consecutive columns (sttode_tlinear_tab: the decoder MLPs' per-agent layer-1 part W1[:, pf] pf + b1, shared by an agent's K samples)
    const float* asrc; long ldas; int adiv;
    int evec;   // I % 4 == 0 and Y, bias, mask 16-byte aligned: the epilogue runs on 16-byte pieces
};

static __device__ __forceinline__ f32x4 ld_guard4(const float* row, int j, int J, bool rowok, bool vec) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (!rowok) return v;
    if (vec && j + 3 < J) return ld4(row + j);
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (j + r < J) v[r] = row[j + r];
    return v;
}

static __device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case 1: return fmaxf(v, 0.f);
        case 2: return tanhf(v);
        case 3: return 1.0f / (1.0f + expf(-v));
        default: return v;
    }
}

// WG = 4 waves; a wave owns CT column tiles x RT output tiles, and ``ksplit`` waves of the WG share one such block, splitting
// the reduction range (partials combined through LDS).  Operands of U k-steps are fetched back to back before their MFMAs, so
// one memory round trip is paid per 16*U reduction indices.  Two instantiations:
//   <1,1,8>  latency mode (few columns: 32 .. 1024): 16 x 16 block per WG, 4-way K split -> one or two round trips per launch;
//   <4,4,2>  throughput mode (NBA / long batches): 64 columns x 64 outputs per wave, weight fragments reused over 4 column tiles.
template <int RT, int CT, int U>
static __device__ __forceinline__ void tlinear_body(const TLin& a, int ksplit, int bx, int by, f32x4 (*part)[RT * CT][64]) {
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4, wave = threadIdx.x >> 6;
    const int blocks_per_wg = 4 / ksplit;
    const int oblock = by * blocks_per_wg + wave / ksplit, ksub = wave % ksplit;
    const int it0 = oblock * RT;
    const bool active = it0 * 16 < a.I;
    int col[CT];
    bool colok[CT];
    const float* xrow[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        col[t] = (bx * CT + t) * 16 + c;
        colok[t] = col[t] < a.cols;
        xrow[t] = a.X + (long)((colok[t] ? col[t] : 0) / a.xdiv) * a.ldx;
    }
    f32x4 acc[RT][CT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int t = 0; t < CT; ++t) acc[i][t] = splat4(0.f);
    if (active) {
        for (int j0 = ksub * 16 * U; j0 < a.J; j0 += 16 * U * ksplit) {
            f32x4 b[U][CT], w[U][RT];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = j0 + 16 * u + 4 * q;
#pragma unroll
                for (int t = 0; t < CT; ++t) b[u][t] = ld_guard4(xrow[t], j, a.J, colok[t], a.xvec);
#pragma unroll
                for (int i = 0; i < RT; ++i) {
                    const int row = (it0 + i) * 16 + c;  // A-operand row held by this lane
                    f32x4 wv = {0.f, 0.f, 0.f, 0.f};
                    if (row < a.I) {
                        if (!a.trans) {
                            wv = ld_guard4(a.W + (long)row * a.ldw, j, a.J, true, a.wvec);
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (j + r < a.J) wv[r] = a.W[(long)(j + r) * a.ldw + row];
                        }
                    }
                    w[u][i] = wv;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int t = 0; t < CT; ++t) acc[i][t] = mfma_k16(acc[i][t], w[u][i], b[u][t]);
        }
    }
    if (ksplit > 1) {
        // waves of one block are consecutive: block leader = wave - ksub; partial slot = (leader's block) * (ksplit-1) + ksub - 1
        const int slot = (wave / ksplit) * (ksplit - 1) + ksub - 1;
        if (ksub > 0)
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int t = 0; t < CT; ++t) part[slot][i * CT + t][lane] = acc[i][t];
        __syncthreads();
        if (ksub == 0)
            for (int k = 1; k < ksplit; ++k)
#pragma unroll
                for (int i = 0; i < RT; ++i)
#pragma unroll
                    for (int t = 0; t < CT; ++t) acc[i][t] += part[(wave / ksplit) * (ksplit - 1) + k - 1][i * CT + t][lane];
    }
    if (!active || ksub != 0) return;
    if (a.evec) {
        // every operand of the epilogue in 16-byte pieces, all requested before the first is used (element by element, each load under its
        // own bounds check is followed by its own wait: 4-12 dependent L2 round trips in a kernel that lasts 5-9 us at scene sizes)
        f32x4 bv[RT], yv[RT][CT], mv[RT][CT];
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const int o = (it0 + i) * 16 + 4 * q, oc = o < a.I ? o : 0;   // (I % 4 == 0: a piece is inside or outside as a whole)
            if (a.bias) bv[i] = ld4(a.bias + oc);
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                const long cc = colok[t] ? col[t] : 0;
                if (a.accumulate) yv[i][t] = ld4(a.asrc + (cc / a.adiv) * a.ldas + oc);
                if (a.mask) mv[i][t] = ld4(a.mask + cc * a.ldm + oc);
            }
        }
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const int o = (it0 + i) * 16 + 4 * q;
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                f32x4 v = acc[i][t];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = v[r];
                    if (a.bias) x += bv[i][r];
                    if (a.accumulate) x += yv[i][t][r];
                    x = act_apply(x, a.act);
                    if (a.mask && !(mv[i][t][r] > 0.f)) x = 0.f;
                    v[r] = x;
                }
                if (colok[t] && o < a.I) st4(a.Y + (long)col[t] * a.ldy + o, v);
            }
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        if (!colok[t]) continue;
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const int o = (it0 + i) * 16 + 4 * q;
            if (o >= a.I) continue;
            float* yp = a.Y + (long)col[t] * a.ldy + o;
            f32x4 v = acc[i][t];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (o + r >= a.I) continue;
                float x = v[r];
                if (a.bias) x += a.bias[o + r];
                if (a.accumulate) x += a.asrc[(long)(col[t] / a.adiv) * a.ldas + o + r];
                x = act_apply(x, a.act);
                if (a.mask && !(a.mask[(long)col[t] * a.ldm + o + r] > 0.f)) x = 0.f;
                v[r] = x;
            }
            if (a.yvec && o + 3 < a.I) st4(yp, v);
            else
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (o + r < a.I) yp[r] = v[r];
        }
    }
}

template <int RT, int CT, int U>
__global__ __launch_bounds__(256) void tlinear_kernel(TLin a, int ksplit) {
    __shared__ f32x4 part[3][RT * CT][64];
    tlinear_body<RT, CT, U>(a, ksplit, blockIdx.x, blockIdx.y, part);
}

// ---------------------------------------------------------------------------------------------------
// tgemm (round 4): the three products of nn.Linear's training step at BATCH sizes (more than 2048 columns: NBA batches, scene batches)
//     forward          Y [c][i]  = act(sum_j X[c / xdiv][j] W[i][j] + b[i])         (train.py:83 -> model/STTODE.py:553-568)
//     input gradient   dX[c][k]  = mask(sum_n dY[c][n] W[n][k] (+ dX[c][k]))
//     weight gradient  dW[n][k] += sum_c dY[c][n] X[c / xdiv][k],   db[n] += sum_c dY[c][n]
// as ONE LDS-tiled kernel  C[m][n] (+)= sum_k A(m, k) B(n, k).  The generic kernels above read every MFMA operand straight from global
// memory with per-lane 16-byte (or, for the transposed operands, four strided 4-byte) loads and no look-ahead: 0.35-0.39 of the fp32 MFMA
// peak, bound by operand-load latency (profiles/r03).  Here a workgroup owns a 64 x 64 tile of C; per 32-deep k tile all 256 threads
// fetch the two 64 x 32 operand panels with coalesced 16-byte loads -- along k where k is the contiguous index, along the row index and
// transposed on the way into LDS where it is not -- one k tile AHEAD of the MFMAs (registers -> the other LDS buffer), and every wave
// computes a 32 x 32 block with v_mfma_f32_32x32x2_f32 from 16-byte LDS reads (rows padded to 36 words: conflict-free).
// MFMA step 4g + r consumes the k pair (8g + r, 8g + 4 + r): both operands are read as f32x4 at k = 8g + 4h + (0..3) by lane half h.
// ---------------------------------------------------------------------------------------------------
struct TG {
    const float* A; const float* B; float* C;
    long lda, ldb, ldc;
    int M, N, Kt;            // C is M x N, the reduction runs over Kt
    int adiv, bkdiv;         // row of A = m / adiv (A not transposed: tlinear's broadcast rows); reduction index of B = k / bkdiv (twgrad's X rows)
    int ones_row;            // twgrad: B(n == ones_row, .) = 1 -- the bias gradient rides as one more column of dW; -1: none
    int avec, bvec, cvec;    // operand / result rows 16-byte aligned
    int evec;                // mode 0: N % 4 == 0 and C, bias, mask 16-byte aligned -- the epilogue runs on 16-byte pieces
    int fast;                // operands fit tg_fetch_fast (tg_fast below)
    long long* dbg;          // diagnostic (sttode_tgemm_debug_buffer): [workgroup][4] stamps of the 100 MHz clock -- start, first tile in LDS, reduction done, end
    const float* bias; const float* mask; long ldm; int act, accumulate;   // mode 0 (tlinear) epilogue
    const float* asrc; long ldas; int acdiv;                               // `accumulate` adds row (m / acdiv) of asrc (TLin::asrc)
    float* db; float* scratch; int S, kchunk, mode;                        // mode 1 (twgrad): split s = blockIdx.z reduces k in [s kchunk, (s + 1) kchunk)
};

typedef float tg_f32x16 __attribute__((ext_vector_type(16)));

// one 64 x 32 operand panel: 2 x f32x4 per thread.  T = false: memory is [row][k] (k contiguous): thread -> (row, 4 k); T = true: memory is
// [k][row] (row contiguous): thread -> (k, 4 rows), transposed when stored to LDS.
template <bool T>
static __device__ __forceinline__ void tg_fetch(f32x4 (&v)[2], const float* __restrict__ src, long ld, int row0, int rows, int rdiv, int k0, int kend,
                                                int kdiv, int ones_row, bool vec) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int idx = (int)threadIdx.x + 256 * p;
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        if (!T) {
            const int row = row0 + (idx >> 3), k = k0 + (idx & 7) * 4;
            if (row < rows && k < kend) {
                const float* q = src + (long)(row / rdiv) * ld + k;
                if (vec && k + 3 < kend) x = ld4(q);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (k + e < kend) x[e] = q[e];
                }
            }
        } else {
            const int k = k0 + (idx >> 4), row = row0 + (idx & 15) * 4;
            if (k < kend && row < rows + (ones_row >= 0 ? 1 : 0)) {
                const float* q = src + (long)(k / kdiv) * ld + row;
                if (vec && row + 3 < rows) x = ld4(q);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (row + e < rows) x[e] = q[e];
                        else if (row + e == ones_row) x[e] = 1.0f;
                }
            }
        }
        v[p] = x;
    }
}
// LDS panel of one operand (2304 floats).  T = false: [64 rows][36] (k contiguous, rows padded to 36 words: 16-byte stores and 16-byte
// fragment reads, conflict-free).  T = true: [32 k][68] (rows contiguous: the transposed source's 16-byte pieces are stored as they are;
// the fragment is read as four 4-byte words, lanes on consecutive rows -- transposing on the way IN, four scalar stores at a stride of
// 36 words, is an 8-way bank conflict: measured 36 us per weight gradient against 34.5 us for the generic kernel).
#define TG_PANEL 2304
template <bool T>
static __device__ __forceinline__ void tg_store(const f32x4 (&v)[2], float* S) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int idx = (int)threadIdx.x + 256 * p;
        if (!T) *reinterpret_cast<f32x4*>(S + (idx >> 3) * 36 + (idx & 7) * 4) = v[p];
        else *reinterpret_cast<f32x4*>(S + (idx >> 4) * 68 + (idx & 15) * 4) = v[p];
    }
}
// the fragment of MFMA steps 4q .. 4q + 3 for row `row` (0..63) of the panel: k = 8q + 4h + (0..3)
template <bool T>
static __device__ __forceinline__ f32x4 tg_frag(const float* S, int row, int q, int h) {
    if (!T) return *reinterpret_cast<const f32x4*>(S + row * 36 + 8 * q + 4 * h);
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = S[(8 * q + 4 * h + e) * 68 + row];
    return r;
}

// Branch-free form of tg_fetch for the shapes the training step is made of (g.fast: 16-byte aligned operands, no broadcast rows, the
// contiguous index a multiple of 4): addresses are clamped into the operand instead of tested, pieces beyond [.., kend) are zeroed by a
// select.  Without branches the compiler counts outstanding loads exactly, and a tile can be requested TWO tiles ahead: a 64 x 64 tile
// needs 16 KB per 32-deep step for 262 kFLOP -- at the MFMA rate that is 38 GB/s per CU, 9.6 TB/s chip-wide out of L2 -- and with one
// tile in flight per workgroup the step lasted one loaded L2 round trip instead (measured 30 us for a product with 12 us of MFMA).
template <bool T>
static __device__ __forceinline__ void tg_fetch_fast(f32x4 (&v)[2], const float* __restrict__ src, long ld, int row0, int rows, int k0, int kend) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int idx = (int)threadIdx.x + 256 * p;
        if (!T) {
            const int row = row0 + (idx >> 3), k = k0 + (idx & 7) * 4;
            v[p] = ld4(src + (long)(row < rows ? row : rows - 1) * ld + (k < kend ? k : kend - 4));
        } else {
            const int k = k0 + (idx >> 4), row = row0 + (idx & 15) * 4;
            v[p] = ld4(src + (long)(k < kend ? k : kend - 1) * ld + (row + 3 < rows ? row : rows - 4));
        }
    }
}
// ... and what the bounds tests would have done, applied when the tile goes to LDS (not at the request: the selects would wait for the data)
template <bool T>
static __device__ __forceinline__ void tg_store_fast(f32x4 (&v)[2], float* S, int row0, int k0, int kend, int ones_row) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int idx = (int)threadIdx.x + 256 * p;
        f32x4 x = v[p];
        if (!T) {
            if (!(k0 + (idx & 7) * 4 < kend)) x = splat4(0.f);
            *reinterpret_cast<f32x4*>(S + (idx >> 3) * 36 + (idx & 7) * 4) = x;
        } else {
            const int row = row0 + (idx & 15) * 4;
            if (ones_row >= 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (row + e == ones_row) x[e] = 1.0f;
            }
            if (!(k0 + (idx >> 4) < kend)) x = splat4(0.f);
            *reinterpret_cast<f32x4*>(S + (idx >> 4) * 68 + (idx & 15) * 4) = x;
        }
    }
}

template <bool AT, bool BT>
static __device__ __forceinline__ void tg_mma_tile(tg_f32x16& acc, const float* Sa, const float* Sb, int mt, int nt, int c, int h) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 b = tg_frag<AT>(Sa, mt * 32 + c, q, h);     // MFMA columns = m
        const f32x4 a = tg_frag<BT>(Sb, nt * 32 + c, q, h);     // MFMA rows = n
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], b[r], acc, 0, 0, 0);
    }
}

static __device__ __forceinline__ void tg_epilogue(const TG& g, const tg_f32x16& acc, int m0, int n0, int mt, int nt, int c, int h, int bz);
// one 64 x 64 tile of C (tile indices bx, by; bz: the split of the reduction in mode 1) by the calling workgroup
template <bool AT, bool BT>
static __device__ __forceinline__ void tgemm_body(const TG& g, int bx, int by, int bz, float (*As)[TG_PANEL], float (*Bs)[TG_PANEL]) {
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5, wave = threadIdx.x >> 6;
    const int m0 = bx * 64, n0 = by * 64;
    const int kbeg = g.mode == 1 ? bz * g.kchunk : 0;
    const int kend = g.mode == 1 ? (kbeg + g.kchunk < g.Kt ? kbeg + g.kchunk : g.Kt) : g.Kt;
    const int mt = wave & 1, nt = wave >> 1;
    tg_f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    f32x4 va[2], vb[2];
    const int brows = g.N - (g.ones_row >= 0 ? 1 : 0);
    const int wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (g.dbg && threadIdx.x == 0) g.dbg[4 * wg] = __builtin_amdgcn_s_memrealtime();
    if (g.fast && kbeg < kend) {
        // two tiles ahead: register set 0 / 1 holds tile t / t + 1 on its way to LDS buffer 0 / 1 (requests beyond the last tile read one
        // clamped piece and give zeros); lds_barrier(): the workgroup barrier WITHOUT the vmcnt(0) of __syncthreads(), which would drain
        // the requests of the tile after next at every step
        f32x4 ua[2], ub[2];
        const int P = (kend - kbeg + 31) / 32;
        tg_fetch_fast<AT>(va, g.A, g.lda, m0, g.M, kbeg, kend);
        tg_fetch_fast<BT>(vb, g.B, g.ldb, n0, brows, kbeg, kend);
        tg_fetch_fast<AT>(ua, g.A, g.lda, m0, g.M, kbeg + 32, kend);
        tg_fetch_fast<BT>(ub, g.B, g.ldb, n0, brows, kbeg + 32, kend);
        __builtin_amdgcn_sched_barrier(0);
        tg_store_fast<AT>(va, As[0], m0, kbeg, kend, -1);
        tg_store_fast<BT>(vb, Bs[0], n0, kbeg, kend, g.ones_row);
        lds_barrier();
        if (g.dbg && threadIdx.x == 0) g.dbg[4 * wg + 1] = __builtin_amdgcn_s_memrealtime();
        int t = 0;
        for (; t + 2 <= P; t += 2) {
            const int k1 = kbeg + 32 * (t + 1);
            tg_fetch_fast<AT>(va, g.A, g.lda, m0, g.M, k1 + 32, kend);
            tg_fetch_fast<BT>(vb, g.B, g.ldb, n0, brows, k1 + 32, kend);
            __builtin_amdgcn_sched_barrier(0);               // (the scheduler otherwise sinks the requests to their first use, behind the MFMAs)
            tg_mma_tile<AT, BT>(acc, As[0], Bs[0], mt, nt, c, h);
            __builtin_amdgcn_sched_barrier(0);
            tg_store_fast<AT>(ua, As[1], m0, k1, kend, -1);
            tg_store_fast<BT>(ub, Bs[1], n0, k1, kend, g.ones_row);
            lds_barrier();
            tg_fetch_fast<AT>(ua, g.A, g.lda, m0, g.M, k1 + 64, kend);
            tg_fetch_fast<BT>(ub, g.B, g.ldb, n0, brows, k1 + 64, kend);
            __builtin_amdgcn_sched_barrier(0);
            tg_mma_tile<AT, BT>(acc, As[1], Bs[1], mt, nt, c, h);
            __builtin_amdgcn_sched_barrier(0);
            tg_store_fast<AT>(va, As[0], m0, k1 + 32, kend, -1);
            tg_store_fast<BT>(vb, Bs[0], n0, k1 + 32, kend, g.ones_row);
            lds_barrier();
        }
        if (t < P) tg_mma_tile<AT, BT>(acc, As[0], Bs[0], mt, nt, c, h);
    } else {
    tg_fetch<AT>(va, g.A, g.lda, m0, g.M, g.adiv, kbeg, kend, 1, -1, g.avec);
    tg_fetch<BT>(vb, g.B, g.ldb, n0, brows, 1, kbeg, kend, g.bkdiv, g.ones_row, g.bvec);
    tg_store<AT>(va, As[0]);
    tg_store<BT>(vb, Bs[0]);
    __syncthreads();
    int buf = 0;
    for (int k0 = kbeg; k0 < kend; k0 += 32) {
        const bool more = k0 + 32 < kend;
        if (more) {   // the next k tile travels while this one is multiplied
            tg_fetch<AT>(va, g.A, g.lda, m0, g.M, g.adiv, k0 + 32, kend, 1, -1, g.avec);
            tg_fetch<BT>(vb, g.B, g.ldb, n0, brows, 1, k0 + 32, kend, g.bkdiv, g.ones_row, g.bvec);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b = tg_frag<AT>(As[buf], mt * 32 + c, q, h);     // MFMA columns = m
            const f32x4 a = tg_frag<BT>(Bs[buf], nt * 32 + c, q, h);     // MFMA rows = n
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], b[r], acc, 0, 0, 0);
        }
        if (more) {
            tg_store<AT>(va, As[buf ^ 1]);
            tg_store<BT>(vb, Bs[buf ^ 1]);
        }
        __syncthreads();
        buf ^= 1;
    }
    }
    if (g.dbg && threadIdx.x == 0) g.dbg[4 * wg + 2] = __builtin_amdgcn_s_memrealtime();
    tg_epilogue(g, acc, m0, n0, mt, nt, c, h, bz);
    if (g.dbg) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) g.dbg[4 * wg + 3] = __builtin_amdgcn_s_memrealtime();
    }
}
static __device__ __forceinline__ void tg_epilogue(const TG& g, const tg_f32x16& acc, int m0, int n0, int mt, int nt, int c, int h, int bz) {
    // lane (c, h): m = m0 + 32 mt + c; register 4a + b <-> n = n0 + 32 nt + 8a + 4h + b
    const int m = m0 + mt * 32 + c;
    if (m >= g.M) return;
    if (g.mode == 1 && g.S > 1) {
        // split reduction: the partial tile goes to scratch [split][M][N]; the splits are added in order by a reduction launch (deterministic).
        // (Combining inside the launch -- last workgroup of a tile, ticket counter -- was built and measured: the agent-scope release every
        // workgroup needs before its ticket writes the whole L2 back on this part, 183 us per weight gradient against 36 us + 10 us.)
        float* part = g.scratch + ((long)bz * g.M + m) * g.N;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int n = n0 + nt * 32 + 8 * a + 4 * h;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (n + e < g.N) part[n + e] = acc[4 * a + e];
        }
        return;
    }
    if (g.mode == 0 && g.evec) {
        // every operand of the epilogue in 16-byte pieces, all requested before the first is used (element by element under its bounds
        // check each load is followed by its own wait: 16-48 dependent L2 round trips per lane, 5-10 us of a 25-us product)
        f32x4 bv[4], yv[4], mv[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int n = n0 + nt * 32 + 8 * a + 4 * h;
            const int nc = n < g.N ? n : 0;                   // (N % 4 == 0: a piece is inside or outside as a whole)
            if (g.bias) bv[a] = ld4(g.bias + nc);
            if (g.accumulate) yv[a] = ld4(g.asrc + (long)(m / g.acdiv) * g.ldas + nc);
            if (g.mask) mv[a] = ld4(g.mask + (long)m * g.ldm + nc);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int n = n0 + nt * 32 + 8 * a + 4 * h;
            f32x4 v = {acc[4 * a], acc[4 * a + 1], acc[4 * a + 2], acc[4 * a + 3]};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = v[e];
                if (g.bias) x += bv[a][e];
                if (g.accumulate) x += yv[a][e];
                x = act_apply(x, g.act);
                if (g.mask && !(mv[a][e] > 0.f)) x = 0.f;
                v[e] = x;
            }
            if (n < g.N) st4(g.C + (long)m * g.ldc + n, v);
        }
        return;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int n = n0 + nt * 32 + 8 * a + 4 * h;
        if (n >= g.N) continue;
        f32x4 v = {acc[4 * a], acc[4 * a + 1], acc[4 * a + 2], acc[4 * a + 3]};
        if (g.mode == 0) {
            float* yp = g.C + (long)m * g.ldc + n;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (n + e >= g.N) continue;
                float x = v[e];
                if (g.bias) x += g.bias[n + e];
                if (g.accumulate) x += g.asrc[(long)(m / g.acdiv) * g.ldas + n + e];
                x = act_apply(x, g.act);
                if (g.mask && !(g.mask[(long)m * g.ldm + n + e] > 0.f)) x = 0.f;
                v[e] = x;
            }
            if (g.cvec && n + 3 < g.N) st4(yp, v);
            else
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < g.N) yp[e] = v[e];
        } else {
            const int K = g.N - 1;   // the last column of C is the bias gradient
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (n + e >= g.N) continue;
                if (n + e < K) g.C[(long)m * g.ldc + n + e] += v[e];        // (S == 1; the split case returned above)
                else if (g.db) g.db[m] += v[e];
            }
        }
    }
}

template <bool AT, bool BT>
__global__ __launch_bounds__(256) void tgemm_kernel(TG g) {
    __shared__ __attribute__((aligned(16))) float As[2][TG_PANEL];
    __shared__ __attribute__((aligned(16))) float Bs[2][TG_PANEL];
    tgemm_body<AT, BT>(g, blockIdx.x, blockIdx.y, blockIdx.z, As, Bs);
}

// One launch for a layer's backward at batch sizes: blocks [0, nw) are the tiles x splits of the weight gradient dW = dY^T [X | 1],
// the remaining blocks the tiles of the input gradient dX = dY W -- two products that share nothing but dY and have the chip to themselves
// for 25-30 us each when launched one after the other (profiles/r04/train_shapes_before.txt: 0.31 of peak for the pair + its reduction).
__global__ __launch_bounds__(256) void tgemm_bwd_kernel(TG gw, int gxw, int gyw, int nw, TG gx, int gxx) {
    __shared__ __attribute__((aligned(16))) float As[2][TG_PANEL];
    __shared__ __attribute__((aligned(16))) float Bs[2][TG_PANEL];
    int id = blockIdx.x;
    if (id < nw) tgemm_body<true, true>(gw, id % gxw, (id / gxw) % gyw, id / (gxw * gyw), As, Bs);
    else {
        id -= nw;
        tgemm_body<false, true>(gx, id % gxx, id / gxx, 0, As, Bs);
    }
}

// Several independent products in ONE launch (sttode_tgemm_group): a 2-GFLOP product is one round of ~930 workgroups on 1024 slots and
// pays ~7 us of start skew, first tile and store burst around 16 us of MFMA (profiles/r04/tgemm_workgroup_trace.txt); with the decoder's
// decoder_x / decoder_y layers (same input, separate weights) or a layer's dX / dW side by side, a later product's workgroups start as an
// earlier one's finish.  Problem p owns blocks [blk0[p], blk0[p + 1]); kind: 0 forward, 1 input gradient, 2 weight gradient.
#define TG_MULTI_MAX 4
struct TGMulti { TG g[TG_MULTI_MAX]; int blk0[TG_MULTI_MAX + 1]; int gx[TG_MULTI_MAX], gy[TG_MULTI_MAX], kind[TG_MULTI_MAX]; int n; };
__global__ __launch_bounds__(256) void tgemm_multi_kernel(TGMulti M) {
    __shared__ __attribute__((aligned(16))) float As[2][TG_PANEL];
    __shared__ __attribute__((aligned(16))) float Bs[2][TG_PANEL];
    int p = 0;
    while (p + 1 < M.n && (int)blockIdx.x >= M.blk0[p + 1]) ++p;
    p = __builtin_amdgcn_readfirstlane(p);
    // the problem's descriptor out of the kernel-argument segment (uniform index: scalar loads; indexing the by-value struct would park
    // all four descriptors in registers first)
    TG g;
    {
        const __attribute__((address_space(4))) int* src = (const __attribute__((address_space(4))) int*)(
            (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(TGMulti, g) + (size_t)p * sizeof(TG));
        int* dst = reinterpret_cast<int*>(&g);
#pragma unroll
        for (unsigned i = 0; i < sizeof(TG) / 4; ++i) dst[i] = src[i];
    }
    const int id = (int)blockIdx.x - M.blk0[p], gx = M.gx[p], gy = M.gy[p], kind = M.kind[p];
    const int bx = id % gx, by = (id / gx) % gy, bz = id / (gx * gy);
    if (kind == 0) tgemm_body<false, false>(g, bx, by, bz, As, Bs);
    else if (kind == 1) tgemm_body<false, true>(g, bx, by, bz, As, Bs);
    else tgemm_body<true, true>(g, bx, by, bz, As, Bs);
}

// Deferred reductions of split weight gradients: up to TG_RED_MAX of them are added into their dW / db by ONE launch (twenty 10-us launches
// per NBA-size step otherwise).  Item i owns blocks [blk0[i], blk0[i + 1]).
#define TG_RED_MAX 16
struct TGRedItem { const float* part; float* dW; float* db; long ldw; long per; int K1, S, blk0; };
struct TGRed { TGRedItem it[TG_RED_MAX]; int n; };
__global__ __launch_bounds__(256) void tgemm_reduce_kernel(TGRed r) {
    int i = 0;
    while (i + 1 < r.n && (int)blockIdx.x >= r.it[i + 1].blk0) ++i;
    const TGRedItem& t = r.it[i];
    const long e = (long)(blockIdx.x - t.blk0) * 256 + threadIdx.x;
    if (e >= t.per) return;
    const int n = (int)(e / t.K1), k = (int)(e % t.K1);
    float tot = 0.f;
    for (int s = 0; s < t.S; ++s) tot += t.part[(long)s * t.per + e];
    if (k < t.K1 - 1) t.dW[(long)n * t.ldw + k] += tot;
    else if (t.db) t.db[n] += tot;
}

#ifndef TLIN_MEDIUM_BELOW
#define TLIN_MEDIUM_BELOW 4096   // throughput-mode wave count below which the 32 x 32 tiling is used instead
#endif

// columns above which the LDS-tiled kernel takes over from the generic ones (STTODE_TGEMM_MIN_COLS: experiments; measured at 1024: the
// one-scene step the same 1.04-1.08 ms, and the NBA 16 x 11 gradient yardstick fails -- profiles/r04/tgemm_min_cols_ab.txt)
// Round 5: the BACKWARD products (input gradient, weight gradient) switch at 600 columns (STTODE_TGEMM_MIN_COLS_BWD).  With the decoder's
// backward over the live columns only, an NBA-size step's backward products have 2 n = 704 columns -- below 2048, on the generic kernels:
// 1.155 -> 1.086 ms per step with the LDS-tiled kernel (profiles/r05/train_tgemm_min_cols_ab.txt, measured with both thresholds at 600;
// the forward products keep 2048: at 600 the forward of a 32-agent scene (672 columns) changes its summation order and three gradient
// yardsticks on the known ill-conditioned rows move from 0.8 to 1.05-1.8 of their bounds, for no gain at one scene per step).
static inline int tg_min_cols() {
    static const int v = getenv("STTODE_TGEMM_MIN_COLS") ? atoi(getenv("STTODE_TGEMM_MIN_COLS")) : 2048;
    return v;
}
static inline int tg_min_cols_bwd() {
    static const int v = getenv("STTODE_TGEMM_MIN_COLS_BWD") ? atoi(getenv("STTODE_TGEMM_MIN_COLS_BWD")) : 600;
    return v;
}
static inline int aligned16(const void* p, long ld) { return (((size_t)p) % 16 == 0) && (ld % 4 == 0); }
// tg_fetch_fast: aligned operands without broadcast rows; the contiguous index of each operand (k, or the row index of a transposed one)
// a multiple of 4 and at least 4; every split of the reduction a multiple of 4 long
static inline int tg_fast(const TG& g, bool AT, bool BT) {
    const int brows = g.N - (g.ones_row >= 0 ? 1 : 0);
    if (!g.avec || !g.bvec || g.adiv != 1 || g.bkdiv != 1 || g.Kt < 4) return 0;
    if (AT ? (g.M % 4 != 0 || g.M < 4) : g.Kt % 4 != 0) return 0;
    if (BT ? (brows % 4 != 0 || brows < 4) : g.Kt % 4 != 0) return 0;
    return 1;
}
static long long* g_tg_dbg = nullptr;
extern "C" int sttode_tgemm_debug_buffer(void* p) { g_tg_dbg = (long long*)p; return 0; }   // diagnostic: >= grid * 4 int64 (NULL: off); stand-alone launches only
static inline int tg_evec(const TG& g) {
    return g.N % 4 == 0 && aligned16(g.C, g.ldc) && (!g.bias || aligned16(g.bias, 4)) && (!g.mask || aligned16(g.mask, g.ldm)) &&
           (!g.accumulate || aligned16(g.asrc, g.ldas));
}

// ---- split weight gradients of the LDS-tiled kernel: where the partial sums go and when they are added up --------------------------------
// Default: each weight gradient is followed by its own reduction launch (partial sums in the call's scratch).  Between
// sttode_twgrad_defer(1, buf, floats) and sttode_twgrad_defer(0, ..) (the training engine brackets a backward pass with them) the partial
// sums are bump-allocated from `buf` instead -- a buffer nothing else writes -- and the reductions run as ONE launch per TG_RED_MAX
// gradients, or earlier: buf full, a destination that is already pending, another stream.  Host-side state only; inside a hipGraph capture
// the flush is captured like any other launch.
static std::mutex g_red_mu;
// (per HOST THREAD: a training step -- its group brackets, its backward pass with the deferred reductions -- is issued by one thread; two
// threads that train two models must not see each other's open group)
static thread_local struct { TGRed r; int blocks; long used; float* buf; long cap; void* stream; bool defer; } g_red = {{}, 0, 0, nullptr, 0, nullptr, false};

static void tg_red_flush_locked() {
    if (g_red.r.n > 0) hipLaunchKernelGGL(tgemm_reduce_kernel, dim3((unsigned)g_red.blocks), dim3(256), 0, (hipStream_t)g_red.stream, g_red.r);
    g_red.r.n = 0; g_red.blocks = 0; g_red.used = 0;
}

// ---- grouped launches (sttode_tgemm_group): batch-size products queued between group(1) and group(0) leave as ONE tgemm_multi_kernel launch ----
static thread_local struct {
    TGMulti M; int gz[TG_MULTI_MAX];
    struct { float* dW; long ldw; float* db; } post[TG_MULTI_MAX];   // weight gradients: their split sums are queued for reduction AFTER the launch
    void* stream; bool on;
} g_grp = {};
struct TWg;
static void tg_wgrad_done(const TG& g, float* dW, long ldw, float* db);
static void ts_group_launch_locked();   // the scene-size queue (defined below the kernel it launches)
static void ts_group_launch_locked_forget();
static void ew_group_launch_locked();   // the element-wise queue (defined with its kernel)
static void ew_group_forget();
static void ts_submit(const TLin& a, const TWg* w, int ksplit, int gxA, int nA, int gxW, int gyW, int nB, void* stream);
static void tg_group_launch_locked() {
    TGMulti& M = g_grp.M;
    if (M.n == 0) return;
    hipLaunchKernelGGL(tgemm_multi_kernel, dim3((unsigned)M.blk0[M.n]), dim3(256), 0, (hipStream_t)g_grp.stream, M);
    const int n = M.n;
    M.n = 0;
    for (int i = 0; i < n; ++i)
        if (M.kind[i] == 2) tg_wgrad_done(M.g[i], g_grp.post[i].dW, g_grp.post[i].ldw, g_grp.post[i].db);
}
// queue (group mode) or launch one product; kind: 0 forward, 1 input gradient, 2 weight gradient (its split sums: dW, ldw, db)
static void tg_submit(const TG& g, int kind, int gx, int gy, int gz, void* stream, float* dW = nullptr, long ldw = 0, float* db = nullptr) {
    if (g_grp.on) {
        TGMulti& M = g_grp.M;
        if (M.n == TG_MULTI_MAX || (M.n > 0 && g_grp.stream != stream)) tg_group_launch_locked();
        const int i = M.n++;
        if (i == 0) M.blk0[0] = 0;
        M.g[i] = g; M.g[i].dbg = nullptr; M.kind[i] = kind; M.gx[i] = gx; M.gy[i] = gy; g_grp.gz[i] = gz;
        M.blk0[i + 1] = M.blk0[i] + gx * gy * gz;
        g_grp.post[i].dW = dW; g_grp.post[i].ldw = ldw; g_grp.post[i].db = db;
        g_grp.stream = stream;
        return;
    }
    const dim3 grid(gx, gy, gz);
    if (kind == 0) hipLaunchKernelGGL((tgemm_kernel<false, false>), grid, dim3(256), 0, (hipStream_t)stream, g);
    else if (kind == 1) hipLaunchKernelGGL((tgemm_kernel<false, true>), grid, dim3(256), 0, (hipStream_t)stream, g);
    else {
        hipLaunchKernelGGL((tgemm_kernel<true, true>), grid, dim3(256), 0, (hipStream_t)stream, g);
        tg_wgrad_done(g, dW, ldw, db);
    }
}
extern "C" int sttode_tgemm_group(int on) {
    std::lock_guard<std::mutex> lk(g_red_mu);
    if (on < 0) { g_grp.M.n = 0; ts_group_launch_locked_forget(); ew_group_forget(); }   // error paths: forget what is queued
    tg_group_launch_locked();
    ts_group_launch_locked();
    ew_group_launch_locked();
    g_grp.on = on > 0;
    if (int rc = stt_trunk_group(on)) return rc;   // (a fused trunk forward queued in this group, train_trunk.hip)
    STT_HIP(hipGetLastError());
    return 0;
}

// fills g for dW (+)= dY^T [X | 1] with the reduction over the columns split S ways (about want_blocks workgroups); false: no room for partial sums
static bool tg_wgrad_fill(TG& g, const float* dY, long ldy, const float* X, long ldx, int xdiv, float* dW, long ldw, float* db, int cols, int N,
                          int K, float* scratch, long scratch_floats, int want_blocks, void* stream) {
    const long per = (long)N * (K + 1);
    const int tiles = ((N + 63) / 64) * ((K + 1 + 63) / 64);
    int S = (want_blocks + tiles - 1) / tiles;
    if (S > 64) S = 64;
    if (S > (cols + 127) / 128) S = (cols + 127) / 128;      // >= 128 columns per split
    if (S < 1) S = 1;
    if (g_red.r.n > 0 && (g_red.stream != stream || !g_red.defer)) { tg_group_launch_locked(); tg_red_flush_locked(); }
    const bool defer = g_red.defer && g_red.buf && g_red.cap >= 2 * per;
    if (defer && g_red.r.n > 0) {
        bool again = g_red.r.n == TG_RED_MAX || (S > 1 && g_red.used + per * S > g_red.cap);
        for (int i = 0; i < g_red.r.n && !again; ++i) {   // one launch adds every pending gradient: none of them may share a destination
            const TGRedItem& t = g_red.r.it[i];
            const float* lo = t.dW; const float* hi = t.dW + (t.per / t.K1) * t.ldw;
            again = (dW < hi && lo < dW + (long)N * ldw) || (db && db == t.db);
        }
        if (again) { tg_group_launch_locked(); tg_red_flush_locked(); }   // (also in front of an unsplit gradient to a pending destination: it adds into dW itself)
    }
    if (defer && g_grp.M.n > 0) {         // queued, not yet launched gradients of the open group count as pending destinations too
        bool again = S > 1 && g_red.used + per * S > g_red.cap;
        for (int i = 0; i < g_grp.M.n && !again; ++i)
            if (g_grp.M.kind[i] == 2) {
                const float* lo = g_grp.post[i].dW; const float* hi = lo + (long)g_grp.M.g[i].M * g_grp.post[i].ldw;
                again = (dW < hi && lo < dW + (long)N * ldw) || (db && db == g_grp.post[i].db);
            }
        if (again) { tg_group_launch_locked(); tg_red_flush_locked(); }
    }
    if (!defer && g_grp.M.n > 0) {        // without a buffer of its own every split gradient uses the call's scratch: one per launch
        for (int i = 0; i < g_grp.M.n; ++i)
            if (g_grp.M.kind[i] == 2) { tg_group_launch_locked(); break; }
    }
    float* part = defer ? g_red.buf + g_red.used : scratch;
    const long room = defer ? g_red.cap - g_red.used : scratch_floats;
    if (S > 1 && (!part || per * S > room)) S = part ? (int)(room / per) : 1;
    if (S < 1) return false;
    g_red.stream = stream;
    g.A = dY; g.lda = ldy; g.B = X; g.ldb = ldx; g.C = dW; g.ldc = ldw;
    g.M = N; g.N = K + 1; g.Kt = cols; g.adiv = 1; g.bkdiv = xdiv; g.ones_row = K;
    g.avec = aligned16(dY, ldy); g.bvec = aligned16(X, ldx); g.cvec = 0;
    g.bias = nullptr; g.mask = nullptr; g.ldm = 0; g.act = 0; g.accumulate = 0; g.asrc = nullptr; g.ldas = 0; g.acdiv = 1;
    g.db = db; g.scratch = part; g.S = S; g.mode = 1; g.evec = 0; g.fast = tg_fast(g, true, true); g.dbg = nullptr;
    if (defer && S > 1) g_red.used += per * S;     // reserved now: a second gradient of the same group must not get the same piece
    g.kchunk = ((cols + S - 1) / S + 31) / 32 * 32;
    return true;
}
// after the launch that wrote g's partial sums: queue (or run) their reduction
static void tg_wgrad_done(const TG& g, float* dW, long ldw, float* db) {
    if (g.S <= 1) return;
    const long per = (long)g.M * g.N;
    if (g_red.r.n == TG_RED_MAX) tg_red_flush_locked();
    TGRedItem& t = g_red.r.it[g_red.r.n++];
    t.part = g.scratch; t.dW = dW; t.db = db; t.ldw = ldw; t.per = per; t.K1 = g.N; t.S = g.S; t.blk0 = g_red.blocks;
    g_red.blocks += (int)((per + 255) / 256);
    const bool in_buf = g_red.buf && g.scratch >= g_red.buf && g.scratch < g_red.buf + g_red.cap;
    if (!g_red.defer || !in_buf) tg_red_flush_locked();
}
extern "C" int sttode_twgrad_defer(int on, float* buf, long floats) {
    std::lock_guard<std::mutex> lk(g_red_mu);
    if (on < 0) { g_red.r.n = 0; g_red.blocks = 0; g_red.used = 0; }      // error paths: forget what is pending
    tg_red_flush_locked();
    g_red.defer = on > 0 && buf && floats > 0;
    g_red.buf = g_red.defer ? buf : nullptr; g_red.cap = g_red.defer ? floats : 0;
    STT_HIP(hipGetLastError());
    return 0;
}
extern "C" int sttode_twgrad_flush(void) {
    std::lock_guard<std::mutex> lk(g_red_mu);
    tg_red_flush_locked();
    STT_HIP(hipGetLastError());
    return 0;
}

static int tlinear_impl(const float* X, long ldx, int xdiv, const float* W, long ldw, int trans, const float* bias, const float* mask, long ldm,
                        float* Y, long ldy, int cols, int J, int I, int act, int accumulate, const float* asrc, long ldas, int adiv, void* stream);
extern "C" int sttode_tlinear(const float* X, long ldx, int xdiv, const float* W, long ldw, int trans, const float* bias,
                              const float* mask, long ldm, float* Y, long ldy, int cols, int J, int I, int act, int accumulate,
                              void* stream) {
    return tlinear_impl(X, ldx, xdiv, W, ldw, trans, bias, mask, ldm, Y, ldy, cols, J, I, act, accumulate, Y, ldy, 1, stream);
}
// Y[c] = act(W X[c] + tab[c / tdiv] (+ bias)): nn.Linear whose input is cat(shared, own) with the shared part's product -- the same for tdiv
// consecutive columns -- precomputed as a table (the decoder MLPs' layer 1, model/utils.py:86-95 on cat(past_feature_rep, z, state),
// model/STTODE.py:71-75,322-328: tab = W1[:, pf] pf + b1 per AGENT, W = W1[:, z | state]; half the layer's products, as in the inference chain)
extern "C" int sttode_tlinear_tab(const float* X, long ldx, const float* W, long ldw, const float* bias, const float* tab, long ldt, int tdiv,
                                  float* Y, long ldy, int cols, int J, int I, int act, void* stream) {
    STT_REQUIRE(tab && tdiv > 0 && ldt >= I, "sttode_tlinear_tab: bad table");
    return tlinear_impl(X, ldx, 1, W, ldw, 0, bias, nullptr, 0, Y, ldy, cols, J, I, act, 1, tab, ldt, tdiv, stream);
}
static int tlinear_impl(const float* X, long ldx, int xdiv, const float* W, long ldw, int trans, const float* bias, const float* mask, long ldm,
                        float* Y, long ldy, int cols, int J, int I, int act, int accumulate, const float* asrc, long ldas, int adiv, void* stream) {
    STT_REQUIRE(X && W && Y, "sttode_tlinear: null pointer");
    STT_REQUIRE(cols > 0 && J > 0 && I > 0 && xdiv > 0, "sttode_tlinear: cols, J, I, xdiv must be positive");
    STT_REQUIRE(act >= 0 && act <= 3, "sttode_tlinear: act must be 0 none | 1 relu | 2 tanh | 3 sigmoid");
    STT_REQUIRE(ldx >= J && ldy >= I && ldw >= (trans ? I : J), "sttode_tlinear: leading dimension smaller than the row length");
    TLin a;
    a.X = X; a.W = W; a.bias = bias; a.mask = mask; a.Y = Y;
    a.ldx = ldx; a.ldw = ldw; a.ldy = ldy; a.ldm = ldm;
    a.cols = cols; a.J = J; a.I = I; a.trans = trans; a.act = act; a.accumulate = accumulate; a.xdiv = xdiv;
    a.asrc = asrc; a.ldas = ldas; a.adiv = adiv;
    a.xvec = aligned16(X, ldx); a.wvec = aligned16(W, ldw); a.yvec = aligned16(Y, ldy);
    a.evec = I % 4 == 0 && a.yvec && (!bias || aligned16(bias, 4)) && (!mask || aligned16(mask, ldm)) && (!accumulate || aligned16(asrc, ldas));
    static const bool tg_on = !(getenv("STTODE_TGEMM") && atoi(getenv("STTODE_TGEMM")) == 0);   // STTODE_TGEMM=0: the generic kernels (A/B)
    if (tg_on && cols > (trans ? tg_min_cols_bwd() : tg_min_cols())) {   // batch sizes: the LDS-tiled kernel (trans: an input gradient)
        TG g;
        g.A = X; g.lda = ldx; g.B = W; g.ldb = ldw; g.C = Y; g.ldc = ldy;
        g.M = cols; g.N = I; g.Kt = J; g.adiv = xdiv; g.bkdiv = 1; g.ones_row = -1;
        g.avec = a.xvec; g.bvec = a.wvec; g.cvec = a.yvec;
        g.bias = bias; g.mask = mask; g.ldm = ldm; g.act = act; g.accumulate = accumulate; g.asrc = asrc; g.ldas = ldas; g.acdiv = adiv;
        g.db = nullptr; g.scratch = nullptr; g.S = 1; g.kchunk = 0; g.mode = 0;
        g.evec = tg_evec(g); g.fast = tg_fast(g, false, trans != 0); g.dbg = g_tg_dbg;
        // (NB = 2, 64 x 128 tiles, measured SLOWER at the NBA step's shapes -- 36-38 us against 19-25 us per product: 55 KB of LDS leave two
        // workgroups per CU to hide the panel loads instead of four -- and is not instantiated)
        std::lock_guard<std::mutex> lk(g_red_mu);
        tg_submit(g, trans ? 1 : 0, (cols + 63) / 64, (I + 63) / 64, 1, stream);
        STT_HIP(hipGetLastError());
        return 0;
    }
    if (cols <= 1024) {
        // latency mode: one 16 x 16 block per WG, reduction split over up to 4 waves (128 indices per round trip and wave)
        const int ksplit = J > 256 ? 4 : (J > 128 ? 2 : 1);
        const int blocks_per_wg = 4 / ksplit;
        dim3 grid((cols + 15) / 16, ((I + 15) / 16 + blocks_per_wg - 1) / blocks_per_wg);
        if (g_grp.on) {   // an open group: queued, leaves with the group's other scene-size layers as one launch
            std::lock_guard<std::mutex> lk(g_red_mu);
            ts_submit(a, nullptr, ksplit, (int)grid.x, (int)(grid.x * grid.y), 0, 0, 0, stream);
            STT_HIP(hipGetLastError());
            return 0;
        }
        hipLaunchKernelGGL((tlinear_kernel<1, 1, 8>), grid, dim3(256), 0, (hipStream_t)stream, a, ksplit);
    } else if ((long)((cols + 63) / 64) * ((I + 63) / 64) < TLIN_MEDIUM_BELOW) {
        // medium mode: 32 columns x 32 outputs per wave -- 4x the waves of the throughput tiling, for launches that would
        // otherwise leave most SIMDs empty (e.g. 7040 columns x 512 outputs = 880 throughput-mode waves on 1024 SIMDs)
        const int ksplit = (I <= 32 && J > 64) ? 4 : ((I <= 64 && J > 64) ? 2 : 1);
        const int outs_per_wg = 128 / ksplit;
        dim3 grid((cols + 31) / 32, (I + outs_per_wg - 1) / outs_per_wg);
        hipLaunchKernelGGL((tlinear_kernel<2, 2, 4>), grid, dim3(256), 0, (hipStream_t)stream, a, ksplit);
    } else {
        const int ksplit = (I <= 64 && J > 64) ? 4 : ((I <= 128 && J > 64) ? 2 : 1);
        const int outs_per_wg = 256 / ksplit;
        dim3 grid((cols + 63) / 64, (I + outs_per_wg - 1) / outs_per_wg);
        hipLaunchKernelGGL((tlinear_kernel<4, 4, 2>), grid, dim3(256), 0, (hipStream_t)stream, a, ksplit);
    }
    STT_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// twgrad: WG = 4 waves own a 32 x 32 block of [dW | db] for one column split; waves interleave 16-column chunks
// ---------------------------------------------------------------------------------------------------
struct TWg {
    const float* dY; const float* X; float* dW; float* db; float* scratch;
    long ldy, ldx, ldw;
    int cols, N, K, xdiv, S, chunks_per_split;
};

static __device__ __forceinline__ void twgrad_body(const TWg& a, int bx, int by, int bz, float (*red)[32][33]) {
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4, wave = threadIdx.x >> 6;
    const int n0 = bx * 32, k0 = by * 32, s = bz;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = splat4(0.f);
    const int ch0 = s * a.chunks_per_split, ch1 = min(ch0 + a.chunks_per_split, (a.cols + 15) / 16);
    for (int ch = ch0 + wave; ch < ch1; ch += 4) {
        f32x4 av[2], bv[2];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int cc = ch * 16 + 4 * q + r;
            const bool ok = cc < a.cols;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int n = n0 + 16 * i + c;
                av[i][r] = (ok && n < a.N) ? a.dY[(long)cc * a.ldy + n] : 0.f;
                const int k = k0 + 16 * i + c;
                bv[i][r] = !ok ? 0.f : (k < a.K ? a.X[(long)(cc / a.xdiv) * a.ldx + k] : (k == a.K ? 1.0f : 0.f));
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = mfma_k16(acc[i][j], av[i], bv[j]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][16 * i + 4 * q + r][16 * j + c] = acc[i][j][r];
    __syncthreads();
    for (int e = threadIdx.x; e < 32 * 32; e += 256) {
        const int rn = e >> 5, rk = e & 31, n = n0 + rn, k = k0 + rk;
        if (n >= a.N || k > a.K) continue;
        const float tot = ((red[0][rn][rk] + red[1][rn][rk]) + red[2][rn][rk]) + red[3][rn][rk];
        if (a.S == 1) {
            if (k < a.K) a.dW[(long)n * a.ldw + k] += tot;
            else if (a.db) a.db[n] += tot;
        } else {
            a.scratch[((long)s * a.N + n) * (a.K + 1) + k] = tot;
        }
    }
}

__global__ __launch_bounds__(256) void twgrad_kernel(TWg a) {
    __shared__ float red[4][32][33];
    twgrad_body(a, blockIdx.x, blockIdx.y, blockIdx.z, red);
}

// One launch for a layer's backward at training-scene sizes: blocks [0, nA) compute the input gradient (tlinear latency mode),
// the remaining blocks the weight / bias gradient.  The two halves are independent (dX must not alias dY or X).
__global__ __launch_bounds__(256) void tbwd_kernel(TLin a, int ksplit, int gxA, int nA, TWg w, int gxW, int gyW) {
    __shared__ __attribute__((aligned(16))) char sm[4 * 32 * 33 * 4];
    int id = blockIdx.x;
    if (id < nA) {
        tlinear_body<1, 1, 8>(a, ksplit, id % gxA, id / gxA, reinterpret_cast<f32x4(*)[1][64]>(sm));
    } else {
        id -= nA;
        twgrad_body(w, id % gxW, (id / gxW) % gyW, id / (gxW * gyW), reinterpret_cast<float(*)[32][33]>(sm));
    }
}

__global__ void twgrad_reduce_kernel(TWg a) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long per = (long)a.N * (a.K + 1);
    if (e >= per) return;
    const int n = (int)(e / (a.K + 1)), k = (int)(e % (a.K + 1));
    float tot = 0.f;
    for (int s = 0; s < a.S; ++s) tot += a.scratch[(long)s * per + e];
    if (k < a.K) a.dW[(long)n * a.ldw + k] += tot;
    else if (a.db) a.db[n] += tot;
}

// Scene sizes (cols <= 1024), grouped (sttode_tgemm_group): up to four independent layers -- forward (kind 0) or a whole backward (kind 1:
// input-gradient blocks, then weight-gradient blocks, as tbwd_kernel) -- in ONE launch.  A one-scene training step is bound by the NUMBER
// of launches (~5 us per dependent graph node whatever it does): decoder_x / decoder_y of a block and the two encoder trunks walk through
// the same layers with different weights.
#define TS_MULTI_MAX 4
struct TSProb { TLin a; TWg w; int ksplit, gxA, nA, gxW, gyW, kind; };
struct TSMulti { TSProb p[TS_MULTI_MAX]; int blk0[TS_MULTI_MAX + 1]; int n; };
__global__ __launch_bounds__(256) void tsmall_multi_kernel(TSMulti M) {
    __shared__ __attribute__((aligned(16))) char sm[4 * 32 * 33 * 4];
    int p = 0;
    while (p + 1 < M.n && (int)blockIdx.x >= M.blk0[p + 1]) ++p;
    p = __builtin_amdgcn_readfirstlane(p);
    TSProb P;   // the problem's descriptor out of the kernel-argument segment (uniform index: scalar loads)
    {
        const __attribute__((address_space(4))) int* src = (const __attribute__((address_space(4))) int*)(
            (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(TSMulti, p) + (size_t)p * sizeof(TSProb));
        int* dst = reinterpret_cast<int*>(&P);
#pragma unroll
        for (unsigned i = 0; i < sizeof(TSProb) / 4; ++i) dst[i] = src[i];
    }
    int id = (int)blockIdx.x - M.blk0[p];
    if (P.kind == 0 || id < P.nA) tlinear_body<1, 1, 8>(P.a, P.ksplit, id % P.gxA, id / P.gxA, reinterpret_cast<f32x4(*)[1][64]>(sm));
    else {
        id -= P.nA;
        twgrad_body(P.w, id % P.gxW, (id / P.gxW) % P.gyW, id / (P.gxW * P.gyW), reinterpret_cast<float(*)[32][33]>(sm));
    }
}

extern "C" int sttode_twgrad(const float* dY, long ldy, const float* X, long ldx, int xdiv, float* dW, long ldw, float* db,
                             int cols, int N, int K, float* scratch, long scratch_floats, void* stream) {
    STT_REQUIRE(dY && X && dW, "sttode_twgrad: null pointer");
    STT_REQUIRE(cols > 0 && N > 0 && K > 0 && xdiv > 0, "sttode_twgrad: cols, N, K, xdiv must be positive");
    STT_REQUIRE(ldy >= N && ldx >= K && ldw >= K, "sttode_twgrad: leading dimension smaller than the row length");
    TWg a;
    a.dY = dY; a.X = X; a.dW = dW; a.db = db; a.scratch = scratch;
    a.ldy = ldy; a.ldx = ldx; a.ldw = ldw; a.cols = cols; a.N = N; a.K = K; a.xdiv = xdiv;
    const int chunks = (cols + 15) / 16;
    const long per = (long)N * (K + 1);
    static const bool tg_on = !(getenv("STTODE_TGEMM") && atoi(getenv("STTODE_TGEMM")) == 0);
    if (tg_on && cols > tg_min_cols_bwd()) {   // batch sizes: the LDS-tiled kernel, reduction over the columns split so that the chip is full
        std::lock_guard<std::mutex> lk(g_red_mu);
        TG g;
        if (tg_wgrad_fill(g, dY, ldy, X, ldx, xdiv, dW, ldw, db, cols, N, K, scratch, scratch_floats, 480, stream)) {
            tg_submit(g, 2, (N + 63) / 64, (K + 1 + 63) / 64, g.S, stream, dW, ldw, db);
            STT_HIP(hipGetLastError());
            return 0;
        }
    }
    int S = chunks <= 64 ? 1 : (chunks + 31) / 32;    // >= 512 columns per split; up to 1024 columns one workgroup per tile (no reduce launch)
    if (S > 64) S = 64;
    if (!scratch || per * S > scratch_floats) S = scratch && scratch_floats >= 2 * per ? (int)(scratch_floats / per) : 1;
    if (S < 1) S = 1;
    a.S = S;
    a.chunks_per_split = (chunks + S - 1) / S;
    dim3 grid((N + 31) / 32, (K + 1 + 31) / 32, S);
    if (g_grp.on && S == 1 && cols <= 1024) {   // an open group: a weight gradient alone is a layer backward without input-gradient blocks
        std::lock_guard<std::mutex> lk(g_red_mu);
        ts_submit(TLin{}, &a, 1, 1, 0, (int)grid.x, (int)grid.y, (int)(grid.x * grid.y), stream);
        STT_HIP(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(twgrad_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    if (S > 1) hipLaunchKernelGGL(twgrad_reduce_kernel, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    STT_HIP(hipGetLastError());
    return 0;
}

// Backward of one nn.Linear in (at most) two launches: dX = mask(dY W[:, :Kdx] (+ dX)) and dW += dY^T X, db += sum dY.
// Small column counts (the launch-bound regime) take the fused kernel; otherwise the two stand-alone entry points run.
extern "C" int sttode_tlinear_bwd(const float* dY, long ldy, const float* W, long ldw, const float* mask, long ldm, float* dX,
                                  long lddx, int Kdx, int accumulate, const float* X, long ldx, int xdiv, float* dW, long ldgw,
                                  float* db, int cols, int N, int K, float* scratch, long scratch_floats, void* stream) {
    STT_REQUIRE(dY && W && dX && X && dW, "sttode_tlinear_bwd: null pointer");
    STT_REQUIRE(cols > 0 && N > 0 && K > 0 && Kdx > 0 && Kdx <= K && xdiv > 0, "sttode_tlinear_bwd: bad sizes");
    static const bool tg_on = !(getenv("STTODE_TGEMM") && atoi(getenv("STTODE_TGEMM")) == 0);
    static const bool fuse_on = !(getenv("STTODE_TGEMM_BWD") && atoi(getenv("STTODE_TGEMM_BWD")) == 0);   // =0: the two products as two launches (A/B)
    if (tg_on && fuse_on && cols > tg_min_cols_bwd() && xdiv == 1) {   // batch sizes: both products of the layer's backward in ONE launch
        STT_REQUIRE(ldy >= N && ldx >= K && ldgw >= K && ldw >= K && lddx >= Kdx, "sttode_tlinear_bwd: leading dimension smaller than the row length");
        std::lock_guard<std::mutex> lk(g_red_mu);
        TG gx;
        gx.A = dY; gx.lda = ldy; gx.B = W; gx.ldb = ldw; gx.C = dX; gx.ldc = lddx;
        gx.M = cols; gx.N = Kdx; gx.Kt = N; gx.adiv = 1; gx.bkdiv = 1; gx.ones_row = -1;
        gx.avec = aligned16(dY, ldy); gx.bvec = aligned16(W, ldw); gx.cvec = aligned16(dX, lddx);
        gx.bias = nullptr; gx.mask = mask; gx.ldm = ldm; gx.act = 0; gx.accumulate = accumulate; gx.asrc = dX; gx.ldas = lddx; gx.acdiv = 1;
        gx.db = nullptr; gx.scratch = nullptr; gx.S = 1; gx.kchunk = 0; gx.mode = 0;
        gx.evec = tg_evec(gx); gx.fast = tg_fast(gx, false, true); gx.dbg = nullptr;
        const int gxx = (cols + 63) / 64, nx = gxx * ((Kdx + 63) / 64);
        TG gw;
        if (tg_wgrad_fill(gw, dY, ldy, X, ldx, 1, dW, ldgw, db, cols, N, K, scratch, scratch_floats,
                          g_grp.on ? 400 : (nx < 680 ? 1000 - nx : 320), stream)) {
            const int gxw = (N + 63) / 64, gyw = (K + 1 + 63) / 64, nw = gxw * gyw * gw.S;
            if (g_grp.on) {   // (an open group: the two products join it as two of its problems)
                tg_submit(gw, 2, gxw, gyw, gw.S, stream, dW, ldgw, db);
                tg_submit(gx, 1, gxx, (Kdx + 63) / 64, 1, stream);
            } else {
                hipLaunchKernelGGL(tgemm_bwd_kernel, dim3(nw + nx), dim3(256), 0, (hipStream_t)stream, gw, gxw, gyw, nw, gx, gxx);
                tg_wgrad_done(gw, dW, ldgw, db);
            }
            STT_HIP(hipGetLastError());
            return 0;
        }
    }
    if (cols > 1024 || xdiv != 1) {
        if (int rc = sttode_tlinear(dY, ldy, 1, W, ldw, 1, nullptr, mask, ldm, dX, lddx, cols, N, Kdx, 0, accumulate, stream)) return rc;
        return sttode_twgrad(dY, ldy, X, ldx, xdiv, dW, ldgw, db, cols, N, K, scratch, scratch_floats, stream);
    }
    STT_REQUIRE(ldy >= N && ldx >= K && ldgw >= K && ldw >= K && lddx >= Kdx, "sttode_tlinear_bwd: leading dimension smaller than the row length");
    TLin a;
    a.X = dY; a.W = W; a.bias = nullptr; a.mask = mask; a.Y = dX;
    a.ldx = ldy; a.ldw = ldw; a.ldy = lddx; a.ldm = ldm;
    a.cols = cols; a.J = N; a.I = Kdx; a.trans = 1; a.act = 0; a.accumulate = accumulate; a.xdiv = 1; a.asrc = dX; a.ldas = lddx; a.adiv = 1;
    a.xvec = aligned16(dY, ldy); a.wvec = aligned16(W, ldw); a.yvec = aligned16(dX, lddx);
    a.evec = Kdx % 4 == 0 && a.yvec && (!mask || aligned16(mask, ldm));
    const int ksplit = N > 256 ? 4 : (N > 128 ? 2 : 1);
    const int blocks_per_wg = 4 / ksplit;
    const int gxA = (cols + 15) / 16, gyA = ((Kdx + 15) / 16 + blocks_per_wg - 1) / blocks_per_wg;
    TWg w;
    w.dY = dY; w.X = X; w.dW = dW; w.db = db; w.scratch = scratch;
    w.ldy = ldy; w.ldx = ldx; w.ldw = ldgw; w.cols = cols; w.N = N; w.K = K; w.xdiv = 1;
    const int chunks = (cols + 15) / 16;
    const long per = (long)N * (K + 1);
    int S = chunks <= 64 ? 1 : (chunks + 31) / 32;
    if (!scratch || per * S > scratch_floats) S = scratch && scratch_floats >= 2 * per ? (int)(scratch_floats / per) : 1;
    if (S < 1) S = 1;
    w.S = S;
    w.chunks_per_split = (chunks + S - 1) / S;
    const int gxW = (N + 31) / 32, gyW = (K + 1 + 31) / 32;
    const int nA = gxA * gyA, nB = gxW * gyW * S;
    if (g_grp.on && S == 1) {
        std::lock_guard<std::mutex> lk(g_red_mu);
        ts_submit(a, &w, ksplit, gxA, nA, gxW, gyW, nB, stream);
        STT_HIP(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(tbwd_kernel, dim3(nA + nB), dim3(256), 0, (hipStream_t)stream, a, ksplit, gxA, nA, w, gxW, gyW);
    if (S > 1) hipLaunchKernelGGL(twgrad_reduce_kernel, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w);
    STT_HIP(hipGetLastError());
    return 0;
}

static thread_local struct { TSMulti M; void* stream; } g_ts = {};
static void ts_group_launch_locked_forget() { g_ts.M.n = 0; }
static void ts_group_launch_locked() {
    if (g_ts.M.n == 0) return;
    hipLaunchKernelGGL(tsmall_multi_kernel, dim3((unsigned)g_ts.M.blk0[g_ts.M.n]), dim3(256), 0, (hipStream_t)g_ts.stream, g_ts.M);
    g_ts.M.n = 0;
}
static void ts_submit(const TLin& a, const TWg* w, int ksplit, int gxA, int nA, int gxW, int gyW, int nB, void* stream) {
    TSMulti& M = g_ts.M;
    if (M.n == TS_MULTI_MAX || (M.n > 0 && g_ts.stream != stream)) ts_group_launch_locked();
    const int i = M.n++;
    if (i == 0) M.blk0[0] = 0;
    TSProb& P = M.p[i];
    P.a = a; P.ksplit = ksplit; P.gxA = gxA; P.nA = nA; P.gxW = gxW; P.gyW = gyW; P.kind = w ? 1 : 0;
    if (w) P.w = *w; else P.w = TWg{};
    M.blk0[i + 1] = M.blk0[i] + nA + nB;
    g_ts.stream = stream;
}

// ---------------------------------------------------------------------------------------------------
// The decoder's layer-1 input prefix of BOTH decompose blocks in one launch: row c = (agent a, sample k) of inp0 / inp1 [n K1, ld] gets
// cat(past_feature[a] (128), z (32)) with z = the posterior draw qz[a] for k = 0 and the prior draw eps[a, k - 1] otherwise
// (model/STTODE.py:322-331, 553-566; the blocks' own state fills columns 160.. later).  Replaces two repeat_interleave copies per block
// and the two that assembled z: six launches of a launch-bound step.
// ---------------------------------------------------------------------------------------------------
__global__ void decoder_inputs_kernel(float* inp0, float* inp1, long ld, const float* pf, long ldpf, const float* qz, const float* eps, int n, int K1,
                                      int pfw, int zd) {
    const int q4 = (pfw + zd) / 4;                                    // float4 pieces of a row's prefix cat(pf [pfw = 2 hidden_dim], z [zdim]): 40 at the defaults
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (long)n * K1 * q4) return;
    const int f = (int)(id % q4) * 4;
    const long c = id / q4;
    const int a = (int)(c / K1), k = (int)(c % K1);
    const float* src = f < pfw ? pf + (long)a * ldpf + f : (k == 0 ? qz + (long)a * zd : eps + ((long)a * (K1 - 1) + k - 1) * zd) + (f - pfw);
    const f32x4 v = {src[0], src[1], src[2], src[3]};
    st4(inp0 + c * ld + f, v);
    if (inp1) st4(inp1 + c * ld + f, v);
}
extern "C" int sttode_decoder_inputs(float* inp0, float* inp1, long ld, const float* pf, long ldpf, const float* qz, const float* eps, int n,
                                     int K1, int pfw, int zd, void* stream) {
    STT_REQUIRE(inp0 && pf && qz && eps && n > 0 && K1 >= 1 && pfw >= 0 && zd > 0 && pfw % 4 == 0 && zd % 4 == 0 && ld >= pfw + zd && ld % 4 == 0 &&
                ldpf >= pfw, "sttode_decoder_inputs: bad argument (pfw, zd multiples of 4; ld >= pfw + zd)");
    STT_REQUIRE(((size_t)inp0 | (size_t)inp1) % 16 == 0, "sttode_decoder_inputs: inp0 / inp1 must be 16-byte aligned");
    const long tot = (long)n * K1 * ((pfw + zd) / 4);
    hipLaunchKernelGGL(decoder_inputs_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, inp0, inp1, ld, pf, ldpf, qz, eps, n, K1,
                       pfw, zd);
    STT_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// row shuffles
// ---------------------------------------------------------------------------------------------------
__global__ void rows_copy_kernel(float* dst, long ldd, const float* src, long lds, int rows, int width, int div, int mod) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)rows * width) return;
    const int r = (int)(e / width), f = (int)(e % width);
    dst[(long)r * ldd + f] = src[(long)((r / div) % mod) * lds + f];
}
// dst[a, f] (+)= sum_{k<K} src[a*K + k, f]
__global__ void rows_reduce_kernel(float* dst, long ldd, const float* src, long lds, int rows_out, int width, int K, int accumulate) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)rows_out * width) return;
    const int r = (int)(e / width), f = (int)(e % width);
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += src[((long)r * K + k) * lds + f];
    float* d = dst + (long)r * ldd + f;
    *d = accumulate ? *d + s : s;
}
extern "C" int sttode_rows_copy(float* dst, long ldd, const float* src, long lds, int rows, int width, int div, int mod, void* stream) {
    STT_REQUIRE(dst && src && rows > 0 && width > 0 && div > 0 && mod > 0, "sttode_rows_copy: bad argument");
    const long tot = (long)rows * width;
    hipLaunchKernelGGL(rows_copy_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dst, ldd, src, lds, rows, width, div, mod);
    STT_HIP(hipGetLastError());
    return 0;
}
extern "C" int sttode_rows_reduce(float* dst, long ldd, const float* src, long lds, int rows_out, int width, int K, int accumulate,
                                  void* stream) {
    STT_REQUIRE(dst && src && rows_out > 0 && width > 0 && K > 0, "sttode_rows_reduce: bad argument");
    const long tot = (long)rows_out * width;
    hipLaunchKernelGGL(rows_reduce_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dst, ldd, src, lds, rows_out, width, K, accumulate);
    STT_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// element-wise pieces.  op codes of sttode_train_ewise(op, p0..p5, count, i0, i1, f0):
// ---------------------------------------------------------------------------------------------------
enum {
    EW_MUL = 0,        // p0[i] = p1[i] * p2[i]                                 (dropout mask, gate product)
    EW_AXPY = 1,       // p0[i] += f0 * p1[i]
    EW_GATE_BWD = 2,   // given dout=p0, t=p1 (tanh out), s=p2 (sigmoid out): du=p3 = dout*s*(1-t^2), dv=p4 = dout*t*s*(1-s)
    EW_EULER_FWD = 3,  // p0 = relu(p1 + f0 * p2)
    EW_EULER_BWD = 4,  // d = dout(p0) * (out(p1) > 0): dx(p3) += d ; dy(p4) = f0 * d
    EW_RSAMPLE = 5,    // params p1 [rows, 2*i0] (mu | logvar), eps p2 [rows, i0] -> z p0 = mu + eps * exp(logvar / 2)
    EW_RELU_BWD = 6,   // p0[i] = p1[i] * (p2[i] > 0)
    EW_FILL = 7,       // p0[i] = f0
    EW_CUR_ADD = 9,    // p0[c, d] += p1[c / K, d % 2] with row length i0, K = (int)f0   ("+ cur_location", model/STTODE.py:343-344)
    EW_TANH_BWD = 10,  // p0[i] = p1[i] * (1 - p2[i]^2)      (p2 = tanh output)
    EW_LATENT_BWD = 11,  // sampler.py:51-53: dz=p0, dlogvar=p1, A=p2, eps p3 (mode i0: 0 none | 1 shared [nz] | 2 per agent) -> dA=p4
    EW_SUM_CUR = 12,   // p0[c, d] = p1 + p2 (+ p3[c / K, d % 2] if p3)   row length i0, K = (int)f0  (Decoder.forward :336-344)
    EW_SCALE_ADD = 14,       // p0[i] = f0 * p0[i] + (p1 ? p1[i] : 0)
    EW_AXPY_ROWS = 15,       // p0[r, c] += f0 * p1[r * ld + c], c < width: width = i0 & 0xffff, ld = i0 >> 16 (a column block of a wider matrix)
    EW_EULER_BWD_CAT = 13,   // op 4 reading dout = cat(dx0 | dode) as rows of p0 with leading dimension i0: d = dode * (out(p1) > 0); p3 = dx0 + d; p4 = f0 * d
    EW_RSAMPLE_BWD = 8,  // dz=p0 (in), params p1, eps p2 -> dparams p3 [rows, 2*i0]: dmu += dz ; dlogvar += dz * eps * exp(logvar/2) / 2
};

static __device__ __forceinline__ void ewise_body(int op, float* p0, const float* p1, const float* p2, float* p3, float* p4, long count, int i0,
                                                  float f0, long i) {
    if (i >= count) return;
    switch (op) {
        case EW_MUL: p0[i] = p1[i] * p2[i]; break;
        case EW_AXPY: p0[i] += f0 * p1[i]; break;
        case EW_GATE_BWD: {
            const float d = p0[i], t = p1[i], s = p2[i];
            p3[i] = d * s * (1.0f - t * t);
            p4[i] = d * t * s * (1.0f - s);
        } break;
        case EW_EULER_FWD: p0[i] = fmaxf(p1[i] + f0 * p2[i], 0.f); break;
        case EW_EULER_BWD: {
            const float d = p1[i] > 0.f ? p0[i] : 0.f;
            p3[i] += d;
            p4[i] = f0 * d;
        } break;
        case EW_SCALE_ADD: p0[i] = f0 * p0[i] + (p1 ? p1[i] : 0.f); break;
        case EW_AXPY_ROWS: {
            const int width = i0 & 0xffff, ld = i0 >> 16;
            p0[i] += f0 * p1[(i / width) * ld + i % width];
        } break;
        case EW_EULER_BWD_CAT: {   // i0 = ld | (D << 16); D = 0 means 64 (rounds 3-4 callers)
            const int D = (i0 >> 16) ? (i0 >> 16) : 64, ld = i0 & 0xffff;
            const long r = i / D;
            const int c = (int)(i % D);
            const float d = p1[i] > 0.f ? p0[r * ld + D + c] : 0.f;
            p3[i] = p0[r * ld + c] + d;
            p4[i] = f0 * d;
        } break;
        case EW_RSAMPLE: {
            const long r = i / i0;
            const int d = (int)(i % i0);
            p0[i] = p1[r * 2 * i0 + d] + p2[i] * expf(0.5f * p1[r * 2 * i0 + i0 + d]);
        } break;
        case EW_RELU_BWD: p0[i] = p2[i] > 0.f ? p1[i] : 0.f; break;
        case EW_FILL: p0[i] = f0; break;
        case EW_TANH_BWD: p0[i] = p1[i] * (1.0f - p2[i] * p2[i]); break;
        case EW_LATENT_BWD: {
            // z = A * eps + b, logvar = log(A^2 + 1e-8); f0 = K * nz (row length of A viewed [n, K*nz]), nz = i0 >> 2, mode = i0 & 3
            const int mode = i0 & 3, nz = i0 >> 2;
            const float a = p2[i];
            float e = 0.f;
            if (mode == 1) e = p3[i % nz];
            else if (mode == 2) e = p3[(i / (long)f0) * nz + i % nz];
            p4[i] = p0[i] * e + p1[i] * 2.0f * a / (a * a + 1e-8f);
        } break;
        case EW_SUM_CUR: {
            float v = p1[i] + p2[i];
            if (p3) v += p3[((i / i0) / (int)f0) * 2 + (i % i0) % 2];
            p0[i] = v;
        } break;
        case EW_CUR_ADD: {
            const long c = i / i0;
            p0[i] += p1[(c / (int)f0) * 2 + (i % i0) % 2];
        } break;
        case EW_RSAMPLE_BWD: {
            const long r = i / i0;
            const int d = (int)(i % i0);
            p3[r * 2 * i0 + d] += p0[i];
            p3[r * 2 * i0 + i0 + d] += p0[i] * p2[i] * 0.5f * expf(0.5f * p1[r * 2 * i0 + i0 + d]);
        } break;
    }
}

__global__ void ewise_kernel(int op, float* p0, const float* p1, const float* p2, float* p3, float* p4, long count, int i0, float f0) {
    ewise_body(op, p0, p1, p2, p3, p4, count, i0, f0, (long)blockIdx.x * blockDim.x + threadIdx.x);
}
// up to four independent element-wise pieces in one launch (sttode_tgemm_group: the same piece of the two encoder trunks)
#define EW_MULTI_MAX 4
struct EwProb { float* p0; const float* p1; const float* p2; float* p3; float* p4; long count; int op, i0; float f0; int blk0; };
struct EwMulti { EwProb p[EW_MULTI_MAX]; int n, blocks; };
__global__ void ewise_multi_kernel(EwMulti M) {
#pragma unroll
    for (int k = 0; k < EW_MULTI_MAX; ++k) {
        if (k >= M.n) break;
        const EwProb& e = M.p[k];
        const int last = k + 1 < M.n ? M.p[k + 1 < EW_MULTI_MAX ? k + 1 : k].blk0 : M.blocks;
        if ((int)blockIdx.x >= e.blk0 && (int)blockIdx.x < last)
            ewise_body(e.op, e.p0, e.p1, e.p2, e.p3, e.p4, e.count, e.i0, e.f0, (long)((int)blockIdx.x - e.blk0) * blockDim.x + threadIdx.x);
    }
}
static thread_local struct { EwMulti M; void* stream; } g_ewq = {};
static void ew_group_launch_locked() {
    if (g_ewq.M.n == 0) return;
    hipLaunchKernelGGL(ewise_multi_kernel, dim3((unsigned)g_ewq.M.blocks), dim3(256), 0, (hipStream_t)g_ewq.stream, g_ewq.M);
    g_ewq.M.n = 0; g_ewq.M.blocks = 0;
}
static void ew_group_forget() { g_ewq.M.n = 0; g_ewq.M.blocks = 0; }

extern "C" int sttode_train_ewise(int op, float* p0, const float* p1, const float* p2, float* p3, float* p4, long count, int i0,
                                  float f0, void* stream) {
    STT_REQUIRE(op >= 0 && op <= EW_AXPY_ROWS && p0 && count > 0, "sttode_train_ewise: bad argument");
    if (g_grp.on && count <= (1L << 24)) {   // an open group: queued, leaves with the group's other pieces
        std::lock_guard<std::mutex> lk(g_red_mu);
        EwMulti& M = g_ewq.M;
        if (M.n == EW_MULTI_MAX || (M.n > 0 && g_ewq.stream != stream)) ew_group_launch_locked();
        EwProb& e = M.p[M.n++];
        e.p0 = p0; e.p1 = p1; e.p2 = p2; e.p3 = p3; e.p4 = p4; e.count = count; e.op = op; e.i0 = i0; e.f0 = f0; e.blk0 = M.blocks;
        M.blocks += (int)((count + 255) / 256);
        g_ewq.stream = stream;
        return 0;
    }
    hipLaunchKernelGGL(ewise_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, op, p0, p1, p2, p3, p4, count, i0, f0);
    STT_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// LayerNorm(x + r) over 64 features, one wave per row (lane = feature); backward with per-WG partials
// ---------------------------------------------------------------------------------------------------
static __device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// D = hidden_dim (32 / 64 / 128: LayerNorm over the model dimension, hypertransformer.py:119-120); one wave per row, lane l holds elements
// l, l + 64 (D = 128) or is idle beyond D (D = 32).  D = 64: one element per lane, the sums of rounds 1-4.
template <int D>
__global__ __launch_bounds__(256) void add_ln_fwd_kernel(const float* x, const float* r, const float* gamma, const float* beta,
                                                         float* y, float* xhat, float* rstd, int rows) {
    constexpr int NE = D > 64 ? D / 64 : 1;
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bool on = lane < D;
    float v[NE], tot = 0.f;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const long o = (long)row * D + lane + 64 * e;
        v[e] = on ? x[o] + (r ? r[o] : 0.f) : 0.f;
        tot += v[e];
    }
    const float mean = wsum(tot) * (1.0f / D);
    float d[NE], sq = 0.f;
#pragma unroll
    for (int e = 0; e < NE; ++e) { d[e] = on ? v[e] - mean : 0.f; sq += d[e] * d[e]; }
    const float rs = 1.0f / sqrtf(wsum(sq) * (1.0f / D) + 1e-5f);
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        if (!on) continue;
        const long o = (long)row * D + lane + 64 * e;
        const float xh = d[e] * rs;
        xhat[o] = xh;
        y[o] = xh * gamma[lane + 64 * e] + beta[lane + 64 * e];
    }
    if (lane == 0) rstd[row] = rs;
}

// dsum = grad wrt (x + r); dgamma / dbeta accumulated deterministically: WG g sums its rows, a single last pass adds the
// per-WG partials in order (grid is small: rows <= a few thousand).
template <int D>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* dy, const float* xhat, const float* rstd, const float* gamma,
                                                     float* dsum, float* part, int rows, int rows_per_wg, float* dgamma, float* dbeta) {
    constexpr int NE = D > 64 ? D / 64 : 1;
    __shared__ float sg[4][NE * 64], sb[4][NE * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = blockIdx.x * rows_per_wg, r1 = min(r0 + rows_per_wg, rows);
    const bool on = lane < D;
    float ag[NE], ab[NE], g[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) { ag[e] = ab[e] = 0.f; g[e] = on ? gamma[lane + 64 * e] : 0.f; }
    for (int row = r0 + wave; row < r1; row += 4) {
        float dd[NE], xh[NE], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const long o = (long)row * D + lane + 64 * e;
            dd[e] = on ? dy[o] : 0.f;
            xh[e] = on ? xhat[o] : 0.f;
            ag[e] += dd[e] * xh[e];
            ab[e] += dd[e];
            const float dh = dd[e] * g[e];
            s1 += dh;
            s2 += dh * xh[e];
        }
        const float m1 = wsum(s1) * (1.0f / D), m2 = wsum(s2) * (1.0f / D);
#pragma unroll
        for (int e = 0; e < NE; ++e)
            if (on) dsum[(long)row * D + lane + 64 * e] = rstd[row] * (dd[e] * g[e] - m1 - xh[e] * m2);
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) { sg[wave][lane + 64 * e] = ag[e]; sb[wave][lane + 64 * e] = ab[e]; }
    __syncthreads();
    if (wave == 0 && on) {
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int c = lane + 64 * e;
            const float tg = ((sg[0][c] + sg[1][c]) + sg[2][c]) + sg[3][c], tb = ((sb[0][c] + sb[1][c]) + sb[2][c]) + sb[3][c];
            if (dgamma) {   // a single workgroup (rows <= 64: scene sizes): no partials, no second launch
                dgamma[c] += tg;
                dbeta[c] += tb;
            } else {
                part[(long)blockIdx.x * 2 * D + c] = tg;
                part[(long)blockIdx.x * 2 * D + D + c] = tb;
            }
        }
    }
}
__global__ void ln_bwd_reduce_kernel(const float* part, int G, float* dgamma, float* dbeta, int D) {
    const int t = threadIdx.x;  // 2 D threads
    float s = 0.f;
    for (int g = 0; g < G; ++g) s += part[(long)g * 2 * D + t];
    if (t < D) dgamma[t] += s;
    else dbeta[t - D] += s;
}

#define LN_DISPATCH(D_, CALL)                                                                          \
    do {                                                                                               \
        if ((D_) == 64) { constexpr int DD = 64; CALL; }                                               \
        else if ((D_) == 32) { constexpr int DD = 32; CALL; }                                          \
        else if ((D_) == 128) { constexpr int DD = 128; CALL; }                                        \
        else STT_REQUIRE(false, "LayerNorm kernels: hidden_dim must be 32, 64 or 128");                \
    } while (0)

extern "C" int sttode_add_ln_fwd(const float* x, const float* r, const float* gamma, const float* beta, float* y, float* xhat,
                                 float* rstd, int rows, int D, void* stream) {
    STT_REQUIRE(x && gamma && beta && y && xhat && rstd && rows > 0, "sttode_add_ln_fwd: bad argument");
    LN_DISPATCH(D, hipLaunchKernelGGL(add_ln_fwd_kernel<DD>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, r, gamma, beta, y, xhat, rstd, rows));
    STT_HIP(hipGetLastError());
    return 0;
}
extern "C" int sttode_ln_bwd(const float* dy, const float* xhat, const float* rstd, const float* gamma, float* dsum, float* dgamma,
                             float* dbeta, int rows, int D, float* scratch, long scratch_floats, void* stream) {
    STT_REQUIRE(dy && xhat && rstd && gamma && dsum && dgamma && dbeta && scratch && rows > 0, "sttode_ln_bwd: bad argument");
    // one workgroup up to 64 rows (scene sizes: no partials, no second launch); beyond that 16 rows per workgroup (round 5: 6 workgroups for the
    // 352 rows of an NBA-size step took 11 us; the reduction launch adds the per-workgroup partials in order either way)
    int G = rows <= 64 ? 1 : (rows + 15) / 16;
    if (G > 256) G = 256;
    STT_REQUIRE(scratch_floats >= (long)G * 2 * D, "sttode_ln_bwd: scratch too small");
    const int rpw = (rows + G - 1) / G;
    LN_DISPATCH(D, hipLaunchKernelGGL(ln_bwd_kernel<DD>, dim3(G), dim3(256), 0, (hipStream_t)stream, dy, xhat, rstd, gamma, dsum, scratch, rows, rpw,
                                      G == 1 ? dgamma : nullptr, G == 1 ? dbeta : nullptr));
    if (G > 1) hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3(1), dim3(2 * D), 0, (hipStream_t)stream, scratch, G, dgamma, dbeta, D);
    STT_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// GRU cell (torch.nn.GRU gate order r | z | n, model/STTODE.py:68): gi = W_ih e_t + b_ih (rows m*Tp + t), gh = W_hh h + b_hh
//   r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r * gh_n), h' = (1 - z) * n + z * h
// tape per step: r, z, n, gh_n  [m, 4*96]
// ---------------------------------------------------------------------------------------------------
__global__ void gru_cell_fwd_kernel(const float* gi, long ldgi, const float* gh, const float* hprev, float* hnew, float* tape, int m) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)m * 96) return;
    const int c = (int)(e / 96), f = (int)(e % 96);
    const float* gic = gi + (long)c * ldgi;
    const float* ghc = gh + (long)c * 288;
    const float r = 1.0f / (1.0f + expf(-(gic[f] + ghc[f])));
    const float z = 1.0f / (1.0f + expf(-(gic[96 + f] + ghc[96 + f])));
    const float hn = ghc[192 + f];
    const float n = tanhf(gic[192 + f] + r * hn);
    const float hp = hprev ? hprev[e] : 0.f;
    hnew[e] = (1.0f - z) * n + z * hp;
    float* t = tape + (long)c * 384;
    t[f] = r; t[96 + f] = z; t[192 + f] = n; t[288 + f] = hn;
}
// dh: grad wrt h' (in) ; writes dgi [m, 288] (rows with ld ldgi), dgh [m, 288], dhprev = dh * z (out, overwrites)
__global__ void gru_cell_bwd_kernel(const float* dh, const float* tape, const float* hprev, float* dgi, long ldgi, float* dgh,
                                    float* dhprev, int m) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)m * 96) return;
    const int c = (int)(e / 96), f = (int)(e % 96);
    const float* t = tape + (long)c * 384;
    const float r = t[f], z = t[96 + f], n = t[192 + f], hn = t[288 + f];
    const float hp = hprev ? hprev[e] : 0.f;
    const float d = dh[e];
    const float dn = d * (1.0f - z), dz = d * (hp - n);
    const float dnp = dn * (1.0f - n * n);
    const float drp = dnp * hn * r * (1.0f - r);
    const float dzp = dz * z * (1.0f - z);
    float* gi = dgi + (long)c * ldgi;
    float* gh = dgh + (long)c * 288;
    gi[f] = drp; gi[96 + f] = dzp; gi[192 + f] = dnp;
    gh[f] = drp; gh[96 + f] = dzp; gh[192 + f] = dnp * r;
    dhprev[e] = d * z;
}
extern "C" int sttode_gru_cell_fwd(const float* gi, long ldgi, const float* gh, const float* hprev, float* hnew, float* tape, int m,
                                   void* stream) {
    STT_REQUIRE(gi && gh && hnew && tape && m > 0, "sttode_gru_cell_fwd: bad argument");
    const long tot = (long)m * 96;
    hipLaunchKernelGGL(gru_cell_fwd_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gi, ldgi, gh, hprev, hnew, tape, m);
    STT_HIP(hipGetLastError());
    return 0;
}
extern "C" int sttode_gru_cell_bwd(const float* dh, const float* tape, const float* hprev, float* dgi, long ldgi, float* dgh,
                                   float* dhprev, int m, void* stream) {
    STT_REQUIRE(dh && tape && dgi && dgh && dhprev && m > 0, "sttode_gru_cell_bwd: bad argument");
    const long tot = (long)m * 96;
    hipLaunchKernelGGL(gru_cell_bwd_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dh, tape, hprev, dgi, ldgi, dgh, dhprev, m);
    STT_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// Whole-sequence GRU for the training step: columns are independent, so ONE launch runs all Tp steps (forward) or the whole
// BPTT (backward).  WG = 16 columns x 6 waves; wave j owns hidden features [16j, 16j+16) of all three gates.  The W_hh
// fragments a wave needs (18 f32x4 forward: rows of its 3 gate tiles; 18 backward: its 16 columns of W_hh as the A operand
// of dh_prev += dgh W_hh) stay in REGISTERS for all steps; h (forward) / dgh (backward) is exchanged through LDS once per step.
// ---------------------------------------------------------------------------------------------------
#define GSEQ_LDH 100   // padded row length of the h exchange buffer (floats)
#define GSEQ_LDG 292   // padded row length of the dgh exchange buffer

__global__ __launch_bounds__(384) void gru_seq_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ Whh,
                                                          const float* __restrict__ bhh, float* __restrict__ H,
                                                          float* __restrict__ tapes, float* __restrict__ hfinal, long ldhf, int m,
                                                          int Tp) {
    __shared__ float sH[16 * GSEQ_LDH];
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4, j = threadIdx.x >> 6;
    const int col = blockIdx.x * 16 + c;
    const bool ok = col < m;
    const int f = 16 * j + 4 * q;                       // first of this lane's 4 hidden features
    f32x4 w[3][6], bias[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
#pragma unroll
        for (int T = 0; T < 6; ++T) w[g][T] = ld4(Whh + (long)(g * 96 + 16 * j + c) * 96 + 16 * T + 4 * q);
        bias[g] = ld4(bhh + g * 96 + f);
    }
    for (int i = threadIdx.x; i < 16 * GSEQ_LDH; i += 384) sH[i] = 0.f;
    if (ok) st4(H + (long)col * 96 + f, splat4(0.f));   // H[0] = h_{-1} = 0: the backward pass reads it (the caller need not zero H)
    __syncthreads();
    // the input-gate rows of step t + 1 travel while step t runs (requested inside the step they cost an L2 round trip per step: 8 of them
    // were a third of the launch at scene sizes)
    const float* gic0 = gi + (long)(ok ? col : 0) * Tp * 288;
    f32x4 gr_n = ld4(gic0 + f), gz_n = ld4(gic0 + 96 + f), gn_n = ld4(gic0 + 192 + f);
    for (int t = 0; t < Tp; ++t) {
        const f32x4 gr = gr_n, gz = gz_n, gn = gn_n;
        if (t + 1 < Tp) {
            const float* gn1 = gic0 + (long)(t + 1) * 288;
            gr_n = ld4(gn1 + f); gz_n = ld4(gn1 + 96 + f); gn_n = ld4(gn1 + 192 + f);
        }
        f32x4 acc[3] = {bias[0], bias[1], bias[2]};
#pragma unroll
        for (int T = 0; T < 6; ++T) {
            const f32x4 b = ld4(sH + c * GSEQ_LDH + 16 * T + 4 * q);
#pragma unroll
            for (int g = 0; g < 3; ++g) acc[g] = mfma_k16(acc[g], w[g][T], b);
        }
        const f32x4 hp = ld4(sH + c * GSEQ_LDH + f);
        f32x4 hn = hp;
        if (ok) {
            f32x4 r, z, n;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // (the hardware's exp2 / rcp forms of chain.hpp, as in the inference GRU: absolute error ~1e-7; expf / tanhf / IEEE division
                // were ~1 500 vector instructions per lane and step -- most of a 4.7-us step at scene sizes)
                r[e] = sigmoidf_(gr[e] + acc[0][e]);
                z[e] = sigmoidf_(gz[e] + acc[1][e]);
                n[e] = tanhf_(gn[e] + r[e] * acc[2][e]);
                hn[e] = (1.0f - z[e]) * n[e] + z[e] * hp[e];
            }
            float* tp = tapes + ((long)t * m + col) * 384;
            st4(tp + f, r); st4(tp + 96 + f, z); st4(tp + 192 + f, n); st4(tp + 288 + f, acc[2]);
            st4(H + ((long)(t + 1) * m + col) * 96 + f, hn);
            if (hfinal && t == Tp - 1) st4(hfinal + (long)col * ldhf + f, hn);
        }
        __syncthreads();                                // every wave has read h_{t-1}
        st4(sH + c * GSEQ_LDH + f, hn);
        __syncthreads();
    }
}

__global__ __launch_bounds__(384) void gru_seq_bwd_kernel(const float* __restrict__ dh_last, long lddh, const float* __restrict__ tapes,
                                                          const float* __restrict__ H, const float* __restrict__ Whh,
                                                          float* __restrict__ dgi, float* __restrict__ dgh, int m, int Tp) {
    __shared__ float sG[16 * GSEQ_LDG];
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4, j = threadIdx.x >> 6;
    const int col = blockIdx.x * 16 + c;
    const bool ok = col < m;
    const int f = 16 * j + 4 * q;
    // A operand of dh_prev[:, 16j..16j+16) += dgh W_hh[:, 16j..]:  A[i = 16j + c][k] = W_hh[k][16j + c], k = 16T + 4q + r
    f32x4 w[18];
#pragma unroll
    for (int T = 0; T < 18; ++T)
#pragma unroll
        for (int r = 0; r < 4; ++r) w[T][r] = Whh[(long)(16 * T + 4 * q + r) * 96 + 16 * j + c];
    f32x4 dh = ok ? ld4(dh_last + (long)col * lddh + f) : splat4(0.f);
    // (requesting the tape of step t - 1 during step t, as the forward launch does with its rows, was measured: the same 30 us at scene
    // sizes and 47 -> 55 us at NBA size -- five more live f32x4 per lane)
    for (int t = Tp - 1; t >= 0; --t) {
        f32x4 dr = splat4(0.f), dz = dr, dn = dr, dhn = dr, dhz = dr;
        f32x4 r = dr, z = dr, n = dr, hn = dr, hp = dr;
        if (ok) {
            const float* tp = tapes + ((long)t * m + col) * 384;
            r = ld4(tp + f); z = ld4(tp + 96 + f); n = ld4(tp + 192 + f); hn = ld4(tp + 288 + f);
            hp = ld4(H + ((long)t * m + col) * 96 + f);
        }
        if (ok) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = dh[e];
                const float dnp = d * (1.0f - z[e]) * (1.0f - n[e] * n[e]);
                dn[e] = dnp;
                dr[e] = dnp * hn[e] * r[e] * (1.0f - r[e]);
                dz[e] = d * (hp[e] - n[e]) * z[e] * (1.0f - z[e]);
                dhn[e] = dnp * r[e];
                dhz[e] = d * z[e];
            }
            float* gi = dgi + ((long)col * Tp + t) * 288;
            st4(gi + f, dr); st4(gi + 96 + f, dz); st4(gi + 192 + f, dn);
            float* gh = dgh + ((long)t * m + col) * 288;
            st4(gh + f, dr); st4(gh + 96 + f, dz); st4(gh + 192 + f, dhn);
        }
        st4(sG + c * GSEQ_LDG + f, dr);
        st4(sG + c * GSEQ_LDG + 96 + f, dz);
        st4(sG + c * GSEQ_LDG + 192 + f, dhn);
        __syncthreads();
        f32x4 acc = dhz;
#pragma unroll
        for (int T = 0; T < 18; ++T) acc = mfma_k16(acc, w[T], ld4(sG + c * GSEQ_LDG + 16 * T + 4 * q));
        dh = acc;
        __syncthreads();                                // sG is rewritten by the next step
    }
}

// ---------------------------------------------------------------------------------------------------
// The same two launches for FEW columns (round 5; m <= gseq_small_max(), default 1024: one scene per step, the live columns of a backward
// pass, the per-agent first block).  The kernels above put 16 columns on a 16-wide MFMA tile and all of a tile's W_hh products on ONE CU:
// 885 kFLOP per step = 1.4-1.9 us of that CU's matrix pipe per step of the recurrence, whatever m is -- with m = 32 columns two CUs work
// and 254 idle, 28-33 us per launch, four launches per training step (20 % of a one-scene step).  Here a workgroup owns FOUR columns and
// the products run on the vector ALUs with the weight row (forward: W_hh[row][0..95]; backward: 72 of W_hh[..][f]'s 288) in registers and
// h / the gate gradients broadcast from LDS: 4x more workgroups, ~0.5 us per step.  Same tape layout, same arithmetic per element; the
// 96- / 288-term sums run as four interleaved partial sums instead of the MFMA's blocked order (differences at fp32 rounding).
// ---------------------------------------------------------------------------------------------------
#define GSEQ_SC 4
__global__ __launch_bounds__(384) void gru_seq_fwd_small_kernel(const float* __restrict__ gi, const float* __restrict__ Whh,
                                                                const float* __restrict__ bhh, float* __restrict__ H,
                                                                float* __restrict__ tapes, float* __restrict__ hfinal, long ldhf, int m,
                                                                int Tp) {
    __shared__ __attribute__((aligned(16))) float sH[GSEQ_SC][96];
    __shared__ float sA[GSEQ_SC][288];
    const int tid = threadIdx.x;
    const bool mv = tid < 288;                       // matvec thread: one row of W_hh
    float w[96];
    float bias = 0.f;
    if (mv) {
#pragma unroll
        for (int k = 0; k < 24; ++k) {
            const f32x4 v = ld4(Whh + (long)tid * 96 + 4 * k);
            w[4 * k] = v[0]; w[4 * k + 1] = v[1]; w[4 * k + 2] = v[2]; w[4 * k + 3] = v[3];
        }
        bias = bhh[tid];
    } else {
#pragma unroll
        for (int k = 0; k < 96; ++k) w[k] = 0.f;
    }
    const int c = tid / 96, f = tid % 96;            // gate thread: (column, feature)
    const int col = blockIdx.x * GSEQ_SC + c;
    const bool ok = col < m;
    sH[c][f] = 0.f;
    if (ok) H[(long)col * 96 + f] = 0.f;             // H[0] = h_{-1} = 0: the backward pass reads it
    __syncthreads();
    const float* gic = gi + (long)(ok ? col : 0) * Tp * 288;
    for (int t = 0; t < Tp; ++t) {
        const float gr = gic[(long)t * 288 + f], gz = gic[(long)t * 288 + 96 + f], gn = gic[(long)t * 288 + 192 + f];   // (in flight under the products)
        if (mv) {
            f32x4 acc[GSEQ_SC];                       // four partial sums per column (k mod 4): short dependency chains, blocked like the MFMA's sums
#pragma unroll
            for (int cc = 0; cc < GSEQ_SC; ++cc) acc[cc] = splat4(0.f);
#pragma unroll
            for (int k = 0; k < 24; ++k) {
#pragma unroll
                for (int cc = 0; cc < GSEQ_SC; ++cc) {
                    const f32x4 h4 = *reinterpret_cast<const f32x4*>(&sH[cc][4 * k]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[cc][e] = fmaf(w[4 * k + e], h4[e], acc[cc][e]);
                }
            }
#pragma unroll
            for (int cc = 0; cc < GSEQ_SC; ++cc) sA[cc][tid] = bias + ((acc[cc][0] + acc[cc][1]) + (acc[cc][2] + acc[cc][3]));
        }
        __syncthreads();
        const float a2 = sA[c][192 + f], hp = sH[c][f];
        const float r = sigmoidf_(gr + sA[c][f]);
        const float z = sigmoidf_(gz + sA[c][96 + f]);
        const float n = tanhf_(gn + r * a2);
        const float hn = (1.0f - z) * n + z * hp;
        if (ok) {
            float* tp = tapes + ((long)t * m + col) * 384;
            tp[f] = r; tp[96 + f] = z; tp[192 + f] = n; tp[288 + f] = a2;
            H[((long)(t + 1) * m + col) * 96 + f] = hn;
            if (hfinal && t == Tp - 1) hfinal[(long)col * ldhf + f] = hn;
        }
        sH[c][f] = ok ? hn : 0.f;
        __syncthreads();
    }
}
__global__ __launch_bounds__(384) void gru_seq_bwd_small_kernel(const float* __restrict__ dh_last, long lddh, const float* __restrict__ tapes,
                                                                const float* __restrict__ H, const float* __restrict__ Whh,
                                                                float* __restrict__ dgi, float* __restrict__ dgh, int m, int Tp) {
    __shared__ __attribute__((aligned(16))) float sG[GSEQ_SC][288];   // (dr | dz | dhn) of the step
    __shared__ float sP[4][GSEQ_SC][96];                              // partial products of the four k ranges
    const int tid = threadIdx.x;
    const int c = tid / 96, f = tid % 96;            // element thread (column, feature); product thread (k range c, feature f)
    const int col = blockIdx.x * GSEQ_SC + c;
    const bool ok = col < m;
    float w[72];                                      // W_hh[72 c + k][f]: this thread's quarter of the 288-term sum for feature f
#pragma unroll
    for (int k = 0; k < 72; ++k) w[k] = Whh[(long)(72 * c + k) * 96 + f];
    float dh = ok ? dh_last[(long)col * lddh + f] : 0.f;
    for (int t = Tp - 1; t >= 0; --t) {
        float dr = 0.f, dz = 0.f, dn = 0.f, dhn = 0.f, dhz = 0.f;
        if (ok) {
            const float* tp = tapes + ((long)t * m + col) * 384;
            const float r = tp[f], z = tp[96 + f], n = tp[192 + f], hn = tp[288 + f];
            const float hp = H[((long)t * m + col) * 96 + f];
            const float dnp = dh * (1.0f - z) * (1.0f - n * n);
            dn = dnp;
            dr = dnp * hn * r * (1.0f - r);
            dz = dh * (hp - n) * z * (1.0f - z);
            dhn = dnp * r;
            dhz = dh * z;
            float* gi = dgi + ((long)col * Tp + t) * 288;
            gi[f] = dr; gi[96 + f] = dz; gi[192 + f] = dn;
            float* gh = dgh + ((long)t * m + col) * 288;
            gh[f] = dr; gh[96 + f] = dz; gh[192 + f] = dhn;
        }
        sG[c][f] = dr; sG[c][96 + f] = dz; sG[c][192 + f] = dhn;
        __syncthreads();
        f32x4 acc[GSEQ_SC];
#pragma unroll
        for (int cc = 0; cc < GSEQ_SC; ++cc) acc[cc] = splat4(0.f);
#pragma unroll
        for (int k = 0; k < 18; ++k) {
#pragma unroll
            for (int cc = 0; cc < GSEQ_SC; ++cc) {
                const f32x4 g4 = *reinterpret_cast<const f32x4*>(&sG[cc][72 * c + 4 * k]);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[cc][e] = fmaf(g4[e], w[4 * k + e], acc[cc][e]);
            }
        }
#pragma unroll
        for (int cc = 0; cc < GSEQ_SC; ++cc) sP[c][cc][f] = (acc[cc][0] + acc[cc][1]) + (acc[cc][2] + acc[cc][3]);
        __syncthreads();
        dh = dhz + (((sP[0][c][f] + sP[1][c][f]) + sP[2][c][f]) + sP[3][c][f]);
        // (sG and sP are rewritten only after the next step's first barrier / this step's readers are past the second one)
    }
}
static inline int gseq_small_max() {
    static const int v = getenv("STTODE_GRU_SMALL_MAX") ? atoi(getenv("STTODE_GRU_SMALL_MAX")) : 1024;
    return v;
}

extern "C" int sttode_gru_seq_fwd(const float* gi, const float* Whh, const float* bhh, float* H, float* tapes, float* hfinal,
                                  long ldhf, int m, int Tp, void* stream) {
    STT_REQUIRE(gi && Whh && bhh && H && tapes && m > 0 && Tp > 0, "sttode_gru_seq_fwd: bad argument");
    STT_REQUIRE(((size_t)Whh) % 16 == 0 && ((size_t)gi) % 16 == 0, "sttode_gru_seq_fwd: pointers must be 16-byte aligned");
    STT_REQUIRE(!hfinal || (((size_t)hfinal) % 16 == 0 && ldhf % 4 == 0 && ldhf >= 96), "sttode_gru_seq_fwd: hfinal must be 16-byte aligned rows of >= 96 floats");
    if (m <= gseq_small_max())
        hipLaunchKernelGGL(gru_seq_fwd_small_kernel, dim3((m + GSEQ_SC - 1) / GSEQ_SC), dim3(384), 0, (hipStream_t)stream, gi, Whh, bhh, H, tapes, hfinal, ldhf, m, Tp);
    else
        hipLaunchKernelGGL(gru_seq_fwd_kernel, dim3((m + 15) / 16), dim3(384), 0, (hipStream_t)stream, gi, Whh, bhh, H, tapes, hfinal, ldhf, m, Tp);
    STT_HIP(hipGetLastError());
    return 0;
}
extern "C" int sttode_gru_seq_bwd(const float* dh_last, long lddh, const float* tapes, const float* H, const float* Whh, float* dgi,
                                  float* dgh, int m, int Tp, void* stream) {
    STT_REQUIRE(dh_last && tapes && H && Whh && dgi && dgh && m > 0 && Tp > 0, "sttode_gru_seq_bwd: bad argument");
    STT_REQUIRE(((size_t)dh_last) % 16 == 0 && lddh % 4 == 0 && lddh >= 96, "sttode_gru_seq_bwd: dh_last must be 16-byte aligned rows of >= 96 floats");
    if (m <= gseq_small_max())
        hipLaunchKernelGGL(gru_seq_bwd_small_kernel, dim3((m + GSEQ_SC - 1) / GSEQ_SC), dim3(384), 0, (hipStream_t)stream, dh_last, lddh, tapes, H, Whh, dgi, dgh, m, Tp);
    else
        hipLaunchKernelGGL(gru_seq_bwd_kernel, dim3((m + 15) / 16), dim3(384), 0, (hipStream_t)stream, dh_last, lddh, tapes, H, Whh, dgi, dgh, m, Tp);
    STT_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// conv1d(2 -> 32, k = 3, pad = 1) + relu over x [m, T, 2] (model/STTODE.py:65); e [m, T, 32]
// x = xa[c / adiv] - (xb ? xb[c] : 0)   (x_true - x_hat of the previous block)
// ---------------------------------------------------------------------------------------------------
__global__ void conv_fwd_kernel(const float* xa, int adiv, const float* xb, const float* w, const float* b, float* x, float* e, int m, int T) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (long)m * T * 32) return;
    const int oc = (int)(id % 32), t = (int)((id / 32) % T), c = (int)(id / (32L * T));
    float acc = b[oc];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int tt = t + k - 1;
        if (tt < 0 || tt >= T) continue;
#pragma unroll
        for (int ic = 0; ic < 2; ++ic) {
            float v = xa[((long)(c / adiv) * T + tt) * 2 + ic];
            if (xb) v -= xb[((long)c * T + tt) * 2 + ic];
            acc += w[(oc * 2 + ic) * 3 + k] * v;
            if (oc == 0 && k == 1) x[((long)c * T + tt) * 2 + ic] = v;  // k == 1: tt == t, every (c, t) written once
        }
    }
    e[id] = fmaxf(acc, 0.f);
}
// de already masked by relu.  dx[c, t, ic] = sum_{oc,k} w[oc,ic,k] * de[c, t - k + 1, oc]
__global__ void conv_bwd_x_kernel(const float* de, const float* w, float* dx, int m, int T) {
    const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (long)m * T * 2) return;
    const int ic = (int)(id % 2), t = (int)((id / 2) % T), c = (int)(id / (2L * T));
    float acc = 0.f;
    for (int k = 0; k < 3; ++k) {
        const int te = t - k + 1;
        if (te < 0 || te >= T) continue;
        const float* d = de + ((long)c * T + te) * 32;
        for (int oc = 0; oc < 32; ++oc) acc += w[(oc * 2 + ic) * 3 + k] * d[oc];
    }
    dx[id] = acc;
}
// dW[oc,ic,k] += sum_{c,t} de[c,t,oc] * x[c,t+k-1,ic], db[oc] += sum de.  A WG walks a contiguous slab of (c,t) rows: thread
// (row lane 0..7, oc 0..31) reads de[row][oc] (one 128-byte line per row across the 32 oc threads) and the row's 3 x 2 inputs,
// keeps its 6 weight partials + 1 bias partial in registers, the 8 row lanes are combined through LDS and every WG writes one
// partial vector [224]; a second single-WG pass adds the partials in order (deterministic).
__global__ __launch_bounds__(256) void conv_bwd_w_kernel(const float* de, const float* x, float* part, int m, int T, int rows_per_wg, float* dw,
                                                         float* db) {   // dw != nullptr: ONE workgroup, its sums go straight into dw / db (no reduction launch)
    __shared__ float red[8][224];
    const int oc = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const long rows = (long)m * T;
    const long r0 = (long)blockIdx.x * rows_per_wg, r1 = min(r0 + rows_per_wg, rows);
    float w[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, b = 0.f;
    for (long r = r0 + rl; r < r1; r += 8) {
        const int t = (int)(r % T);
        const float d = de[r * 32 + oc];
        b += d;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int tt = t + k - 1;
            if (tt < 0 || tt >= T) continue;
            const float* xr = x + (r - t + tt) * 2;
            w[0 * 3 + k] += d * xr[0];
            w[1 * 3 + k] += d * xr[1];
        }
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) red[rl][oc * 6 + j] = w[j];      // dw index = (oc*2 + ic)*3 + k = oc*6 + ic*3 + k
    red[rl][192 + oc] = b;
    __syncthreads();
    if (threadIdx.x < 224) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += red[i][threadIdx.x];
        if (dw) {
            if (threadIdx.x < 192) dw[threadIdx.x] += s;
            else db[threadIdx.x - 192] += s;
        } else {
            part[(long)blockIdx.x * 224 + threadIdx.x] = s;
        }
    }
}
// one wave per output j: lane l adds the partials g = l, l + 64, ... in order, then a fixed xor-shuffle tree (deterministic); a single
// 224-thread block walking up to 256 partials one after the other took 60 us at NBA batch sizes
__global__ __launch_bounds__(64) void conv_bwd_w_reduce_kernel(const float* part, int G, float* dw, float* db) {
    const int j = blockIdx.x, lane = threadIdx.x;
    float s = 0.f;
    for (int g = lane; g < G; g += 64) s += part[(long)g * 224 + j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) {
        if (j < 192) dw[j] += s;
        else db[j - 192] += s;
    }
}
extern "C" int sttode_conv_fwd(const float* xa, int adiv, const float* xb, const float* w, const float* b, float* x, float* e, int m,
                               int T, void* stream) {
    STT_REQUIRE(xa && w && b && x && e && m > 0 && T > 0 && adiv > 0, "sttode_conv_fwd: bad argument");
    const long tot = (long)m * T * 32;
    hipLaunchKernelGGL(conv_fwd_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, xa, adiv, xb, w, b, x, e, m, T);
    STT_HIP(hipGetLastError());
    return 0;
}
extern "C" int sttode_conv_bwd(const float* de, const float* x, const float* w, float* dx, float* dw, float* db, int m, int T,
                               float* scratch, long scratch_floats, void* stream) {
    STT_REQUIRE(de && x && w && dw && db && scratch && m > 0 && T > 0, "sttode_conv_bwd: bad argument");
    if (dx) {
        const long tot = (long)m * T * 2;
        hipLaunchKernelGGL(conv_bwd_x_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, de, w, dx, m, T);
    }
    const long rows = (long)m * T;
    int G = (int)((rows + 63) / 64);     // 8 rows per thread at scene sizes (512 per workgroup made an 11-workgroup launch of 43 us)
    if (G > 256) G = 256;
    STT_REQUIRE(scratch_floats >= (long)G * 224, "sttode_conv_bwd: scratch too small");
    if (rows <= 128) {                   // very few rows (16 trips of the row loop): one workgroup adds into dw / db itself -- one launch, not two
        hipLaunchKernelGGL(conv_bwd_w_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, de, x, scratch, m, T, (int)rows, dw, db);
        STT_HIP(hipGetLastError());
        return 0;
    }
    const int rpw = (int)((rows + G - 1) / G);
    hipLaunchKernelGGL(conv_bwd_w_kernel, dim3(G), dim3(256), 0, (hipStream_t)stream, de, x, scratch, m, T, rpw, (float*)nullptr, (float*)nullptr);
    hipLaunchKernelGGL(conv_bwd_w_reduce_kernel, dim3(224), dim3(64), 0, (hipStream_t)stream, scratch, G, dw, db);
    STT_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// geodesic self-attention backward (hyptransformerlib.py:191-300 with the untransposed-scores quirk :261-265):
//   out_i = sum_j P_ij v_j,  P_ij = softmax_j( -acos(clamp(khat_i . qhat_j)) ),  rows i = keys, columns j = queries.
// one WG per (slot, head); token (l, slot) is row l*Nb + slot of qkv [L*Nb, 192] = (q | k | v); L <= 1024.
// ---------------------------------------------------------------------------------------------------
// HD = hidden_dim / 8 (4 / 8 / 16); token rows are [q | k | v] of 3 * 8 * HD floats.  HD = 8: the sums of rounds 1-4.
template <int HD>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* qkv, const float* dO, float* dqkv, int L, int Nb) {
    constexpr int DM = 8 * HD;
    extern __shared__ float sm[];
    float* kh = sm;              // [L][HD] normalised keys
    float* qh = kh + L * HD;      // normalised queries
    float* vv = qh + L * HD;
    float* dd = vv + L * HD;      // dO
    float* rinv = dd + L * HD;    // [L] 1 / row sum of exp
    float* rdot = rinv + L;      // [L] sum_j P_ij dP_ij
    float* kn = rdot + L;        // [L] 1/|k|
    float* qn = kn + L;          // [L] 1/|q|
    const int slot = blockIdx.x / 8, h = blockIdx.x % 8;
    for (int l = threadIdx.x; l < L; l += blockDim.x) {
        const float* row = qkv + ((long)l * Nb + slot) * (3 * DM) + h * HD;
        float q[HD], k[HD], sq = 0.f, sk = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) { q[d] = row[d]; k[d] = row[DM + d]; sq += q[d] * q[d]; sk += k[d] * k[d]; }
        const float iq = 1.0f / sqrtf(sq), ik = 1.0f / sqrtf(sk);
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            qh[l * HD + d] = q[d] * iq;
            kh[l * HD + d] = k[d] * ik;
            vv[l * HD + d] = row[2 * DM + d];
            dd[l * HD + d] = dO[((long)l * Nb + slot) * DM + h * HD + d];
        }
        qn[l] = iq;
        kn[l] = ik;
    }
    __syncthreads();
    const float lo = -1.0f + 1e-4f, hi = 1.0f - 1e-4f;
    // pass 1 (thread = key row i): softmax denominator, sum_j P dP, and dkhat_i
    for (int i = threadIdx.x; i < L; i += blockDim.x) {
        float se = 0.f, sp = 0.f;
        for (int j = 0; j < L; ++j) {
            float dot = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) { dot += kh[i * HD + d] * qh[j * HD + d]; dp += dd[i * HD + d] * vv[j * HD + d]; }
            const float ex = expf(-acosf(fminf(fmaxf(dot, lo), hi)));
            se += ex;
            sp += ex * dp;
        }
        rinv[i] = 1.0f / se;
        rdot[i] = sp / se;
        float dk[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) dk[d] = 0.f;
        for (int j = 0; j < L; ++j) {
            float dot = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) { dot += kh[i * HD + d] * qh[j * HD + d]; dp += dd[i * HD + d] * vv[j * HD + d]; }
            const bool inside = dot > lo && dot < hi;
            const float cl = fminf(fmaxf(dot, lo), hi);
            const float P = expf(-acosf(cl)) * rinv[i];
            const float dS = P * (dp - rdot[i]);
            const float g = inside ? dS / sqrtf(1.0f - cl * cl) : 0.f;   // d(-acos x)/dx = 1/sqrt(1-x^2)
#pragma unroll
            for (int d = 0; d < HD; ++d) dk[d] += g * qh[j * HD + d];
        }
        float pr = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) pr += dk[d] * kh[i * HD + d];
        float* o = dqkv + ((long)i * Nb + slot) * (3 * DM) + DM + h * HD;
#pragma unroll
        for (int d = 0; d < HD; ++d) o[d] = (dk[d] - kh[i * HD + d] * pr) * kn[i];
    }
    __syncthreads();
    // pass 2 (thread = query column j): dqhat_j, dv_j
    for (int j = threadIdx.x; j < L; j += blockDim.x) {
        float dq[HD], dv[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) dq[d] = dv[d] = 0.f;
        for (int i = 0; i < L; ++i) {
            float dot = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) { dot += kh[i * HD + d] * qh[j * HD + d]; dp += dd[i * HD + d] * vv[j * HD + d]; }
            const bool inside = dot > lo && dot < hi;
            const float cl = fminf(fmaxf(dot, lo), hi);
            const float P = expf(-acosf(cl)) * rinv[i];
            const float dS = P * (dp - rdot[i]);
            const float g = inside ? dS / sqrtf(1.0f - cl * cl) : 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) { dq[d] += g * kh[i * HD + d]; dv[d] += P * dd[i * HD + d]; }
        }
        float pr = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) pr += dq[d] * qh[j * HD + d];
        float* o = dqkv + ((long)j * Nb + slot) * (3 * DM) + h * HD;
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            o[d] = (dq[d] - qh[j * HD + d] * pr) * qn[j];
            o[2 * DM + d] = dv[d];
        }
    }
}
// The same backward with the L x L pair work spread over the whole workgroup (round 5): the kernel above gives every key row / query
// column ONE thread that walks all L partners three times (exp, acos, sqrt per pair) -- at the NBA training batch (L = 32: 32 active
// lanes per workgroup) 36 us per trunk, 5 % of the step.  Here a thread owns pairs (phases A, C) or one (row, d) output element (phase D);
// the per-pair terms are computed once and kept in LDS.  Every sum over partners runs in the order of the kernel above and every term
// is the same expression; only the HD-term tangent projection is a lane butterfly instead of a loop (differences at fp32 rounding).
// LDS: 3 L^2 + L (4 HD + 4) floats (L <= ~100).
template <int HD>
__global__ __launch_bounds__(256) void attn_bwd_pairs_kernel(const float* qkv, const float* dO, float* dqkv, int L, int Nb) {
    constexpr int DM = 8 * HD;
    extern __shared__ float sm[];
    float* kh = sm;
    float* qh = kh + L * HD;
    float* vv = qh + L * HD;
    float* dd = vv + L * HD;
    float* rinv = dd + L * HD;
    float* rdot = rinv + L;
    float* kn = rdot + L;
    float* qn = kn + L;
    float* E = qn + L;            // [L][L] exp(-acos(.)), then P
    float* DP = E + L * L;        // [L][L] dO_i . v_j, then g
    float* G0 = DP + L * L;       // [L][L] sqrt(1 - x^2) inside the clamp, else 0
    const int slot = blockIdx.x / 8, h = blockIdx.x % 8, t = threadIdx.x;
    for (int l = t; l < L; l += 256) {
        const float* row = qkv + ((long)l * Nb + slot) * (3 * DM) + h * HD;
        float q[HD], k[HD], sq = 0.f, sk = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) { q[d] = row[d]; k[d] = row[DM + d]; sq += q[d] * q[d]; sk += k[d] * k[d]; }
        const float iq = 1.0f / sqrtf(sq), ik = 1.0f / sqrtf(sk);
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            qh[l * HD + d] = q[d] * iq;
            kh[l * HD + d] = k[d] * ik;
            vv[l * HD + d] = row[2 * DM + d];
            dd[l * HD + d] = dO[((long)l * Nb + slot) * DM + h * HD + d];
        }
        qn[l] = iq;
        kn[l] = ik;
    }
    __syncthreads();
    const float lo = -1.0f + 1e-4f, hi = 1.0f - 1e-4f;
    for (int p = t; p < L * L; p += 256) {                     // A: per pair (i = key row, j = query column)
        const int i = p / L, j = p % L;
        float dot = 0.f, dp = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) { dot += kh[i * HD + d] * qh[j * HD + d]; dp += dd[i * HD + d] * vv[j * HD + d]; }
        const bool inside = dot > lo && dot < hi;
        const float cl = fminf(fmaxf(dot, lo), hi);
        E[p] = expf(-acosf(cl));
        DP[p] = dp;
        G0[p] = inside ? sqrtf(1.0f - cl * cl) : 0.f;            // (>= 0.014 inside the clamp: 0 marks 'outside')
    }
    __syncthreads();
    for (int i = t; i < L; i += 256) {                         // B: row sums, partners in order
        float se = 0.f, sp = 0.f;
        for (int j = 0; j < L; ++j) { const float ex = E[i * L + j]; se += ex; sp += ex * DP[i * L + j]; }
        rinv[i] = 1.0f / se;
        rdot[i] = sp / se;
    }
    __syncthreads();
    for (int p = t; p < L * L; p += 256) {                     // C: P and g per pair
        const int i = p / L;
        const float P = E[p] * rinv[i];
        const float dS = P * (DP[p] - rdot[i]);
        const float g0 = G0[p];
        E[p] = P;
        DP[p] = g0 != 0.f ? dS / g0 : 0.f;                     // d(-acos x)/dx = 1/sqrt(1-x^2)
    }
    __syncthreads();
    // D: one output element per thread: (row, d); the tangent projection needs the row's HD elements: they sit in HD consecutive lanes
    for (int e0 = 0; e0 < L * HD; e0 += 256) {
        const int e = e0 + t;
        const bool on = e < L * HD;
        const int r = on ? e / HD : 0, d = e % HD;
        float dk = 0.f, dq = 0.f, dv = 0.f;
        for (int j = 0; j < L; ++j) dk += DP[r * L + j] * qh[j * HD + d];
        for (int i = 0; i < L; ++i) { dq += DP[i * L + r] * kh[i * HD + d]; dv += E[i * L + r] * dd[i * HD + d]; }
        float pk = dk * kh[r * HD + d], pq = dq * qh[r * HD + d];
#pragma unroll
        for (int sh = 1; sh < HD; sh <<= 1) { pk += __shfl_xor(pk, sh, 64); pq += __shfl_xor(pq, sh, 64); }
        if (on) {
            float* o = dqkv + ((long)r * Nb + slot) * (3 * DM) + h * HD + d;
            o[0] = (dq - qh[r * HD + d] * pq) * qn[r];
            o[DM] = (dk - kh[r * HD + d] * pk) * kn[r];
            o[2 * DM] = dv;
        }
    }
}
extern "C" int sttode_mhgsa_attn_bwd(const float* qkv, const float* dO, float* dqkv, int L, int Nb, int head_dim, void* stream) {
    STT_REQUIRE(qkv && dO && dqkv && L > 0 && Nb > 0, "sttode_mhgsa_attn_bwd: bad argument");
    STT_REQUIRE(head_dim == 4 || head_dim == 8 || head_dim == 16, "sttode_mhgsa_attn_bwd: head_dim must be 4, 8 or 16 (hidden_dim 32 / 64 / 128)");
    const size_t shm = (size_t)L * (4 * head_dim + 4) * sizeof(float);
    STT_REQUIRE(shm <= 160 * 1024, "sttode_mhgsa_attn_bwd: attention length too long for the training backward (L (4 head_dim + 4) floats of LDS)");
#define ATTB_GO(HD)                                                                                                                   \
    do {                                                                                                                              \
        STT_SET_LDS_ONCE(attn_bwd_kernel<HD>, 160 * 1024);                                                                            \
        hipLaunchKernelGGL(attn_bwd_kernel<HD>, dim3(Nb * 8), dim3(L < 256 ? ((L + 63) / 64) * 64 : 256), shm, (hipStream_t)stream, qkv, dO, dqkv, L, Nb); \
    } while (0)
    const size_t shm2 = shm + (size_t)3 * L * L * sizeof(float);
#define ATTB_PAIRS(HD)                                                                                                                \
    do {                                                                                                                              \
        STT_SET_LDS_ONCE(attn_bwd_pairs_kernel<HD>, 160 * 1024);                                                                      \
        hipLaunchKernelGGL(attn_bwd_pairs_kernel<HD>, dim3(Nb * 8), dim3(256), shm2, (hipStream_t)stream, qkv, dO, dqkv, L, Nb);        \
    } while (0)
    static const bool pairs = !(getenv("STTODE_ATTN_BWD_PAIRS") && atoi(getenv("STTODE_ATTN_BWD_PAIRS")) == 0);
    if (pairs && L >= 4 && shm2 <= 150 * 1024) {
        if (head_dim == 8) ATTB_PAIRS(8); else if (head_dim == 4) ATTB_PAIRS(4); else ATTB_PAIRS(16);
    } else {
        if (head_dim == 8) ATTB_GO(8); else if (head_dim == 4) ATTB_GO(4); else ATTB_GO(16);
    }
#undef ATTB_PAIRS
#undef ATTB_GO
    STT_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// Adam (train.py:122 torch.optim.Adam(model.parameters(), lr); its step at train.py:66,87) for ALL parameters of the model in ONE launch:
//     m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)      (g += wd p first)
// torch's fused implementation walks the 88 small tensors with multi_tensor_apply: 3 launches of 41-44 us each per step (131 us of a 2.3-ms
// NBA-size step, 140 us of a 1.05-ms one-scene step: profiles/r05/prof_train_nba_kernel_stats_before_adam.csv); the whole update moves
// 1.6 M parameters x 4 tensors = 26 MB.  Here a device table lists the tensors (parameter, first / second moment, gradient offset, element
// count, first chunk); block b finds its tensor by binary search over the chunk prefix and updates one 1024-element chunk in 16-byte pieces.
// Gradients are addressed as gbase + offset: the training engine hands out every step's gradients as views of ONE flat buffer with a fixed
// layout, so the table is uploaded once and only gbase changes.
// ---------------------------------------------------------------------------------------------------
struct AdamItem { float* p; float* m; float* v; long goff; long numel; long chunk0; };   // goff: floats from gbase; chunk0: first 1024-element chunk
#define ADAM_CHUNK 1024
__global__ __launch_bounds__(256) void adam_step_kernel(const AdamItem* __restrict__ items, int n, const float* __restrict__ gbase, float lr_over_bc1,
                                                        float omb1, float b2, float omb2, float eps, float inv_bc2_sqrt, float wd) {
    const long b = blockIdx.x;
    int lo = 0, hi = n - 1;                       // largest t with chunk0[t] <= b
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (items[mid].chunk0 <= b) lo = mid; else hi = mid - 1;
    }
    const AdamItem it = items[lo];
    const long e0 = (b - it.chunk0) * ADAM_CHUNK + 4 * (long)threadIdx.x;
    if (e0 >= it.numel) return;
    const float* g = gbase + it.goff;
    float pv[4], mv[4], vv[4], gv[4];
    const bool vec = e0 + 3 < it.numel && ((((size_t)(it.p + e0)) | ((size_t)(it.m + e0)) | ((size_t)(it.v + e0)) | ((size_t)(g + e0))) & 15) == 0;
    const int cnt = it.numel - e0 < 4 ? (int)(it.numel - e0) : 4;
    if (vec) {
        const f32x4 P = ld4(it.p + e0), M = ld4(it.m + e0), V = ld4(it.v + e0), G = ld4(g + e0);
#pragma unroll
        for (int r = 0; r < 4; ++r) { pv[r] = P[r]; mv[r] = M[r]; vv[r] = V[r]; gv[r] = G[r]; }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r < cnt) { pv[r] = it.p[e0 + r]; mv[r] = it.m[e0 + r]; vv[r] = it.v[e0 + r]; gv[r] = g[e0 + r]; }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (r >= cnt) break;
        float gr = gv[r];
        if (wd != 0.f) gr = fmaf(wd, pv[r], gr);
        mv[r] = mv[r] + omb1 * (gr - mv[r]);                         // exp_avg.lerp_(grad, 1 - beta1): 1 - beta formed in double on the host, as torch does
        vv[r] = vv[r] * b2 + omb2 * gr * gr;                         // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
        const float denom = sqrtf(vv[r]) * inv_bc2_sqrt + eps;
        pv[r] = pv[r] - lr_over_bc1 * (mv[r] / denom);
    }
    if (vec) {
        st4(it.p + e0, f32x4{pv[0], pv[1], pv[2], pv[3]});
        st4(it.m + e0, f32x4{mv[0], mv[1], mv[2], mv[3]});
        st4(it.v + e0, f32x4{vv[0], vv[1], vv[2], vv[3]});
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r < cnt) { it.p[e0 + r] = pv[r]; it.m[e0 + r] = mv[r]; it.v[e0 + r] = vv[r]; }
    }
}
// items: DEVICE array of n AdamItem (6 x 8 bytes each: p, m, v pointers, goff, numel, chunk0), chunk0 ascending from 0; chunks = their total
extern "C" int sttode_adam_step(const void* items, int n, long chunks, const float* gbase, double lr, double beta1, double beta2, double eps,
                                double weight_decay, long step, void* stream) {
    STT_REQUIRE(items && n > 0 && chunks > 0 && chunks < (1L << 31) && step >= 1, "sttode_adam_step: bad argument");
    STT_REQUIRE(lr >= 0. && beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1. && eps >= 0., "sttode_adam_step: bad hyper-parameter");
    // (hyper-parameters as doubles: torch forms 1 - beta, the bias corrections and the step size in Python floats and rounds once)
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)chunks), dim3(256), 0, (hipStream_t)stream, (const AdamItem*)items, n, gbase,
                       (float)(lr / bc1), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)(1.0 / sqrt(bc2)), (float)weight_decay);
    STT_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// losses (model/STTODE.py:372-395) -- values and their gradients; out[] slots written by single-WG reductions
// ---------------------------------------------------------------------------------------------------
static __device__ float block_sum(float v, float* red) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}
// out[0] = scale * sum (pred - target)^2 ; dpred = 2 * scale * gscale * (pred - target)
__global__ __launch_bounds__(256) void sqerr_kernel(const float* pred, const float* target, long count, float scale, float* out, float* dpred) {
    __shared__ float red[256];
    float acc = 0.f;
    for (long i = threadIdx.x; i < count; i += 256) {
        const float d = pred[i] - target[i];
        acc += d * d;
        if (dpred) dpred[i] = 2.0f * scale * d;
    }
    const float s = block_sum(acc, red);
    if (threadIdx.x == 0) out[0] = s * scale;
}
// KL(q || N(0, I)) in the two-distribution form (utils/dist.py:26-29 with p.sigma = 1), sum / denom, clamp_min(min_clip).
// dparams [rows, 2*zd] = gradient of the CLAMPED value (zero when the clamp is active).
// one WG per scene (scene_ptr NULL: one WG over all rows with the given denominator).  vals[s] = clamp_min(KL_s / denom_s, min_clip),
// denom_s = agents of the scene (B = 1 per scene, model/STTODE.py:378-382); dparams carries the gradient of the clamped value.
__global__ __launch_bounds__(256) void kl_kernel(const float* params, const int* scene_ptr, int rows, int zd, float denom, float min_clip,
                                                 float* vals, float* dparams) {
    __shared__ float red[256];
    const float ps = 1.0f + 1e-8f;
    const long r0 = scene_ptr ? scene_ptr[blockIdx.x] : 0, r1 = scene_ptr ? scene_ptr[blockIdx.x + 1] : rows;
    if (scene_ptr) denom = (float)(r1 - r0);
    float acc = 0.f;
    for (long i = r0 * zd + threadIdx.x; i < r1 * zd; i += 256) {
        const long r = i / zd;
        const int d = (int)(i % zd);
        const float mu = params[r * 2 * zd + d], lv = params[r * 2 * zd + zd + d];
        const float t1 = mu / ps, t2 = expf(0.5f * lv) / ps;
        acc += 0.5f * (t1 * t1 + t2 * t2) - 0.5f - logf(t2);
    }
    const float s = block_sum(acc, red) / denom;
    const bool live = s >= min_clip;   // clamp_min_: gradient passes where input >= min (torch convention)
    if (threadIdx.x == 0) vals[blockIdx.x] = live ? s : min_clip;
    if (dparams) {
        for (long i = r0 * zd + threadIdx.x; i < r1 * zd; i += 256) {
            const long r = i / zd;
            const int d = (int)(i % zd);
            const float mu = params[r * 2 * zd + d], lv = params[r * 2 * zd + zd + d];
            const float t2 = expf(0.5f * lv) / ps;
            dparams[r * 2 * zd + d] = live ? (mu / (ps * ps)) / denom : 0.f;
            dparams[r * 2 * zd + zd + d] = live ? (0.5f * t2 * t2 - 0.5f) / denom : 0.f;  // d/dlv [0.5 t2^2 - log t2], dt2/dlv = t2/2
        }
    }
}
// best-of-K: per agent min_k sum_{t,xy} (target - pred)^2 (first minimum, like torch.min), mean over agents.
// one wave per agent (lane = sample k, K <= 64), then a single-WG mean over the per-agent minima (fixed order).
__global__ __launch_bounds__(64) void diverse_agent_kernel(const float* pred, const float* target, const int* scene_ptr,
                                                           const int* agent_scene, int n, int K, int D, float* best, float* dpred) {
    const int a = blockIdx.x, lane = threadIdx.x;
    // weight of this agent in the objective: 1 / (agents of its scene) (mean over the scene, :390-395); one scene: 1 / n
    float wgt = 1.0f / (float)n;
    if (scene_ptr) { const int sc = agent_scene[a]; wgt = 1.0f / (float)(scene_ptr[sc + 1] - scene_ptr[sc]); }
    float s = 3.4e38f;
    if (lane < K) {
        s = 0.f;
        for (int d = 0; d < D; ++d) {
            const float t = target[(long)a * D + d] - pred[((long)a * K + lane) * D + d];
            s += t * t;
        }
    }
    float bs = s;
    int bk = lane;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float os = __shfl_xor(bs, o, 64);
        const int ok = __shfl_xor(bk, o, 64);
        if (os < bs || (os == bs && ok < bk)) { bs = os; bk = ok; }
    }
    if (lane == 0) best[a] = bs * wgt;
    if (dpred)
        for (int i = lane; i < K * D; i += 64) {
            const int k = i / D, d = i % D;
            const long idx = ((long)a * K + k) * D + d;
            dpred[idx] = k == bk ? 2.0f * (pred[idx] - target[(long)a * D + d]) * wgt : 0.f;
        }
}
__global__ __launch_bounds__(256) void sum_kernel(const float* v, int n, float* out) {
    __shared__ float red[256];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) acc += v[i];
    const float s = block_sum(acc, red);
    if (threadIdx.x == 0) out[0] = s;
}
// The whole objective of forward() (model/STTODE.py:372-395,553-568) for ONE decoder pass over K1 = 1 + K samples per agent (sample 0
// decoded from the posterior draw, samples 1..K from the prior draws), in two launches:
//   objective_kernel   blocks [0, n): agent a -- squared errors of sample 0 (prediction and recovered past), best-of-K over samples
//                      1..K, and every gradient row of the agent (dpred [K1,D], drec [K1,Dp]: zero rows where a sample does not enter);
//                      blocks [n, n + nb): the KL blocks of kl_kernel;   partial sums -> scratch
//   objective_sum      fixed-order sums of the partials -> out[0..4] = (mse, recover, kl, diverse, their sum)
struct ObjArgs {
    const float* pred; const float* rec; const float* fut; const float* past; const float* qzp; const int* scene_ptr; const int* agent_scene;
    float* dpred; float* drec; float* dqzp; float* part;   // part: [3][n] agent partials, then [nb] KL values
    int n, K1, D, Dp, zd, nb;
    int kl_rows;   // != 0 (no scene_ptr: ONE KL value over all n rows): every agent's block sums its own row's KL terms into part[3 n + a] and writes
                   // the row's UNCLAMPED gradient; objective_sum adds them up, applies the clamp and -- where it is active -- zeroes dqzp.  (Round 4:
                   // one block walked all n zd terms twice, 43 us of a 2.3-ms NBA-size step.)
    float scale_mse, scale_rec, kl_denom, min_clip;
    // live form (best != nullptr; sttode_loss_objective_live): of an agent's K1 decoder columns only sample 0 (posterior: mse + recover terms)
    // and the best of the prior samples (the min over K of loss_diverse, model/STTODE.py:390-395) receive a gradient -- every other
    // column's is exactly zero.  The gradients are written for those two columns only, dpred2 [n][2][D] / drec2 [n][2][Dp] (row 1 of
    // drec2 = 0), and best [n] names the second one (1..K); dpred / drec are not written.
    int* best; float* dpred2; float* drec2;
};
__global__ __launch_bounds__(256) void objective_kernel(ObjArgs o) {
    __shared__ float red[256];
    if ((int)blockIdx.x >= o.n) {     // KL block (same arithmetic as kl_kernel): a batch of scenes (one block per scene) -- or the single-scene case
                                      // when the agents' blocks did not take the term over (o.kl_rows == 0)
        const int b = blockIdx.x - o.n;
        const float ps = 1.0f + 1e-8f;
        const long r0 = o.scene_ptr ? o.scene_ptr[b] : 0, r1 = o.scene_ptr ? o.scene_ptr[b + 1] : o.n;
        const float denom = o.scene_ptr ? (float)(r1 - r0) : o.kl_denom;
        const int zd = o.zd;
        float acc = 0.f;
        for (long i = r0 * zd + threadIdx.x; i < r1 * zd; i += 256) {
            const long r = i / zd;
            const int d = (int)(i % zd);
            const float mu = o.qzp[r * 2 * zd + d], lv = o.qzp[r * 2 * zd + zd + d];
            const float t1 = mu / ps, t2 = expf(0.5f * lv) / ps;
            acc += 0.5f * (t1 * t1 + t2 * t2) - 0.5f - logf(t2);
        }
        const float sm = block_sum(acc, red) / denom;
        const bool live = sm >= o.min_clip;
        if (threadIdx.x == 0) o.part[3 * (long)o.n + b] = live ? sm : o.min_clip;
        for (long i = r0 * zd + threadIdx.x; i < r1 * zd; i += 256) {
            const long r = i / zd;
            const int d = (int)(i % zd);
            const float mu = o.qzp[r * 2 * zd + d], lv = o.qzp[r * 2 * zd + zd + d];
            const float t2 = expf(0.5f * lv) / ps;
            o.dqzp[r * 2 * zd + d] = live ? (mu / (ps * ps)) / denom : 0.f;
            o.dqzp[r * 2 * zd + zd + d] = live ? (0.5f * t2 * t2 - 0.5f) / denom : 0.f;
        }
        return;
    }
    const int a = blockIdx.x, t = threadIdx.x, K1 = o.K1, D = o.D, Dp = o.Dp;
    const float* pa = o.pred + (long)a * K1 * D;
    const float* ra = o.rec + (long)a * K1 * Dp;
    // sample 0: squared errors + gradients
    float e0 = 0.f, e1 = 0.f;
    const bool live = o.best != nullptr;
    float* dp0 = live ? o.dpred2 + (long)a * 2 * D : o.dpred + (long)a * K1 * D;
    float* dr0 = live ? o.drec2 + (long)a * 2 * Dp : o.drec + (long)a * K1 * Dp;
    for (int d = t; d < D; d += 256) { const float df = pa[d] - o.fut[(long)a * D + d]; e0 += df * df; dp0[d] = 2.0f * o.scale_mse * df; }
    for (int d = t; d < Dp; d += 256) { const float df = ra[d] - o.past[(long)a * Dp + d]; e1 += df * df; dr0[d] = 2.0f * o.scale_rec * df; }
    for (int i = Dp + t; i < (live ? 2 : K1) * Dp; i += 256) dr0[i] = 0.f;
    const float s0 = block_sum(e0, red), s1 = block_sum(e1, red);
    // samples 1..K: first minimum of the summed squared error (torch.min), weight 1 / (agents of the scene)
    float wgt = 1.0f / (float)o.n;
    if (o.scene_ptr) { const int sc = o.agent_scene[a]; wgt = 1.0f / (float)(o.scene_ptr[sc + 1] - o.scene_ptr[sc]); }
    __shared__ float sv[64];
    __shared__ int sk;
    if (t < 64) {
        float sq = 3.4e38f;
        if (t + 1 < K1) {
            sq = 0.f;
            for (int d = 0; d < D; ++d) { const float df = o.fut[(long)a * D + d] - pa[(long)(t + 1) * D + d]; sq += df * df; }
        }
        float bs = sq;
        int bk = t;
#pragma unroll
        for (int sh = 32; sh > 0; sh >>= 1) {
            const float os = __shfl_xor(bs, sh, 64);
            const int ok = __shfl_xor(bk, sh, 64);
            if (os < bs || (os == bs && ok < bk)) { bs = os; bk = ok; }
        }
        if (t == 0) { sv[0] = bs; sk = bk; }
    }
    __syncthreads();
    const int bk = sk + 1;
    if (live) {
        for (int d = t; d < D; d += 256) dp0[D + d] = 2.0f * (pa[(long)bk * D + d] - o.fut[(long)a * D + d]) * wgt;
        if (t == 0) o.best[a] = bk;
    } else {
        for (int i = D + t; i < K1 * D; i += 256) {
            const int k = i / D, d = i % D;
            o.dpred[(long)a * K1 * D + i] = k == bk ? 2.0f * (pa[i] - o.fut[(long)a * D + d]) * wgt : 0.f;
        }
    }
    if (t == 0) { o.part[a] = s0; o.part[o.n + a] = s1; o.part[2 * (long)o.n + a] = sv[0] * wgt; }
    if (o.kl_rows) {                  // this agent's row of the KL term (arithmetic of kl_kernel), gradient as if the clamp were inactive
        const float ps = 1.0f + 1e-8f;
        const int zd = o.zd;
        float acc = 0.f;
        for (int d = t; d < zd; d += 256) {
            const float mu = o.qzp[(long)a * 2 * zd + d], lv = o.qzp[(long)a * 2 * zd + zd + d];
            const float t1 = mu / ps, t2 = expf(0.5f * lv) / ps;
            acc += 0.5f * (t1 * t1 + t2 * t2) - 0.5f - logf(t2);
            o.dqzp[(long)a * 2 * zd + d] = (mu / (ps * ps)) / o.kl_denom;
            o.dqzp[(long)a * 2 * zd + zd + d] = (0.5f * t2 * t2 - 0.5f) / o.kl_denom;
        }
        const float sk_ = block_sum(acc, red);
        if (t == 0) o.part[3 * (long)o.n + a] = sk_;
    }
}
__global__ __launch_bounds__(256) void objective_sum_kernel(const float* part, int n, int nb, float scale_mse, float scale_rec, float* out,
                                                           int kl_rows, float kl_denom, float min_clip, float* dqzp, int zd) {
    __shared__ float red[256];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) { a0 += part[i]; a1 += part[n + i]; a2 += part[2 * (long)n + i]; }
    for (int i = threadIdx.x; i < (kl_rows ? kl_rows : nb); i += 256) a3 += part[3 * (long)n + i];
    const float s0 = block_sum(a0, red), s1 = block_sum(a1, red), s2 = block_sum(a2, red);
    float s3 = block_sum(a3, red);
    if (kl_rows) {                    // (uniform) the agents' row sums -> clamp_min(KL / denom, min_clip); an active clamp passes no gradient
        s3 /= kl_denom;
        const bool live = s3 >= min_clip;
        if (!live) {
            s3 = min_clip;
            for (long i = threadIdx.x; i < (long)kl_rows * 2 * zd; i += 256) dqzp[i] = 0.f;
        }
    }
    if (threadIdx.x == 0) {
        const float l0 = s0 * scale_mse, l1 = s1 * scale_rec;
        out[0] = l0; out[1] = l1; out[2] = s3; out[3] = s2;
        out[4] = ((l0 + l1) + s3) + s2;   // total_loss (model/STTODE.py:568)
    }
}
static int loss_objective_impl(const float* pred, const float* rec, const float* fut, const float* past, const float* qzp,
                               const int* scene_ptr, const int* agent_scene, int S, int n, int K1, int D, int Dp, int zd,
                               float scale_mse, float scale_rec, float kl_denom, float min_clip, float* out, float* dpred,
                               float* drec, float* dqzp, int* best, float* scratch, long scratch_floats, void* stream) {
    STT_REQUIRE(pred && rec && fut && past && qzp && out && dpred && drec && dqzp && scratch, "sttode_loss_objective: null pointer");
    STT_REQUIRE(n > 0 && K1 >= 2 && K1 <= 65 && D > 0 && Dp > 0 && zd > 0, "sttode_loss_objective: bad sizes (2 <= K1 <= 65)");
    STT_REQUIRE(scene_ptr ? (S > 0 && agent_scene) : kl_denom > 0.f, "sttode_loss_objective: scene_ptr needs S > 0 and agent_scene, otherwise kl_denom > 0");
    const int nb = scene_ptr ? S : 0;             // KL blocks: one per scene of a batch; the single-value case rides in the agents' blocks
    const int kl_rows = scene_ptr ? 0 : n;
    STT_REQUIRE(3L * n + (scene_ptr ? S : n) <= scratch_floats, "sttode_loss_objective: scratch too small (3 n + max(S, n) floats)");
    ObjArgs o;
    o.pred = pred; o.rec = rec; o.fut = fut; o.past = past; o.qzp = qzp; o.scene_ptr = scene_ptr; o.agent_scene = agent_scene;
    o.dpred = best ? nullptr : dpred; o.drec = best ? nullptr : drec; o.dqzp = dqzp; o.part = scratch;
    o.best = best; o.dpred2 = best ? dpred : nullptr; o.drec2 = best ? drec : nullptr;
    o.n = n; o.K1 = K1; o.D = D; o.Dp = Dp; o.zd = zd; o.nb = nb; o.kl_rows = kl_rows;
    o.scale_mse = scale_mse; o.scale_rec = scale_rec; o.kl_denom = kl_denom; o.min_clip = min_clip;
    hipLaunchKernelGGL(objective_kernel, dim3(n + nb), dim3(256), 0, (hipStream_t)stream, o);
    hipLaunchKernelGGL(objective_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, scratch, n, nb, scale_mse, scale_rec, out, kl_rows, kl_denom,
                       min_clip, dqzp, zd);
    STT_HIP(hipGetLastError());
    return 0;
}
extern "C" int sttode_loss_objective(const float* pred, const float* rec, const float* fut, const float* past, const float* qzp,
                                     const int* scene_ptr, const int* agent_scene, int S, int n, int K1, int D, int Dp, int zd,
                                     float scale_mse, float scale_rec, float kl_denom, float min_clip, float* out, float* dpred,
                                     float* drec, float* dqzp, float* scratch, long scratch_floats, void* stream) {
    return loss_objective_impl(pred, rec, fut, past, qzp, scene_ptr, agent_scene, S, n, K1, D, Dp, zd, scale_mse, scale_rec, kl_denom, min_clip, out,
                               dpred, drec, dqzp, nullptr, scratch, scratch_floats, stream);
}
extern "C" int sttode_loss_objective_live(const float* pred, const float* rec, const float* fut, const float* past, const float* qzp,
                                          const int* scene_ptr, const int* agent_scene, int S, int n, int K1, int D, int Dp, int zd,
                                          float scale_mse, float scale_rec, float kl_denom, float min_clip, float* out, float* dpred2,
                                          float* drec2, float* dqzp, int* best, float* scratch, long scratch_floats, void* stream) {
    STT_REQUIRE(best, "sttode_loss_objective_live: null pointer");
    return loss_objective_impl(pred, rec, fut, past, qzp, scene_ptr, agent_scene, S, n, K1, D, Dp, zd, scale_mse, scale_rec, kl_denom, min_clip, out,
                               dpred2, drec2, dqzp, best, scratch, scratch_floats, stream);
}

// ---------------------------------------------------------------------------------------------------
// The rows of the decoder's tape that carry a gradient (sttode_loss_objective_live): per agent, of its K1 trajectory columns, sample 0 and
// sample best[a] -> rows 2 a and 2 a + 1 of a compact copy.  Every tensor of the tape is rows of `row` floats, `outer` planes of them
// (the GRU's per-step planes [T][columns][..]); up to STT_GATHER_MAX tensors per launch, one workgroup per (tensor, plane, output row).
// ---------------------------------------------------------------------------------------------------
#define STT_GATHER_MAX 32
struct GatherItem { const float* src; float* dst; long src_plane; long dst_plane; int row; int outer; };   // planes in floats
struct GatherArgs { GatherItem it[STT_GATHER_MAX]; int blk0[STT_GATHER_MAX + 1]; int count; const int* best; int n, K1; };
__global__ __launch_bounds__(128) void live_rows_gather_kernel(GatherArgs g) {
    int p = 0;
    while (p + 1 < g.count && (int)blockIdx.x >= g.blk0[p + 1]) ++p;
    const GatherItem& it = g.it[p];
    const int local = (int)blockIdx.x - g.blk0[p], rows = 2 * g.n;
    const int o = local / rows, r = local % rows, a = r >> 1;
    int sidx = (r & 1) ? g.best[a] : 0;
    sidx = sidx < 0 ? 0 : (sidx >= g.K1 ? g.K1 - 1 : sidx);     // (a `best` nobody wrote must not turn into an out-of-bounds read)
    const float* src = it.src + (long)o * it.src_plane + ((long)a * g.K1 + sidx) * it.row;
    float* dst = it.dst + (long)o * it.dst_plane + (long)r * it.row;
    if ((it.row & 3) == 0 && ((((size_t)src) | ((size_t)dst)) & 15) == 0) {
        for (int i = threadIdx.x; i < it.row / 4; i += 128) reinterpret_cast<f32x4*>(dst)[i] = reinterpret_cast<const f32x4*>(src)[i];
    } else {
        for (int i = threadIdx.x; i < it.row; i += 128) dst[i] = src[i];
    }
}
extern "C" int sttode_live_rows_gather(const void* items_, int count, const int* best, int n, int K1, void* stream) {
    const GatherItem* items = (const GatherItem*)items_;
    STT_REQUIRE(items && best && count > 0 && count <= STT_GATHER_MAX && n > 0 && K1 >= 2, "sttode_live_rows_gather: null pointer or bad counts (at most 32 tensors)");
    GatherArgs g;
    g.count = count; g.best = best; g.n = n; g.K1 = K1;
    long blocks = 0;
    for (int i = 0; i < count; ++i) {
        STT_REQUIRE(items[i].src && items[i].dst && items[i].row > 0 && items[i].outer > 0, "sttode_live_rows_gather: bad item");
        g.it[i] = items[i];
        g.blk0[i] = (int)blocks;
        blocks += (long)items[i].outer * 2 * n;
    }
    STT_REQUIRE(blocks < (1L << 31), "sttode_live_rows_gather: too many rows");
    g.blk0[count] = (int)blocks;
    hipLaunchKernelGGL(live_rows_gather_kernel, dim3((unsigned)blocks), dim3(128), 0, (hipStream_t)stream, g);
    STT_HIP(hipGetLastError());
    return 0;
}
extern "C" int sttode_loss_sqerr(const float* pred, const float* target, long count, float scale, float* out, float* dpred, void* stream) {
    STT_REQUIRE(pred && target && out && count > 0, "sttode_loss_sqerr: bad argument");
    hipLaunchKernelGGL(sqerr_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pred, target, count, scale, out, dpred);
    STT_HIP(hipGetLastError());
    return 0;
}
extern "C" int sttode_loss_kl(const float* params, const int* scene_ptr, int S, int rows, int zd, float denom, float min_clip, float* out,
                              float* dparams, float* scratch, void* stream) {
    STT_REQUIRE(params && out && scratch && rows > 0 && zd > 0, "sttode_loss_kl: bad argument");
    STT_REQUIRE(scene_ptr ? S > 0 : denom > 0.f, "sttode_loss_kl: scene_ptr needs S > 0, otherwise denom > 0");
    const int nb = scene_ptr ? S : 1;
    hipLaunchKernelGGL(kl_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, params, scene_ptr, rows, zd, denom, min_clip, scratch, dparams);
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, scratch, nb, out);
    STT_HIP(hipGetLastError());
    return 0;
}
extern "C" int sttode_loss_diverse(const float* pred, const float* target, const int* scene_ptr, const int* agent_scene, int n, int K,
                                   int D, float* out, float* dpred, float* scratch, void* stream) {
    STT_REQUIRE(pred && target && out && scratch && n > 0 && K > 0 && K <= 64 && D > 0, "sttode_loss_diverse: bad argument (K <= 64, scratch >= n floats)");
    STT_REQUIRE(!scene_ptr || agent_scene, "sttode_loss_diverse: scene_ptr needs agent_scene");
    hipLaunchKernelGGL(diverse_agent_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, pred, target, scene_ptr, agent_scene, n, K, D, scratch, dpred);
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, scratch, n, out);
    STT_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// Values a replayed step hands to the HOST in the middle of its graph (STTODENet.forward() returns four Python floats,
// model/STTODE.py:568 -- `.item()` there is a device synchronisation at the END of whatever is queued; the loss values exist after the
// forward half of the step).  publish: one lane copies n values into pinned host memory with system-scope stores, then bumps a DEVICE
// counter (a replayed graph cannot carry a per-replay argument) and stores the new count into the host's sequence word (release).
// wait: the host spins on that word -- no stream or event is involved, so the rest of the graph (the backward pass) keeps running
// while the caller goes on to zero_grad / backward / optimizer.step and queues them behind it.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void publish_kernel(const float* __restrict__ vals, int n, float* host_vals, unsigned* dev_seq, unsigned* host_seq) {
    if (threadIdx.x != 0) return;
    for (int i = 0; i < n; ++i) __hip_atomic_store(host_vals + i, vals[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned s = *dev_seq + 1u;
    *dev_seq = s;
    __hip_atomic_store(host_seq, s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
extern "C" int sttode_publish_values(const float* vals, int n, float* host_vals, unsigned* dev_seq, unsigned* host_seq, void* stream) {
    STT_REQUIRE(vals && host_vals && dev_seq && host_seq && n > 0 && n <= 64, "sttode_publish_values: null pointer or n outside 1..64");
    hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, vals, n, host_vals, dev_seq, host_seq);
    STT_HIP(hipGetLastError());
    return 0;
}
extern "C" int sttode_wait_value(const unsigned* host_seq, long want, double timeout_s) {
    STT_REQUIRE(host_seq && timeout_s > 0, "sttode_wait_value: null pointer or no time-out");
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        if (__atomic_load_n(host_seq, __ATOMIC_ACQUIRE) == (unsigned)want) return 0;
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
        if ((spins & 1023u) == 1023u &&
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) {
            stt_set_error("sttode_wait_value: the sequence word did not reach the expected count in time");
            return 1;
        }
    }
}
