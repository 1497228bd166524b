// Device primitives for the "column chain" formulation used by every dense layer of the path.
//
// CDNA4 design (not a translation of the reference's row-major addmm calls):
//   * work items (trajectories / agents) sit on the 16 LANE COLUMNS of v_mfma_f32_16x16x4_f32,
//     features sit in REGISTERS: a 16-feature x 16-column tile is one f32x4 per lane.
//       lane l:  c = l & 15 (column), q = l >> 4 (row group);  reg r  <->  feature 16*tile + 4*q + r
//   * a layer is  Y^T[features x cols] = W[features x K] * X^T[K x cols]:  the WEIGHTS are the MFMA
//     A operand and the ACTIVATIONS are the B operand.  The accumulator layout of one layer IS the
//     B-operand layout of the next (k = 16*T + 4*q + r), so a whole MLP / GRU chain runs in registers
//     with no LDS round trip and no transposes; relu / sigmoid / tanh are applied on the accumulators.
//   * weights are packed once on the host in fragment order ("PK16", sttode_amd/packing.py):
//       P[((it*KT + T)*64 + lane)*4 + r] = W[16*it + (lane&15)][16*T + 4*(lane>>4) + r]
//     so one wave's A fragment for 4 consecutive MFMAs is a single contiguous 1 KiB read
//     (ds_read_b128 / global_load_dwordx4, lane-linear, conflict-free by construction).
//   * fp32-in / fp32-accumulate MFMA is bit-exact fmaf chaining: same precision class as the
//     reference's fp32 addmm (tolerance 1e-4 relative is met with ~1e-6 to spare).
#pragma once
#include <hip/hip_runtime.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define STT_WAVE 64
// Compiler-only fence: stops hipcc hoisting every fragment load of an unrolled region to its top
// (which blows the VGPR budget and spills).  No instruction is emitted.
#define STT_FENCE() asm volatile("" ::: "memory")

__device__ __forceinline__ f32x4 mfma_k16(f32x4 acc, const f32x4 w, const f32x4 b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[0], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[2], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[3], b[3], acc, 0, 0, 0);
    return acc;
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, const f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

__device__ __forceinline__ f32x4 relu4(f32x4 v) {
    v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
    return v;
}

// Gate nonlinearities on the hardware transcendental unit: v_exp_f32 (2^x, ~1 ulp) and v_rcp_f32 (~1 ulp).
// Absolute error ~1e-7 on outputs in [-1, 1]: two orders of magnitude inside the 1e-4 parity bar, and ~4x fewer
// VALU instructions than expf + IEEE division (the gate math otherwise steals ~40% of the GRU's MFMA issue time).
// Saturation: exp2 -> inf gives rcp -> 0; exp2 -> 0 gives rcp(1) = 1.
__device__ __forceinline__ float sigmoidf_(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float tanhf_(float x) {
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}
// Pre-scaled forms: the GRU packs its gate rows already multiplied by -log2(e) (r, z) and 2*log2(e) (n), so the
// argument arrives as t = -x*log2(e) resp. t = 2*x*log2(e) and the multiply disappears from the (non-overlappable) VALU work.
__device__ __forceinline__ float sigmoid_prescaled(float t) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t)); }
__device__ __forceinline__ float tanh_prescaled(float t) { return fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t)), 1.0f); }

__device__ __forceinline__ f32x4 splat4(float v) { f32x4 r = {v, v, v, v}; return r; }

// Sum over the 4 row groups (lanes c, c+16, c+32, c+48) -> every lane of a column gets the column total.
__device__ __forceinline__ float colsum_q(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// LayerNorm over 64 features held as 4 tiles (x[T][r] <-> feature 16T+4q+r), per column; gamma/beta row-major [64].
__device__ __forceinline__ void layernorm64(f32x4 (&x)[4], const float* __restrict__ gamma, const float* __restrict__ beta, int q) {
    float s = 0.f;
#pragma unroll
    for (int T = 0; T < 4; ++T) s += (x[T][0] + x[T][1]) + (x[T][2] + x[T][3]);
    const float mean = colsum_q(s) * (1.0f / 64.0f);
    float v = 0.f;
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float d = x[T][r] - mean; v += d * d; }
    const float rstd = 1.0f / sqrtf(colsum_q(v) * (1.0f / 64.0f) + 1e-5f);
#pragma unroll
    for (int T = 0; T < 4; ++T) {
        const f32x4 g = ld4(gamma + 16 * T + 4 * q), b = ld4(beta + 16 * T + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) x[T][r] = (x[T][r] - mean) * rstd * g[r] + b[r];
    }
}

// LDS byte address of a pointer into shared memory (generic -> address space 3)
// Workgroup barrier for LDS exchange ONLY: LDS operations complete (lgkmcnt), global loads in flight stay in flight.  __syncthreads()
// carries a workgroup-scope fence, i.e. `s_waitcnt vmcnt(0)` in front of every s_barrier: a weight prefetch issued groups ahead would be
// drained at the next exchange and every loop iteration would expose a full L2 round trip (measured in the latency forms: 1.8 us per
// 0.97-us group of MFMAs).  Not for hand-offs through global memory.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p;
}
// LDS-DMA issued from inline asm, so hipcc does NOT track it: with the builtin form it inserts s_waitcnt vmcnt(0) in front of
// later LDS reads it cannot prove disjoint from the DMA's destination (here: every tile read of the other ring buffer), which
// serialises DMA and MFMA -- measured 3.4x the MFMA time per chunk.  The completion is counted by hand: ChainStream::end() waits
// vmcnt(0) before the workgroup barrier (the guide's recipe: cdna_hip_programming.md §5.7, glds16_asm).  `lds_dst` is the
// wave-uniform destination byte address; the hardware adds lane * 16.
__device__ __forceinline__ void glds16_asm(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
