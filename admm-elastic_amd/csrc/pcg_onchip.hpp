// pcg_onchip.hpp -- the whole Jacobi-PCG solve of the ADMM global step as ONE persistent launch.
//
// Replaces the prefactored LDLT solve of src/LinearSolver.hpp:87-90 (same system, same stop rule as the
// two-kernels-per-iteration path in kernels.hpp; that path stays as the fallback for systems that do not
// fit).  Why: at the BASELINE sizes one CG iteration moves ~90 MB (matrix 32 MB + 14 vector passes) through
// L2 / Infinity Cache although the whole problem fits ON CHIP: 256 CUs x (160 KB LDS + 512 KB VGPRs).
//   * block = one CU, wave = one 64-row SELL slice, thread = one vertex (3 dofs);
//   * the thread's matrix row lives in LDS for the whole solve (loaded once, column-major per wave so the
//     reads are conflict-free), its x / u / p / s / dinv / m entries live in registers;
//   * the only per-iteration memory traffic is the search-space vector u: 32 B per vertex published with
//     write-through (sc1) stores and gathered with sc1 loads (per-XCD L2s are not coherent), plus one
//     64-byte record of partial dot products per block;
//   * two grid barriers per iteration (Chronopoulos-Gear CG has ONE reduction point: u visible -> SpMV ->
//     partials visible -> alpha/beta), XCD-hierarchical counters, relaxed agent-scope polling, bounded
//     spins (a barrier that cannot complete aborts the solve with an error instead of hanging the GPU);
//   * every block reduces the partial records in the same fixed order, so all blocks take the same
//     convergence decision and the result is deterministic run to run.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels.hpp"

namespace admm_k {

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

struct OcArgs {
    int n_rows, n_slices;
    const int *ptr, *w, *col; const double *val;   // SELL-64 of Ahat
    const double *m, *dinv, *b;
    double *x, *u_out;
    double *ubuf;       // [64 n_slices][4] published u (x, y, z, pad)
    double *part;       // [G][8]  per-block partial sums (gamma[3], delta[3])
    double *part_b;     // [G][4]  per-block partial sums of b . dinv . b
    unsigned *bar;      // barrier words, 16-word (64 B) stride: [0..7] group counters, [8] top, [9..16] generations, [17] abort
    int *counters; CgScal *scal; int *sig;
    unsigned long long *prof;   // diagnosis only (ADMM_HIP_OC_PROF=1): [64][8] timestamps of block 0
    int spb, wl, G, max_iters, seq;
    double tol2;
};

constexpr int kOcScratch = 2048;          // bytes of LDS ahead of the matrix slab
constexpr unsigned kOcSpinLimit = 4000000u;

__device__ __forceinline__ void oc_store_sc1(__amdgpu_buffer_rsrc_t rs, int byte_off, double a, double b) {
    union { double d[2]; v4u v; } t; t.d[0] = a; t.d[1] = b;
    __builtin_amdgcn_raw_buffer_store_b128(t.v, rs, byte_off, 0, 16 /* sc1: write-through */);
}
__device__ __forceinline__ void oc_store_sc1(__amdgpu_buffer_rsrc_t rs, int byte_off, double a) {
    union { double d; v2u v; } t; t.d = a;
    __builtin_amdgcn_raw_buffer_store_b64(t.v, rs, byte_off, 0, 16);
}
__device__ __forceinline__ void oc_load_sc1(__amdgpu_buffer_rsrc_t rs, int byte_off, double *g) {
    union { double d[2]; v4u v; } lo; union { double d; v2u v; } hi;
    lo.v = __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16);
    hi.v = __builtin_amdgcn_raw_buffer_load_b64(rs, byte_off + 16, 0, 16);
    g[0] = lo.d[0]; g[1] = lo.d[1]; g[2] = hi.d;
}
__device__ __forceinline__ double oc_load_sc1_f64(__amdgpu_buffer_rsrc_t rs, int byte_off) {
    union { double d; v2u v; } t;
    t.v = __builtin_amdgcn_raw_buffer_load_b64(rs, byte_off, 0, 16);
    return t.d;
}

// Grid barrier: every payload store before it was a write-through (sc1) store, so no release fence is
// needed -- every wave drains its stores, one lane arrives.  Hierarchical: blocks of group (blockIdx & 7)
// -- the XCD the block runs on, by observation; correctness does not depend on it -- count on their own
// word, the last of a group counts on the top word, the last of all publishes the generation to all groups.
__device__ __forceinline__ bool oc_barrier(unsigned *bar, unsigned epoch, int G, int *ok_lds, int *sig) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const int x = (int)blockIdx.x & 7;
        const unsigned nx = (unsigned)((G + 7 - x) >> 3);
        const unsigned ng = (unsigned)(G < 8 ? G : 8);
        const unsigned old = __hip_atomic_fetch_add(bar + 16 * x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1u == nx * epoch) {
            const unsigned t = __hip_atomic_fetch_add(bar + 16 * 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t + 1u == ng * epoch)
                for (unsigned g = 0; g < ng; ++g)
                    __hip_atomic_store(bar + 16 * (9 + g), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        int ok = 1;
        unsigned spins = 0;
        while (__hip_atomic_load(bar + 16 * (9 + x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > kOcSpinLimit || __hip_atomic_load(bar + 16 * 17, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(bar + 16 * 17, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(sig + 2, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                ok = 0;
                break;
            }
        }
        *ok_lds = ok;
    }
    __syncthreads();
    return *ok_lds != 0;
}

// block-wide sums of NQ quantities over nw waves; result valid in every thread, fixed summation order
template <int NQ>
__device__ __forceinline__ void oc_block_sum(double *q, double *red, int nw) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const double s = wave_sum(q[i]);
        if (lane == 0) red[wv * NQ + i] = s;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        double s = 0.0;
        for (int w = 0; w < nw; ++w) s += red[w * NQ + i];
        q[i] = s;
    }
    __syncthreads();
}

// acc = sum_k Ahat(row, k) * in[col_k] for the three axes of this thread's row.  Columns [0, wl_s) come from
// the LDS slab, the rest (only when a slice is wider than the slab) from global memory.  Two batches of four
// gathers are kept in flight.
template <bool FROM_UBUF>
__device__ __forceinline__ void oc_row(__amdgpu_buffer_rsrc_t rs, const double *__restrict__ xin, const double *lv, const int *lc,
                                       int wl_s, int w, const int *__restrict__ cpg, const double *__restrict__ vpg, double *acc) {
    acc[0] = acc[1] = acc[2] = 0.0;
    if (w == 0) return;
    int c[4]; double v[4]; double g[12];
    auto fetch = [&](int k, int *cc, double *vv) {
        if (k < wl_s) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { cc[i] = lc[64 * (k + i)]; vv[i] = lv[64 * (k + i)]; }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) { cc[i] = cpg[64 * (k + i)]; vv[i] = vpg[64 * (k + i)]; }
        }
    };
    auto gather = [&](const int *cc, double *gg) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (FROM_UBUF) oc_load_sc1(rs, cc[i] * 32, gg + 3 * i);
            else { const double *p = xin + 3 * (size_t)cc[i]; gg[3 * i] = p[0]; gg[3 * i + 1] = p[1]; gg[3 * i + 2] = p[2]; }
        }
    };
    fetch(0, c, v);
    gather(c, g);
    for (int k = 4; k < w; k += 4) {
        int cn[4]; double vn[4]; double gn[12];
        fetch(k, cn, vn);
        gather(cn, gn);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[0] = fma(v[i], g[3 * i], acc[0]); acc[1] = fma(v[i], g[3 * i + 1], acc[1]); acc[2] = fma(v[i], g[3 * i + 2], acc[2]);
            c[i] = cn[i]; v[i] = vn[i];
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) g[i] = gn[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        acc[0] = fma(v[i], g[3 * i], acc[0]); acc[1] = fma(v[i], g[3 * i + 1], acc[1]); acc[2] = fma(v[i], g[3 * i + 2], acc[2]);
    }
}

template <int MAXT>
__global__ __launch_bounds__(MAXT) void k_pcg_onchip(OcArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *red = (double *)smem;                 // [16 * 9]
    int *ok_lds = (int *)(smem + 16 * 9 * 8);     // barrier verdict
    const int T = (int)blockDim.x, tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = T >> 6;
    double *lv_all = (double *)(smem + kOcScratch);
    int *lc_all = (int *)(lv_all + (size_t)a.spb * a.wl * 64);
    const double *lv = lv_all + (size_t)wv * a.wl * 64 + lane;
    const int *lc = lc_all + (size_t)wv * a.wl * 64 + lane;

    const int s = __builtin_amdgcn_readfirstlane((int)blockIdx.x * a.spb + wv);
    const bool live_slice = s < a.n_slices;
    const int row = s * 64 + lane;
    const bool live = live_slice && row < a.n_rows;
    const int w = live_slice ? a.w[s] : 0;
    const int base = live_slice ? a.ptr[s] : 0;
    const int wl_s = w < a.wl ? w : a.wl;
    const int *cpg = a.col + base + lane;
    const double *vpg = a.val + base + lane;
    {   // the thread's matrix row -> LDS, once per solve
        double *lvw = lv_all + (size_t)wv * a.wl * 64 + lane;
        int *lcw = lc_all + (size_t)wv * a.wl * 64 + lane;
        for (int k = 0; k < wl_s; ++k) { lvw[64 * k] = vpg[64 * k]; lcw[64 * k] = cpg[64 * k]; }
    }
    __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc((void *)a.ubuf, 0, a.n_slices * 64 * 32, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc((void *)a.part, 0, a.G * 64, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void *)a.part_b, 0, a.G * 32, 0x00020000);

    double rx[3], ru[3], rp[3], rsv[3], rd[3], rm[3], rid[3];
    double q[9];
    {   // u0 = dinv (b - A x0), partial b . dinv . b
        double acc[3];
        oc_row<false>(rs_u, a.x, lv, lc, wl_s, w, cpg, vpg, acc);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const size_t i = 3 * (size_t)(live ? row : 0) + j;
            const double bi = live ? a.b[i] : 0.0;
            rx[j] = live ? a.x[i] : 0.0; rd[j] = live ? a.dinv[i] : 0.0; rm[j] = live ? a.m[i] : 0.0;
            rid[j] = live ? fast_rcp(rd[j]) : 0.0;
            const double ri = bi - fma(rm[j], rx[j], acc[j]);
            ru[j] = rd[j] * ri;
            rp[j] = 0.0; rsv[j] = 0.0;
            q[j] = bi * rd[j] * bi;
        }
        if (live_slice) { oc_store_sc1(rs_u, row * 32, ru[0], ru[1]); oc_store_sc1(rs_u, row * 32 + 16, ru[2]); }
        oc_block_sum<3>(q, red, nw);
        if (tid < 3) oc_store_sc1(rs_b, (int)blockIdx.x * 32 + 8 * tid, tid == 0 ? q[0] : (tid == 1 ? q[1] : q[2]));
    }

    unsigned epoch = 0;
    double g_prev[3] = {0, 0, 0}, a_prev[3] = {0, 0, 0}, gb[3] = {0, 0, 0}, g_last[3] = {0, 0, 0};
    int iters = 0;
    bool conv = false, aborted = false;
    const bool prof = a.prof && blockIdx.x == 0 && tid == 0;
#define OC_STAMP(slot) do { if (prof && it < 64) a.prof[it * 8 + (slot)] = wall_clock64(); } while (0)
    for (int it = 0; it < a.max_iters; ++it) {
        OC_STAMP(0);
        if (!oc_barrier(a.bar, ++epoch, a.G, ok_lds, a.sig)) { aborted = true; break; }
        OC_STAMP(1);
        double acc[3], rw[3];
        oc_row<true>(rs_u, nullptr, lv, lc, wl_s, w, cpg, vpg, acc);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            rw[j] = fma(rm[j], ru[j], acc[j]);
            q[j] = ru[j] * ru[j] * rid[j];
            q[3 + j] = rw[j] * ru[j];
        }
        {   // block partials -> this block's record
            const int ln = tid & 63;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const double sm = wave_sum(q[i]);
                if (ln == 0) red[wv * 6 + i] = sm;
            }
            __syncthreads();
            if (tid < 6) {
                double sm = 0.0;
                for (int ww = 0; ww < nw; ++ww) sm += red[ww * 6 + tid];
                oc_store_sc1(rs_p, (int)blockIdx.x * 64 + 8 * tid, sm);
            }
        }
        OC_STAMP(2);
        if (!oc_barrier(a.bar, ++epoch, a.G, ok_lds, a.sig)) { aborted = true; break; }
        OC_STAMP(3);
        // every block reduces all records in the same order (waves 0..3 only hold data)
#pragma unroll
        for (int i = 0; i < 9; ++i) q[i] = 0.0;
        const int nrw = nw < 4 ? nw : 4;
        if (wv < nrw) {
            for (int i = tid; i < a.G; i += 64 * nrw) {
#pragma unroll
                for (int kk = 0; kk < 6; ++kk) q[kk] += oc_load_sc1_f64(rs_p, i * 64 + 8 * kk);
                if (it == 0) {
#pragma unroll
                    for (int kk = 0; kk < 3; ++kk) q[6 + kk] += oc_load_sc1_f64(rs_b, i * 32 + 8 * kk);
                }
            }
        }
        if (it == 0) oc_block_sum<9>(q, red, nrw); else oc_block_sum<6>(q, red, nrw);
        OC_STAMP(4);
        double alpha[3], beta[3];
        conv = true;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (it == 0) gb[j] = q[6 + j];
            g_last[j] = q[j];
            conv = conv && (q[j] <= a.tol2 * gb[j] + 1e-300);
        }
        if (conv) break;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double g = q[j], d = q[3 + j];
            if (it == 0) { beta[j] = 0.0; alpha[j] = (d > 0.0) ? g / d : 0.0; }
            else {
                beta[j] = (g_prev[j] > 0.0) ? g / g_prev[j] : 0.0;
                const double den = (a_prev[j] != 0.0) ? d - beta[j] * g / a_prev[j] : d;
                alpha[j] = (den > 0.0) ? g / den : 0.0;
            }
            g_prev[j] = g; a_prev[j] = alpha[j];
            const double pi = fma(beta[j], rp[j], ru[j]);
            const double si = fma(beta[j], rsv[j], rw[j]);
            rp[j] = pi; rsv[j] = si;
            rx[j] = fma(alpha[j], pi, rx[j]);
            ru[j] = fma(-alpha[j] * rd[j], si, ru[j]);      // u = M^-1 (r - alpha s)
        }
        ++iters;
        if (live_slice) { oc_store_sc1(rs_u, row * 32, ru[0], ru[1]); oc_store_sc1(rs_u, row * 32 + 16, ru[2]); }
        OC_STAMP(5);
    }
#undef OC_STAMP
    if (live) {
#pragma unroll
        for (int j = 0; j < 3; ++j) { a.x[3 * (size_t)row + j] = rx[j]; a.u_out[3 * (size_t)row + j] = ru[j]; }
    }
    if (blockIdx.x == 0 && tid == 0) {
        CgScal o;
#pragma unroll
        for (int j = 0; j < 3; ++j) { o.gamma[j] = g_last[j]; o.alpha[j] = a_prev[j]; o.gamma_b[j] = gb[j]; }
        o.converged = (conv && !aborted) ? 1 : 0; o.iters = iters; o.seq = a.seq; o.pad_ = 0;
        a.scal[0] = o;
        atomicAdd(a.counters, iters);
        if (o.converged) {
            atomicAdd(a.counters + 4, 1);
            atomicMax(a.counters + 3, iters);
            a.counters[8 + (a.seq & 63)] = iters;
        }
    }
}

} // namespace admm_k
