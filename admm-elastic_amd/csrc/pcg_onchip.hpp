// pcg_onchip.hpp -- the whole Jacobi-PCG solve of the ADMM global step as ONE persistent launch.
//
// Replaces the prefactored LDLT solve of src/LinearSolver.hpp:87-90 (same system, same stop rule as the
// two-kernels-per-iteration path in kernels.hpp; that path stays as the fallback for systems that do not
// fit).  Why: at the BASELINE sizes one CG iteration moves ~90 MB (matrix 32 MB + 14 vector passes) through
// L2 / Infinity Cache although the whole problem fits ON CHIP: 256 CUs x (160 KB LDS + 512 KB VGPRs).
//   * block = one CU, wave = one 64-row SELL slice, thread = one vertex (3 dofs);
//   * the thread's matrix row lives in LDS for the whole solve (loaded once, column-major per wave so the
//     reads are conflict-free ds_reads), its x / u / w / p / s / z / dinv / m entries live in registers;
//   * the only per-iteration memory traffic is one vector: 24 B per vertex published per axis (SoA) with
//     16-byte write-through (sc1) stores and gathered with sc1 loads (per-XCD L2s are not coherent), plus one
//     record of six partial dot products per block;
//   * ONE grid barrier per iteration: the recurrences are those of pipelined CG (Ghysels & Vanroose 2014),
//     whose dot products (r,u), (w,u) use vectors that exist BEFORE the SpMV n = A M^-1 w, so a block
//     publishes its slice of M^-1 w and its partial dot products together, crosses one barrier, and then
//     gathers and reduces at the same time.  (Measured per iteration: Chronopoulos-Gear with two barriers
//     17.6 us; pipelined 9.0 us = 1.0 us draining the write-through stores + 2.4 us barrier + 3.1 us gather and
//     reduction + the rest.)  Barrier: eight per-group counters, one fire-and-forget arrival, relaxed
//     agent-scope polling, bounded spins (a barrier that cannot complete aborts the solve with an error instead
//     of hanging the GPU); two counter sets alternate between solves so nothing is cleared between launches.
//     The gather does not wait for that barrier: a block announces its published slice with a (solve, phase)
//     flag and starts gathering as soon as the <= 64 blocks whose rows its matrix rows reference have
//     announced theirs (host-built neighbour lists; 1.6 us instead of 3.4 us); the barrier completes behind
//     the gather and only gates the reduction of the partial sums.  8.2 us per iteration;
//   * the recycled (Galerkin) warm start of the ADMM loop is the first phase of the same launch and the new
//     (correction, A correction) pair is written in its epilogue;
//   * pipelined CG carries w = A u by recurrence, so its recursive residual can drift from the true one.
//     The stop rule is therefore applied to the TRUE residual: when the recurrence reports convergence the
//     kernel recomputes r = b - A x, tests  r.M^-1 r <= tol^2 b.M^-1 b  (the rule of the launch path), and
//     if the test fails, restarts CG from that true residual.  Pipelined CG is also known to stagnate -- and
//     then blow up -- when asked for residuals near the FP64 floor of a system, so it is only used down to a
//     relative residual of 1e-9 (kOcPipeFloor): below that, or after 50 iterations without a new residual
//     minimum (200 while the residual is still far from that floor), the kernel continues SEAMLESSLY (same x, u, p,
//     gamma) in the CLASSIC Hestenes-Stiefel form: p = u + beta p, s = A p, alpha = gamma / (p . s) -- three
//     synchronisations per iteration, but p . A p is computed directly.  (The end game used to run the
//     Chronopoulos-Gear recurrences, two synchronisations; their alpha = gamma / (delta - beta gamma / alpha') is a
//     difference of nearly equal numbers near the floor, and on free nearly incompressible bodies -- cond ~ 1e7,
//     the systems UzawaCG hands down with sparse right-hand sides C^T d -- it sent x to 1e254.  The end game is a
//     few iterations of a solve, so the third synchronisation costs nothing measurable.)  A verification that fails
//     twice without a 4x improvement means the FP64 floor has been reached: the solve stops as converged, as a
//     recurrence-only CG would;
//   * the three axes are independent systems (A = Ahat (x) I3) with their own b . M^-1 b; an axis whose right-hand
//     side vanishes (C^T d of a floor contact has no x / z part) is measured against the largest axis, and b = 0
//     returns x = 0 without iterating;
//   * sums that turn non-finite, or grow 1e8x (in norm) above the best residual seen, send the solve back to the
//     ENTRY x (still in global memory: x is written once, in the epilogue) and on in the classic form; a second
//     failure returns the entry x, reported as unconverged -- never a non-finite vector;
//   * preconditioner: Jacobi, or -- on 2-colourable meshes, the default -- a symmetric Gauss-Seidel sweep over the rows and
//     columns a block owns (MODE 2, block_prec below): it reads only what is already on the CU, so it adds no exchange
//     (the exchange, not the reduction, bounds the iteration: see the Chebyshev mode, MODE 1, which trades reductions
//     for exchanges and is slower), and it cuts the iterations 1.55x (124.8 -> 80.6 per solve at 1 M tets);
//   * every block reduces the partial records in the same fixed order, so all blocks take the same
//     decisions and the result is deterministic run to run.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels.hpp"

namespace admm_k {

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
// LDS-qualified element types: pointers built by arithmetic on the dynamic LDS base otherwise decay to generic
// (flat) pointers and every matrix access becomes a flat_load instead of a ds_read
typedef __attribute__((address_space(3))) double LdsD;
typedef __attribute__((address_space(3))) int LdsI;

struct OcArgs {
    int n_rows, n_slices;
    const int *ptr, *w, *col; const double *val;   // SELL-64 of Ahat
    const double *m, *dinv, *b;
    double *x, *u_out;
    double *ubuf;       // [2][3][64 n_slices] published vector, per axis, double-buffered by phase parity
    double *part;       // [2][8][G] per-block partial sums, double-buffered by phase parity
    unsigned *bar;      // [2 sets][32 * 16] barrier words, 16-word (64 B) stride: [0..7] group counters, [17] abort
    int *counters; CgScal *scal; int *sig;
    // neighbour hand-off of the pipelined iteration: a block may gather as soon as the blocks it reads from have
    // published (their flags carry (solve, phase)); the grid barrier completes behind the gather
    const int *nbr;               // [G][64] blocks whose rows this block's matrix rows reference (-1 = none); nullptr: feature off
    unsigned long long *flags;    // [G] (8 words apart) last published (seq << 32 | phase) of every block
    unsigned long long *prof;   // diagnosis only (ADMM_HIP_OC_PROF=1): [64][8] timestamps of block prof_block
    int prof_block;
    int spb, wl, G, max_iters, seq;
    // recycled (Galerkin) warm start inside the launch (replaces k_rc_resid / dots / solve / apply / record):
    int rc_on;          // 1: project the initial error on the stored pairs, and store this solve's pair at the end
    RcBasis rc;         // the (up to kRc) pairs (E_j, R_j = A E_j) to project on
    double *rc_xs, *rc_r0;       // scratch: x and residual at entry (read back at the end for the new pair)
    double *rc_Eslot, *rc_Rslot; // where this solve's pair goes
    double *rc_part;             // [3 * kRcQ rounded up to 72][G] block partial sums of the projection
    double tol2;
    // Chebyshev polynomial of D^-1 A as preconditioner (poly_m = degree, 0 / 1 = plain Jacobi): u = sum_k d_k with
    // d_0 = D^-1 r / theta, d_{k+1} = c1[k] d_k + c2[k] (res_k - D^-1 A d_k)  (Saad, Iterative Methods, Alg. 12.1)
    int poly_m;
    double cheb_inv_theta, cheb_c1[8], cheb_c2[8];
    // block-local symmetric Gauss-Seidel preconditioner (MODE 2): colour (0 / 1) of every row of a 2-colourable Ahat
    const signed char *row_color;
};

constexpr int kOcSubK = 4;                // aggregates per block of the two-level preconditioner (= admm_host::kOcSub, pcg_onchip2.hpp)
constexpr int kOcScratch = 6144;          // bytes of LDS scratch ahead of the per-wave staging area and the matrix slab
constexpr int kOcStage = 3 * 64 * 8;      // per-wave staging area (publish transposition)
constexpr unsigned kOcSpinLimit = 4000000u;
constexpr double kOcPipeFloor = 1e-18;    // squared relative residual below which the pipelined recurrences are not trusted
#ifndef ADMM_OC_TRIG
#define ADMM_OC_TRIG 0.9
#endif
constexpr double kOcTrig = ADMM_OC_TRIG;  // the recurrence must report gamma <= kOcTrig tol^2 b.M^-1 b before the true residual is checked
constexpr int kOcStagnation = 50;         // pipelined iterations without a new residual minimum before switching, once the
                                          // residual is within 100x of kOcPipeFloor (rounding-driven stagnation); above that
                                          // level plateaus of the residual norm are ordinary CG behaviour (measured: 13 of 40
                                          // solves of the 1M-tet bench plateau for > 50 iterations) and the window is 4x longer

__device__ __forceinline__ void oc_store_sc1(__amdgpu_buffer_rsrc_t rs, int byte_off, double a, double b) {
    union { double d[2]; v4u v; } t; t.d[0] = a; t.d[1] = b;
    __builtin_amdgcn_raw_buffer_store_b128(t.v, rs, byte_off, 0, 16 /* sc1: write-through */);
}
__device__ __forceinline__ void oc_store_sc1(__amdgpu_buffer_rsrc_t rs, int byte_off, double a) {
    union { double d; v2u v; } t; t.d = a;
    __builtin_amdgcn_raw_buffer_store_b64(t.v, rs, byte_off, 0, 16);
}
// published vectors are stored per axis (SoA): a wave's gather of one neighbour column is then three fully
// coalesced 512-byte requests (12 cache lines) instead of 32 lines of a padded 32-byte AoS record
__device__ __forceinline__ void oc_load_sc1(__amdgpu_buffer_rsrc_t rs, int byte_off, int axis_stride, double *g) {
    union { double d; v2u v; } t0, t1, t2;
    t0.v = __builtin_amdgcn_raw_buffer_load_b64(rs, byte_off, 0, 16);
    t1.v = __builtin_amdgcn_raw_buffer_load_b64(rs, byte_off + axis_stride, 0, 16);
    t2.v = __builtin_amdgcn_raw_buffer_load_b64(rs, byte_off + 2 * axis_stride, 0, 16);
    g[0] = t0.d; g[1] = t1.d; g[2] = t2.d;
}
__device__ __forceinline__ double oc_load_sc1_f64(__amdgpu_buffer_rsrc_t rs, int byte_off) {
    union { double d; v2u v; } t;
    t.v = __builtin_amdgcn_raw_buffer_load_b64(rs, byte_off, 0, 16);
    return t.d;
}

// Grid barrier: every payload store before it was a write-through (sc1) store, so no release fence is
// needed -- every wave drains its stores, one lane arrives.  Eight monotonic counters, one per group
// (blockIdx & 7 = the XCD the block runs on, by observation; correctness does not depend on it): a block
// arrives with ONE non-returning atomic on its group's counter (fire and forget: 32 arrivals per word, no
// second level to wait for) and lanes 0..7 of wave 0 poll the eight counters with relaxed agent-scope loads
// until each has reached (blocks in the group) x epoch.  Measured against the two-level form (per-group
// counter -> top counter -> generation word): see DESIGN.md.
__device__ __forceinline__ bool oc_barrier(unsigned *bar, unsigned epoch, int G, int *ok_lds, int *sig) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x < 64) {
        const int lane = (int)threadIdx.x;
        if (lane == 0) __hip_atomic_fetch_add(bar + 16 * ((int)blockIdx.x & 7), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int x = lane & 7;
        const unsigned need = (unsigned)((G + 7 - x) >> 3) * epoch;      // 0 for groups without blocks
        unsigned *word = bar + 16 * (lane < 8 ? x : 17);                  // lane 8 watches the abort word in the same load
        int ok = 1;
        unsigned spins = 0;
        while (true) {
            const unsigned v = (lane < 9) ? __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            if (__all(lane >= 8 || v >= need)) break;
            if (++spins > kOcSpinLimit || __any(lane == 8 && v != 0u)) {
                if (lane == 0) {
                    __hip_atomic_store(bar + 16 * 17, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(sig + 2, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (lane == 0) *ok_lds = ok;
    }
    __syncthreads();
    return *ok_lds != 0;
}

// Pipelined iteration, first half of the synchronisation: drain, announce (flag for the neighbours + arrival on the
// grid barrier, both fire-and-forget), then wait only for the blocks this block gathers from.
// (ARRIVE = false: flag and wait only -- pcg_onchip2.hpp crosses its grid barrier after the gather)
template <bool ARRIVE = true>
__device__ __forceinline__ bool oc_announce_and_wait_neighbours(unsigned *bar, unsigned long long *flags, const int *nbr, unsigned seq, unsigned epoch,
                                                                int *ok_lds, int *sig) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x < 64) {
        const int lane = (int)threadIdx.x;
        const unsigned long long tag = ((unsigned long long)seq << 32) | epoch;
        if (lane == 0) {
            __hip_atomic_store(flags + 8 * blockIdx.x, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ARRIVE) __hip_atomic_fetch_add(bar + 16 * ((int)blockIdx.x & 7), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const int nb = nbr[64 * blockIdx.x + lane];
        int ok = 1;
        unsigned spins = 0;
        while (true) {
            const unsigned long long v = (nb >= 0) ? __hip_atomic_load(flags + 8 * nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tag;
            if (__all(v >= tag)) break;
            if (++spins > kOcSpinLimit || ((spins & 255u) == 0u && __hip_atomic_load(bar + 16 * 17, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                if (lane == 0) {
                    __hip_atomic_store(bar + 16 * 17, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(sig + 2, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (lane == 0) *ok_lds = ok;
    }
    __syncthreads();
    return *ok_lds != 0;
}
// ... second half: the grid barrier this block has already arrived at (wave 0 polls the eight counters)
__device__ __forceinline__ bool oc_barrier_wait(unsigned *bar, unsigned epoch, int G, int *ok_lds, int *sig) {
    if (threadIdx.x < 64) {
        const int lane = (int)threadIdx.x;
        const int x = lane & 7;
        const unsigned need = (unsigned)((G + 7 - x) >> 3) * epoch;
        unsigned *word = bar + 16 * (lane < 8 ? x : 17);
        int ok = 1;
        unsigned spins = 0;
        while (true) {
            const unsigned v = (lane < 9) ? __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            if (__all(lane >= 8 || v >= need)) break;
            if (++spins > kOcSpinLimit || __any(lane == 8 && v != 0u)) {
                if (lane == 0) {
                    __hip_atomic_store(bar + 16 * 17, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(sig + 2, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (lane == 0) *ok_lds = ok;
    }
    __syncthreads();
    return *ok_lds != 0;
}

// Block totals of q[0..5] -> this block's record of the given parity (SoA: quantity-major, so the readers
// are coalesced).  Wave level: two halving butterfly steps inside each quad (8 -> 4 -> 2 quantities per lane),
// then two rotation steps over the 16-lane row, all with DPP; the four rows of a wave and the waves of the
// block are summed through LDS by six threads in a fixed order.  One __syncthreads; `red` may be reused after
// the next barrier.
__device__ __forceinline__ void oc_publish_partials(const double *q6, double *red /* [16][4][8] */, int nw, __amdgpu_buffer_rsrc_t rs_p, int par, int G,
                                                    double q7 = 0.0, int nq = 6) {
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const double q8[8] = {q6[0], q6[1], q6[2], q6[3], q6[4], q6[5], q7, 0.0};
    double b[2];
    row_sum8(q8, b[0], b[1]);
    if ((lane & 15) < 4) {
        const int id = 4 * (lane & 1) + (lane & 2);
        double *dst = red + ((wv * 4 + (lane >> 4)) * 8 + id);
        dst[0] = b[0]; dst[1] = b[1];
    }
    __syncthreads();
    if (tid < nq) {
        double sm = 0.0;
        for (int r = 0; r < 4 * nw; ++r) sm += red[r * 8 + tid];
        oc_store_sc1(rs_p, ((par * 8 + tid) * G + (int)blockIdx.x) * 8, sm);
    }
}

// acc = sum_k Ahat(row, k) * in[col_k] for the three axes of this thread's row.  Columns [0, wl_s) come from
// the LDS slab, the rest (only when a slice is wider than the slab) from global memory.  DEEP keeps two
// batches of four gathers in flight (the <= 768-thread variant has the registers for it).
template <bool FROM_UBUF, bool DEEP>
__device__ __forceinline__ void oc_row(__amdgpu_buffer_rsrc_t rs, int buf_off, int axis_stride, const double *__restrict__ xin, const LdsD *lv, const LdsI *lc,
                                       int wl_s, int w, const int *__restrict__ cpg, const double *__restrict__ vpg, double *acc) {
    acc[0] = acc[1] = acc[2] = 0.0;
    if (w == 0) return;
    double g[12];
    // the matrix values are read when a batch is consumed (LDS latency is small next to the gathers), so only
    // the gathered vector entries occupy registers while they are in flight
    auto gather = [&](int k, double *gg) {
        int cc[4];
        if (k < wl_s) {
#pragma unroll
            for (int i = 0; i < 4; ++i) cc[i] = lc[64 * (k + i)];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) cc[i] = cpg[64 * (k + i)];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (FROM_UBUF) oc_load_sc1(rs, buf_off + cc[i] * 8, axis_stride, gg + 3 * i);
            else { const double *p = xin + 3 * (size_t)cc[i]; gg[3 * i] = p[0]; gg[3 * i + 1] = p[1]; gg[3 * i + 2] = p[2]; }
        }
    };
    auto consume = [&](int k, const double *gg) {
        double vv[4];
        if (k < wl_s) {
#pragma unroll
            for (int i = 0; i < 4; ++i) vv[i] = lv[64 * (k + i)];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) vv[i] = vpg[64 * (k + i)];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[0] = fma(vv[i], gg[3 * i], acc[0]); acc[1] = fma(vv[i], gg[3 * i + 1], acc[1]); acc[2] = fma(vv[i], gg[3 * i + 2], acc[2]);
        }
    };
    gather(0, g);
    if (DEEP) {
        for (int k = 4; k < w; k += 4) {
            double gn[12];
            gather(k, gn);
            consume(k - 4, g);
#pragma unroll
            for (int i = 0; i < 12; ++i) g[i] = gn[i];
        }
        consume(w - 4, g);
    } else {
        for (int k = 4; k < w; k += 4) {
            consume(k - 4, g);
            gather(k, g);
        }
        consume(w - 4, g);
    }
}

// MODE 0: Jacobi-preconditioned pipelined CG (the default).  MODE 1: the build with the Chebyshev-preconditioned loop.
// MODE 2: the build with the block-local symmetric Gauss-Seidel preconditioner.  Own instantiations: their extra live
// values must not cost the default kernel registers.
template <int MAXT, int MODE = 0>
__global__ __launch_bounds__(MAXT) void k_pcg_onchip(OcArgs a) {
    constexpr bool DEEP = MAXT <= 768;
    constexpr bool POLY = MODE == 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *red = (double *)smem;                   // [16][4][8] row partials of every wave
    double *bc = (double *)(smem + 4096 + 1024);           // [8] reduced scalars of the current phase
    double *sc = (double *)(smem + 4096 + 1088);           // [8] gamma_prev[3], alpha_prev[3]            (thread 0 only)
    double *gbl = (double *)(smem + 4096 + 1216);          // [4] b . M^-1 b per axis
    double *glast = (double *)(smem + 4096 + 1248);        // [4] last gamma per axis (reporting)
    int *ok_lds = (int *)(smem + 4096 + 1280);             // barrier verdict
    double *ctl = (double *)(smem + 4096 + 1296);          // [0] best ratio, [1] ratio of the last failed verification (thread 0 only);
                                                    // [2..4] alpha, [5..7] beta of this iteration (broadcast); [8..10] 1 / (b . M^-1 b)
    int *ictl = (int *)(smem + 4096 + 1392);               // [0] iterations since best, [1] failed verifications (thread 0 only); [2] action (broadcast)
    const int T = (int)blockDim.x, tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = T >> 6;
    if (a.prof && (int)blockIdx.x == a.prof_block && tid == 0) a.prof[63 * 8 + 0] = wall_clock64();
    LdsD *stg = (LdsD *)(smem + kOcScratch) + wv * (kOcStage / 8);   // this wave's staging area
    LdsD *lv_all = (LdsD *)(smem + kOcScratch + nw * kOcStage);
    LdsI *lc_all = (LdsI *)(lv_all + a.spb * a.wl * 64);
    const LdsD *lv = lv_all + wv * a.wl * 64 + lane;
    const LdsI *lc = lc_all + wv * a.wl * 64 + lane;

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
        LdsD *lvw = lv_all + wv * a.wl * 64 + lane;
        LdsI *lcw = lc_all + wv * a.wl * 64 + lane;
        for (int k = 0; k < wl_s; ++k) { lvw[64 * k] = vpg[64 * k]; lcw[64 * k] = cpg[64 * k]; }
    }
    if (a.prof && (int)blockIdx.x == a.prof_block && tid == 0) a.prof[63 * 8 + 1] = wall_clock64();
    const int as = a.n_slices * 64 * 8;     // bytes of one axis of a published vector
    const int ub = 3 * as;                  // bytes of one published-vector buffer
    __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc((void *)a.ubuf, 0, 2 * ub, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc((void *)a.part, 0, 2 * 8 * a.G * 8, 0x00020000);

    double rx[3], ru[3], rw[3], rp[3], rsv[3], rz[3], rd[3], rm[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const size_t i = 3 * (size_t)(live ? row : 0) + j;
        rx[j] = live ? a.x[i] : 0.0; rd[j] = live ? a.dinv[i] : 0.0; rm[j] = live ? a.m[i] : 0.0;
        ru[j] = rw[j] = rp[j] = rsv[j] = rz[j] = 0.0;
    }
    // Two sets of barrier counters, used by alternate solves: this launch counts on set (seq & 1) and clears the
    // other one for the next launch (nobody touches it meanwhile), so no memset is needed between solves.
    unsigned *const bar = a.bar + 32 * 16 * (a.seq & 1);
    if (blockIdx.x == 0 && tid < 9) a.bar[32 * 16 * ((a.seq & 1) ^ 1) + 16 * (tid < 8 ? tid : 17)] = 0u;
    unsigned ph = 0;     // publish phase: buffer parity = ph & 1, barrier epoch = ph
    const bool prof = a.prof && (int)blockIdx.x == a.prof_block && tid == 0;
    int prof_n = 0;
#define OC_STAMP(slot) do { if (prof && prof_n < 62) a.prof[prof_n * 8 + (slot)] = wall_clock64(); } while (0)

    // Publish this wave's 64 x 3 values per axis with 16-byte write-through stores (8-byte sc1 stores cost
    // 2.7x per byte), transposed through the wave's LDS staging area: lanes 0..31 store the x pairs and then
    // the z pairs, lanes 32..63 the y pairs.
    auto publish = [&](const double *v) {
        if (!live_slice) return;
        stg[lane] = v[0]; stg[64 + lane] = v[1]; stg[128 + lane] = v[2];
        const double xy_x = stg[2 * lane], xy_y = stg[2 * lane + 1];                     // x pairs | y pairs
        const int base = (int)(ph & 1u) * ub + s * 512 + (lane & 31) * 16;
        oc_store_sc1(rs_u, base + (lane >= 32 ? as : 0), xy_x, xy_y);
        if (lane < 32) {
            oc_store_sc1(rs_u, base + 2 * as, stg[128 + 2 * lane], stg[129 + 2 * lane]);
        }
    };
    // after the barrier of phase ph: out = A (published vector), bc[0..5] = the six global sums
    int nsum = 6;      // sums per record (7 in the polynomial mode: + the Jacobi-norm residual)
    int rec_par = -1;  // record buffer to reduce from; -1 = the phase parity (one phase per iteration)
    auto gather_and_reduce = [&](const double *self, double *out, bool do_gather, bool do_reduce) {
        const int vpar = (int)(ph & 1u);
        const int par = rec_par >= 0 ? rec_par : vpar;
        double rec[4] = {0.0, 0.0, 0.0, 0.0};
        if (do_reduce && wv < nsum) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int g = lane + 64 * j; if (g < a.G) rec[j] = oc_load_sc1_f64(rs_p, ((par * 8 + wv) * a.G + g) * 8); }
        }
        if (do_gather) {
            double acc[3];
            oc_row<true, DEEP>(rs_u, vpar * ub, as, nullptr, lv, lc, wl_s, w, cpg, vpg, acc);
#pragma unroll
            for (int j = 0; j < 3; ++j) out[j] = fma(rm[j], self[j], acc[j]);
        }
        if (!do_reduce) return;
        if (wv < nsum) {
            const double sm = wave_sum(((rec[0] + rec[1]) + rec[2]) + rec[3]);
            if (lane == 0) bc[wv] = sm;
        }
        for (int k = wv + nw; k < nsum; k += nw) {   // blocks with fewer waves than sums
            double sm = 0.0;
            for (int g = lane; g < a.G; g += 64) sm += oc_load_sc1_f64(rs_p, ((par * 8 + k) * a.G + g) * 8);
            sm = wave_sum(sm);
            if (lane == 0) bc[k] = sm;
        }
        __syncthreads();
    };

    // MODE 2 -- out = M^-1 v with M = (D + L) D^-1 (D + U) restricted to THIS BLOCK'S rows and columns, the rows ordered by
    // their colour (Ahat is 2-colourable: a row's off-diagonal neighbours all have the other colour).  Everything it reads
    // is already on this CU -- the matrix rows in the LDS slab, the vector in the staging area -- so it costs no exchange,
    // and it cuts the CG iterations by ~1.6x (285 -> 180 on the 1 M-tet cube, experiments/block_ssor_proto.py):
    //   forward:  colour 0: y = v / d;  colour 1: y = (v - sum_local a_ij y_j) / d;   backward:  colour 1: z = y;
    //   colour 0: z = y - (sum_local a_ij z_j) / d.      Each row does ONE local row pass; three block barriers.
    const bool bssor = MODE == 2 && a.row_color != nullptr && a.nbr != nullptr && kOcTrig * a.tol2 >= kOcPipeFloor;
    double rr[3] = {0.0, 0.0, 0.0}, rq[3] = {0.0, 0.0, 0.0};   // MODE 2: explicit residual r and q = M^-1 s
    const int blk0 = (int)blockIdx.x * a.spb * 64, nloc = a.spb * 64;
    const int mycol = (MODE == 2 && bssor && live) ? (int)a.row_color[row] : 0;
    unsigned lmask = 0u;     // which of this row's (<= 32) stored entries are off-diagonal non-zeros inside the block
    if (MODE == 2 && bssor && live) {
        for (int k = 0; k < w && k < 32; ++k) {
            const int c = (k < wl_s) ? (int)lc[64 * k] : cpg[64 * k];
            const double val = (k < wl_s) ? (double)lv[64 * k] : vpg[64 * k];
            const int cl = c - blk0;
            if (val != 0.0 && c != row && cl >= 0 && cl < nloc) lmask |= 1u << k;
        }
    }
    auto block_prec = [&](const double *v, double *out) {
        LdsD *sall = (LdsD *)(smem + kOcScratch);        // the staging areas of all waves: [wave][axis][lane]
        const int my = wv * 192 + lane;
        double y[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) { y[j] = v[j] * rd[j]; sall[my + 64 * j] = y[j]; }
        __syncthreads();
        double acc[3] = {0.0, 0.0, 0.0};
        auto local_sum = [&]() {
            for (unsigned m = lmask; m != 0u; m &= m - 1u) {
                const int k = __builtin_ctz(m);
                const int cl = ((k < wl_s) ? (int)lc[64 * k] : cpg[64 * k]) - blk0;
                const double val = (k < wl_s) ? (double)lv[64 * k] : vpg[64 * k];
                const int at = (cl >> 6) * 192 + (cl & 63);
                acc[0] = fma(val, sall[at], acc[0]); acc[1] = fma(val, sall[at + 64], acc[1]); acc[2] = fma(val, sall[at + 128], acc[2]);
            }
        };
        if (live && mycol == 1) {
            local_sum();
#pragma unroll
            for (int j = 0; j < 3; ++j) y[j] = (v[j] - acc[j]) * rd[j];
        }
        __syncthreads();
        if (live && mycol == 1) {
#pragma unroll
            for (int j = 0; j < 3; ++j) sall[my + 64 * j] = y[j];
        }
        __syncthreads();
        if (live && mycol == 0) {
            local_sum();
#pragma unroll
            for (int j = 0; j < 3; ++j) y[j] = fma(-rd[j], acc[j], y[j]);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) out[j] = y[j];
        __syncthreads();     // the staging areas are reused by publish().  (Two of these four block barriers are not needed for
                             // correctness -- a colour-1 row writes only its own slot and reads only colour-0 slots -- but the
                             // version without them measured 3 % slower: 914 vs 943 ADMM it/s.)
    };
    // All scalar decisions (stop tests, alpha/beta, mode switches) are taken by thread 0, whose bookkeeping lives
    // in LDS, and broadcast through LDS: uniform values would otherwise occupy registers in every lane, and the
    // action code read back with readfirstlane keeps the control flow provably uniform.
    int iters = 0, pipe_iters = 0;
    bool conv = false, aborted = false;
    auto action = [&]() -> int { __syncthreads(); return __builtin_amdgcn_readfirstlane(ictl[2]); };
    // u = M^-1 (b - A x) from the x held in registers; gathers x from `xin` (plain loads, kernel start) or from
    // the published copy; leaves gamma_true (and optionally b . M^-1 b) in q[0..5]
    auto true_residual = [&](bool from_global, bool with_bnorm, double *q) -> bool {
        double acc[3];
        if (from_global) oc_row<false, DEEP>(rs_u, 0, 0, a.x, lv, lc, wl_s, w, cpg, vpg, acc);
        else {
            ++ph; publish(rx);
            if (!oc_barrier(bar, ph, a.G, ok_lds, a.sig)) return false;
            oc_row<true, DEEP>(rs_u, (int)(ph & 1u) * ub, as, nullptr, lv, lc, wl_s, w, cpg, vpg, acc);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double bj = live ? a.b[3 * (size_t)row + j] : 0.0;
            const double ri = bj - fma(rm[j], rx[j], acc[j]);
            ru[j] = rd[j] * ri;
            q[j] = ru[j] * ri;                               // r . M^-1 r
            q[3 + j] = with_bnorm ? bj * rd[j] * bj : 0.0;   // b . M^-1 b
        }
        return true;
    };
    do {
        // ---- start: TRUE residual of the warm start, stop test, w = A u ----------------------------------
        {
            double q[6];
            if (!a.rc_on) true_residual(true, true, q);
            else {
                // ---- recycled warm start (see k_rc_* in kernels.hpp for the launch-path version) ----------------
                // r0 = b - A x0; A-orthogonal projection of the error on the stored exact pairs (E_j, A E_j):
                // per axis G c = g with G_ij = E_i . R_j, g_i = E_i . r0;  x += E c,  r0 -= R c.
                double acc[3], ri[3], bj[3];
                const int cnt = a.rc.cnt;
                double e[kRc][3], r[kRc][3];   // issued first: the HBM latency of the pairs hides behind the x gather
#pragma unroll
                for (int jj = 0; jj < kRc; ++jj)
#pragma unroll
                    for (int ax = 0; ax < 3; ++ax) {
                        const bool on = live && jj < cnt;
                        e[jj][ax] = on ? a.rc.E[jj][3 * (size_t)row + ax] : 0.0;
                        r[jj][ax] = on ? a.rc.R[jj][3 * (size_t)row + ax] : 0.0;
                    }
                if (prof) a.prof[62 * 8 + 0] = wall_clock64();
                oc_row<false, DEEP>(rs_u, 0, 0, a.x, lv, lc, wl_s, w, cpg, vpg, acc);
                if (prof) a.prof[62 * 8 + 1] = wall_clock64();
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    bj[j] = live ? a.b[3 * (size_t)row + j] : 0.0;
                    ri[j] = bj[j] - fma(rm[j], rx[j], acc[j]);
                    if (live) { a.rc_xs[3 * (size_t)row + j] = rx[j]; a.rc_r0[3 * (size_t)row + j] = ri[j]; }
                }
                if (cnt > 0) {
                    __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc((void *)a.rc_part, 0, 72 * a.G * 8, 0x00020000);
                    // block totals of the 3 x kRcQ products: three rounds of 24 quantities (three DPP row reductions each)
                    // through two alternating buffers in the still unused staging area -- one barrier per round
#pragma unroll
                    for (int g24 = 0; g24 < 3; ++g24) {
                        double *buf = (double *)(smem + kOcScratch) + (g24 & 1) * (4 * nw * 24);   // [4 nw rows][24]; two of them = nw * kOcStage bytes
#pragma unroll
                        for (int h = 0; h < 3; ++h) {
                            double q8[8];
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                const int f = 24 * g24 + 8 * h + i, ax = f / kRcQ, qi = f % kRcQ;     // compile-time after unrolling
                                q8[i] = (f >= 3 * kRcQ) ? 0.0
                                      : (qi < kRc * kRc) ? e[qi / kRc][ax < 3 ? ax : 0] * r[qi % kRc][ax < 3 ? ax : 0]
                                      : (qi < kRc * kRc + kRc) ? e[(qi - kRc * kRc) % kRc][ax < 3 ? ax : 0] * ri[ax < 3 ? ax : 0]
                                      : (qi == kRc * kRc + kRc) ? ri[ax < 3 ? ax : 0] * rd[ax < 3 ? ax : 0] * ri[ax < 3 ? ax : 0]
                                      : bj[ax < 3 ? ax : 0] * rd[ax < 3 ? ax : 0] * bj[ax < 3 ? ax : 0];
                            }
                            double b0, b1;
                            row_sum8(q8, b0, b1);
                            if ((lane & 15) < 4) {
                                double *dst = buf + (wv * 4 + (lane >> 4)) * 24 + 8 * h + 4 * (lane & 1) + (lane & 2);
                                dst[0] = b0; dst[1] = b1;
                            }
                        }
                        __syncthreads();
                        for (int t = tid; t < 192; t += T) {   // t = quantity (24) x part (8): eight partial sums per quantity, combined by three shuffles
                            const int qn = t >> 3, part = t & 7;
                            double sm = 0.0;
                            for (int rr = part; rr < 4 * nw; rr += 8) sm += buf[rr * 24 + qn];
                            sm += __shfl_xor(sm, 1, 64); sm += __shfl_xor(sm, 2, 64); sm += __shfl_xor(sm, 4, 64);
                            if (part == 0) oc_store_sc1(rs_r, ((24 * g24 + qn) * a.G + (int)blockIdx.x) * 8, sm);
                        }
                    }
                    __syncthreads();
                    if (prof) a.prof[62 * 8 + 2] = wall_clock64();
                    ++ph;
                    if (!oc_barrier(bar, ph, a.G, ok_lds, a.sig)) { aborted = true; break; }
                    if (prof) a.prof[62 * 8 + 3] = wall_clock64();
                    {   // every block adds the G partials of every sum in the same order; wave wv takes sums wv, wv + nw, ...
                        double v[6][4];
#pragma unroll
                        for (int t = 0; t < 6; ++t) {
                            const int k = wv + nw * t;
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int g = lane + 64 * j;
                                v[t][j] = (k < 3 * kRcQ && g < a.G) ? oc_load_sc1_f64(rs_r, (k * a.G + g) * 8) : 0.0;
                            }
                        }
#pragma unroll
                        for (int t = 0; t < 6; ++t) {
                            const int k = wv + nw * t;
                            if (k < 3 * kRcQ) {
                                const double sm = wave_sum(((v[t][0] + v[t][1]) + v[t][2]) + v[t][3]);
                                if (lane == 0) red[k] = sm;
                            }
                        }
                        for (int k = wv + 6 * nw; k < 3 * kRcQ; k += nw) {   // blocks with fewer than 11 waves
                            double sm = 0.0;
                            for (int g = lane; g < a.G; g += 64) sm += oc_load_sc1_f64(rs_r, (k * a.G + g) * 8);
                            sm = wave_sum(sm);
                            if (lane == 0) red[k] = sm;
                        }
                    }
                    __syncthreads();
                    if (prof) a.prof[62 * 8 + 4] = wall_clock64();
                    double *coefL = red + 3 * kRcQ + 2;   // [3][kRc]
                    if (tid < 3) {
                        bool skip = true;    // r0 already meets the tolerance on every axis: the pairs must not perturb x
                        for (int ax = 0; ax < 3; ++ax) skip = skip && (red[ax * kRcQ + 20] <= a.tol2 * red[ax * kRcQ + 21] + 1e-300);
                        double c[kRc];
                        rc_cholesky(red + kRcQ * tid, cnt, skip, c);
#pragma unroll
                        for (int i = 0; i < kRc; ++i) coefL[tid * kRc + i] = c[i];
                    }
                    __syncthreads();
#pragma unroll
                    for (int ax = 0; ax < 3; ++ax)
#pragma unroll
                        for (int jj = 0; jj < kRc; ++jj) {
                            const double c = coefL[ax * kRc + jj];
                            rx[ax] = fma(c, e[jj][ax], rx[ax]);
                            ri[ax] = fma(-c, r[jj][ax], ri[ax]);
                        }
                    __syncthreads();   // coefL / red are reused below
                    if (prof) a.prof[62 * 8 + 5] = wall_clock64();
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    ru[j] = rd[j] * ri[j];
                    q[j] = ru[j] * ri[j];
                    q[3 + j] = bj[j] * rd[j] * bj[j];
                }
            }
            if (MODE == 2 && bssor) {     // r explicitly, u = M^-1 r with the block preconditioner (q above stays the Jacobi norm)
#pragma unroll
                for (int j = 0; j < 3; ++j) rr[j] = live ? ru[j] * fast_rcp(rd[j]) : 0.0;
                block_prec(rr, ru);
            }
            ++ph; publish(ru);
            oc_publish_partials(q, red, nw, rs_p, (int)(ph & 1u), a.G);
        }
        if (!oc_barrier(bar, ph, a.G, ok_lds, a.sig)) { aborted = true; break; }
        gather_and_reduce(ru, rw, true, true);
        if (tid == 0) {
            // The three axes are independent systems with their own b . M^-1 b.  An axis whose right-hand side vanishes
            // (C^T d of a floor contact has no x / z part) or is > 15 orders below the largest one is measured against
            // the largest one: its own norm cannot scale a stop test.  b = 0 altogether: the solution is x = 0.
            const double gmax = fmax(bc[3], fmax(bc[4], bc[5]));
            bool c0 = true;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                gbl[j] = fmax(bc[3 + j], 1e-30 * gmax);
                ctl[8 + j] = gmax > 0.0 ? 1.0 / gbl[j] : 0.0;
                glast[j] = bc[j];
                c0 = c0 && (bc[j] <= a.tol2 * gbl[j] + 1e-300);
            }
            ctl[0] = 1e300; ctl[1] = 0.0; ictl[0] = 0; ictl[1] = 0; ictl[2] = !(gmax > 0.0) ? 3 : c0 ? 1 : 0;
        }
        {
            const int act0 = action();
            if (act0 == 3) {
#pragma unroll
                for (int j = 0; j < 3; ++j) { rx[j] = 0.0; ru[j] = 0.0; }
                conv = true; break;
            }
            if (act0 == 1) {
                if (MODE == 2 && bssor) {   // the epilogue expects u = D^-1 r
#pragma unroll
                    for (int j = 0; j < 3; ++j) ru[j] = rd[j] * rr[j];
                }
                conv = true; break;
            }
        }
        bool fresh = true;
        int restarts = 0;
        if (prof) a.prof[63 * 8 + 2] = wall_clock64();
        // Decision on the sums of this iteration, taken by lanes 0..2 of wave 0 (one axis each) and broadcast through LDS.
        // Returns the action: 0 update, 1 verify on the true residual, 2 non-finite / runaway sums; +4: leave the
        // pipelined form after the update.  Side effects: alpha / beta of this iteration in ctl[2..7].
        auto decide = [&](const bool pipelined) -> int {
            if (wv == 0) {
                const int j = lane < 3 ? lane : 0;
                const double g = bc[j], d = bc[3 + j], gbj = gbl[j];
                const double ratio = g * ctl[8 + j];                       // gamma / (b . M^-1 b)
                const unsigned long long m3 = 7ull;
                // finite, and not 1e8x (in norm) above the best residual seen: CG's residual norm is not monotone, but
                // growth of that size is a recurrence that has lost the iterate
                const bool finite = (__ballot(g < 1e290 && ratio < 1e290 && !(ratio > 1e16 * ctl[0])) & m3) == m3;
                const bool below_trig = (__ballot(g <= kOcTrig * a.tol2 * gbj + 1e-300) & m3) == m3;
                const bool below_tol = (__ballot(g <= a.tol2 * gbj + 1e-300) & m3) == m3;
                const bool below_floor = (__ballot(g <= kOcPipeFloor * gbj + 1e-300) & m3) == m3;
                double rmax = fmax(ratio, __shfl(ratio, 1, 64));
                rmax = fmax(rmax, __shfl(ratio, 2, 64));                   // valid in lane 0
                int act = 0;
                if (!finite) act = 2;
                else if (pipelined ? (below_trig && kOcTrig * a.tol2 >= kOcPipeFloor) : below_tol) act = 1;
                else {
                    int since = 0;
                    if (lane == 0) {
                        since = ictl[0] + 1;
                        if (rmax < ctl[0]) { ctl[0] = rmax; since = 0; }
                        ictl[0] = since;
                    }
                    since = __shfl(since, 0, 64);
                    const double best = __shfl(ctl[0], 0, 64);
                    if (pipelined && (below_floor || since >= (best <= 100.0 * kOcPipeFloor ? kOcStagnation : 4 * kOcStagnation))) act = 4;
                    if (lane < 3) {
                        double alpha = 0.0, beta;
                        if (fresh) { beta = 0.0; if (pipelined) alpha = (d > 0.0) ? g * fast_rcp(d) : 0.0; }
                        else {
                            const double gp = sc[j];
                            beta = (gp > 0.0) ? g * fast_rcp(gp) : 0.0;
                            if (pipelined) {
                                const double ap = sc[3 + j];
                                const double den = (ap != 0.0) ? d - beta * g * fast_rcp(ap) : d;
                                alpha = (den > 0.0) ? g * fast_rcp(den) : 0.0;
                            }
                        }
                        sc[j] = g; sc[3 + j] = alpha; glast[j] = g;
                        ctl[2 + j] = alpha; ctl[5 + j] = beta;
                    }
                }
                if (lane == 0) ictl[2] = act;
            }
            return action();
        };
        // TRUE residual at the current x (into u).  1: it meets the tolerance, or the FP64 floor is reached (a failed
        // verification that did not improve on the previous one by 4x); 0: it does not -- CG restarts from it; -1: aborted
        auto verify = [&]() -> int {
            double q[6];
            if (!true_residual(false, false, q)) return -1;
            ++ph;
            oc_publish_partials(q, red, nw, rs_p, (int)(ph & 1u), a.G);
            if (!oc_barrier(bar, ph, a.G, ok_lds, a.sig)) return -1;
            gather_and_reduce(nullptr, nullptr, false, true);
            if (tid == 0) {
                bool ok = true;
                double tr = 0.0;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    glast[j] = bc[j];
                    ok = ok && (bc[j] <= a.tol2 * gbl[j] + 1e-300);
                    tr = fmax(tr, bc[j] * ctl[8 + j]);
                }
                const bool done = ok || (ictl[1] >= 1 && !(tr <= 0.25 * ctl[1]));
                ctl[1] = tr; ictl[1] += 1;
                ctl[0] = tr; ictl[0] = 0;     // the runaway test measures against the TRUE residual from here on
                ictl[2] = done ? 1 : 0;
            }
            return action() == 1 ? 1 : 0;
        };
        bool entry_restart = false, go_classic = false;
        // ---- Chebyshev-preconditioned CG (Chronopoulos-Gear single-reduction form) ---------------------------------------
        // The iteration of this kernel costs one exchange of a vector (publish + neighbour hand-off + gather: no grid
        // barrier) plus one grid-wide reduction round, and the reduction round is the larger half.  u = P(D^-1 A) D^-1 r
        // with a degree-m Chebyshev polynomial costs m - 1 more exchanges per iteration and cuts the number of
        // iterations -- i.e. of reduction rounds -- by about m + 1 at ~1.2x the matrix-vector products.  r lives in the
        // registers of the pipelined form's z.  Stop test: the Jacobi-norm residual r . D^-1 r of every iteration rides in
        // a seventh sum (the three axes added up after scaling by their 1 / b.D^-1 b: at most 3x stricter than the
        // per-axis rule); it triggers the same true-residual verification as before.  Anything irregular (runaway or
        // non-finite sums, failed verification) hands over to the classic Jacobi form below with a restart.
        const bool poly = POLY && a.poly_m >= 2 && a.nbr != nullptr && kOcTrig * a.tol2 >= kOcPipeFloor;
        if (POLY && poly) {
            double *chb = (double *)(smem + 5632);     // [16] c1[8], c2[8] (indexed by a run-time k: LDS, not registers)
            if (tid < 16) chb[tid] = tid < 8 ? a.cheb_c1[tid & 7] : a.cheb_c2[tid & 7];
            __syncthreads();
            nsum = 7;
#pragma unroll
            for (int j = 0; j < 3; ++j) rz[j] = live ? ru[j] * fast_rcp(rd[j]) : 0.0;       // r (u = D^-1 r so far)
            double rho_best = 1e300;
            while (iters < a.max_iters) {
                double res[3], dd[3], t[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) { res[j] = rd[j] * rz[j]; dd[j] = res[j] * a.cheb_inv_theta; ru[j] = 0.0; }
                bool fail = false;
                for (int k = 0; k < a.poly_m; ++k) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) ru[j] += dd[j];
                    if (k == a.poly_m - 1) break;
                    ++ph; publish(dd);
                    if (!oc_announce_and_wait_neighbours(bar, a.flags, a.nbr, (unsigned)a.seq, ph, ok_lds, a.sig)) { fail = true; break; }
                    gather_and_reduce(dd, t, true, false);                                   // t = A d
                    const double c1 = chb[k], c2 = chb[8 + k];
#pragma unroll
                    for (int j = 0; j < 3; ++j) { res[j] = fma(-rd[j], t[j], res[j]); dd[j] = fma(c1, dd[j], c2 * res[j]); }
                }
                if (fail) { aborted = true; break; }
                ++ph; publish(ru);
                if (!oc_announce_and_wait_neighbours(bar, a.flags, a.nbr, (unsigned)a.seq, ph, ok_lds, a.sig)) { aborted = true; break; }
                gather_and_reduce(ru, rw, true, false);                                      // w = A u
                double q[6], rho = 0.0;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    q[j] = rz[j] * ru[j];                                                    // gamma = r . u
                    q[3 + j] = rw[j] * ru[j];                                                // delta = w . u
                    rho = fma(rz[j] * rd[j] * rz[j], ctl[8 + j], rho);                       // sum_j r.D^-1 r / b.D^-1 b
                }
                // records alternate by ITERATION (an iteration is m + 1 phases: with the phase parity a block that runs
                // ahead through the neighbour-synchronised exchanges could overwrite a record others still reduce)
                ++ph;
                rec_par = iters & 1;
                oc_publish_partials(q, red, nw, rs_p, rec_par, a.G, rho, 7);
                if (!oc_barrier(bar, ph, a.G, ok_lds, a.sig)) { rec_par = -1; aborted = true; break; }
                gather_and_reduce(nullptr, nullptr, false, true);
                rec_par = -1;
                if (wv == 0) {   // lanes 0..2 = one axis each
                    const int j = lane < 3 ? lane : 0;
                    const double g = bc[j], d = bc[3 + j], rr = bc[6];
                    const unsigned long long m3 = 7ull;
                    const bool finite = (__ballot(g < 1e290 && g >= 0.0 && d < 1e290) & m3) == m3 && rr < 1e290 && !(rr > 1e16 * rho_best);
                    int act = 0;
                    if (!finite) act = 2;
                    else if (rr <= kOcTrig * a.tol2) act = 1;
                    else if (lane < 3) {
                        double alpha, beta;
                        if (fresh) { beta = 0.0; alpha = (d > 0.0) ? g / d : 0.0; }
                        else {
                            const double gp = sc[j], ap = sc[3 + j];
                            beta = (gp > 0.0) ? g / gp : 0.0;
                            const double den = (ap != 0.0) ? d - beta * g / ap : d;
                            alpha = (den > 0.0) ? g / den : 0.0;
                        }
                        sc[j] = g; sc[3 + j] = alpha;
                        ctl[2 + j] = alpha; ctl[5 + j] = beta;
                    }
                    rho_best = fmin(rho_best, rr);
                    if (lane == 0) ictl[2] = act;
                }
                const int act = action();
                if (act == 2) { entry_restart = true; go_classic = true; break; }
                if (act == 1) {
                    const int v = verify();
                    if (v < 0) { aborted = true; break; }
                    if (v == 1) { conv = true; break; }
                    go_classic = true; fresh = true;
                    break;
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const double alpha = ctl[2 + j], beta = ctl[5 + j];
                    rp[j] = fma(beta, rp[j], ru[j]);
                    rsv[j] = fma(beta, rsv[j], rw[j]);
                    rx[j] = fma(alpha, rp[j], rx[j]);
                    rz[j] = fma(-alpha, rsv[j], rz[j]);
                }
                __syncthreads();   // ctl / bc are rewritten by the next iteration
                ++iters; ++pipe_iters; fresh = false;
            }
            nsum = 6;
            if (!conv && !go_classic && !aborted) {   // iteration cap: leave u = D^-1 r behind (epilogue, recycled pair)
#pragma unroll
                for (int j = 0; j < 3; ++j) ru[j] = rd[j] * rz[j];
            }
            if (aborted || conv || !go_classic) break;
        }
        // ---- pipelined CG with the block-local symmetric Gauss-Seidel preconditioner (MODE 2) --------------------------------
        // The general recurrences of Ghysels & Vanroose: r and q = M^-1 s are carried explicitly (with a diagonal M they
        // are u / d and d s).  Same single synchronisation per iteration; the stop test is the Jacobi-norm residual riding
        // in the seventh sum, as in the Chebyshev mode.
        if (MODE == 2 && bssor) {
            nsum = 7;
            double rho_best = 1e300;
            while (iters < a.max_iters) {
                double mm[3], rn[3] = {0.0, 0.0, 0.0}, q[6], rho = 0.0;
                block_prec(rw, mm);                                                          // m = M^-1 w
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    q[j] = rr[j] * ru[j];                                                    // gamma = r . u
                    q[3 + j] = rw[j] * ru[j];                                                // delta = w . u
                    rho = fma(rr[j] * rd[j] * rr[j], ctl[8 + j], rho);
                }
                ++ph; publish(mm);
                oc_publish_partials(q, red, nw, rs_p, (int)(ph & 1u), a.G, rho, 7);
                if (!oc_announce_and_wait_neighbours(bar, a.flags, a.nbr, (unsigned)a.seq, ph, ok_lds, a.sig)) { aborted = true; break; }
                gather_and_reduce(mm, rn, true, false);                                      // n = A m
                if (!oc_barrier_wait(bar, ph, a.G, ok_lds, a.sig)) { aborted = true; break; }
                gather_and_reduce(nullptr, nullptr, false, true);
                if (wv == 0) {
                    const int j = lane < 3 ? lane : 0;
                    const double g = bc[j], d = bc[3 + j], rs = bc[6];
                    const unsigned long long m3 = 7ull;
                    const bool finite = (__ballot(g < 1e290 && g >= 0.0 && d < 1e290) & m3) == m3 && rs < 1e290 && !(rs > 1e16 * rho_best);
                    int act = 0;
                    if (!finite) act = 2;
                    else if (rs <= kOcTrig * a.tol2) act = 1;
                    else if (lane < 3) {
                        double alpha, beta;
                        if (fresh) { beta = 0.0; alpha = (d > 0.0) ? g / d : 0.0; }
                        else {
                            const double gp = sc[j], ap = sc[3 + j];
                            beta = (gp > 0.0) ? g / gp : 0.0;
                            const double den = (ap != 0.0) ? d - beta * g / ap : d;
                            alpha = (den > 0.0) ? g / den : 0.0;
                        }
                        sc[j] = g; sc[3 + j] = alpha;
                        ctl[2 + j] = alpha; ctl[5 + j] = beta;
                    }
                    rho_best = fmin(rho_best, rs);
                    if (lane == 0) ictl[2] = act;
                }
                const int act = action();
                if (act == 2) { entry_restart = true; go_classic = true; break; }
                if (act == 1) {
                    const int v = verify();
                    if (v < 0) { aborted = true; break; }
                    if (v == 1) { conv = true; break; }
                    go_classic = true; fresh = true;
                    break;
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const double alpha = ctl[2 + j], beta = ctl[5 + j];
                    rz[j] = fma(beta, rz[j], rn[j]);
                    rq[j] = fma(beta, rq[j], mm[j]);
                    rsv[j] = fma(beta, rsv[j], rw[j]);
                    rp[j] = fma(beta, rp[j], ru[j]);
                    rx[j] = fma(alpha, rp[j], rx[j]);
                    rr[j] = fma(-alpha, rsv[j], rr[j]);
                    ru[j] = fma(-alpha, rq[j], ru[j]);
                    rw[j] = fma(-alpha, rz[j], rw[j]);
                }
                ++iters; ++pipe_iters; fresh = false;
            }
            nsum = 6;
            if (!conv && !go_classic && !aborted) {
#pragma unroll
                for (int j = 0; j < 3; ++j) ru[j] = rd[j] * rr[j];
            }
            if (aborted || conv || !go_classic) break;
        }
        // ---- pipelined CG (Ghysels-Vanroose): one synchronisation per iteration, while its recurrences are trusted ----
        while (!poly && !(MODE == 2 && bssor) && iters < a.max_iters) {
            OC_STAMP(0);
            double rn[3] = {0.0, 0.0, 0.0};
            {
                double mm[3], q[6];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    mm[j] = rd[j] * rw[j];
                    q[j] = (live ? ru[j] * ru[j] * fast_rcp(rd[j]) : 0.0);   // gamma = r . u
                    q[3 + j] = rw[j] * ru[j];                                 // delta = w . u
                }
                ++ph; publish(mm);
                oc_publish_partials(q, red, nw, rs_p, (int)(ph & 1u), a.G);
                OC_STAMP(1);
                if (a.prof) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); OC_STAMP(5); }
                if (a.nbr) {
                    // gather as soon as the blocks this one reads from have published; the grid barrier completes
                    // behind the gather and only gates the reduction of the partial sums
                    if (!oc_announce_and_wait_neighbours(bar, a.flags, a.nbr, (unsigned)a.seq, ph, ok_lds, a.sig)) { aborted = true; break; }
                    OC_STAMP(2);
                    gather_and_reduce(mm, rn, true, false);       // n = A M^-1 w
                    if (!oc_barrier_wait(bar, ph, a.G, ok_lds, a.sig)) { aborted = true; break; }
                    gather_and_reduce(nullptr, nullptr, false, true);   // the sums
                } else {
                    if (!oc_barrier(bar, ph, a.G, ok_lds, a.sig)) { aborted = true; break; }
                    OC_STAMP(2);
                    gather_and_reduce(mm, rn, true, true);        // n = A M^-1 w, and the sums
                }
            }
            OC_STAMP(3);
            const int act = decide(true);
            if (act == 2) { entry_restart = true; go_classic = true; break; }
            if (act == 1) {
                const int v = verify();
                if (v < 0) { aborted = true; break; }
                if (v == 1) { conv = true; break; }
                go_classic = true; fresh = true;     // the true residual replaces the recursive one: restart (beta = 0)
                break;
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double alpha = ctl[2 + j], beta = ctl[5 + j];
                rsv[j] = fma(beta, rsv[j], rw[j]);
                rp[j] = fma(beta, rp[j], ru[j]);
                rx[j] = fma(alpha, rp[j], rx[j]);
                ru[j] = fma(-alpha * rd[j], rsv[j], ru[j]);     // u = M^-1 (r - alpha s)
                rz[j] = fma(beta, rz[j], rn[j]);
                rw[j] = fma(-alpha, rz[j], rw[j]);
            }
            ++iters; ++pipe_iters; fresh = false;
            OC_STAMP(4);
            if (prof) ++prof_n;
            if (act & 4) { go_classic = true; break; }            // seamless: same x, u, p, gamma
        }
        if (aborted || conv || !go_classic) break;
        // ---- classic (Hestenes-Stiefel) CG, three synchronisations per iteration: gamma = r . u, then p, s = A p and
        // delta = p . s.  delta is computed directly (no delta - beta gamma / alpha cancellation as in the one-reduction
        // forms), which keeps the end game stable on ill-conditioned systems (free nearly incompressible bodies: cond ~ 1e7)
        while (iters < a.max_iters) {
            if (entry_restart) {
                // Non-finite or runaway sums.  The entry x is still in global memory (x is written once, at the end):
                // go back to it and restart from its true residual.  A second failure gives up: the solve is reported
                // as unconverged and hands back the entry x, never a non-finite vector.
                entry_restart = false;
#pragma unroll
                for (int j = 0; j < 3; ++j) rx[j] = live ? a.x[3 * (size_t)row + j] : 0.0;
                double q[6];
                if (!true_residual(false, false, q)) { aborted = true; break; }
                if (tid == 0) { ctl[0] = 1e300; ictl[0] = 0; }
                if (++restarts > 1) break;
                fresh = true;
            }
            double q[6];
#pragma unroll
            for (int j = 0; j < 3; ++j) { q[j] = (live ? ru[j] * ru[j] * fast_rcp(rd[j]) : 0.0); q[3 + j] = 0.0; }
            ++ph;
            oc_publish_partials(q, red, nw, rs_p, (int)(ph & 1u), a.G);
            if (!oc_barrier(bar, ph, a.G, ok_lds, a.sig)) { aborted = true; break; }
            gather_and_reduce(nullptr, nullptr, false, true);
            const int act = decide(false);
            if (act == 2) { entry_restart = true; continue; }
            if (act == 1) {
                const int v = verify();
                if (v < 0) { aborted = true; break; }
                if (v == 1) { conv = true; break; }
                fresh = true;
                continue;
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) rp[j] = fma(ctl[5 + j], rp[j], ru[j]);    // p = u + beta p
            ++ph; publish(rp);
            if (!oc_barrier(bar, ph, a.G, ok_lds, a.sig)) { aborted = true; break; }
            gather_and_reduce(rp, rsv, true, false);                               // s = A p
#pragma unroll
            for (int j = 0; j < 3; ++j) { q[j] = 0.0; q[3 + j] = rp[j] * rsv[j]; }
            ++ph;
            oc_publish_partials(q, red, nw, rs_p, (int)(ph & 1u), a.G);
            if (!oc_barrier(bar, ph, a.G, ok_lds, a.sig)) { aborted = true; break; }
            gather_and_reduce(nullptr, nullptr, false, true);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double delta = bc[3 + j];
                const double alpha = (delta > 0.0) ? glast[j] / delta : 0.0;
                rx[j] = fma(alpha, rp[j], rx[j]);
                ru[j] = fma(-alpha * rd[j], rsv[j], ru[j]);
            }
            __syncthreads();   // bc / glast are rewritten by the next iteration's reduction and decision
            ++iters; fresh = false;
        }
    } while (false);
    if (prof) a.prof[63 * 8 + 3] = wall_clock64();
#undef OC_STAMP
    if (live) {
#pragma unroll
        for (int j = 0; j < 3; ++j) { a.x[3 * (size_t)row + j] = rx[j]; a.u_out[3 * (size_t)row + j] = ru[j]; }
        if (a.rc_on) {   // this solve's pair: e = x - x_entry, A e = r_entry - r_final (exact: u carries the true residual)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const size_t i = 3 * (size_t)row + j;
                a.rc_Eslot[i] = rx[j] - a.rc_xs[i];
                a.rc_Rslot[i] = a.rc_r0[i] - ru[j] * fast_rcp(rd[j]);
            }
        }
    }
    if (blockIdx.x == 0 && tid == 0) {
        CgScal o;
#pragma unroll
        for (int j = 0; j < 3; ++j) { o.gamma[j] = glast[j]; o.alpha[j] = 0.0; o.gamma_b[j] = gbl[j]; }
        o.alpha[0] = (double)ictl[1]; o.alpha[1] = (double)ph; o.alpha[2] = ctl[1] / a.tol2;   // diagnosis: verifications done, synchronisation phases, true gamma ratio / tol^2 at the last verification
        o.converged = (conv && !aborted) ? 1 : 0; o.iters = iters; o.seq = a.seq; o.pad_ = pipe_iters;
        a.scal[0] = o;
        atomicAdd(a.counters, iters);
        if (o.converged) {
            atomicAdd(a.counters + 4, 1);
            atomicMax(a.counters + 3, iters);
            a.counters[8 + (a.seq & 63)] = iters;
        }
        if (prof) a.prof[63 * 8 + 4] = wall_clock64();
    }
}

} // namespace admm_k
