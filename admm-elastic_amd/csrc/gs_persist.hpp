// gs_persist.hpp -- NodalMultiColorGS::solve (src/NodalMultiColorGS.hpp:60-146, 180-262) as ONE persistent launch.
//
// The colour kernels of kernels.hpp (k_gs_color*) spend a solve's time in launches: 30 sweeps x 2-3 launches x ~5.2 us at
// configs[1] / configs[4], with a few microseconds of work in all of them together.  Here every CU keeps one compact block of the
// mesh (host plan: oc_plan.cpp build_gs_plan) for the whole solve -- its rows' matrix entries, right-hand side, its part of x and
// the values of the neighbouring blocks' rows it references, all in LDS -- and a colour phase costs ONE neighbour hand-off:
//
//   * every block publishes the boundary rows of the colour it has just swept as TAGGED GRANULES, 16 bytes = {lo32(value), stamp,
//     hi32(value), stamp}, written with one write-through (sc1) store: the two 8-byte halves are single-copy atomic, so a reader that
//     sees the phase's stamp in both halves has the value -- no drain, no flag, no second round trip (experiments/sync_latency.hip:
//     1.6-2.4 us per phase against 2.2-3.1 us for data + drain + flag + poll and ~5.2 us for a kernel launch inside a hipGraph);
//   * a consumer polls exactly the granules of its halo entries of that colour (sc1 loads), with bounded spins: a hand-off that
//     cannot complete aborts the solve (sig[2]) instead of hanging the GPU;
//   * two outbox slots per node alternate between sweeps (a block can run at most one sweep ahead of a neighbour: it needs the
//     neighbour's values of every sweep), four slots for the per-sweep residual partials;
//   * no grid barrier anywhere on the normal path.
//
// Semantics kept exactly (the sweeps are bit-identical to k_gs_color's: same row order of the sums, same update expressions):
// omega over-relaxation (:210), pins first (:111-117), passive obstacles by the constrained segment update without over-relaxation
// (:218-262), at most max_iters sweeps, residual test |b - A x|^2 / |b|^2 < tol^2 after every sweep (:136-140).  The test is
// evaluated one sweep late from the blocks' partial sums (every block reduces the same numbers in the same order: identical
// verdicts), the sweeps in between run speculatively; if a sweep k did meet the tolerance the blocks meet at a (rare) all-to-all and
// REPLAY the solve from the untouched input for exactly k + 1 sweeps without tests -- deterministic, no roll-back copies in LDS.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels.hpp"
#include "oc_sync.hpp"

namespace admm_k {

constexpr int kGspMaxCK = 12, kGspHdrK = 64, kGspT = 256;
constexpr unsigned kGspSpin = 3000000u;
typedef __attribute__((address_space(3))) unsigned short LdsU16;
typedef __attribute__((address_space(3))) int LdsI32;
typedef __attribute__((address_space(3))) unsigned char LdsU8;

struct GspArgs {
    int G, C;
    const int *hdr, *orig, *out_idx, *halo_box, *halo_orig;   // host_setup.hpp: GsPlan
    const double *diag, *vals; const unsigned short *cols;
    const double *m, *b; double *x;
    const int *pin_flag; const double *pin_xyz;               // per node (pin_flag nullptr: no pins)
    double omega, tol2;
    int max_sweeps, check;
    unsigned seq;                                             // solve number of the context (stamps)
    v4u *box;                                                 // [outbox nodes][2 sweep parities][3 axes] granules
    v4u *part;                                                // [G][4 sweep slots][2] granules: |r|^2, |b|^2 of the block
    v4u *meet;                                                // [G] granules: the blocks' rendez-vous before a replay
    unsigned *abort_word;                                     // raised by the first block that gives up
    int *done, *sweeps, *total;                               // counters[1], [2], [0] of the context (as the colour kernels)
    int *sig;                                                 // host-visible: sig[2] = 1 when the solve was aborted
};

__device__ __forceinline__ v4u gsp_pack(double v, unsigned s) {
    union { double d; unsigned u[2]; } t; t.d = v;
    v4u g; g.x = t.u[0]; g.y = s; g.z = t.u[1]; g.w = s;
    return g;
}
__device__ __forceinline__ double gsp_val(v4u g) { union { double d; unsigned u[2]; } t; t.u[0] = g.x; t.u[1] = g.z; return t.d; }
__device__ __forceinline__ bool gsp_ok(v4u g, unsigned s) { return g.y == s && g.w == s; }
__device__ __forceinline__ v4u gsp_load(__amdgpu_buffer_rsrc_t rs, int byte_off) {
    asm volatile("" ::: "memory");      // a polling load: never hoisted out of its loop, never merged with the previous round's
    return __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16 /* sc1 */);
}
__device__ __forceinline__ void gsp_store(__amdgpu_buffer_rsrc_t rs, int byte_off, v4u g) {
    __builtin_amdgcn_raw_buffer_store_b128(g, rs, byte_off, 0, 16 /* sc1: write-through */);
}

__global__ __launch_bounds__(kGspT) void k_gs_persist(GspArgs a, Obstacles ob) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = (int)blockIdx.x, t = (int)threadIdx.x;
    // ---- LDS: scratch | x [L][3] | b [n_own][3] | a_ii [n_own][3] | values | columns | outbox node per row | halo source | pin flags
    //      (host_setup.hpp: gsp_lds_bytes computes the same offsets)
    LdsD *scr = (LdsD *)smem;                       // [0..15] reduction partials, [16..31] verdict sums
    LdsI32 *ih = (LdsI32 *)(smem + 256);            // the block's header (64 ints)
    LdsI32 *ctl = (LdsI32 *)(smem + 512);           // [0] abort seen, [1] verdict
    if (t < kGspHdrK) ih[t] = a.hdr[b * kGspHdrK + t];
    if (t == 0) { ctl[0] = 0; ctl[1] = 0; }
    __syncthreads();
    const int n_own = ih[0], n_halo = ih[1], row_base = ih[2], halo_base = ih[3], ent_base = ih[4], ob_base = ih[5], ent_count = ih[7];
    const int L = n_own + n_halo, C = a.C;
    LdsD *xl = (LdsD *)(smem + 1024);
    LdsD *bl = xl + 3 * L;
    LdsD *al = bl + 3 * n_own;
    LdsD *vl = al + 3 * n_own;
    LdsU16 *cl = (LdsU16 *)(vl + ent_count);
    LdsI32 *ol = (LdsI32 *)((__attribute__((address_space(3))) char *)cl + (2 * ent_count + 7) / 8 * 8);
    LdsI32 *hl = (LdsI32 *)((__attribute__((address_space(3))) char *)ol + (4 * n_own + 7) / 8 * 8);
    LdsU8 *pl = (LdsU8 *)((__attribute__((address_space(3))) char *)hl + (4 * n_halo + 7) / 8 * 8);
    const __amdgpu_buffer_rsrc_t rbox = soa_rsrc(a.box), rpart = soa_rsrc(a.part), rmeet = soa_rsrc(a.meet);

    // ---- fill: matrix, right-hand side, diagonal, lists (once), x (again before a replay) ----
    for (int i = t; i < ent_count; i += kGspT) { vl[i] = a.vals[(size_t)ent_base + i]; cl[i] = a.cols[(size_t)ent_base + i]; }
    for (int i = t; i < n_own; i += kGspT) {
        const int v = a.orig[row_base + i];
        const double d = a.diag[row_base + i];
        ol[i] = a.out_idx[row_base + i];
        pl[i] = (a.pin_flag && a.pin_flag[v]) ? 1 : 0;
#pragma unroll
        for (int q = 0; q < 3; ++q) { bl[3 * i + q] = a.b[3 * (size_t)v + q]; al[3 * i + q] = d + a.m[3 * (size_t)v + q]; }
    }
    for (int i = t; i < n_halo; i += kGspT) hl[i] = a.halo_box[halo_base + i];
    auto load_x = [&]() {
        for (int i = t; i < n_own; i += kGspT) {
            const int v = a.orig[row_base + i];
#pragma unroll
            for (int q = 0; q < 3; ++q) xl[3 * i + q] = a.x[3 * (size_t)v + q];
        }
        for (int i = t; i < n_halo; i += kGspT) {
            const int v = a.halo_orig[halo_base + i];
#pragma unroll
            for (int q = 0; q < 3; ++q) xl[3 * (n_own + i) + q] = a.x[3 * (size_t)v + q];
        }
    };
    load_x();
    __syncthreads();

    auto give_up = [&]() {      // (one thread) tell everybody, and the host
        __hip_atomic_store(a.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.sig + 2, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    };
    // after a __syncthreads: did any thread of the block fail a poll?
    auto block_failed = [&]() -> bool { return ctl[0] != 0; };

    // the halo entries of colour `cp`, published with stamp `want` in slot `par`
    auto fetch_halo = [&](int cp, int par, unsigned want) {
        const int h0 = ih[21 + cp], h1 = ih[21 + cp + 1];
        for (int hh = h0 + t; hh < h1; hh += kGspT) {
            const int off = ((hl[hh] * 2 + par) * 3) * 16;
            v4u g0, g1, g2;
            unsigned spins = 0;
            while (true) {
                g0 = gsp_load(rbox, off); g1 = gsp_load(rbox, off + 16); g2 = gsp_load(rbox, off + 32);
                if (gsp_ok(g0, want) && gsp_ok(g1, want) && gsp_ok(g2, want)) break;
                if (++spins > kGspSpin || ((spins & 127u) == 0u && __hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                    if (!ctl[0]) { ctl[0] = 1; give_up(); }
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            xl[3 * (n_own + hh)] = gsp_val(g0); xl[3 * (n_own + hh) + 1] = gsp_val(g1); xl[3 * (n_own + hh) + 2] = gsp_val(g2);
        }
    };
    // sum_k Ahat(row, k) x_k of row i of colour c (entries in CSR order: the sums of k_gs_color)
    auto row_sum = [&](int c, int i, double *acc) {
        const int n_c = ih[8 + c + 1] - ih[8 + c], W = ih[34 + c];
        const LdsD *vv = vl + ih[46 + c] + i;
        const LdsU16 *cc = cl + ih[46 + c] + i;
        acc[0] = acc[1] = acc[2] = 0.0;
        for (int k = 0; k < W; ++k) {
            const int col = cc[k * n_c];
            const double av = vv[k * n_c];
            acc[0] = fma(av, xl[3 * col], acc[0]); acc[1] = fma(av, xl[3 * col + 1], acc[1]); acc[2] = fma(av, xl[3 * col + 2], acc[2]);
        }
    };
    auto sweep_colour = [&](int c, int par, unsigned stamp) {
        const int r0 = ih[8 + c], n_c = ih[8 + c + 1] - r0;
        for (int i = t; i < n_c; i += kGspT) {
            double LUx[3];
            row_sum(c, i, LUx);
            const int li = r0 + i;
            double nx[3];
            if (pl[li]) { // :111-117
                const int v = a.orig[row_base + li];
#pragma unroll
                for (int q = 0; q < 3; ++q) nx[q] = a.pin_xyz[3 * (size_t)v + q];
            } else {
                const double bi[3] = {bl[3 * li], bl[3 * li + 1], bl[3 * li + 2]};
                const double aii[3] = {al[3 * li], al[3 * li + 1], al[3 * li + 2]};
                const double cx[3] = {xl[3 * li], xl[3 * li + 1], xl[3 * li + 2]};
                gs_relax(ob, a.omega, bi, LUx, aii, cx, nx);
            }
            xl[3 * li] = nx[0]; xl[3 * li + 1] = nx[1]; xl[3 * li + 2] = nx[2];
            const int o = ol[li];
            if (o >= 0) {
                const int off = (((ob_base + o) * 2 + par) * 3) * 16;
                gsp_store(rbox, off, gsp_pack(nx[0], stamp)); gsp_store(rbox, off + 16, gsp_pack(nx[1], stamp)); gsp_store(rbox, off + 32, gsp_pack(nx[2], stamp));
            }
        }
    };
    // |b - A x|^2 and |b|^2 over the block's rows at the current (complete) state; published for sweep `k` with stamp `sp`
    auto publish_partial = [&](int k, unsigned sp) {
        double q2[2] = {0.0, 0.0};
        for (int c = 0; c < C; ++c) {
            const int r0 = ih[8 + c], n_c = ih[8 + c + 1] - r0;
            for (int i = t; i < n_c; i += kGspT) {
                double acc[3];
                row_sum(c, i, acc);
                const int li = r0 + i;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const double bi = bl[3 * li + q];
                    const double r = bi - fma(al[3 * li + q], xl[3 * li + q], acc[q]);
                    q2[0] = fma(r, r, q2[0]); q2[1] = fma(bi, bi, q2[1]);
                }
            }
        }
        const int lane = t & 63, wv = t >> 6;
        const double s0 = wave_sum(q2[0]), s1 = wave_sum(q2[1]);
        if (lane == 0) { scr[wv] = s0; scr[4 + wv] = s1; }
        __syncthreads();
        if (t == 0) {
            const double r2 = scr[0] + scr[1] + scr[2] + scr[3], b2 = scr[4] + scr[5] + scr[6] + scr[7];
            const int off = ((b * 4 + (k & 3)) * 2) * 16;
            gsp_store(rpart, off, gsp_pack(r2, sp)); gsp_store(rpart, off + 16, gsp_pack(b2, sp));
        }
    };
    // the residual test of sweep k from all blocks' partials (fixed order: every block gets the same verdict).  All threads call.
    auto verdict = [&](int k, unsigned sp) -> bool {
        double r2 = 0.0, b2 = 0.0;
        if (t < 64) {            // wave 0: lane l sums blocks l, l + 64, ... in order; then the wave sum
            for (int j = t; j < a.G; j += 64) {
                const int off = ((j * 4 + (k & 3)) * 2) * 16;
                v4u g0, g1;
                unsigned spins = 0;
                while (true) {
                    g0 = gsp_load(rpart, off); g1 = gsp_load(rpart, off + 16);
                    if (gsp_ok(g0, sp) && gsp_ok(g1, sp)) break;
                    if (++spins > kGspSpin || ((spins & 127u) == 0u && __hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                        if (!ctl[0]) { ctl[0] = 1; give_up(); }
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                r2 += gsp_val(g0); b2 += gsp_val(g1);
            }
            r2 = wave_sum(r2); b2 = wave_sum(b2);
            if (t == 0) ctl[1] = (r2 / b2 < a.tol2) ? 1 : 0;
        }
        __syncthreads();
        const bool conv = ctl[1] != 0;
        __syncthreads();
        return conv;
    };

    // n sweeps; with_tests: the residual test of every sweep, evaluated one sweep late.  Returns the first sweep that met the
    // tolerance among 0 .. n-3 (-1: none -- the last two are settled by the caller), -2 after an abort.
    auto run = [&](int n, bool with_tests, unsigned stamp0, unsigned part0) -> int {
        for (int sweep = 0; sweep < n; ++sweep) {
            for (int c = 0; c < C; ++c) {
                const int p = sweep * C + c;
                if (p > 0) fetch_halo(c > 0 ? c - 1 : C - 1, (c > 0 ? sweep : sweep - 1) & 1, stamp0 + (unsigned)(p - 1));
                __syncthreads();
                if (block_failed()) return -2;
                if (with_tests && c == 0 && sweep > 0) {
                    publish_partial(sweep - 1, part0 + (unsigned)(sweep - 1));
                    if (sweep >= 2) {
                        const bool conv = verdict(sweep - 2, part0 + (unsigned)(sweep - 2));
                        if (block_failed()) return -2;
                        if (conv) return sweep - 2;
                    }
                }
                sweep_colour(c, sweep & 1, stamp0 + (unsigned)p);
            }
        }
        // the values of the last colour of the last sweep: the block's state is complete again
        if (n > 0 && a.G > 1) fetch_halo(C - 1, (n - 1) & 1, stamp0 + (unsigned)(n * C - 1));
        __syncthreads();
        if (block_failed()) return -2;
        return -1;
    };

    const unsigned stamp0 = a.seq * 4096u + 1u;         // phases of the first run: stamp0 + p; of a replay: stamp0 + 2048 + p
    const int n = a.max_sweeps;
    int failed_tests = n, conv_flag = 0;                // what the counters get: sweeps whose test failed, done
    int first = run(n, a.check != 0, stamp0, stamp0);
    if (first == -2) return;
    if (a.check && first == -1 && n >= 1) {             // the last two sweeps' tests
        publish_partial(n - 1, stamp0 + (unsigned)(n - 1));
        if (n >= 2 && verdict(n - 2, stamp0 + (unsigned)(n - 2))) first = n - 2;
        if (block_failed()) return;
        if (first == -1 && verdict(n - 1, stamp0 + (unsigned)(n - 1))) { failed_tests = n - 1; conv_flag = 1; }    // the state is already the answer
        if (block_failed()) return;
    }
    if (first >= 0 && first < n - 1) {
        // sweep `first` met the tolerance: the reference stopped there.  Everybody meets (nobody may still be reading this run's
        // granules, x in memory is still the input), then the solve is replayed for first + 1 sweeps without tests.
        failed_tests = first; conv_flag = 1;
        const unsigned ms = stamp0 + 4000u;
        if (t == 0) gsp_store(rmeet, b * 16, gsp_pack(0.0, ms));
        if (t < 64) {
            for (int j = t; j < a.G; j += 64) {
                unsigned spins = 0;
                while (true) {
                    const v4u g = gsp_load(rmeet, j * 16);
                    if (gsp_ok(g, ms)) break;
                    if (++spins > kGspSpin || ((spins & 127u) == 0u && __hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                        if (!ctl[0]) { ctl[0] = 1; give_up(); }
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
        }
        __syncthreads();
        if (block_failed()) return;
        load_x();
        __syncthreads();
        if (run(first + 1, false, stamp0 + 2048u, 0u) == -2) return;
    } else if (first == n - 1 && first >= 0) { failed_tests = n - 1; conv_flag = 1; }
    // ---- write the block's rows back, counters ----
    for (int i = t; i < n_own; i += kGspT) {
        const int v = a.orig[row_base + i];
#pragma unroll
        for (int q = 0; q < 3; ++q) a.x[3 * (size_t)v + q] = xl[3 * i + q];
    }
    if (b == 0 && t == 0) {
        if (conv_flag) *a.done = 1;
        atomicAdd(a.sweeps, failed_tests); atomicAdd(a.total, failed_tests);
    }
}

} // namespace admm_k
