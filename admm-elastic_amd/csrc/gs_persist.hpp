// gs_persist.hpp -- NodalMultiColorGS::solve (src/NodalMultiColorGS.hpp:60-146, 180-262) as ONE persistent launch.
//
// The colour kernels of kernels.hpp (k_gs_color*) spend a solve's time in launches: 30 sweeps x 2-3 launches x ~5.2 us at
// configs[1] / configs[4], with a few microseconds of work in all of them together.  Here every CU keeps one compact block of the
// mesh (host plan: oc_plan.cpp build_gs_plan) for the whole solve -- its rows' matrix entries, right-hand side, its part of x and
// the values of the neighbouring blocks' rows it references, all in LDS -- and a colour phase costs ONE neighbour hand-off:
//
//   * every block publishes the boundary rows of the colour it has just swept as TAGGED GRANULES: 16-byte write-through (sc1) stores whose
//     two 8-byte halves (each single-copy atomic) carry the phase's stamp next to their payload, so a reader that sees the stamp in every half
//     has the value -- no drain, no flag, no second round trip (experiments/sync_latency.hip: 1.6-2.4 us per phase against 2.2-3.1 us for
//     data + drain + flag + poll and ~5.2 us for a kernel launch inside a hipGraph);
//   * ROUND 6 -- what a hand-off costs is how the granules fill SECTORS (profiles/r06_gs_granules_ab.txt).  Rounds 4-5 wrote one granule per
//     value ({lo32, stamp, hi32, stamp}: 48 bytes per row) at [node][parity][axis]: every store instruction of a wave left one 16-byte half
//     sector per row, 96 bytes apart, and the phase's hand-off took 1.4 us.  Now (a) a row's x | y | z travel as ONE 192-bit string in TWO
//     granules = four 8-byte units of 48 payload bits + the low 16 bits of the stamp (gsp_pack3; enough to tell the phases of a solve and
//     sixteen consecutive solves apart, and every slot is rewritten in every solve), and (b) the outbox is laid out [parity][granule][node]:
//     the rows a wave publishes have consecutive outbox nodes (the plan numbers a block's boundary rows colour by colour), so one store
//     instruction fills whole sectors and whole lines, and the consumer's lanes (halo lists sorted by source block and row) read them the
//     same way.  Hand-off 1.4 -> 0.70 us per phase: cube100k_gs 4 330 -> 5 260, cloth200k_gs_floor 3 090 -> 3 300 ADMM it/s, same bits.
//     (Measured and dropped in the same session: two or three polls of a granule in flight at a time -- the wait is the neighbours' stores
//     draining, not the sampling instant: -1 % / -8 %; one granule per lane behind a block barrier -- 4 460 on the cube.)
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

constexpr int kGspMaxCK = 12, kGspHdrK = 64, kGspT = 256, kGspRowsTarget = 192;
constexpr unsigned kGspSpin = 3000000u;
typedef __attribute__((address_space(3))) unsigned short LdsU16;
typedef __attribute__((address_space(3))) int LdsI32;
typedef __attribute__((address_space(3))) unsigned char LdsU8;
typedef __attribute__((address_space(3))) Obstacles LdsObst;
static_assert(sizeof(Obstacles) % 4 == 0 && sizeof(Obstacles) <= 384, "the LDS copy of the obstacles sits at bytes 640..1023 of the scratch area");

struct GspArgs {
    int G, C;
    const int *hdr, *orig, *out_idx, *halo_box, *halo_orig;   // host_setup.hpp: GsPlan
    const double *diag, *vals; const unsigned short *cols;
    const double *m, *b; double *x;
    const int *pin_flag; const double *pin_xyz;               // per node (pin_flag nullptr: no pins); flag 2 = slide pin, normal in pin_nrm
    const double *pin_nrm;
    double omega, tol2;
    int max_sweeps, check;
    unsigned seq;                                             // solve number of the context (stamps)
    v4u *box;                                                 // [outbox nodes][2 sweep parities][3 axes] granules (gsp_box_off)
    int n_box;                                                // outbox nodes of all blocks
    v4u *part;                                                // [G][4 sweep slots][2] granules: |r|^2, |b|^2 of the block
    v4u *meet;                                                // [G] granules: the blocks' rendez-vous before a replay
    unsigned *abort_word;                                     // raised by the first block that gives up
    int *done, *sweeps, *total;                               // counters[1], [2], [0] of the context (as the colour kernels)
    int *sig;                                                 // host-visible: sig[2] = 1 when the solve was aborted
    const Obstacles *ob;                                      // passive obstacles (device copy: 80 SGPRs as a by-value argument)
    unsigned long long *prof; int prof_block;                 // diagnosis (ADMM_HIP_GSP_PROF=1): wall-clock ticks per part of a phase
    unsigned long long *proj;                                 // rows projected onto a passive obstacle since create (admm_hip_contact_totals)
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

#ifndef ADMM_GSP_PACK2
#define ADMM_GSP_PACK2 1
#endif
// ADMM_GSP_PACK2: the three values of a boundary row in TWO granules instead of three -- four 8-byte units (each single-copy atomic) of 48 payload
// bits + the low 16 bits of the stamp: x | y | z as one 192-bit string cut into four.  16 bits tell the phases of one solve apart and sixteen
// consecutive solves (stamp = solve x 4096 + 1 + phase, phase < 4048); every slot is rewritten in every solve.
__device__ __forceinline__ void gsp_pack3(const double *v, unsigned stamp, v4u &ga, v4u &gb) {
    union { double d; unsigned u[2]; } x, y, z; x.d = v[0]; y.d = v[1]; z.d = v[2];
    const unsigned s = stamp << 16;
    ga.x = x.u[0];                              ga.y = (x.u[1] & 0xffffu) | s;
    ga.z = (x.u[1] >> 16) | (y.u[0] << 16);     ga.w = (y.u[0] >> 16) | s;
    gb.x = y.u[1];                              gb.y = (z.u[0] & 0xffffu) | s;
    gb.z = (z.u[0] >> 16) | (z.u[1] << 16);     gb.w = (z.u[1] >> 16) | s;
}
__device__ __forceinline__ bool gsp_ok3(v4u ga, v4u gb, unsigned stamp) {
    const unsigned s = stamp & 0xffffu;
    return (ga.y >> 16) == s && (ga.w >> 16) == s && (gb.y >> 16) == s && (gb.w >> 16) == s;
}
__device__ __forceinline__ void gsp_unpack3(v4u ga, v4u gb, double *v) {
    union { double d; unsigned u[2]; } x, y, z;
    x.u[0] = ga.x;                              x.u[1] = (ga.y & 0xffffu) | (ga.z << 16);
    y.u[0] = (ga.z >> 16) | (ga.w << 16);       y.u[1] = gb.x;
    z.u[0] = (gb.y & 0xffffu) | (gb.z << 16);   z.u[1] = (gb.z >> 16) | (gb.w << 16);
    v[0] = x.d; v[1] = y.d; v[2] = z.d;
}
constexpr int kGspGran = ADMM_GSP_PACK2 ? 2 : 3;      // granules per (outbox node, sweep parity)
#ifndef ADMM_GSP_SOA
#define ADMM_GSP_SOA 1
#endif
// Byte offset of granule k of outbox node `node`, sweep parity `par`.  ADMM_GSP_SOA: [parity][granule][node] -- the rows a wave publishes have
// consecutive outbox nodes (the plan numbers a block's boundary rows colour by colour), so ONE store instruction covers whole 32-byte sectors and
// whole lines, and the consumer's lanes (halo lists sorted by source block and row) read them the same way; 0: [node][parity][granule], a wave's
// store instruction touches one 16-byte half sector per row, 96 bytes apart (half-written sectors drain slowly: pcg_onchip2.hpp, publish()).
__device__ __forceinline__ int gsp_box_off(int node, int par, int k, int n_nodes) {
#if ADMM_GSP_SOA
    return ((par * kGspGran + k) * n_nodes + node) * 16;
#else
    (void)n_nodes;
    return ((node * 2 + par) * kGspGran + k) * 16;
#endif
}

#ifndef ADMM_GSP_PIPE
#define ADMM_GSP_PIPE 0
#endif
#ifndef ADMM_GSP_PIPE_D
#define ADMM_GSP_PIPE_D 3
#endif
#ifndef ADMM_GSP_PIPE_D0
#define ADMM_GSP_PIPE_D0 0
#endif
constexpr int kGspPipeD = ADMM_GSP_PIPE_D, kGspPipeD0 = ADMM_GSP_PIPE_D0, kGspPipeChain = 10;      // (s_sleep periods of 64 clocks; checks in the chain)
struct GspSet2 { v4u g[2]; };
__device__ __forceinline__ void gsp_poll2(__amdgpu_buffer_rsrc_t rs, int off, int off1, GspSet2 &p) { p.g[0] = gsp_load(rs, off); p.g[1] = gsp_load(rs, off1); }
template <int K> __device__ __forceinline__ bool gsp_chain2(__amdgpu_buffer_rsrc_t rs, int off, int off1, unsigned want, const GspSet2 &pa, const GspSet2 &pb, GspSet2 &out) {
    if (gsp_ok3(pa.g[0], pa.g[1], want)) { out = pa; return true; }
    if constexpr (K == 0) { out = pa; return false; }
    else {
        GspSet2 pn;
        gsp_poll2(rs, off, off1, pn);
        return gsp_chain2<K - 1>(rs, off, off1, want, pb, pn, out);
    }
}

struct GspObstOne {      // Obstacles with exactly one entry (k_gs_persist: in SGPRs; kernels.hpp: ObstSingle)
    static constexpr bool kSingle = true;
    static constexpr int n = 1;
    int kind[1]; double par[1][4]; const double *gmeta, *gdata;
};

__global__ __launch_bounds__(kGspT) void k_gs_persist(GspArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = (int)blockIdx.x, t = (int)threadIdx.x;
#ifdef ADMM_GSP_PROF_FINE
    const unsigned long long tk0 = wall_clock64();
#endif
    // ---- LDS: scratch | x [L][3] | x of the previous sweep [L][3] | b [n_own][3] | a_ii [n_own][3] | values | columns | outbox node per
    //      row | halo source | pin flags | 1 / a_ii [n_own][3]   (host_setup.hpp: gsp_lds_bytes computes the same offsets)
    LdsD *scr = (LdsD *)smem;                       // [0..3] wave sums of the residual partial, [8] |b|^2 of the block
    LdsI32 *ih = (LdsI32 *)(smem + 256);            // the block's header (64 ints)
    LdsI32 *ctl = (LdsI32 *)(smem + 512);           // [0] abort seen, [1] a sweep met the tolerance, [2] which one (verdict(): written and read across a barrier);
                                                    // [4 + 2 q], [5 + 2 q]: the same from verdict_wave of a phase of parity q -- written in phase p, read in phase
                                                    // p + 1 behind that phase's barrier, so a wave that is late after a barrier never sees a verdict of its own phase
    if (t < kGspHdrK) ih[t] = a.hdr[b * kGspHdrK + t];
    // the passive obstacles, once per launch: read through the argument pointer, the count, kind and parameters of every obstacle were
    // three DEPENDENT global round trips inside every row update of every phase (the polling loads' memory clobbers forbid hoisting them)
    LdsObst *obl = (LdsObst *)(smem + 640);
    if (t < (int)(sizeof(Obstacles) / 4)) ((LdsI32 *)obl)[t] = ((const int *)a.ob)[t];
    if (t == 0) {
        ctl[1] = 0; ctl[2] = 0; ctl[4] = 0; ctl[5] = 0; ctl[6] = 0; ctl[7] = 0; ctl[10] = 0;
        // a solve of this context has been given up and the host has not recovered yet (steps are issued asynchronously): nothing may
        // run on that state -- every later launch leaves at once, the host replays them after its next synchronisation
        ctl[0] = __hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1 : 0;
        if (ctl[0]) __hip_atomic_store(a.sig + 2, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    if (ctl[0]) return;
    const bool has_ob = __builtin_amdgcn_readfirstlane(obl->n) > 0;      // (uniform, in an SGPR: no LDS round trip per row for scenes without obstacles)
    // ONE obstacle (the usual scene: a floor): kind and parameters in SGPRs -- read from the LDS copy they are three dependent round trips in front of
    // every row's update.  gs_relax is a template on the obstacle container: the same arithmetic, the same bits.
    GspObstOne ob1;
    const bool one_ob = __builtin_amdgcn_readfirstlane(obl->n) == 1;
    {
        auto rfl_d = [](double v) -> double {
            return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
        };
        ob1.kind[0] = one_ob ? __builtin_amdgcn_readfirstlane(obl->kind[0]) : 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) ob1.par[0][q] = one_ob ? rfl_d(obl->par[0][q]) : 0.0;
        ob1.gmeta = a.ob->gmeta; ob1.gdata = a.ob->gdata;      // (kernel-argument memory: scalar loads, once)
    }
    const int n_own = ih[0], n_halo = ih[1], row_base = ih[2], halo_base = ih[3], ent_base = ih[4], ob_base = ih[5], ent_count = ih[7];
    const int L = n_own + n_halo, C = a.C;
    LdsD *xl = (LdsD *)(smem + 1024);
    LdsD *xo = xl + 3 * L;
    LdsD *bl = xo + 3 * L;
    LdsD *al = bl + 3 * n_own;
    LdsD *vl = al + 3 * n_own;
    LdsU16 *cl = (LdsU16 *)(vl + ent_count);
    LdsI32 *ol = (LdsI32 *)((__attribute__((address_space(3))) char *)cl + (2 * ent_count + 7) / 8 * 8);
    LdsI32 *hl = (LdsI32 *)((__attribute__((address_space(3))) char *)ol + (4 * n_own + 7) / 8 * 8);
    LdsU8 *pl = (LdsU8 *)((__attribute__((address_space(3))) char *)hl + (4 * n_halo + 7) / 8 * 8);
    LdsD *il = (LdsD *)((__attribute__((address_space(3))) char *)pl + (n_own + 7) / 8 * 8);      // 1 / a_ii [n_own][3]
    const __amdgpu_buffer_rsrc_t rbox = soa_rsrc(a.box), rpart = soa_rsrc(a.part), rmeet = soa_rsrc(a.meet);
    const int lane = t & 63, wv = t >> 6;

    // ---- fill: matrix, right-hand side, diagonal, lists (once), x (again before a replay) ----
    // Round 6: BATCHED.  The lists are two levels deep (row -> vertex -> b, m, x, pin) and the fill of 52-77 blocks is pure latency: written as
    // plain loops it was one dependent round trip after the other (6-8 us per solve, ADMM_GSP_PROF_FINE).  Now every thread first asks for the
    // first level of its rows and halo entries t, t + 256, streams the matrix through registers eight entries at a time while that is under way,
    // then asks for the whole second level at once.  Rows / entries beyond 512 per block take the plain loops (same order of the |b|^2 sum).
    {
        constexpr int NS = 2;
        int vo[NS], oi[NS], vh[NS], hb[NS]; double dg[NS]; bool ok_o[NS], ok_h[NS];
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const int i = t + kGspT * u;
            ok_o[u] = i < n_own; ok_h[u] = i < n_halo;
            vo[u] = ok_o[u] ? a.orig[row_base + i] : 0; dg[u] = ok_o[u] ? a.diag[row_base + i] : 0.0; oi[u] = ok_o[u] ? a.out_idx[row_base + i] : -1;
            vh[u] = ok_h[u] ? a.halo_orig[halo_base + i] : 0; hb[u] = ok_h[u] ? a.halo_box[halo_base + i] : 0;
        }
        for (int i0 = t; i0 < ent_count; i0 += 8 * kGspT) {
            double v8[8]; unsigned short c8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + kGspT * u; const bool ok = i < ent_count; v8[u] = ok ? a.vals[(size_t)ent_base + i] : 0.0; c8[u] = ok ? a.cols[(size_t)ent_base + i] : (unsigned short)0; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + kGspT * u; if (i < ent_count) { vl[i] = v8[u]; cl[i] = c8[u]; } }
        }
        double bq[NS][3], mq[NS][3], xq[NS][3], xh[NS][3]; int pf[NS];
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            pf[u] = (ok_o[u] && a.pin_flag) ? a.pin_flag[vo[u]] : 0;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                bq[u][q] = ok_o[u] ? a.b[3 * (size_t)vo[u] + q] : 0.0; mq[u][q] = ok_o[u] ? a.m[3 * (size_t)vo[u] + q] : 1.0; xq[u][q] = ok_o[u] ? a.x[3 * (size_t)vo[u] + q] : 0.0;
                xh[u][q] = ok_h[u] ? a.x[3 * (size_t)vh[u] + q] : 0.0;
            }
        }
        double bb = 0.0;
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const int i = t + kGspT * u;
            if (ok_o[u]) {
                ol[i] = oi[u]; pl[i] = (unsigned char)pf[u];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const double bi = bq[u][q], aq = dg[u] + mq[u][q];
                    bl[3 * i + q] = bi; al[3 * i + q] = aq; il[3 * i + q] = 1.0 / aq; xl[3 * i + q] = xq[u][q];
                    bb = fma(bi, bi, bb);
                }
            }
            if (ok_h[u]) {
                hl[i] = hb[u];
#pragma unroll
                for (int q = 0; q < 3; ++q) xl[3 * (n_own + i) + q] = xh[u][q];
            }
        }
        for (int i = t + NS * kGspT; i < n_own; i += kGspT) {
            const int v = a.orig[row_base + i];
            const double d = a.diag[row_base + i];
            ol[i] = a.out_idx[row_base + i];
            pl[i] = a.pin_flag ? (unsigned char)a.pin_flag[v] : 0;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const double bi = a.b[3 * (size_t)v + q];
                const double aq = d + a.m[3 * (size_t)v + q];
                bl[3 * i + q] = bi; al[3 * i + q] = aq; il[3 * i + q] = 1.0 / aq; xl[3 * i + q] = a.x[3 * (size_t)v + q];
                bb = fma(bi, bi, bb);
            }
        }
        for (int i = t + NS * kGspT; i < n_halo; i += kGspT) {
            hl[i] = a.halo_box[halo_base + i];
            const int v = a.halo_orig[halo_base + i];
#pragma unroll
            for (int q = 0; q < 3; ++q) xl[3 * (n_own + i) + q] = a.x[3 * (size_t)v + q];
        }
        bb = wave_sum(bb);
        if (lane == 0) scr[4 + wv] = bb;
    }
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
    __syncthreads();      // (x came with the fill; load_x: before a replay)
    if (t == 0) scr[8] = scr[4] + scr[5] + scr[6] + scr[7];     // |b|^2 of the block's rows (constant during the solve)

    auto give_up = [&]() {      // (one thread) tell everybody, and the host
        __hip_atomic_store(a.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.sig + 2, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    };
    auto poll_failed = [&](unsigned &spins) -> bool {   // one more unsuccessful poll: give up?
        if (++spins > kGspSpin || ((spins & 127u) == 0u && __hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
            if (!ctl[0]) { ctl[0] = 1; give_up(); }
            return true;
        }
        __builtin_amdgcn_s_sleep(1);
        return false;
    };
    // after a __syncthreads: did any thread of the block fail a poll?
    auto block_failed = [&]() -> bool { return ctl[0] != 0; };

    // the halo entries of colour `cp`, published with stamp `want` in slot `par`; keep_old: park the values they replace in xo
    auto fetch_halo = [&](int cp, int par, unsigned want, bool keep_old) {
        const int h0 = ih[21 + cp], h1 = ih[21 + cp + 1];
        for (int hh = h0 + t; hh < h1; hh += kGspT) {
            const int off = gsp_box_off(hl[hh], par, 0, a.n_box), off1 = gsp_box_off(hl[hh], par, 1, a.n_box), off2 = gsp_box_off(hl[hh], par, kGspGran - 1, a.n_box);
            v4u g0, g1, g2;
            unsigned spins = 0;
#if ADMM_GSP_PACK2 && ADMM_GSP_PIPE >= 2
            // (experiment, OFF: ADMM_GSP_PIPE polls in flight, the oldest checked and re-issued -- a CHAIN of checks, not a loop: register sets carried
            // around a back edge are copied there, and the copy waits for every poll in flight.  profiles/r06_gs_granules_ab.txt)
            bool have = false;
            {
                GspSet2 p0, p1, out;
                if (kGspPipeD0 > 0) __builtin_amdgcn_s_sleep(kGspPipeD0);
                gsp_poll2(rbox, off, off1, p0);
                __builtin_amdgcn_s_sleep(kGspPipeD);
                gsp_poll2(rbox, off, off1, p1);
                have = gsp_chain2<kGspPipeChain>(rbox, off, off1, want, p0, p1, out);
                g0 = out.g[0]; g1 = out.g[1];
            }
            while (!have) {
                g0 = gsp_load(rbox, off); g1 = gsp_load(rbox, off1);
                if (gsp_ok3(g0, g1, want)) break;
                if (poll_failed(spins)) break;
            }
#elif ADMM_GSP_PACK2
            while (true) {
                g0 = gsp_load(rbox, off); g1 = gsp_load(rbox, off1);
                if (gsp_ok3(g0, g1, want)) break;
                if (poll_failed(spins)) break;
            }
#else
            while (true) {
                g0 = gsp_load(rbox, off); g1 = gsp_load(rbox, off1); g2 = gsp_load(rbox, off2);
                if (gsp_ok(g0, want) && gsp_ok(g1, want) && gsp_ok(g2, want)) break;
                if (poll_failed(spins)) break;
            }
#endif
            const int j = 3 * (n_own + hh);
            if (keep_old) { xo[j] = xl[j]; xo[j + 1] = xl[j + 1]; xo[j + 2] = xl[j + 2]; }
#if ADMM_GSP_PACK2
            { double v3[3]; gsp_unpack3(g0, g1, v3); xl[j] = v3[0]; xl[j + 1] = v3[1]; xl[j + 2] = v3[2]; }
#else
            xl[j] = gsp_val(g0); xl[j + 1] = gsp_val(g1); xl[j + 2] = gsp_val(g2);
#endif
        }
    };
    // cur = sum_k Ahat(row, k) x_k of row i of colour c (entries in CSR order: the sums of k_gs_color).  BOTH: also old = the same sum
    // with the previous sweep's values (xo) of the neighbours whose colour is below the row's (bit 15 of the column, set by the plan)
    // (W is a multiple of four -- the plan pads rows with 0 x own value -- and the entries are taken EIGHT, then four, at a time: all
    // columns and values of a group in one LDS round trip, all gathers in a second one, then the fma chain in CSR order.  One wave per SIMD
    // hides nothing: the round trips of a row are its time.)
    auto row_sum = [&](int c, int i, double *cur, double *old, bool both) {
        const int n_c = ih[8 + c + 1] - ih[8 + c], W = ih[34 + c];
        const LdsD *vv = vl + ih[46 + c] + i;
        const LdsU16 *cc = cl + ih[46 + c] + i;
        cur[0] = cur[1] = cur[2] = 0.0;
        auto group = [&](int k, auto n_tag) {
            constexpr int N = decltype(n_tag)::value;
            int col[N]; double av[N], g[3 * N];
#pragma unroll
            for (int u = 0; u < N; ++u) { col[u] = 3 * (cc[(k + u) * n_c] & 0x7fff); av[u] = vv[(k + u) * n_c]; }
#pragma unroll
            for (int u = 0; u < N; ++u) { g[3 * u] = xl[col[u]]; g[3 * u + 1] = xl[col[u] + 1]; g[3 * u + 2] = xl[col[u] + 2]; }
#pragma unroll
            for (int u = 0; u < N; ++u) { cur[0] = fma(av[u], g[3 * u], cur[0]); cur[1] = fma(av[u], g[3 * u + 1], cur[1]); cur[2] = fma(av[u], g[3 * u + 2], cur[2]); }
        };
        auto group_both = [&](int k, auto n_tag) {
            constexpr int N = decltype(n_tag)::value;
            int raw[N]; double av[N], g[3 * N], h[3 * N];
#pragma unroll
            for (int u = 0; u < N; ++u) { raw[u] = cc[(k + u) * n_c]; av[u] = vv[(k + u) * n_c]; }
#pragma unroll
            for (int u = 0; u < N; ++u) {
                const int col = 3 * (raw[u] & 0x7fff);
                g[3 * u] = xl[col]; g[3 * u + 1] = xl[col + 1]; g[3 * u + 2] = xl[col + 2];
                h[3 * u] = xo[col]; h[3 * u + 1] = xo[col + 1]; h[3 * u + 2] = xo[col + 2];
            }
#pragma unroll
            for (int u = 0; u < N; ++u) {
                const bool lo = (raw[u] & 0x8000) != 0;
                cur[0] = fma(av[u], g[3 * u], cur[0]); cur[1] = fma(av[u], g[3 * u + 1], cur[1]); cur[2] = fma(av[u], g[3 * u + 2], cur[2]);
                old[0] = fma(av[u], lo ? h[3 * u] : g[3 * u], old[0]); old[1] = fma(av[u], lo ? h[3 * u + 1] : g[3 * u + 1], old[1]);
                old[2] = fma(av[u], lo ? h[3 * u + 2] : g[3 * u + 2], old[2]);
            }
        };
        int k = 0;
        if (!both) {
            for (; k + 8 <= W; k += 8) group(k, std::integral_constant<int, 8>());
            if (k < W) group(k, std::integral_constant<int, 4>());
            return;
        }
        old[0] = old[1] = old[2] = 0.0;
        for (; k + 8 <= W; k += 8) group_both(k, std::integral_constant<int, 8>());
        if (k < W) group_both(k, std::integral_constant<int, 4>());
    };
    // One colour of one sweep.  role (residual test riding on the sweep, as in k_gs_color2 / k_gs_colorN): 0 none; 1 PRE -- the
    // row's residual of the PREVIOUS sweep before it moves (first colour: nothing has moved yet; middle colours: the neighbours that
    // have, by their parked values); 2 POST -- the residual of THIS sweep right after the update (last colour: every neighbour is
    // final).  Returns the thread's sum of squared residuals.
#ifdef ADMM_GSP_PROF_FINE      // (experiments: where the time of "rows + publish" goes -- thread 0 of the profiled block, 100 MHz wall clock)
    const bool proff = a.prof != nullptr && b == a.prof_block && t == 0;
    unsigned long long pf[6] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull}, tf = 0ull;
#define GSP_FLAP0() do { if (proff) tf = wall_clock64(); } while (0)
#define GSP_FLAP(k) do { if (proff) { const unsigned long long now_ = wall_clock64(); pf[k] += now_ - tf; tf = now_; } } while (0)
#else
#define GSP_FLAP0()
#define GSP_FLAP(k)
#endif
    auto sweep_colour = [&](int c, int par, unsigned stamp, int role, bool keep_old, bool first_sweep) -> double {
        const int r0 = ih[8 + c], n_c = ih[8 + c + 1] - r0;
        const bool both = role == 1 && c > 0;
        double rs = 0.0;
        for (int i = t; i < n_c; i += kGspT) {
            double LUx[3], LUo[3];
            GSP_FLAP0();
            row_sum(c, i, LUx, LUo, both);
            GSP_FLAP(0);
            const int li = r0 + i;
            const int pflag = pl[li], o = ol[li];
            const double bi[3] = {bl[3 * li], bl[3 * li + 1], bl[3 * li + 2]};
            const double aii[3] = {al[3 * li], al[3 * li + 1], al[3 * li + 2]};
            const double iaii[3] = {il[3 * li], il[3 * li + 1], il[3 * li + 2]};
            const double cx[3] = {xl[3 * li], xl[3 * li + 1], xl[3 * li + 2]};
            double nx[3];
            if (role == 1) {
#pragma unroll
                for (int q = 0; q < 3; ++q) { const double r = bi[q] - fma(aii[q], cx[q], both ? LUo[q] : LUx[q]); rs = fma(r, r, rs); }
            }
            GSP_FLAP(1);
            if (pflag == 2) {      // slide pin (normal-only constraint): the unrelaxed value projected onto the pin's plane, every sweep
                const int v = a.orig[row_base + li];
                gs_pin_value(2, a.pin_xyz + 3 * (size_t)v, a.pin_nrm + 3 * (size_t)v, bi, LUx, iaii, nx);
            } else if (pflag) { // :111-117
#ifdef ADMM_GSP_OB_GLOBAL
                if (true) {
#else
                if (first_sweep) {
#endif
                    const int v = a.orig[row_base + li];
#pragma unroll
                    for (int q = 0; q < 3; ++q) nx[q] = a.pin_xyz[3 * (size_t)v + q];
                } else {    // the row has held its pin's position since the first sweep of this run: no trip to memory
#pragma unroll
                    for (int q = 0; q < 3; ++q) nx[q] = cx[q];
                }
            }
#ifdef ADMM_GSP_OB_GLOBAL      // (same-box A/B only: the obstacles through the argument pointer, as before round 4's second session)
            else if (gs_relax(*a.ob, a.omega, bi, LUx, iaii, cx, nx)) __hip_atomic_fetch_add(&ctl[10], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
            else if (!has_ob) {      // no passive obstacle in the scene (uniform): :210 alone
#pragma unroll
                for (int q = 0; q < 3; ++q) nx[q] = fma(a.omega, (bi[q] - LUx[q]) * iaii[q], (1.0 - a.omega) * cx[q]);
            }
            else if (one_ob) { if (gs_relax(ob1, a.omega, bi, LUx, iaii, cx, nx)) __hip_atomic_fetch_add(&ctl[10], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
            else if (gs_relax(*obl, a.omega, bi, LUx, iaii, cx, nx)) __hip_atomic_fetch_add(&ctl[10], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (LDS add: rows projected, counted below)
#endif
            GSP_FLAP(2);
            if (o >= 0) {      // the neighbours wait for these: out first, the block's own copies after
                const int off = gsp_box_off(ob_base + o, par, 0, a.n_box), off1 = gsp_box_off(ob_base + o, par, 1, a.n_box);
#if ADMM_GSP_PACK2
                { v4u ga, gb; gsp_pack3(nx, stamp, ga, gb); gsp_store(rbox, off, ga); gsp_store(rbox, off1, gb); }
#else
                gsp_store(rbox, off, gsp_pack(nx[0], stamp)); gsp_store(rbox, off1, gsp_pack(nx[1], stamp)); gsp_store(rbox, gsp_box_off(ob_base + o, par, 2, a.n_box), gsp_pack(nx[2], stamp));
#endif
            }
            if (keep_old) { xo[3 * li] = cx[0]; xo[3 * li + 1] = cx[1]; xo[3 * li + 2] = cx[2]; }
            xl[3 * li] = nx[0]; xl[3 * li + 1] = nx[1]; xl[3 * li + 2] = nx[2];
            if (role == 2) {
#pragma unroll
                for (int q = 0; q < 3; ++q) { const double r = bi[q] - fma(aii[q], nx[q], LUx[q]); rs = fma(r, r, rs); }
            }
            GSP_FLAP(3);
        }
        return rs;
    };
    // residuals of the rows of colours c0 .. c1-1 at the current (complete) state
    auto direct_residual = [&](int c0, int c1) -> double {
        double rs = 0.0;
        for (int c = c0; c < c1; ++c) {
            const int r0 = ih[8 + c], n_c = ih[8 + c + 1] - r0;
            for (int i = t; i < n_c; i += kGspT) {
                double acc[3];
                row_sum(c, i, acc, acc, false);
                const int li = r0 + i;
#pragma unroll
                for (int q = 0; q < 3; ++q) { const double r = bl[3 * li + q] - fma(al[3 * li + q], xl[3 * li + q], acc[q]); rs = fma(r, r, rs); }
            }
        }
        return rs;
    };
    auto park = [&](double rs) { const double s = wave_sum(rs); if (lane == 0) scr[wv] = s; };     // ... a __syncthreads later:
    auto publish_parked = [&](int k, unsigned sp) {       // thread 0: the block's partial of sweep k
        if (t == 0) {
            const double r2 = scr[0] + scr[1] + scr[2] + scr[3];
            const int off = ((b * 4 + (k & 3)) * 2) * 16;
            gsp_store(rpart, off, gsp_pack(r2, sp)); gsp_store(rpart, off + 16, gsp_pack(scr[8], sp));
        }
    };
    // The residual test of sweep k from all blocks' partials; every block adds the same numbers in the same order: identical
    // verdicts.  Wave 0: lane l takes blocks l, l + 64, l + 128, l + 192 (, ...).  `pre`: granules loaded earlier (their latency hidden behind
    // the halo hand-off); whatever has not arrived yet is polled.  All threads call; one block barrier.
    auto verdict = [&](int k, unsigned sp, v4u (*pre)[2], bool have_pre) -> bool {
        if (t < 64) {
            double r2 = 0.0, b2 = 0.0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = t + 64 * u;
                if (j < a.G) {
                    const int off = ((j * 4 + (k & 3)) * 2) * 16;
                    v4u g0, g1;
                    if (have_pre) { g0 = pre[u][0]; g1 = pre[u][1]; }
                    unsigned spins = 0;
                    while (!have_pre || !(gsp_ok(g0, sp) && gsp_ok(g1, sp))) {
                        g0 = gsp_load(rpart, off); g1 = gsp_load(rpart, off + 16);
                        if (gsp_ok(g0, sp) && gsp_ok(g1, sp)) break;
                        if (poll_failed(spins)) break;
                    }
                    r2 += gsp_val(g0); b2 += gsp_val(g1);
                }
            }
            for (int j = t + 256; j < a.G; j += 64) {      // (more than 256 blocks: two per CU)
                const int off = ((j * 4 + (k & 3)) * 2) * 16;
                v4u g0, g1;
                unsigned spins = 0;
                while (true) {
                    g0 = gsp_load(rpart, off); g1 = gsp_load(rpart, off + 16);
                    if (gsp_ok(g0, sp) && gsp_ok(g1, sp)) break;
                    if (poll_failed(spins)) break;
                }
                r2 += gsp_val(g0); b2 += gsp_val(g1);
            }
            r2 = wave_sum(r2); b2 = wave_sum(b2);
            if (t == 0) ctl[1] = (r2 / b2 < a.tol2) ? 1 : 0;
        }
        __syncthreads();
        return ctl[1] != 0;
    };

    // ... the same by ONE wave from granules loaded earlier, without a barrier: the verdict is left in the slot of the phase's parity
    // (ctl[4 + 2 q] met, ctl[5 + 2 q] sweep) for everybody to read after the NEXT phase's block barrier.  (One slot read in the phase it
    // is written in was a race: a wave held up behind the barrier could see the verdict one phase before its siblings and leave alone.)
    auto verdict_wave = [&](int k, unsigned sp, v4u (*pre)[2], int q) {
        double r2 = 0.0, b2 = 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = (t & 63) + 64 * u;
            if (j < a.G) {
                const int off = ((j * 4 + (k & 3)) * 2) * 16;
                v4u g0 = pre[u][0], g1 = pre[u][1];
                unsigned spins = 0;
                while (!(gsp_ok(g0, sp) && gsp_ok(g1, sp))) {
                    g0 = gsp_load(rpart, off); g1 = gsp_load(rpart, off + 16);
                    if (gsp_ok(g0, sp) && gsp_ok(g1, sp)) break;
                    if (poll_failed(spins)) break;
                }
                r2 += gsp_val(g0); b2 += gsp_val(g1);
            }
        }
        for (int j = (t & 63) + 256; j < a.G; j += 64) {      // (more than 256 blocks: not loaded ahead)
            const int off = ((j * 4 + (k & 3)) * 2) * 16;
            v4u g0, g1;
            unsigned spins = 0;
            while (true) {
                g0 = gsp_load(rpart, off); g1 = gsp_load(rpart, off + 16);
                if (gsp_ok(g0, sp) && gsp_ok(g1, sp)) break;
                if (poll_failed(spins)) break;
            }
            r2 += gsp_val(g0); b2 += gsp_val(g1);
        }
        r2 = wave_sum(r2); b2 = wave_sum(b2);
        if ((t & 63) == 0 && r2 / b2 < a.tol2) { ctl[5 + 2 * q] = k; ctl[4 + 2 * q] = 1; }
    };
    // n sweeps.  tests: the residual test of every sweep (:136-140), riding on the sweeps and evaluated two sweeps late.  Returns the
    // first sweep that met the tolerance (0 .. n-1), -1 if none did, -2 after an abort.
    auto run = [&](int n, bool tests, unsigned stamp0) -> int {
        const bool keep_old = tests && C >= 3;
        const int c_pub = C >= 2 ? C - 2 : 0;        // the phase of sweep s + 1 after which the partial of sweep s is complete
        double racc = 0.0;
        int parked = -1;
        const bool prof = a.prof != nullptr && b == a.prof_block && t == 0;
        unsigned long long pt[4] = {0ull, 0ull, 0ull, 0ull}, tk = prof ? wall_clock64() : 0ull;
        const unsigned long long cyc0 = prof ? (unsigned long long)clock64() : 0ull, wc0 = tk;     // shader clock against the 100 MHz wall clock
        auto lap = [&](int k) { if (prof) { const unsigned long long now = wall_clock64(); pt[k] += now - tk; tk = now; } };
        for (int sweep = 0; sweep < n; ++sweep) {
            for (int c = 0; c < C; ++c) {
                const int p = sweep * C + c;
                // The verdict on sweep - 2 is formed by the LAST wave of the block (the one with the fewest rows) while the others
                // sweep, and read by everybody one phase later: no barrier of its own, nothing on the sweeping waves' path.
                const bool due = tests && c == 0 && sweep >= 2;
                const bool judge = due && t >= kGspT - 64;
                v4u pre[4][2];
                if (judge) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int j = (t & 63) + 64 * u;
                        if (j < a.G) { const int off = ((j * 4 + ((sweep - 2) & 3)) * 2) * 16; pre[u][0] = gsp_load(rpart, off); pre[u][1] = gsp_load(rpart, off + 16); }
                    }
                }
                lap(3);
                if (p > 0) fetch_halo(c > 0 ? c - 1 : C - 1, (c > 0 ? sweep : sweep - 1) & 1, stamp0 + (unsigned)(p - 1), keep_old);
                lap(0);
                __syncthreads();
                lap(1);
                if (block_failed()) return -2;
                if (tests && ctl[4 + 2 * ((p + 1) & 1)] != 0) return ctl[5 + 2 * ((p + 1) & 1)];   // the verdict of the PREVIOUS phase: that sweep met the tolerance
                GSP_FLAP0();
                if (parked >= 0) { publish_parked(parked, stamp0 + (unsigned)parked); parked = -1; }
                if (judge) verdict_wave(sweep - 2, stamp0 + (unsigned)(sweep - 2), pre, p & 1);
                GSP_FLAP(4);
                int role = 0;
                if (tests) {
                    if (C >= 2 && c == C - 1) role = 2;
                    else if (sweep > 0) role = 1;
                }
                racc += sweep_colour(c, sweep & 1, stamp0 + (unsigned)p, role, keep_old, sweep == 0);
                GSP_FLAP0();
                if (tests && sweep > 0 && c == c_pub) { park(racc); parked = sweep - 1; racc = 0.0; }
                GSP_FLAP(5);
                lap(2);
            }
        }
        if (prof) { a.prof[0] += pt[0]; a.prof[1] += pt[1]; a.prof[2] += pt[2]; a.prof[3] += pt[3]; a.prof[4] += (unsigned long long)(n * C);
                    a.prof[5] += (unsigned long long)clock64() - cyc0; a.prof[6] += wall_clock64() - wc0; }
        // the values of the last colour of the last sweep: the block's state is complete again
        if (n > 0 && a.G > 1) fetch_halo(C - 1, (n - 1) & 1, stamp0 + (unsigned)(n * C - 1), false);
        __syncthreads();
        if (block_failed()) return -2;
        if (!tests || n < 1) return -1;
        { const int q = ((n - 1) * C) & 1; if (ctl[4 + 2 * q] != 0) return ctl[5 + 2 * q]; }   // the last judged phase, (n - 1) C: read here (C = 1) or seen one phase later already
        if (parked >= 0) publish_parked(parked, stamp0 + (unsigned)parked);
        __syncthreads();                                    // (thread 0 has read the parked sums)
        // the last sweep: its last colour's rows were taken after their update, the others now
        racc += direct_residual(0, C >= 2 ? C - 1 : 1);
        park(racc);
        __syncthreads();
        publish_parked(n - 1, stamp0 + (unsigned)(n - 1));
        if (n >= 2) {
            const bool conv = verdict(n - 2, stamp0 + (unsigned)(n - 2), nullptr, false);
            if (block_failed()) return -2;
            if (conv) return n - 2;
            __syncthreads();                                // (everybody has read the verdict word before it is written again)
        }
        const bool conv = verdict(n - 1, stamp0 + (unsigned)(n - 1), nullptr, false);
        if (block_failed()) return -2;
        return conv ? n - 1 : -1;
    };

    const unsigned stamp0 = a.seq * 4096u + 1u;         // phases (and partials) of the first run: stamp0 + p; of a replay: stamp0 + 2048 + p
    const int n = a.max_sweeps;
    int failed_tests = n, conv_flag = 0;                // what the counters get: sweeps whose test failed, done
#ifdef ADMM_GSP_PROF_FINE
    const unsigned long long tk1 = wall_clock64();
#endif
    const int first = run(n, a.check != 0, stamp0);
    if (first == -2) return;
    if (first >= 0) { failed_tests = first; conv_flag = 1; }
    if (first >= 0 && first < n - 1) {
        // sweep `first` met the tolerance: the reference stopped there.  Everybody meets (nobody may still be reading this run's
        // granules; x in memory is still the input), then the solve is replayed for first + 1 sweeps without tests.
        const unsigned ms = stamp0 + 4000u;
        if (t == 0) gsp_store(rmeet, b * 16, gsp_pack(0.0, ms));
        if (t < 64) {
            for (int j = t; j < a.G; j += 64) {
                unsigned spins = 0;
                while (true) {
                    const v4u g = gsp_load(rmeet, j * 16);
                    if (gsp_ok(g, ms)) break;
                    if (poll_failed(spins)) break;
                }
            }
        }
        __syncthreads();
        if (block_failed()) return;
        load_x();
        if (t == 0) ctl[10] = 0;     // (the projections of the abandoned run do not count)
        __syncthreads();
        if (run(first + 1, false, stamp0 + 2048u) == -2) return;
    }
    // ---- write the block's rows back, counters ----
    for (int i = t; i < n_own; i += kGspT) {
        const int v = a.orig[row_base + i];
#pragma unroll
        for (int q = 0; q < 3; ++q) a.x[3 * (size_t)v + q] = xl[3 * i + q];
    }
    __syncthreads();
    if (t == 0 && a.proj && ctl[10] > 0) atomicAdd(a.proj, (unsigned long long)ctl[10]);
#ifdef ADMM_GSP_PROF_FINE
    if (proff) { for (int k = 0; k < 6; ++k) a.prof[8 + k] += pf[k]; a.prof[14] += tk1 - tk0; a.prof[15] += wall_clock64() - tk1; }
#endif
    if (b == 0 && t == 0) {
        *a.done = conv_flag; *a.sweeps = failed_tests;      // (stored, not accumulated: the launch needs no memset in front of it)
        atomicAdd(a.total, failed_tests);
    }
}

} // namespace admm_k
