// kernels.hpp -- gfx950 kernels of the ADMM hot path (included once by admm_hip.hip).
//
// Data layout in HBM (FP64, int32 indices):
//   node vectors  x, v, m, Mxbar, curr, b, r, p, ...   [n_verts][3] interleaved (the API's layout)
//   per-tet       idx  int4[nt]            one 16-B load per lane, coalesced
//                 u [9][ld], z [9][ld]   SoA, ld = nt + 1: lane t reads component c at c*ld + t -> every load/store of a
//                 wave is one contiguous 512-B run; Binv [9][ld] likewise, but only when the tets do not share one set of rest
//                 positions -- otherwise x0 [n_verts][3] is gathered next to x and Binv recomputed (tet_rest_binv)
//                 sc [ld] = dt^2 w^2, mat int[nt] -> Mat table {mu, lambda, k, kappa, type, table}
//                 corner forces: not stored per tet -- summed per vertex over the block's 256 tets in LDS, one 32-byte record per
//                 (block, vertex) in rec [n_rec + 1][4] (host_setup.hpp: TetChunks)
//   per-tri       idx int4[n], rest [4][ld], u/z [6][ld], cf [9][ld], sc, limits
//   Ahat / vertex->corner incidence: SELL-64 (slice = one wavefront, see host_setup.hpp)
// One lane owns one element (local step) or one vertex row (gather, SpMV): the SoA layout makes all
// streaming traffic fully coalesced 8-B-per-lane accesses, and the only irregular accesses are the
// 24-B position gathers, which hit L2 for any locality-preserving mesh order.
#pragma once
#include <hip/hip_runtime.h>
#include "device_math.hpp"

namespace admm_k {

using namespace admm_dev;

struct Mat { double mu, la, k, kappa; int type, table; };   // KIND 4: type 0..2 = xu:: spline with a compression term kappa; type 3 = tabulated (user-defined) spline `table`; type 4 = stable Neo-Hookean

constexpr int kMaxObst = 8;
struct Obstacles {
    int n;
    int kind[kMaxObst];
    double par[kMaxObst][4];
    const double *gmeta;      // sampled obstacles (kind 3): 10 doubles per grid -- origin xyz, spacing xyz, nodes nx ny nz, first node of its data
    const double *gdata;      // ... and 4 doubles per node: signed distance, normal xyz (x fastest)
};

// scalars of one PCG solve, double-buffered by iteration parity
struct CgScal {
    double gamma[3];
    double alpha[3];
    double gamma_b[3];
    int converged;
    int iters;
    int seq;      // sequence number of the solve these scalars belong to
    int pad_;
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide sum of NQ quantities (256 threads = 4 waves); result valid in all threads
template <int NQ>
__device__ __forceinline__ void block_sum(double *q, double *lds /* [4*NQ] */) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const double s = wave_sum(q[i]);
        if (lane == 0) lds[wv * NQ + i] = s;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NQ; ++i) q[i] = lds[i] + lds[NQ + i] + lds[2 * NQ + i] + lds[3 * NQ + i];
    __syncthreads();
}

// One DPP lane exchange of a double (two 32-bit DPP moves; VALU only, the LDS pipe stays free)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// Sums of 8 quantities over each 16-lane row of a wave, VALU only: two halving butterfly steps inside every quad
// (8 -> 4 -> 2 quantities per lane, exact xor-1 / xor-2 quad permutes), then rotations by 8 and 4 over the row
// (they preserve the two low lane bits).  On return the lanes with (lane & 15) < 4 hold the row sums of the
// quantities id = 4 (lane & 1) + (lane & 2) + {0, 1} in r0, r1.  Fixed order -> deterministic.
__device__ __forceinline__ void row_sum8(const double *q, double &r0, double &r1) {
    const int lane = threadIdx.x & 63;
    const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0;
    double a[4], b[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = (b0 ? q[4 + i] : q[i]) + dpp_f64<0xB1>(b0 ? q[i] : q[4 + i]);
#pragma unroll
    for (int i = 0; i < 2; ++i) b[i] = (b1 ? a[2 + i] : a[i]) + dpp_f64<0x4E>(b1 ? a[i] : a[2 + i]);
#pragma unroll
    for (int i = 0; i < 2; ++i) { b[i] += dpp_f64<0x128>(b[i]); b[i] += dpp_f64<0x124>(b[i]); }
    r0 = b[0]; r1 = b[1];
}

// XCD-aware block remap (MI355X: block b runs on XCD b % 8, each XCD has its own 4 MB L2).  Blocks that
// land on the same XCD are given a CONTIGUOUS range of work, so neighbouring rows / elements -- which
// share gathered cache lines -- hit the same L2 instead of re-fetching through the fabric.  Bijective on
// [0, nb); only a speed-up, never needed for correctness.
__device__ __forceinline__ int xcd_block() {
    const int nb = (int)gridDim.x, b = (int)blockIdx.x;
    const int per = nb >> 3, rem = nb & 7, x = b & 7, i = b >> 3;
    return x * per + (x < rem ? x : rem) + i;
}

// wave-uniform slice index of this wave (one wave = one SELL slice)
__device__ __forceinline__ int wave_slice() {
    return __builtin_amdgcn_readfirstlane(xcd_block() * 4 + (int)(threadIdx.x >> 6));
}

// ---------------------------------------------------------------------------------------------------
// Solver::step prologue (src/Solver.cpp:57-72): gravity, x_bar, M x_bar, curr_x = x_bar
__global__ __launch_bounds__(256) void k_predict(int n3, double dt, double gravity, const double *__restrict__ x,
                                                 double *__restrict__ v, const double *__restrict__ m,
                                                 double *__restrict__ Mxbar, double *__restrict__ curr) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n3) return;
    double vi = v[i];
    if (i % 3 == 1) { vi += dt * gravity; v[i] = vi; }
    const double xb = x[i] + dt * vi;
    Mxbar[i] = m[i] * xb;
    curr[i] = xb;
}

// WindForce::project (src/ExplicitForce.cpp:47-104; Wejchert & Haumann 1991) on the device, for device-resident stepping:
// per triangle the force -alpha_n area v_n |v_n| n (alpha_n = 1000) from the velocity relative to the wind along the unit normal,
// times 0.33 dt ...
__global__ __launch_bounds__(256) void k_wind_tris(int n, const int *__restrict__ tris, const double *__restrict__ x,
                                                   const double *__restrict__ v, double dx, double dy, double dz, double dt,
                                                   double *__restrict__ force /* [3][n + 1], entry n stays zero */) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const int i0 = 3 * tris[3 * t], i1 = 3 * tris[3 * t + 1], i2 = 3 * tris[3 * t + 2];
    const double vr[3] = {(v[i0] + v[i1] + v[i2]) * (1.0 / 3.0) - dx, (v[i0 + 1] + v[i1 + 1] + v[i2 + 1]) * (1.0 / 3.0) - dy,
                          (v[i0 + 2] + v[i1 + 2] + v[i2 + 2]) * (1.0 / 3.0) - dz};
    const double a[3] = {x[i1] - x[i0], x[i1 + 1] - x[i0 + 1], x[i1 + 2] - x[i0 + 2]};
    const double b[3] = {x[i2] - x[i0], x[i2 + 1] - x[i0 + 1], x[i2 + 2] - x[i0 + 2]};
    double nn[3];
    cross3(a, b, nn);
    const double len = sqrt(dot3(nn, nn));
    double f[3] = {0.0, 0.0, 0.0};
    if (len > 0.0) {    // (degenerate triangle: no area, no force)
        const double il = 1.0 / len;
        const double un[3] = {nn[0] * il, nn[1] * il, nn[2] * il};
        const double vn = dot3(un, vr);
        const double sc = -1000.0 * (0.5 * len) * vn * fabs(vn) * 0.33 * dt;
#pragma unroll
        for (int j = 0; j < 3; ++j) f[j] = sc * un[j];
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) force[(size_t)j * (n + 1) + t] = f[j];
}
// ... ADDED TO THE VELOCITY of the triangle's three nodes (the reference does not divide by the node mass; kept): a gather over
// the vertex -> triangle incidence lists, no atomics.  Every triangle sees the velocities of the start of the step (the
// reference's OpenMP loop reads whatever its critical sections have already written: order-dependent there).
__global__ __launch_bounds__(256) void k_wind_nodes(int nv, int n_tris, const int *__restrict__ ptr, const int *__restrict__ w,
                                                    const int *__restrict__ inc, const double *__restrict__ force, double *__restrict__ v) {
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (s * 64 >= nv) return;
    const int vtx = s * 64 + lane;
    const int *il = inc + ptr[s] + lane;
    double acc[3] = {0.0, 0.0, 0.0};
    for (int k = 0; k < w[s]; ++k) {
        const int t = il[64 * k] >> 2;      // (element, corner) code of incidence_sell; padding = the all-zero entry n_tris
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[j] += force[(size_t)j * (n_tris + 1) + t];
    }
    if (vtx < nv) {
#pragma unroll
        for (int j = 0; j < 3; ++j) v[3 * (size_t)vtx + j] += acc[j];
    }
}

// Solver::step epilogue (src/Solver.cpp:105-106)
__global__ __launch_bounds__(256) void k_finish(int n3, double inv_dt, double *__restrict__ x, double *__restrict__ v,
                                                const double *__restrict__ curr) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n3) return;
    const double c = curr[i];
    v[i] = (c - x[i]) * inv_dt;
    x[i] = c;
}

// ---------------------------------------------------------------------------------------------------
// LOCAL STEP, tets: EnergyTerm::update (src/EnergyTerm.hpp:130-140) for every tet of one constitutive
// model, fused with the element's contribution to dt^2 D^T W^2 (z - u) (src/Solver.cpp:98): the block (= CHUNK of 256
// consecutive tets) parks its 4 x 256 corner force vectors in LDS and sums them per vertex into RECORDS (host_setup.hpp:
// TetChunks; fixed order, no atomics, deterministic) that k_gather_rhs sums per vertex -- ~25 B per tet written and ~4
// sectors per vertex gathered instead of 96 B per tet and ~24 scattered corner forces per vertex.
//   F = [x1-x0, x2-x0, x3-x0] Binv  ==  D_i x   (D-block of src/TetEnergyTerm.cpp:50-71)
constexpr int kChunkFanK = 8, kChunkLdK = 257;   // == admm_host::kChunkFan, kChunkLd (checked in admm_hip.hip)
struct TetArgs {
    int ld;
    const int4 *idx; const double *Binv; double *u; double *z; const double *sc; const int *mat_id; const Mat *mats;
    const double *x;
    const double *x0;         // rest positions [nv][3] (the tets' Binv is then recomputed per launch instead of streamed), or nullptr
    const unsigned short *ch_ent; const int *ch_group, *ch_rec; double *rec; int chunk0;   // chunk plan, records [n_rec + 1][4], first chunk of this launch
    const double *spl;        // tabulated user splines (device_math.hpp: kSplineTableDoubles each), KIND 4 / Mat::type 3
    // kernel-level timing (stats only): every wave stores the device wall clock at entry in ts[wave slot] and at exit in
    // ts[ts_n + wave slot]; nullptr = off.  max(exit) - min(entry) is the launch's
    // duration as rocprofv3 reports it, without the dispatch gaps an event pair around the launch also counts.
    unsigned long long *ts; int ts_n;
};
__device__ __forceinline__ void ts_enter(const TetArgs &a) {
    if (a.ts && (threadIdx.x & 63) == 0) a.ts[blockIdx.x * 4 + (threadIdx.x >> 6)] = wall_clock64();
}
__device__ __forceinline__ void ts_exit(const TetArgs &a) {
    // (no wait for the wave's own stores here: holding the wave slot until they have drained made the kernel itself 4 us
    // longer; the stamp therefore precedes the drain of the last stores, ~1 us)
    if (a.ts && (threadIdx.x & 63) == 0) a.ts[a.ts_n + blockIdx.x * 4 + (threadIdx.x >> 6)] = wall_clock64();
}
// per launch slot: min over the waves' entry stamps (0 = wave never ran), max over their exit stamps
__global__ __launch_bounds__(256) void k_ts_reduce(const unsigned long long *__restrict__ ts, int ts_n, int n_launches,
                                                   unsigned long long *__restrict__ out) {
    __shared__ unsigned long long lo[256], hi[256];
    const unsigned long long *base = ts + 2 * (size_t)ts_n * blockIdx.x;
    unsigned long long mn = ~0ull, mx = 0ull;
    for (int i = threadIdx.x; i < ts_n; i += 256) {
        const unsigned long long a = base[i], b = base[ts_n + i];
        if (a != 0ull && a < mn) mn = a;
        if (b > mx) mx = b;
    }
    lo[threadIdx.x] = mn; hi[threadIdx.x] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            if (lo[threadIdx.x + o] < lo[threadIdx.x]) lo[threadIdx.x] = lo[threadIdx.x + o];
            if (hi[threadIdx.x + o] > hi[threadIdx.x]) hi[threadIdx.x] = hi[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && (int)blockIdx.x < n_launches) { out[2 * blockIdx.x] = lo[0]; out[2 * blockIdx.x + 1] = hi[0]; }
}

// Every per-tet array is SoA with the tet index fastest, so all accesses of a thread are "array base + c * ld
// (uniform) + t".  They go through buffer instructions -- descriptor in SGPRs, ONE 32-bit VGPR offset shared by
// all of them, the c * ld part in an SGPR -- instead of 64-bit VGPR address chains, which the compiler kept
// alive from the u loads to the u stores (~20 VGPRs in a kernel that sits on the 168-VGPR / 3-waves-per-SIMD
// edge) and paid for with one 64-bit VALU add per access.
typedef unsigned bv4u __attribute__((ext_vector_type(4)));
typedef unsigned bv2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t soa_rsrc(const void *p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7ffffff0, 0x00020000);
}
// cache policy of the per-element STREAMS (read or written exactly once per launch) -- gfx950 aux bits: 1 = sc0, 2 = nt
// (non-temporal: first in line for eviction), 16 = sc1.  The gathered vertex positions keep the default policy.
#ifndef ADMM_STREAM_LD_AUX
#define ADMM_STREAM_LD_AUX 0
#endif
#ifndef ADMM_STREAM_ST_AUX
#define ADMM_STREAM_ST_AUX 0
#endif
__device__ __forceinline__ double buf_ld_stream(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    union { double d; bv2u v; } t;
    t.v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, ADMM_STREAM_LD_AUX);
    return t.d;
}
__device__ __forceinline__ void buf_st_stream(__amdgpu_buffer_rsrc_t rs, int voff, int soff, double x) {
    union { double d; bv2u v; } t; t.d = x;
    __builtin_amdgcn_raw_buffer_store_b64(t.v, rs, voff, soff, ADMM_STREAM_ST_AUX);
}
__device__ __forceinline__ double buf_ld(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    union { double d; bv2u v; } t;
    t.v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
    return t.d;
}
__device__ __forceinline__ void buf_st(__amdgpu_buffer_rsrc_t rs, int voff, int soff, double x) {
    union { double d; bv2u v; } t; t.d = x;
    __builtin_amdgcn_raw_buffer_store_b64(t.v, rs, voff, soff, 0);
}

// What one tet reads: loaded by tet_load (streams: Binv, u, scale, material id) and tet_gather (the four vertex
// positions); all loads of a tet are issued before any of its arithmetic.
struct TetIn { double Bi[9], ui[9]; int mid; };
struct TetPos { double p[12]; };

__device__ __forceinline__ int4 tet_load_idx(const TetArgs &a, int t) {
    union { int4 i; bv4u v; } q4;
    q4.v = __builtin_amdgcn_raw_buffer_load_b128(soa_rsrc(a.idx), t * 16, 0, 0);
    return q4.i;
}
template <int KIND, bool REST>
__device__ __forceinline__ void tet_load(const TetArgs &a, int t, TetIn &in) {
    const int ld8 = a.ld * 8, t8 = t * 8;   // bytes between two components of an SoA array (the largest offset, 12 ld 8 for cf, stays < 2^31: admm_hip_create rejects more than 22.3 M elements)
    const __amdgpu_buffer_rsrc_t rBinv = soa_rsrc(a.Binv), ru = soa_rsrc(a.u);
#pragma unroll
    for (int c = 0; c < 9; ++c) {
        if (!REST) in.Bi[c] = buf_ld_stream(rBinv, t8, c * ld8);      // REST: Binv comes from the rest positions (tet_rest_binv)
        in.ui[c] = buf_ld_stream(ru, t8, c * ld8);
    }
    in.mid = (KIND == 0) ? 0 : __builtin_amdgcn_raw_buffer_load_b32(soa_rsrc(a.mat_id), t * 4, 0, 0);
}
// Binv = [x1 - x0, x2 - x0, x3 - x0]^-1 of the REST positions (src/TetEnergyTerm.cpp:31-48), recomputed per launch: four
// 24-byte gathers of vertex data (4.4 MB at 1 M tets: L2-resident, next to the same gathers of x) and ~45 FP64 operations
// instead of 72 streamed bytes per tet -- a quarter of the kernel's HBM traffic.  admm_hip_create switches this on only when
// every tet's recomputed Binv reproduces the caller's (host_setup.cpp: tet_rest_positions).
__device__ __forceinline__ void tet_rest_binv(const TetArgs &a, const int4 id, TetIn &in) {
    const __amdgpu_buffer_rsrc_t r0 = soa_rsrc(a.x0);
    const int o[4] = {id.x * 24, id.y * 24, id.z * 24, id.w * 24};
    double p[12];
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int j = 0; j < 3; ++j) p[3 * v + j] = buf_ld(r0, o[v] + 8 * j, 0);
    double e0[3], e1[3], e2[3], c0[3], c1[3], c2[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { e0[j] = p[3 + j] - p[j]; e1[j] = p[6 + j] - p[j]; e2[j] = p[9 + j] - p[j]; }
    cross3(e1, e2, c0); cross3(e2, e0, c1); cross3(e0, e1, c2);
    const double idet = fast_rcp(fma(e0[0], c0[0], fma(e0[1], c0[1], e0[2] * c0[2])));
#pragma unroll
    for (int r = 0; r < 3; ++r) { in.Bi[r * 3 + 0] = c0[r] * idet; in.Bi[r * 3 + 1] = c1[r] * idet; in.Bi[r * 3 + 2] = c2[r] * idet; }
}
__device__ __forceinline__ void tet_gather(const TetArgs &a, const int4 id, TetPos &x) {
    const __amdgpu_buffer_rsrc_t rx = soa_rsrc(a.x);
    const int o[4] = {id.x * 24, id.y * 24, id.z * 24, id.w * 24};
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int j = 0; j < 3; ++j) x.p[3 * v + j] = buf_ld(rx, o[v] + 8 * j, 0);
}

// prox + dual update + corner forces of one tet from its loaded inputs
// LDS block of the local step: rows of kChunkLdK doubles, one column per thread (thread-private, conflict-free 8-byte
// accesses) plus the all-zero padding column 256.  Rows 0..8 park Binv across the prox, rows 9..17 V (StVK); after the prox
// rows 0..11 hold the thread's four corner forces for the chunk's reduction.
typedef __attribute__((address_space(3))) double LdsDk;
// -DADMM_LOCAL_PHASES (experiments only): every wave adds the wall-clock ticks (100 MHz) it spent between the marks of
// tet_compute_store to g_local_phase[]; [7] counts the waves.  experiments/local_phases.py reads them.
#ifdef ADMM_LOCAL_PHASES
constexpr int kPhaseWaves = 1 << 16;
__device__ unsigned long long g_local_phase[8 * kPhaseWaves];      // [wave of the launch][mark]: private slots, no atomics
#define ADMM_PHASE_SLOT() (g_local_phase + 8 * (((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6)) & (kPhaseWaves - 1)))
#define ADMM_PHASE_MARK(k) { const unsigned long long now_ = wall_clock64(); if ((threadIdx.x & 63) == 0) ADMM_PHASE_SLOT()[k] += now_ - ph_t_; ph_t_ = now_; }
#define ADMM_PHASE_BEGIN() unsigned long long ph_t_ = wall_clock64(); if ((threadIdx.x & 63) == 0) ADMM_PHASE_SLOT()[7] += 1ull;
#else
#define ADMM_PHASE_MARK(k)
#define ADMM_PHASE_BEGIN()
#endif
template <int KIND, bool WRITE_Z>
__device__ __forceinline__ void tet_compute_store(const TetArgs &a, int t, bool valid, int chunk, const TetIn &in, const TetPos &x, LdsDk *sL
#ifdef ADMM_LOCAL_PHASES
                                                  , unsigned long long ph_t_
#endif
                                                  ) {
    const int ld8 = a.ld * 8, t8 = t * 8;
    const __amdgpu_buffer_rsrc_t ru = soa_rsrc(a.u);
    LdsDk *sBi = sL + threadIdx.x, *sV = sL + 9 * kChunkLdK + threadIdx.x;     // row c of this thread: [c * kChunkLdK]
#ifdef ADMM_LOCAL_PHASES
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the marks below then separate the wait for the loads from the math)
#endif
    ADMM_PHASE_MARK(0);
    const Mat *__restrict__ mats = a.mats;
    // Binv is needed twice (F = Ds Binv before the prox, corner forces after it).  It is parked in LDS (sBi) in
    // between: thread-private slots, [c][tid] layout (bank-conflict-free 8-B accesses), no VGPRs held
    // across the prox and no second trip to HBM (rocprof FETCH_SIZE showed the re-read going to fabric).
    // sV: V is parked there across the stretch minimisation (StVK).
    double U[9], V[9], S0[3], S1[3];
    {
        double q[9];
        {
            double Ds[9];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double a0 = x.p[j];
                Ds[0 + j] = x.p[3 + j] - a0; Ds[3 + j] = x.p[6 + j] - a0; Ds[6 + j] = x.p[9 + j] - a0;
            }
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int j = 0; j < 3; ++j)   // EnergyTerm.hpp:133-135  zi = D_i x + u_i
                    q[r * 3 + j] = fma(Ds[j], in.Bi[r * 3 + 0], fma(Ds[3 + j], in.Bi[r * 3 + 1], fma(Ds[6 + j], in.Bi[r * 3 + 2], in.ui[r * 3 + j])));
#pragma unroll
            for (int c = 0; c < 9; ++c) sBi[c * kChunkLdK] = in.Bi[c];
        }
        signed_svd3(q, U, S0, V);   // q = U diag(S0) V^T to round-off: q itself is not needed any more
    }
    ADMM_PHASE_MARK(1);
    // the reduction list of this thread's record (first pass): in flight across the prox instead of after the block barrier
    const int g0 = __builtin_amdgcn_readfirstlane(a.ch_group[chunk]), g1 = __builtin_amdgcn_readfirstlane(a.ch_group[chunk + 1]);
    const int r0 = __builtin_amdgcn_readfirstlane(a.ch_rec[chunk]), nrec = __builtin_amdgcn_readfirstlane(a.ch_rec[chunk + 1]) - r0;
    const __amdgpu_buffer_rsrc_t re = soa_rsrc(a.ch_ent);
    union { bv4u v; unsigned short h[8]; } e;
    e.v = __builtin_amdgcn_raw_buffer_load_b128(re, (g0 * 256 + (int)threadIdx.x) * 16, 0, ADMM_STREAM_LD_AUX);
    // dt^2 w^2 of this tet: needed after the prox, fetched across it like the list (two registers less across the SVD: the fused
    // kernel sits exactly on its 128)
    const double s = buf_ld_stream(soa_rsrc(a.sc), valid ? t8 : 0, 0);
#pragma unroll
    for (int i = 0; i < 3; ++i) S1[i] = S0[i];
    if (KIND == 0) {
        prox_stretches<0>(0.0, 0.0, 0.0, S1);
    } else if (KIND == 4) {   // xu:: spline with kappa != 0 (dense-Hessian Newton; rare, not tuned)
        const Mat mt = mats[in.mid];
        if (mt.type == 3) prox_stretches_table(a.spl + (size_t)mt.table * kSplineTableDoubles, mt.k, S1);
        else if (mt.type == 4) prox_stretches_stable_nh(mt.mu, mt.la, mt.k, S1);      // ADMM_TET_STABLE_NH
        else prox_stretches_kappa(mt.type, mt.mu, mt.la, mt.k, mt.kappa, S1);
    } else {
        // NH, StVK and the co-rotated spline fit 4 waves/SIMD (128 VGPRs) with V out of the way during the stretch
        // minimisation AND the general Newton loop outlined (device_math.hpp: newton_stretch_general) -- inlined, that rare
        // path dictated 166 VGPRs = 3 waves/SIMD; forcing 128 then spilled on the common path and was slower (measured in both
        // rounds).  Same box, 1 M tets: 3 waves 68.4 us, 4 waves with spills 71.8 us, 4 waves + V parked + outlined loop 65.4 us.
#ifndef ADMM_PARK_V_NH
#define ADMM_PARK_V_NH 1
#endif
        constexpr bool kParkV = (KIND == 2) || (KIND == 3) || (KIND == 1 && ADMM_PARK_V_NH != 0);
        if (kParkV) {
#pragma unroll
            for (int c = 0; c < 9; ++c) sV[c * kChunkLdK] = V[c];
        }
        const Mat mt = mats[in.mid];
        prox_stretches<KIND>(mt.mu, mt.la, mt.k, S1);
        if (kParkV) {
#pragma unroll
            for (int c = 0; c < 9; ++c) V[c] = sV[c * kChunkLdK];
        }
    }
    ADMM_PHASE_MARK(2);
    // z = U diag(S1) V^T ; u_new = u + D_i x - z = q - z = U diag(S0 - S1) V^T   (EnergyTerm.hpp:137)
    // G = dt^2 w^2 (z - u_new) = s U diag(2 S1 - S0) V^T
    double G[9];
    {
        double du[3], dg[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) { du[i] = S0[i] - S1[i]; dg[i] = s * (2.0 * S1[i] - S0[i]); }
        double un[9];
        usvt(U, du, V, un);
        if (valid) {
#pragma unroll
            for (int c = 0; c < 9; ++c) buf_st_stream(ru, t8, c * ld8, un[c]);
        }
        if (WRITE_Z && valid) {
            double zi[9];
            usvt(U, S1, V, zi);
            const __amdgpu_buffer_rsrc_t rz = soa_rsrc(a.z);
#pragma unroll
            for (int c = 0; c < 9; ++c) buf_st(rz, t8, c * ld8, zi[c]);
        }
        usvt(U, dg, V, G);
    }
    // corner forces: H(j,m) = sum_r G(j,r) Binv(m,r); corner m+1 gets H(:,m), corner 0 gets -sum_m H(:,m).
    double f[12] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const double b0 = sBi[(0 + m) * kChunkLdK], b1 = sBi[(3 + m) * kChunkLdK], b2 = sBi[(6 + m) * kChunkLdK];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double h = fma(G[j], b0, fma(G[3 + j], b1, G[6 + j] * b2));
            f[3 * (m + 1) + j] = h;
            f[j] -= h;
        }
    }
    // The chunk's reduction: every thread parks its corner forces (its own column: nobody else has touched it), then thread j
    // of pass p sums the <= 8 corner forces of record 256 p + j (16 bytes of LDS offsets, host-built) and stores the record as
    // one 32-byte sector.  Lanes past the end of the model's tet range park values no list refers to.
#pragma unroll
    for (int c = 0; c < 12; ++c) sBi[c * kChunkLdK] = f[c];
    ADMM_PHASE_MARK(3);
    __syncthreads();
    ADMM_PHASE_MARK(4);
    {
        const __amdgpu_buffer_rsrc_t rr = soa_rsrc(a.rec);
        const LdsDk *base = sL;
        for (int g = g0; g < g1; ++g) {
            if (g > g0) e.v = __builtin_amdgcn_raw_buffer_load_b128(re, (g * 256 + (int)threadIdx.x) * 16, 0, ADMM_STREAM_LD_AUX);
            double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int i = 0; i < kChunkFanK; ++i) {
                const LdsDk *q = (const LdsDk *)((const __attribute__((address_space(3))) char *)base + e.h[i]);
                s0 += q[0]; s1 += q[kChunkLdK]; s2 += q[2 * kChunkLdK];
            }
            const int j = (g - g0) * 256 + (int)threadIdx.x;
            if (j < nrec) {
                union { double d[2]; bv4u v; } p0; p0.d[0] = s0; p0.d[1] = s1;
                union { double d; bv2u v; } p1; p1.d = s2;
                __builtin_amdgcn_raw_buffer_store_b128(p0.v, rr, (r0 + j) * 32, 0, ADMM_STREAM_ST_AUX);
                __builtin_amdgcn_raw_buffer_store_b64(p1.v, rr, (r0 + j) * 32 + 16, 0, ADMM_STREAM_ST_AUX);
            }
        }
    }
    ADMM_PHASE_MARK(5);
#ifdef ADMM_LOCAL_PHASES
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ADMM_PHASE_MARK(6);
#endif
}

// t_end = end of this constitutive model's tet range.  The whole block takes part (the chunk's reduction synchronises it):
// lanes past the end redo the last tet of the range and store nothing of their own.
template <int KIND, bool WRITE_Z, bool REST>
__device__ __forceinline__ void local_tet_body(const TetArgs &a, int t, int t_end, int chunk, LdsDk *sL) {
    TetIn in; TetPos x;
    ADMM_PHASE_BEGIN();
    const bool valid = t < t_end;
    const int tl = valid ? t : t_end - 1;
    const int4 id = tet_load_idx(a, tl);
    tet_load<KIND, REST>(a, tl, in);
    if (REST) tet_rest_binv(a, id, in);
    tet_gather(a, id, x);
    if (threadIdx.x < 3) sL[threadIdx.x * kChunkLdK + 256] = 0.0;     // the padding column of the reduction lists
#ifdef ADMM_LOCAL_PHASES
    tet_compute_store<KIND, WRITE_Z>(a, t, valid, chunk, in, x, sL, ph_t_);
#else
    tet_compute_store<KIND, WRITE_Z>(a, t, valid, chunk, in, x, sL);
#endif
}

// one constitutive model per launch (used when a scene has a single model, and by the parity entry point)
template <int KIND, bool WRITE_Z, bool REST>
#ifndef ADMM_NH_WAVES
#define ADMM_NH_WAVES 4
#endif
__global__ __launch_bounds__(256, (KIND == 1 ? ADMM_NH_WAVES : KIND == 4 ? 2 : 4)) void k_local_tets(int t0, int t1, TetArgs a) {
    __shared__ double sLm[((KIND == 2 || KIND == 3 || (KIND == 1 && ADMM_PARK_V_NH != 0)) ? 18 : 12) * kChunkLdK];     // rows 0..8: Binv; 9..: V (parked); 0..11: corner forces
    LdsDk *sL = (LdsDk *)sLm;
    const int blk = xcd_block();
    ts_enter(a);
    local_tet_body<KIND, WRITE_Z, REST>(a, t0 + blk * 256 + (int)threadIdx.x, t1, a.chunk0 + blk, sL);
    ts_exit(a);
}

// all models in ONE launch: block ranges [0,nb0) linear, [nb0,nb1) NH, [nb1,nb2) StVK (wave-uniform branch).
// Avoids the ramp-down / ramp-up between per-model launches of a mixed scene.  (Chunks are numbered model by model in this
// order, so the block index is the chunk index.)
template <bool WRITE_Z, bool REST>
__global__ __launch_bounds__(256, ADMM_NH_WAVES) void k_local_tets_fused(int b0, int b1, int b2, int b3, int nb0, int nb1, TetArgs a) {
    __shared__ double sLm[18 * kChunkLdK];
    LdsDk *sL = (LdsDk *)sLm;
    const int blk = xcd_block();
    ts_enter(a);
    if (blk < nb0) {
        local_tet_body<0, WRITE_Z, REST>(a, b0 + blk * 256 + (int)threadIdx.x, b1, a.chunk0 + blk, sL);
    } else if (blk < nb1) {
        local_tet_body<1, WRITE_Z, REST>(a, b1 + (blk - nb0) * 256 + (int)threadIdx.x, b2, a.chunk0 + blk, sL);
    } else {
        local_tet_body<2, WRITE_Z, REST>(a, b2 + (blk - nb1) * 256 + (int)threadIdx.x, b3, a.chunk0 + blk, sL);
    }
    ts_exit(a);
}

// LOCAL STEP, triangles (src/TriEnergyTerm.cpp:54-101): F (3x2) = [x1-x0, x2-x0] rest
template <bool WRITE_Z>
__global__ __launch_bounds__(256) void k_local_tris(int n, int ld, const int4 *__restrict__ idx,
                                                    const double *__restrict__ rest, double *__restrict__ u,
                                                    double *__restrict__ z, const double *__restrict__ sc,
                                                    const double *__restrict__ lmin, const double *__restrict__ lmax,
                                                    const double *__restrict__ x, double *__restrict__ cf) {
    const int t = xcd_block() * 256 + threadIdx.x;
    if (t >= n) return;
    const int4 id = idx[t];
    double R[4], ui[6];
#pragma unroll
    for (int c = 0; c < 4; ++c) R[c] = rest[(size_t)c * ld + t];
#pragma unroll
    for (int c = 0; c < 6; ++c) ui[c] = u[(size_t)c * ld + t];
    const double *p0 = x + 3 * (size_t)id.x, *p1 = x + 3 * (size_t)id.y, *p2 = x + 3 * (size_t)id.z;
    double F[6], q[6], zi[6];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double a = p0[j], e1 = p1[j] - a, e2 = p2[j] - a;
        F[j] = fma(e1, R[0], e2 * R[1]);
        F[3 + j] = fma(e1, R[2], e2 * R[3]);
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) q[c] = F[c] + ui[c];
    prox_tri(q, lmin[t], lmax[t], zi);
    const double s = sc[t];
    double G[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const double un = ui[c] + (F[c] - zi[c]);
        u[(size_t)c * ld + t] = un;
        if (WRITE_Z) z[(size_t)c * ld + t] = zi[c];
        G[c] = s * (zi[c] - un);
    }
    // D = S rest: corner1 coefficient (R0 for col0, R2 for col1), corner2 (R1, R3), corner0 = -(sum)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double h1 = fma(G[j], R[0], G[3 + j] * R[2]);
        const double h2 = fma(G[j], R[1], G[3 + j] * R[3]);
        cf[(size_t)(0 + j) * ld + t] = -(h1 + h2);
        cf[(size_t)(3 + j) * ld + t] = h1;
        cf[(size_t)(6 + j) * ld + t] = h2;
    }
}

// LOCAL STEP, bending hinges (README.md:23-28 TODO of the reference, no reference code; include/admm_hip.h: desc.bend_*).  One EnergyTerm
// per hinge in the mould of EnergyTerm::update (src/EnergyTerm.hpp:130-140): D_i x = sum_k c_k x_{v_k} (3 rows), quadratic energy
// E(z) = kappa / 2 |z|^2  =>  prox(q) = gam q with gam = w^2 / (kappa + w^2), u += D_i x - z; the four corner forces
// dt^2 w^2 c_k (z - u) go to cf [12][ld] and are summed per vertex by k_gather_rhs.  lane = hinge, SoA like the triangles.
template <bool WRITE_Z>
__global__ __launch_bounds__(256) void k_local_bends(int n, int ld, const int4 *__restrict__ idx, const double *__restrict__ coef,
                                                     double *__restrict__ u, double *__restrict__ z, const double *__restrict__ sc,
                                                     const double *__restrict__ gam, const double *__restrict__ x, double *__restrict__ cf) {
    const int t = xcd_block() * 256 + threadIdx.x;
    if (t >= n) return;
    const int4 id = idx[t];
    const int vid[4] = {id.x, id.y, id.z, id.w};
    double c[4], Dx[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        c[k] = coef[(size_t)k * ld + t];
        const double *p = x + 3 * (size_t)vid[k];
#pragma unroll
        for (int j = 0; j < 3; ++j) Dx[j] = fma(c[k], p[j], Dx[j]);
    }
    const double s = sc[t], g = gam[t];
    double G[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double uo = u[(size_t)j * ld + t];
        const double zi = g * (Dx[j] + uo);
        const double un = uo + (Dx[j] - zi);
        u[(size_t)j * ld + t] = un;
        if (WRITE_Z) z[(size_t)j * ld + t] = zi;
        G[j] = s * (zi - un);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 3; ++j) cf[(size_t)(3 * k + j) * ld + t] = c[k] * G[j];
}

// ---------------------------------------------------------------------------------------------------
// RHS: b = [M x_bar] + dt^2 D^T W^2 (z - u)  (src/Solver.cpp:98), gathered per vertex from the corner
// forces, plus the SpringPin terms (src/SpringEnergyTerm.hpp:54-61), whose local step is done here.
struct GatherArgs {
    int nv, n_slices;
    const int *t_ptr, *t_w, *t_inc; const double *t_rec;           // tets: lists of records [.][4] (t_inc == nullptr -> none)
    const int *r_ptr, *r_w, *r_inc; const double *r_cf; int r_ld;  // tris
    const int *h_ptr, *h_w, *h_inc; const double *h_cf; int h_ld;  // bending hinges (h_inc == nullptr -> none)
    const int *vert_pin;       // [nv] pin term index or -1 (nullptr -> no pin terms)
    const double *pin_xyz; const int *pin_active; double *pin_u, *pin_z; double pin_sc; // dt^2 w_pin^2
    const double *pin_nrm;     // [3 per pin term] or nullptr: a non-zero (unit) normal makes the term a SLIDE pin -- prox = projection onto the plane n.(z - p) = 0
    const double *x;           // curr_x (for the pin terms)
    const double *Mxbar;       // added when add_mxbar
    double *b;
    int add_mxbar;             // 1 on a single GPU / on rank 0
    const int *order;          // [nv] vertex gathered by every row (rows sorted by list length inside 512-vertex windows)
};

// sum of the corner forces incident to this lane's vertex (incidence widths are multiples of 8; padding
// points at the all-zero dummy element ld-1).  Software-pipelined, 8 incidences (24 gathers) per round.
// sum of the records (partial sums of a chunk of tets, one 32-byte sector each) of this lane's vertex; widths are multiples of
// 4, padding points at the all-zero record.  Software-pipelined, 4 records per round.
__device__ __forceinline__ void gather_records(const int *__restrict__ inc, int w, const double *__restrict__ rec, double *acc) {
    constexpr int R = 4;
    const __amdgpu_buffer_rsrc_t rr = soa_rsrc(rec);
    int e[R];
#pragma unroll
    for (int i = 0; i < R; ++i) e[i] = inc[64 * i];
    for (int k = R; k <= w; k += R) {
        int en[R];
        if (k < w) {
#pragma unroll
            for (int i = 0; i < R; ++i) en[i] = inc[64 * (k + i)];
        }
        union { double d[2]; bv4u v; } g0[R]; union { double d; bv2u v; } g1[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            g0[i].v = __builtin_amdgcn_raw_buffer_load_b128(rr, e[i] * 32, 0, 0);
            g1[i].v = __builtin_amdgcn_raw_buffer_load_b64(rr, e[i] * 32 + 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < R; ++i) { acc[0] += g0[i].d[0]; acc[1] += g0[i].d[1]; acc[2] += g1[i].d; if (k < w) e[i] = en[i]; }
    }
}

template <bool AOS>
__device__ __forceinline__ void gather_corners(const int *__restrict__ inc, int w, const double *__restrict__ cf, int ld, double *acc) {
    constexpr int R = 8;
    // element (tet e >> 2, corner e & 3): AoS = doubles [16 (e >> 2) + 3 (e & 3), + 3) of the tet's record; SoA = cf[3 c + j][tet]
    const size_t js = AOS ? 1 : (size_t)ld;
    int e[R];
#pragma unroll
    for (int i = 0; i < R; ++i) e[i] = inc[64 * i];
    for (int k = R; k < w; k += R) {
        int en[R];
#pragma unroll
        for (int i = 0; i < R; ++i) en[i] = inc[64 * (k + i)];
        double g[3 * R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const double *p = AOS ? cf + (size_t)(e[i] >> 2) * 16 + 3 * (e[i] & 3) : cf + (size_t)(3 * (e[i] & 3)) * ld + (e[i] >> 2);
            g[3 * i] = p[0]; g[3 * i + 1] = p[js]; g[3 * i + 2] = p[2 * js];
        }
#pragma unroll
        for (int i = 0; i < R; ++i) { acc[0] += g[3 * i]; acc[1] += g[3 * i + 1]; acc[2] += g[3 * i + 2]; e[i] = en[i]; }
    }
    double g[3 * R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const double *p = AOS ? cf + (size_t)(e[i] >> 2) * 16 + 3 * (e[i] & 3) : cf + (size_t)(3 * (e[i] & 3)) * ld + (e[i] >> 2);
        g[3 * i] = p[0]; g[3 * i + 1] = p[js]; g[3 * i + 2] = p[2 * js];
    }
#pragma unroll
    for (int i = 0; i < R; ++i) { acc[0] += g[3 * i]; acc[1] += g[3 * i + 1]; acc[2] += g[3 * i + 2]; }
}

// The SpringPin term of vertex v (src/SpringEnergyTerm.hpp:31-73; EnergyTerm::update, src/EnergyTerm.hpp:130-140): z = pin point (or q with
// the term inactive), u += D x - z, and its share dt^2 w^2 (z - u) of the right-hand side added to acc.  One thread per vertex, once per
// ADMM iteration -- from k_gather_rhs or from the fill phase of k_pcg2 (pcg_onchip2.hpp: the solve that sums its own right-hand side).
__device__ __forceinline__ void pin_term_update(const int *__restrict__ vert_pin, const double *__restrict__ pin_xyz, const int *__restrict__ pin_active,
                                                double *__restrict__ pin_u, double *__restrict__ pin_z, double pin_sc, const double *__restrict__ pin_nrm,
                                                const double *__restrict__ x, int v, bool add, double *acc) {
    const int pi = vert_pin[v];
    if (pi < 0) return;
    const bool act = pin_active[pi] != 0;
    // SLIDE pin (README.md:23-28 TODO of the reference; a SpringPin, src/SpringEnergyTerm.hpp:31-73, whose prox projects onto
    // the plane through the pin's point instead of onto the point): z = q - n (n . (q - p)), q = D x + u
    double sl = 0.0, nrm[3] = {0.0, 0.0, 0.0};
    if (pin_nrm && act) {
#pragma unroll
        for (int j = 0; j < 3; ++j) nrm[j] = pin_nrm[3 * (size_t)pi + j];
        if (nrm[0] != 0.0 || nrm[1] != 0.0 || nrm[2] != 0.0) {
#pragma unroll
            for (int j = 0; j < 3; ++j) sl = fma(nrm[j], x[3 * (size_t)v + j] + pin_u[3 * (size_t)pi + j] - pin_xyz[3 * (size_t)pi + j], sl);
        }
    }
    const bool slide = nrm[0] != 0.0 || nrm[1] != 0.0 || nrm[2] != 0.0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double Dix = x[3 * (size_t)v + j];
        const double uo = pin_u[3 * (size_t)pi + j];
        const double zi = !act ? (Dix + uo) : slide ? fma(-sl, nrm[j], Dix + uo) : pin_xyz[3 * (size_t)pi + j];
        const double un = uo + (Dix - zi);
        pin_u[3 * (size_t)pi + j] = un;
        pin_z[3 * (size_t)pi + j] = zi;
        if (add) acc[j] += pin_sc * (zi - un);
    }
}

__global__ __launch_bounds__(256) void k_gather_rhs(GatherArgs a) {
    const int lane = threadIdx.x & 63;
    const int s = wave_slice();
    if (s >= a.n_slices) return;
    {
        const int r = s * 64 + lane;
        const int v = r < a.nv ? a.order[r] : a.nv;
        double acc[3] = {0.0, 0.0, 0.0};
        if (a.t_inc) gather_records(a.t_inc + a.t_ptr[s] + lane, a.t_w[s], a.t_rec, acc);
        if (a.r_inc) gather_corners<false>(a.r_inc + a.r_ptr[s] + lane, a.r_w[s], a.r_cf, a.r_ld, acc);
        if (a.h_inc) gather_corners<false>(a.h_inc + a.h_ptr[s] + lane, a.h_w[s], a.h_cf, a.h_ld, acc);
        if (v < a.nv) {
            if (a.vert_pin) pin_term_update(a.vert_pin, a.pin_xyz, a.pin_active, a.pin_u, a.pin_z, a.pin_sc, a.pin_nrm, a.x, v, a.add_mxbar != 0, acc);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                double r = acc[j];
                if (a.add_mxbar) r += a.Mxbar[3 * (size_t)v + j];
                a.b[3 * (size_t)v + j] = r;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// GLOBAL STEP (ADMM_LS_LDLT_AS_PCG): Jacobi-preconditioned CG in the Chronopoulos-Gear form (one
// reduction point per iteration -> two kernels per iteration) on A = diag(m) + Ahat (x) I3, the three
// axes carried as three independent systems sharing Ahat (own alpha/beta per axis).
// Replaces the prefactored LDLT solve of src/LinearSolver.hpp:87-90.
struct SellA { int n_rows, n_slices; const int *ptr, *w, *col; const double *val; };

// acc = sum_k Ahat(row,k) in_col for the 3 axes of one row of slice s (slice widths are multiples of 4).
// Software-pipelined: the column/value loads of round k+1 are in flight while round k gathers, so a
// wave keeps 4 index loads + 12 gathers outstanding instead of one dependent chain per non-zero.
__device__ __forceinline__ void sell_row(const SellA &A, int s, int lane, const double *__restrict__ in, double *acc) {
    const int w = A.w[s];
    const int *__restrict__ cp = A.col + A.ptr[s] + lane;
    const double *__restrict__ vp = A.val + A.ptr[s] + lane;
    acc[0] = acc[1] = acc[2] = 0.0;
    int c[4]; double a[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { c[i] = cp[64 * i]; a[i] = vp[64 * i]; }
    for (int k = 4; k < w; k += 4) {
        int cn[4]; double an[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { cn[i] = cp[64 * (k + i)]; an[i] = vp[64 * (k + i)]; }
        double g[12];
#pragma unroll
        for (int i = 0; i < 4; ++i) { const double *p = in + 3 * (size_t)c[i]; g[3 * i] = p[0]; g[3 * i + 1] = p[1]; g[3 * i + 2] = p[2]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[0] = fma(a[i], g[3 * i], acc[0]); acc[1] = fma(a[i], g[3 * i + 1], acc[1]); acc[2] = fma(a[i], g[3 * i + 2], acc[2]);
            c[i] = cn[i]; a[i] = an[i];
        }
    }
    double g[12];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const double *p = in + 3 * (size_t)c[i]; g[3 * i] = p[0]; g[3 * i + 1] = p[1]; g[3 * i + 2] = p[2]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        acc[0] = fma(a[i], g[3 * i], acc[0]); acc[1] = fma(a[i], g[3 * i + 1], acc[1]); acc[2] = fma(a[i], g[3 * i + 2], acc[2]);
    }
}

// u = dinv (b - A x) ; partial_b[3][NB] = sum b * dinv * b.   The residual r itself is never stored: the
// iteration carries only u = M^-1 r (r = u / dinv), which saves two vector passes per iteration.
__global__ __launch_bounds__(256) void k_cg_resid(SellA A, const double *__restrict__ m, const double *__restrict__ dinv,
                                                  const double *__restrict__ b, const double *__restrict__ x,
                                                  double *__restrict__ u,
                                                  double *__restrict__ part_b, int NB, CgScal *__restrict__ sc0, int seq,
                                                  int row_lo = 0, int row_hi = 0x7fffffff) {
    // [row_lo, row_hi): the rows this rank owns in the DISTRIBUTED solve (admm_hip.hip: launch_pcg_dist); rows of other ranks get
    // u = 0 and add nothing to the sums -- a sum all-reduce then assembles the vector and the sums.  Single GPU: every row.
    __shared__ double lds[12];
    if (blockIdx.x == 0 && threadIdx.x == 0) { sc0->converged = 0; sc0->iters = 0; sc0->seq = seq; }
    const int lane = threadIdx.x & 63;
    const int s = wave_slice();
    double q[3] = {0.0, 0.0, 0.0};
    if (s < A.n_slices) {
        const int row = s * 64 + lane;
        const bool own = row >= row_lo && row < row_hi;
        double acc[3] = {0.0, 0.0, 0.0};
        if (__any(own)) sell_row(A, s, lane, x, acc);          // (a slice of another rank's rows costs nothing)
        if (row < A.n_rows) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const size_t i = 3 * (size_t)row + j;
                const double bi = b[i], di = dinv[i];
                const double ri = bi - fma(m[i], x[i], acc[j]);
                u[i] = own ? di * ri : 0.0;
                if (own) q[j] = fma(bi * di, bi, q[j]);
            }
        }
    }
    block_sum<3>(q, lds);
    if (threadIdx.x == 0) { part_b[blockIdx.x] = q[0]; part_b[NB + blockIdx.x] = q[1]; part_b[2 * NB + blockIdx.x] = q[2]; }
}

// w = A u ; partials gamma = r.u = sum u^2 / dinv, delta = w.u   (skipped when the solve has converged)
__global__ __launch_bounds__(256) void k_cg_spmv(SellA A, const double *__restrict__ m, const double *__restrict__ u,
                                                 const double *__restrict__ dinv, double *__restrict__ w,
                                                 double *__restrict__ part, int NB, const CgScal *__restrict__ sc,
                                                 int row_lo = 0, int row_hi = 0x7fffffff) {
    __shared__ double lds[24];
    if (sc->converged) return;
    const int lane = threadIdx.x & 63;
    const int s = wave_slice();
    double q[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (s < A.n_slices) {
        const int row = s * 64 + lane;
        const bool own = row >= row_lo && row < row_hi;        // (distributed solve: see k_cg_resid)
        if (__any(own)) {
            double acc[3];
            sell_row(A, s, lane, u, acc);
            if (row < A.n_rows && own) {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const size_t i = 3 * (size_t)row + j;
                    const double ui = u[i];
                    const double wi = fma(m[i], ui, acc[j]);
                    w[i] = wi;
                    q[j] = fma(ui * ui, fast_rcp(dinv[i]), q[j]);
                    q[3 + j] = fma(wi, ui, q[3 + j]);
                }
            }
        }
    }
    block_sum<6>(q, lds);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) part[i * NB + blockIdx.x] = q[i];
    }
}

// reduce the partials, derive alpha/beta, update p, s, x, u.   One vertex (3 dofs) per thread; the
// thread's own vector entries are loaded BEFORE the partial-sum reduction so that the reduction's latency
// (every block re-reduces the <= ~700 SpMV partials; deterministic, no atomics) hides behind those loads.
__global__ __launch_bounds__(256) void k_cg_vec(int it, int nv, int NB, const double *__restrict__ part,
                                                const double *__restrict__ part_b, const CgScal *__restrict__ prev,
                                                CgScal *__restrict__ next, double tol2, int *__restrict__ total_iters,
                                                const double *__restrict__ dinv, double *__restrict__ p,
                                                double *__restrict__ s, double *__restrict__ x,
                                                double *__restrict__ u, const double *__restrict__ w,
                                                int *__restrict__ sig, int mark_here, int row_lo = 0, int row_hi = 0x7fffffff) {
    __shared__ double lds[36];
    const CgScal pv = *prev;
    // progress mark for the host (pinned memory): the chunk this kernel closes has (almost) drained
    if (mark_here && blockIdx.x == 0 && threadIdx.x == 0) {
        const int m = atomicAdd(total_iters + 5, 1) + 1;
        __hip_atomic_store(sig + 1, m, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (it > 0 && pv.converged) {
        if (blockIdx.x == 0 && threadIdx.x == 0) *next = pv;
        return;
    }
    const int v = xcd_block() * 256 + threadIdx.x;   // same XCD <-> row-range affinity as the SpMV
    const bool live = v < nv;
    const size_t i0 = 3 * (size_t)(live ? v : 0);
    double ru[3], rw[3], rp[3], rs[3], rx[3], rd[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        ru[a] = u[i0 + a]; rw[a] = w[i0 + a]; rx[a] = x[i0 + a]; rd[a] = dinv[i0 + a];
        rp[a] = (it == 0) ? 0.0 : p[i0 + a];
        rs[a] = (it == 0) ? 0.0 : s[i0 + a];
    }
    double q[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < NB; i += 256) {
#pragma unroll
        for (int kk = 0; kk < 6; ++kk) q[kk] += part[kk * NB + i];
        if (it == 0) {
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) q[6 + kk] += part_b[kk * NB + i];
        }
    }
    block_sum<9>(q, lds);
    double gb[3], alpha[3], beta[3];
    bool conv = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        gb[a] = (it == 0) ? q[6 + a] : pv.gamma_b[a];
        conv = conv && (q[a] <= tol2 * gb[a] + 1e-300);
    }
    if (conv) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            CgScal o = pv;
            if (it == 0) { o.iters = 0; }
#pragma unroll
            for (int a = 0; a < 3; ++a) { o.gamma[a] = q[a]; o.gamma_b[a] = gb[a]; o.alpha[a] = 0.0; }
            o.converged = 1;
            *next = o;
            atomicAdd(total_iters + 4, 1);        // solves that met the residual test
            atomicMax(total_iters + 3, o.iters);  // most iterations any solve of this step needed
            total_iters[8 + (pv.seq & 63)] = o.iters; // per-solve log (ring of 64)
            __hip_atomic_store(sig, pv.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); // tell the host
        }
        return;
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double g = q[a], d = q[3 + a];
        if (it == 0) { beta[a] = 0.0; alpha[a] = (d > 0.0) ? g / d : 0.0; }
        else {
            beta[a] = (pv.gamma[a] > 0.0) ? g / pv.gamma[a] : 0.0;
            const double den = (pv.alpha[a] != 0.0) ? d - beta[a] * g / pv.alpha[a] : d;
            alpha[a] = (den > 0.0) ? g / den : 0.0;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        CgScal o;
#pragma unroll
        for (int a = 0; a < 3; ++a) { o.gamma[a] = q[a]; o.alpha[a] = alpha[a]; o.gamma_b[a] = gb[a]; }
        o.converged = 0;
        o.iters = (it == 0 ? 0 : pv.iters) + 1;
        o.seq = pv.seq; o.pad_ = 0;
        *next = o;
        atomicAdd(total_iters, 1);
    }
    if (!live) return;
    if (v < row_lo || v >= row_hi) {      // distributed solve: another rank's row -- only its u is cleared, for the all-reduce that assembles u
        u[i0] = 0.0; u[i0 + 1] = 0.0; u[i0 + 2] = 0.0;
        return;
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double pi = fma(beta[a], rp[a], ru[a]);
        const double si = fma(beta[a], rs[a], rw[a]);
        p[i0 + a] = pi; s[i0 + a] = si;
        x[i0 + a] = fma(alpha[a], pi, rx[a]);
        u[i0 + a] = fma(-alpha[a] * rd[a], si, ru[a]);     // u = M^-1 (r - alpha s)
    }
}

// distributed solve: clear the rows of a node vector this rank does not own (a sum all-reduce then assembles the vector)
__global__ __launch_bounds__(256) void k_keep_own_rows(int nv, int row_lo, int row_hi, double *__restrict__ x) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= nv || (v >= row_lo && v < row_hi)) return;
    x[3 * (size_t)v] = 0.0; x[3 * (size_t)v + 1] = 0.0; x[3 * (size_t)v + 2] = 0.0;
}

// ---------------------------------------------------------------------------------------------------
// RECYCLED WARM START.  ADMM converges linearly, so the correction e_s = x_{s+1} - x_s of one global solve
// is almost a combination of the previous ones.  With A e_j = r0_j known for free (r0_j = the initial
// residual of solve j), the A-orthogonal (Galerkin) projection of the new error onto span{e_j} is
//     x <- x + sum_j c_j e_j,   (E^T R) c = E^T r0,   G_ij = e_i . r0_j  (= e_i^T A e_j),
// done per axis (the three axes are independent systems).  It cuts the initial PCG residual by 2-3 decades
// in the later ADMM iterations of a frame.  Purely an initial guess: the PCG still iterates to pcg_tol.
constexpr int kRc = 4; // recycled pairs in one projection
struct RcBasis { const double *E[kRc]; const double *R[kRc]; int cnt; };

// r0 = b - A x ; xs = x
__global__ __launch_bounds__(256) void k_rc_resid(SellA A, const double *__restrict__ m, const double *__restrict__ b,
                                                  const double *__restrict__ x, double *__restrict__ r0, double *__restrict__ xs) {
    const int lane = threadIdx.x & 63;
    const int s = wave_slice();
    if (s >= A.n_slices) return;
    const int row = s * 64 + lane;
    double acc[3];
    sell_row(A, s, lane, x, acc);
    if (row < A.n_rows) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const size_t i = 3 * (size_t)row + j;
            const double xi = x[i];
            r0[i] = b[i] - fma(m[i], xi, acc[j]);
            xs[i] = xi;
        }
    }
}

// partial sums (per axis) of G_ij = E_i . R_j and g_i = E_i . r0 for the `cnt` stored pairs
// (+ the preconditioned norms of r0 and b, so that the projection can be skipped when r0 already meets pcg_tol)
constexpr int kRcQ = kRc * kRc + kRc + 2;
__global__ __launch_bounds__(256) void k_rc_dots(int nv, RcBasis B, const double *__restrict__ r0, const double *__restrict__ b,
                                                 const double *__restrict__ dinv, double *__restrict__ part, int NBr) {
    const int cnt = B.cnt;
    __shared__ double red[16 * 8 * ((3 * kRcQ + 7) / 8)];
    double q[3][kRcQ];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int i = 0; i < kRcQ; ++i) q[a][i] = 0.0;
    for (int v = blockIdx.x * 256 + threadIdx.x; v < nv; v += gridDim.x * 256) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const size_t i = 3 * (size_t)v + a;
            double e[kRc], r[kRc];
#pragma unroll
            for (int j = 0; j < kRc; ++j) { e[j] = (j < cnt) ? B.E[j][i] : 0.0; r[j] = (j < cnt) ? B.R[j][i] : 0.0; }
            const double rr = r0[i], bb = b[i], di = dinv[i];
            q[a][20] = fma(rr * di, rr, q[a][20]);
            q[a][21] = fma(bb * di, bb, q[a][21]);
#pragma unroll
            for (int ii = 0; ii < kRc; ++ii) {
#pragma unroll
                for (int jj = 0; jj < kRc; ++jj) q[a][ii * kRc + jj] = fma(e[ii], r[jj], q[a][ii * kRc + jj]);
                q[a][16 + ii] = fma(e[ii], rr, q[a][16 + ii]);
            }
        }
    }
    // block totals of the 3 x kRcQ quantities: DPP row sums (nine groups of eight), then the 16 (wave, row)
    // partials of every quantity are added by one thread each
    {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        constexpr int NQ = 3 * kRcQ, NG = (NQ + 7) / 8;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            double q8[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { const int f = 8 * g + i; q8[i] = (f < NQ) ? q[f / kRcQ][f % kRcQ] : 0.0; }
            double r0, r1;
            row_sum8(q8, r0, r1);
            if ((lane & 15) < 4) {
                double *dst = red + (wv * 4 + (lane >> 4)) * (8 * NG) + 8 * g + 4 * (lane & 1) + (lane & 2);
                dst[0] = r0; dst[1] = r1;
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < NQ) {
            double sm = 0.0;
            for (int r = 0; r < 16; ++r) sm += red[r * (8 * NG) + threadIdx.x];
            part[(size_t)threadIdx.x * NBr + blockIdx.x] = sm;
        }
    }
}

// Coefficients of the Galerkin projection for ONE axis from its kRcQ sums S (G_ij = E_i.R_j at S[i*kRc+j],
// g_i = E_i.r0 at S[16+i]): fully unrolled 4x4 masked Cholesky (static indices only: everything stays in
// registers).  A direction is dropped (zero coefficient) when it is absent, numerically null, or dependent on the
// accepted ones; `skip` (r0 already meets the tolerance) zeroes everything.
template <typename SumPtr>
__device__ __forceinline__ void rc_cholesky(SumPtr S, int cnt, bool skip, double *c) {
    double G[kRc][kRc], g[kRc], L[kRc][kRc], y[kRc];
    bool ok[kRc];
    double gmax = 0.0;
#pragma unroll
    for (int i = 0; i < kRc; ++i) {
        g[i] = S[16 + i];
#pragma unroll
        for (int j = 0; j < kRc; ++j) { G[i][j] = 0.5 * (S[i * kRc + j] + S[j * kRc + i]); L[i][j] = 0.0; }
        gmax = fmax(gmax, (i < cnt) ? G[i][i] : 0.0);
    }
#pragma unroll
    for (int i = 0; i < kRc; ++i) {
        double d = G[i][i];
#pragma unroll
        for (int k2 = 0; k2 < i; ++k2) d -= L[i][k2] * L[i][k2];
        ok[i] = !skip && (i < cnt) && (G[i][i] > 1e-12 * gmax) && (d > 1e-10 * G[i][i]) && (d > 0.0);
        const double ipiv = ok[i] ? fast_rsqrt(d) : 1.0;   // hardware seed + 2 Newton steps (IEEE sqrt / div sequences are slow
        const double piv = ok[i] ? d * ipiv : 1.0;         // on one lane, and this runs on the critical path of every solve)
        L[i][i] = piv;
#pragma unroll
        for (int k2 = 0; k2 < i; ++k2) L[i][k2] = ok[i] ? L[i][k2] : 0.0;
#pragma unroll
        for (int j = i + 1; j < kRc; ++j) {
            double v = G[j][i];
#pragma unroll
            for (int k2 = 0; k2 < i; ++k2) v -= L[j][k2] * L[i][k2];
            L[j][i] = ok[i] ? v * ipiv : 0.0;
        }
    }
#pragma unroll
    for (int i = 0; i < kRc; ++i) {
        double v = g[i];
#pragma unroll
        for (int k2 = 0; k2 < i; ++k2) v -= L[i][k2] * y[k2];
        y[i] = ok[i] ? v * fast_rcp(L[i][i]) : 0.0;
    }
#pragma unroll
    for (int i = kRc - 1; i >= 0; --i) {
        double v = y[i];
#pragma unroll
        for (int k2 = i + 1; k2 < kRc; ++k2) v -= L[k2][i] * c[k2];
        c[i] = ok[i] ? v * fast_rcp(L[i][i]) : 0.0;
        if (!(c[i] == c[i])) c[i] = 0.0;
    }
}

// one block: finish the sums, solve the three cnt x cnt systems (symmetrised Cholesky that SKIPS numerically
// null or dependent directions).  If r0 already meets pcg_tol on every axis the coefficients are exactly
// zero: in a stationary state the stored pairs are round-off and must not perturb the iterate.
__global__ __launch_bounds__(1024) void k_rc_solve(int cnt, const double *__restrict__ part, int NBr, double tol2, double *__restrict__ coef) {
    __shared__ double sums[3 * kRcQ];
    __shared__ int skip;
    const int t = threadIdx.x;
    {   // each wave reduces a strided subset of the 3 * kRcQ partial-sum rows
        const int lane = t & 63, wv = t >> 6, nwv = (int)blockDim.x >> 6;
        for (int qi = wv; qi < 3 * kRcQ; qi += nwv) {
            double s = 0.0;
            for (int i = lane; i < NBr; i += 64) s += part[(size_t)qi * NBr + i];
            s = wave_sum(s);
            if (lane == 0) sums[qi] = s;
        }
    }
    __syncthreads();
    if (t == 0) {
        bool conv = true;
        for (int a = 0; a < 3; ++a) conv = conv && (sums[a * kRcQ + 20] <= tol2 * sums[a * kRcQ + 21] + 1e-300);
        skip = conv ? 1 : 0;
    }
    __syncthreads();
    if (t < 3) {
        double c[kRc];
        rc_cholesky(sums + kRcQ * t, cnt, skip != 0, c);
#pragma unroll
        for (int i = 0; i < kRc; ++i) coef[t * kRc + i] = c[i];
    }
}

// x += sum_j c_j E_j
__global__ __launch_bounds__(256) void k_rc_apply(int n3, RcBasis B, const double *__restrict__ coef, double *__restrict__ x) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n3) return;
    const int a = i % 3;
    double acc = x[i];
#pragma unroll
    for (int j = 0; j < kRc; ++j)
        if (j < B.cnt) acc = fma(coef[a * kRc + j], B.E[j][i], acc);
    x[i] = acc;
}

// ---- END PROJECTION ON SOFT MODES (round 5; admm_hip_set_soft_modes) ---------------------------------------------------------------
// A residual-norm stop leaves the error of a PCG solve in the soft modes of the body (error = residual / eigenvalue), every solve of a
// frame lags the same way and the lag accumulates in the velocity: the drift that set the bench tolerance (profiles/r04_drift_*).  With k
// smooth global vectors Z (the lowest eigenvectors of K = M + Ahat, one scalar field for the three axes) the FINAL iterate of a solve is
// corrected by the exact Galerkin step  x += Z (Z^T K Z)^-1 Z^T (b - A x)  (exact pairs (Z, K Z): exact whatever the accuracy of Z), which
// removes the error in span(Z) A-orthogonally.  CPU prototype (experiments/end_deflation_proto.py, 52 k-tet twin, 25 frames): 32 modes cut
// the position error 10-40x at every tolerance; a projection at the START of a solve does not (round 4).
// k_defl_dots: r = b - A x and the partial sums of Z^T r per block; k_defl_solve: one block reduces them and applies (Z^T K Z)^-1;
// k_defl_apply: x += Z y.
constexpr int kDeflMax = 64;
__global__ __launch_bounds__(256) void k_defl_dots(SellA A, const double *__restrict__ m, const double *__restrict__ b, const double *__restrict__ x,
                                                   int k, const double *__restrict__ Z, int nv, double *__restrict__ part, int NB) {
    __shared__ double lds[4][3 * kDeflMax];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int s = wave_slice();
    double r[3] = {0.0, 0.0, 0.0};
    int row = -1;
    if (s < A.n_slices) {
        row = s * 64 + lane;
        double acc[3];
        sell_row(A, s, lane, x, acc);
        if (row < A.n_rows) {
#pragma unroll
            for (int j = 0; j < 3; ++j) { const size_t i = 3 * (size_t)row + j; r[j] = b[i] - fma(m[i], x[i], acc[j]); }
        } else row = -1;
    }
    // eight modes' loads in flight, ONE wave sum per mode and axis, the four waves' sums added once at the end (a block-wide sum per mode
    // -- 24 block barriers -- cost 101 us at 2 M tets: rocprofv3, profiles/r05_a_kernel_stats_blob2m_launch_path.csv)
    for (int q0 = 0; q0 < k; q0 += 8) {
        double z[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) z[i] = (row >= 0 && q0 + i < k) ? Z[(size_t)(q0 + i) * nv + row] : 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (q0 + i < k) {
                const double t0 = wave_sum(z[i] * r[0]), t1 = wave_sum(z[i] * r[1]), t2 = wave_sum(z[i] * r[2]);
                if (lane == 0) { lds[wv][3 * (q0 + i)] = t0; lds[wv][3 * (q0 + i) + 1] = t1; lds[wv][3 * (q0 + i) + 2] = t2; }
            }
        }
    }
    __syncthreads();
    for (int o = threadIdx.x; o < 3 * k; o += 256) part[(size_t)o * NB + blockIdx.x] = (lds[0][o] + lds[1][o]) + (lds[2][o] + lds[3][o]);
}
// The same partial sums from a residual the caller ALREADY HAS (round 6): after a launch-path solve the PCG's own final residual is in memory as
// u = D^-1 r (k_big_scatter; what k_rc_record builds the recycled pair from), so the end projection needs no matrix-vector product of its own --
// at 2 M tets k_defl_dots is 79 us, 45 of them the product (profiles/r06_launch_path_fixed_costs.txt).  Four vertices per thread, the loads
// of eight modes x four vertices in flight, one wave sum per mode and axis; a quarter of the partials for k_defl_solve to add up.
constexpr int kDeflRV = 4;
__global__ __launch_bounds__(256) void k_defl_dots_r(int nv, const double *__restrict__ u, const double *__restrict__ dinv, int k, const double *__restrict__ Z,
                                                     double *__restrict__ part, int NBd) {
    __shared__ double lds[4][3 * kDeflMax];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int v0 = blockIdx.x * (256 * kDeflRV) + threadIdx.x;
    double r[kDeflRV][3];
#pragma unroll
    for (int i = 0; i < kDeflRV; ++i) {
        const int v = v0 + 256 * i;
#pragma unroll
        for (int j = 0; j < 3; ++j) r[i][j] = v < nv ? u[3 * (size_t)v + j] / dinv[3 * (size_t)v + j] : 0.0;
    }
    for (int q0 = 0; q0 < k; q0 += 8) {
        double z[kDeflRV][8];
#pragma unroll
        for (int i = 0; i < kDeflRV; ++i) {
            const int v = v0 + 256 * i;
#pragma unroll
            for (int m = 0; m < 8; ++m) z[i][m] = (v < nv && q0 + m < k) ? Z[(size_t)(q0 + m) * nv + v] : 0.0;
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (q0 + m < k) {
                double t[3] = {0.0, 0.0, 0.0};
#pragma unroll
                for (int i = 0; i < kDeflRV; ++i) { t[0] = fma(z[i][m], r[i][0], t[0]); t[1] = fma(z[i][m], r[i][1], t[1]); t[2] = fma(z[i][m], r[i][2], t[2]); }
                const double t0 = wave_sum(t[0]), t1 = wave_sum(t[1]), t2 = wave_sum(t[2]);
                if (lane == 0) { lds[wv][3 * (q0 + m)] = t0; lds[wv][3 * (q0 + m) + 1] = t1; lds[wv][3 * (q0 + m) + 2] = t2; }
            }
        }
    }
    __syncthreads();
    for (int o = threadIdx.x; o < 3 * k; o += 256) part[(size_t)o * NBd + blockIdx.x] = (lds[0][o] + lds[1][o]) + (lds[2][o] + lds[3][o]);
}
// one block of 1024 threads: wave w adds up the quantities w, w + 16, ... over the blocks (lanes stride the blocks: fixed order), then y = G^-1 d
// (round 6: eight partials of a lane in flight at a time -- as a plain loop every load waited for the one before it: 22 round trips per
// quantity at 2 M tets, 35 us for a 24 x 24 solve; the order of the additions is unchanged)
__global__ __launch_bounds__(1024) void k_defl_solve(int k, const double *__restrict__ part, int NB, const double *__restrict__ Ginv, double *__restrict__ y) {
    __shared__ double d[3 * kDeflMax];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (int)(blockDim.x >> 6);
    for (int q = wv; q < 3 * k; q += nw) {
        double t = 0.0;
        for (int i0 = lane; i0 < NB; i0 += 64 * 8) {
            double v[8];
#pragma unroll
            for (int w = 0; w < 8; ++w) { const int i = i0 + 64 * w; v[w] = i < NB ? part[(size_t)q * NB + i] : 0.0; }
#pragma unroll
            for (int w = 0; w < 8; ++w) { const int i = i0 + 64 * w; if (i < NB) t += v[w]; }
        }
        t = wave_sum(t);
        if (lane == 0) d[q] = t;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < 3 * k; o += (int)blockDim.x) {      // y[q][axis] = sum_p Ginv[q][p] d[p][axis]
        const int q = o / 3, ax = o % 3;
        double acc = 0.0;
        for (int pp = 0; pp < k; ++pp) acc = fma(Ginv[q * k + pp], d[3 * pp + ax], acc);
        y[o] = acc;
    }
}
__global__ __launch_bounds__(256) void k_defl_apply(int nv, int k, const double *__restrict__ Z, const double *__restrict__ y, double *__restrict__ x) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= nv) return;
    double acc[3] = {0.0, 0.0, 0.0};
    for (int q = 0; q < k; ++q) {
        const double z = Z[(size_t)q * nv + v];
        acc[0] = fma(z, y[3 * q], acc[0]); acc[1] = fma(z, y[3 * q + 1], acc[1]); acc[2] = fma(z, y[3 * q + 2], acc[2]);
    }
    x[3 * (size_t)v] += acc[0]; x[3 * (size_t)v + 1] += acc[1]; x[3 * (size_t)v + 2] += acc[2];
}

// after the solve: store the pair (e = x - xs, A e) in a ring slot.  A e = r0 - r_final exactly, and the PCG
// carries r_final = u / dinv, so the pair is exact (up to round-off) even though the solve stopped at pcg_tol.
__global__ __launch_bounds__(256) void k_rc_record(int n3, const double *__restrict__ x, const double *__restrict__ xs,
                                                   const double *__restrict__ r0, const double *__restrict__ u,
                                                   const double *__restrict__ dinv, double *__restrict__ Eslot,
                                                   double *__restrict__ Rslot) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n3) return;
    Eslot[i] = x[i] - xs[i];
    Rslot[i] = r0[i] - u[i] * fast_rcp(dinv[i]);
}

// ---------------------------------------------------------------------------------------------------
// GLOBAL STEP (ADMM_LS_NCMCGS): nodal multi-colour SOR, src/NodalMultiColorGS.hpp:60-146,180-262.
// PassiveCollision::signed_distance of obstacle j at x (src/Collider.hpp:66-83): updates the payload (best, n, p) when its distance
// is not above the current one, like every implementation in src/PassiveObject.hpp does.  Kinds: 0 Floor (:32-45), 1 Sphere (:48-64);
// user-side PassiveCollision subclasses: 2 a plane n.x = d, 3 any object sampled on a grid at Solver::initialize
// (admm_host_sample_obstacle: distance and normal per node, trilinear on the device, contact point = x - dx n; no hit outside the grid).
template <class OB>   // Obstacles in any address space (k_gs_persist keeps its copy in LDS)
__device__ __forceinline__ void obstacle_payload(const OB &ob, int j, const double *x, double &best, double *n, double *p) {
    const int kind = ob.kind[j];
    if (kind == 0) {
        const double dx = x[1] - ob.par[j][0];
        if (!(dx > best)) { best = dx; p[0] = x[0]; p[1] = ob.par[j][0]; p[2] = x[2]; n[0] = 0.0; n[1] = 1.0; n[2] = 0.0; }
    } else if (kind == 1) {
        double d[3] = {x[0] - ob.par[j][0], x[1] - ob.par[j][1], x[2] - ob.par[j][2]};
        const double l = sqrt(dot3(d, d)), dx = l - ob.par[j][3];
        if (!(dx > best)) {
            best = dx;
            const double il = 1.0 / l;
#pragma unroll
            for (int c = 0; c < 3; ++c) { d[c] *= il; p[c] = ob.par[j][c] + d[c] * ob.par[j][3]; n[c] = d[c]; }
        }
    } else if (kind == 2) {
        const double dx = fma(ob.par[j][2], x[2], fma(ob.par[j][1], x[1], ob.par[j][0] * x[0])) - ob.par[j][3];
        if (!(dx > best)) {
            best = dx;
#pragma unroll
            for (int c = 0; c < 3; ++c) { n[c] = ob.par[j][c]; p[c] = fma(-dx, ob.par[j][c], x[c]); }
        }
    } else {
        const double *m = ob.gmeta + 10 * (int)ob.par[j][0];
        double u[3]; int i0[3]; bool inside = true;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            u[c] = (x[c] - m[c]) / m[3 + c];
            inside = inside && u[c] >= 0.0 && u[c] <= m[6 + c] - 1.0;
            i0[c] = min(max((int)floor(u[c]), 0), (int)m[6 + c] - 2);
            u[c] -= (double)i0[c];
        }
        if (!inside) return;
        const int nx = (int)m[6], ny = (int)m[7];
        const double *g = ob.gdata + 4 * ((size_t)m[9] + (size_t)i0[0] + (size_t)nx * ((size_t)i0[1] + (size_t)ny * (size_t)i0[2]));
        double q[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int corner = 0; corner < 8; ++corner) {
            const int a = corner & 1, b = (corner >> 1) & 1, c2 = corner >> 2;
            const double w = (a ? u[0] : 1.0 - u[0]) * (b ? u[1] : 1.0 - u[1]) * (c2 ? u[2] : 1.0 - u[2]);
            const double *gc = g + 4 * ((size_t)a + (size_t)nx * ((size_t)b + (size_t)ny * (size_t)c2));
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = fma(w, gc[k], q[k]);
        }
        const double dx = q[0];
        if (!(dx > best)) {
            best = dx;
            const double il = 1.0 / sqrt(fma(q[3], q[3], fma(q[2], q[2], q[1] * q[1])));
#pragma unroll
            for (int c = 0; c < 3; ++c) { n[c] = q[1 + c] * il; p[c] = fma(-dx, n[c], x[c]); }
        }
    }
}
// (a container that holds exactly ONE obstacle says so -- `static constexpr bool kSingle = true` -- and is read with a constant index: its count, kind
// and parameters then stay in (scalar) registers; indexed by the loop variable they would live in scratch memory)
template <class OB, class = void> struct ObstSingle { static constexpr bool value = false; };
template <class OB> struct ObstSingle<OB, decltype((void)OB::kSingle)> { static constexpr bool value = OB::kSingle; };
template <class OB>
__device__ __forceinline__ bool passive_hit(const OB &ob, const double *x, double *n, double *p) {
    double best = 1.7976931348623157e308; // Payload ctor (src/Collider.hpp:73)
    if constexpr (ObstSingle<OB>::value) {
        obstacle_payload(ob, 0, x, best, n, p);
        return best < 0.0;
    }
    for (int j = 0; j < ob.n; ++j) {
        obstacle_payload(ob, j, x, best, n, p);
        if (best < 0.0) return true; // src/Collider.hpp:143-148: first object with dx < 0 wins
    }
    return false;
}

// One node of one colour of a sweep: over-relaxed Jacobi value (:210), or -- when the relaxed point would be inside a passive
// obstacle -- the constrained segment update of :218-262 (projection of the UNRELAXED value onto the tangent plane at the contact
// point, no over-relaxation).  One definition with every product-sum written as an explicit fma, shared by all sweep kernels
// (k_gs_color, k_gs_color2, k_gs_colorN, k_gs_persist): their results are bit-identical by construction, not by the compiler's
// contraction choices.
// Returns whether the row was projected onto an obstacle (counted: admm_hip_contact_totals).
template <class OB>
__device__ __forceinline__ bool gs_relax(const OB &ob, double omega, const double *bi, const double *LUx, const double *inv_aii,
                                         const double *cx, double *nx) {
    // inv_aii = 1 / a_ii, formed ONCE per row (IEEE division) by every caller -- the persistent kernel keeps it in LDS next to a_ii: three
    // division sequences less on the dependent chain of a row update.  (The reference divides, :210; one rounding of difference per update.)
    double jac[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        jac[q] = (bi[q] - LUx[q]) * inv_aii[q];
        nx[q] = fma(omega, jac[q], (1.0 - omega) * cx[q]); // :210
    }
    double n[3], p[3];
    if (ob.n > 0 && passive_hit(ob, nx, n, p)) { // constrained_segment_update :218-262
        const double dx[3] = {jac[0] - p[0], jac[1] - p[1], jac[2] - p[2]};
        double nn[3] = {0.0, 0.0, 0.0}, uu[3], vv[3];
        if (n[0] > 0.999) nn[2] = 1.0; else nn[0] = 1.0; // orthoG :171-177
        uu[0] = fma(nn[1], n[2], -(nn[2] * n[1])); uu[1] = fma(nn[2], n[0], -(nn[0] * n[2])); uu[2] = fma(nn[0], n[1], -(nn[1] * n[0]));
        double il = 1.0 / sqrt(fma(uu[2], uu[2], fma(uu[1], uu[1], uu[0] * uu[0])));
#pragma unroll
        for (int q = 0; q < 3; ++q) uu[q] *= il;
        vv[0] = fma(n[1], uu[2], -(n[2] * uu[1])); vv[1] = fma(n[2], uu[0], -(n[0] * uu[2])); vv[2] = fma(n[0], uu[1], -(n[1] * uu[0]));
        il = 1.0 / sqrt(fma(vv[2], vv[2], fma(vv[1], vv[1], vv[0] * vv[0])));
#pragma unroll
        for (int q = 0; q < 3; ++q) vv[q] *= il;
        const double t0 = fma(uu[2], dx[2], fma(uu[1], dx[1], uu[0] * dx[0])), t1 = fma(vv[2], dx[2], fma(vv[1], dx[1], vv[0] * dx[0]));
#pragma unroll
        for (int q = 0; q < 3; ++q) nx[q] = fma(uu[q], t0, fma(vv[q], t1, p[q]));
        return true;
    }
    return false;
}

// A pinned node of a sweep (:111-117): its pin's position.  flag 2 = a SLIDE pin (normal-only constraint n . (x - p) = 0, README.md:23-28
// TODO of the reference): the plane-constrained Jacobi value of :218-262 on the pin's own plane -- the unrelaxed value
// D^-1 (b - LUx) projected onto it, G G^T (jac - p) + p = jac - n (n . (jac - p)).
__device__ __forceinline__ void gs_pin_value(int flag, const double *pin, const double *nrm, const double *bi, const double *LUx, const double *inv_aii, double *nx) {
    if (flag != 2) { nx[0] = pin[0]; nx[1] = pin[1]; nx[2] = pin[2]; return; }
    double jac[3], d = 0.0;
#pragma unroll
    for (int q = 0; q < 3; ++q) { jac[q] = (bi[q] - LUx[q]) * inv_aii[q]; d = fma(nrm[q], jac[q] - pin[q], d); }
#pragma unroll
    for (int q = 0; q < 3; ++q) nx[q] = fma(-d, nrm[q], jac[q]);
}

struct GsArgs {
    SellA S;                    // colour-ordered SELL of the off-diagonal non-zeros (host_setup.hpp: GsSell)
    const int *slot_node;       // node of every SELL lane, -1 = padding
    const double *diag;         // Ahat(v,v) per lane
    const double *m;            // [3 nv]
    const double *b; double *x;
    const int *pin_flag; const double *pin_xyz; // per node (nullptr -> no pins); flag 2 = slide pin (its unit normal in pin_nrm)
    const double *pin_nrm;
    double omega;
    int *done;                  // set once the residual test passed
    // residual test of the PREVIOUS sweep, decided here by every block of the first colour kernel of a sweep
    // (deterministic re-reduction of the k_gs_resid partials; saves one launch per sweep)
    const double *part; int NBp; double tol2; int *sweeps; int *total;
    const unsigned char *skip;  // nodes whose rows are not rows of A in this solve (touched by dynamic hits: they are
                                // swept by k_gs_touched); nullptr when there are none
    unsigned long long *proj;   // rows projected onto a passive obstacle since create (admm_hip_contact_totals); may be nullptr
};

// one colour of one sweep: wave = one 64-node slice of that colour, lane = node.  The off-diagonal row sum is
// the same software-pipelined SELL loop as the SpMV (exact zeros are not stored: the reference skips them at
// run time, NodalMultiColorGS.hpp:194; the summation order is the row's column order, like the reference).
__global__ __launch_bounds__(256) void k_gs_color(GsArgs a, int slice0, int nslices, Obstacles ob, int decide) {
    __shared__ double lds[8];
    const int done_flag = *a.done;   // consulted only before something is written (its load overlaps the row gather)
    if (decide) { // first colour of sweep i+1: was sweep i converged?  (NodalMultiColorGS.hpp:136-140)
        if (done_flag) return;
        double q[2] = {0.0, 0.0};
        for (int i = threadIdx.x; i < a.NBp; i += 256) { q[0] += a.part[i]; q[1] += a.part[a.NBp + i]; }
        block_sum<2>(q, lds);
        const bool conv = decide == 2 && (q[0] / q[1] < a.tol2);   // decide == 1: count the sweep only (tol <= 0)
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            if (conv) *a.done = 1; else { atomicAdd(a.sweeps, 1); atomicAdd(a.total, 1); }
        }
        if (conv) return;
    }
    const int lane = threadIdx.x & 63;
    const int ws = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (ws >= nslices) return;
    const int s = slice0 + ws;
    const int v = a.slot_node[(size_t)64 * s + lane];
    double LUx[3];
    sell_row(a.S, s, lane, a.x, LUx);
    if (v < 0 || done_flag) return;
    if (a.skip && a.skip[v]) return;
    const int pflag = a.pin_flag ? a.pin_flag[v] : 0;
    if (pflag == 1) { // :111-117
#pragma unroll
        for (int q = 0; q < 3; ++q) a.x[3 * (size_t)v + q] = a.pin_xyz[3 * (size_t)v + q];
        return;
    }
    const double ad = a.diag[(size_t)64 * s + lane];
    double aii[3], cx[3], bi[3], nx[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) { aii[q] = ad + a.m[3 * (size_t)v + q]; cx[q] = a.x[3 * (size_t)v + q]; bi[q] = a.b[3 * (size_t)v + q]; }
    const double iaii[3] = {1.0 / aii[0], 1.0 / aii[1], 1.0 / aii[2]};
    if (pflag == 2) {     // slide pin
        gs_pin_value(2, a.pin_xyz + 3 * (size_t)v, a.pin_nrm + 3 * (size_t)v, bi, LUx, iaii, nx);
#pragma unroll
        for (int q = 0; q < 3; ++q) a.x[3 * (size_t)v + q] = nx[q];
        return;
    }
    if (gs_relax(ob, a.omega, bi, LUx, iaii, cx, nx) && a.proj) atomicAdd(a.proj, 1ull);
#pragma unroll
    for (int q = 0; q < 3; ++q) a.x[3 * (size_t)v + q] = nx[q];
}

// TWO-COLOUR meshes (bipartite coupling graph, e.g. Kuhn / make_tet_blocks meshes): the per-sweep residual test
// (:136-140) needs no SpMV pass of its own.  After sweep k the rows of the LAST colour see only final neighbour
// values, so their residuals are computed in their own colour kernel right after the update (POST); the rows of the
// FIRST colour are untouched until sweep k+1 starts, so the first-colour kernel of sweep k+1 computes their
// sweep-k residuals before it updates them (PRE).  The test of sweep k is then settled by the LAST-colour kernel
// of sweep k+1 (every block re-reduces both partial arrays, deterministic); if the sweep had converged, the
// first-colour update of sweep k+1 -- done speculatively -- is rolled back from the backup `xb` and the solve
// stops exactly where the reference stops.  Two kernels per sweep instead of three.
struct Gs2Args {
    GsArgs g;
    double *xb;          // [3 nv] backup of the first colour's values (roll-back)
    double *partA;       // [2 (sweep parity)][2][nbA] last-colour partials  (|r|^2, |b|^2)
    double *partB;       // [2][nbB]                   first-colour partials
    int nbA, nbB;
    int s0_first, ns_first;   // slices of the first colour (for the roll-back)
};

template <bool PRE, bool UPDATE, bool POST>
__global__ __launch_bounds__(256) void k_gs_color2(Gs2Args a2, int slice0, int nslices, Obstacles ob, int decide, int parity) {
    const GsArgs &a = a2.g;
    __shared__ double lds[8];
    // `done` is only consulted where something would be written: its load overlaps the row gather instead of
    // standing in front of it (these kernels are a few microseconds of pure latency)
    const int done_flag = *a.done;
    const int lane = threadIdx.x & 63;
    if (decide) {   // last colour of sweep k+1 (decide = k+1, the stamp of this launch): was sweep k converged?
        // `done` set by an EARLIER launch: nothing to do.  Block 0 of THIS launch raises it with this launch's stamp, and a block that
        // is scheduled after that store must still do its grid-strided share of the roll-back below -- it re-derives the verdict from
        // the partials like every other block.
        if (done_flag && done_flag != decide) return;
        double q[2] = {0.0, 0.0};
        const double *pA = a2.partA + (size_t)(parity ^ 1) * 2 * a2.nbA;
        for (int i = threadIdx.x; i < a2.nbA; i += 256) { q[0] += pA[i]; q[1] += pA[a2.nbA + i]; }
        for (int i = threadIdx.x; i < a2.nbB; i += 256) { q[0] += a2.partB[i]; q[1] += a2.partB[a2.nbB + i]; }
        block_sum<2>(q, lds);
        const bool conv = q[0] / q[1] < a.tol2;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            if (conv) *a.done = decide; else { atomicAdd(a.sweeps, 1); atomicAdd(a.total, 1); }
        }
        if (conv) {   // undo the speculative first-colour update of this sweep
            for (int ws = (int)(blockIdx.x * 4 + (threadIdx.x >> 6)); ws < a2.ns_first; ws += (int)gridDim.x * 4) {
                const int v = a.slot_node[(size_t)64 * (a2.s0_first + ws) + lane];
                if (v >= 0) {
#pragma unroll
                    for (int q3 = 0; q3 < 3; ++q3) a.x[3 * (size_t)v + q3] = a2.xb[3 * (size_t)v + q3];
                }
            }
            return;
        }
    }
    const int ws = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    double rs[2] = {0.0, 0.0};
    if (ws < nslices) {
        const int s = slice0 + ws;
        const int v = a.slot_node[(size_t)64 * s + lane];
        double LUx[3];
        sell_row(a.S, s, lane, a.x, LUx);
        if (v >= 0) {
            const double ad = a.diag[(size_t)64 * s + lane];
            const int pflag = a.pin_flag ? a.pin_flag[v] : 0;
            const bool pinned = pflag != 0;
            double aii[3], cx[3], bi[3], nx[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                aii[q] = ad + a.m[3 * (size_t)v + q];
                cx[q] = a.x[3 * (size_t)v + q];
                bi[q] = a.b[3 * (size_t)v + q];
                nx[q] = cx[q];
            }
            if (PRE) {
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const double r = bi[q] - fma(aii[q], cx[q], LUx[q]);
                    rs[0] = fma(r, r, rs[0]); rs[1] = fma(bi[q], bi[q], rs[1]);
                    if (UPDATE && !done_flag) a2.xb[3 * (size_t)v + q] = cx[q];
                }
            }
            if (UPDATE && !done_flag) {
                const double iaii[3] = {1.0 / aii[0], 1.0 / aii[1], 1.0 / aii[2]};
                if (pinned) { // :111-117 (flag 2: slide pin)
                    gs_pin_value(pflag, a.pin_xyz + 3 * (size_t)v, pflag == 2 ? a.pin_nrm + 3 * (size_t)v : a.pin_xyz, bi, LUx, iaii, nx);
                } else if (gs_relax(ob, a.omega, bi, LUx, iaii, cx, nx) && a.proj) atomicAdd(a.proj, 1ull);
#pragma unroll
                for (int q = 0; q < 3; ++q) a.x[3 * (size_t)v + q] = nx[q];
            }
            if (POST) {
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const double r = bi[q] - fma(aii[q], nx[q], LUx[q]);
                    rs[0] = fma(r, r, rs[0]); rs[1] = fma(bi[q], bi[q], rs[1]);
                }
            }
        }
    }
    if (PRE || POST) {
        block_sum<2>(rs, lds);
        if (threadIdx.x == 0 && !done_flag) {
            if (POST) { double *pA = a2.partA + (size_t)parity * 2 * a2.nbA; pA[blockIdx.x] = rs[0]; pA[a2.nbA + blockIdx.x] = rs[1]; }
            else { a2.partB[blockIdx.x] = rs[0]; a2.partB[a2.nbB + blockIdx.x] = rs[1]; }
        }
    }
}

// settles the LAST sweep of the two-colour scheme (its first-colour residuals come from a PRE-only pass)
__global__ __launch_bounds__(256) void k_gs_check2(Gs2Args a2, int parity) {
    __shared__ double lds[8];
    const GsArgs &a = a2.g;
    if (*a.done) return;
    double q[2] = {0.0, 0.0};
    const double *pA = a2.partA + (size_t)parity * 2 * a2.nbA;
    for (int i = threadIdx.x; i < a2.nbA; i += 256) { q[0] += pA[i]; q[1] += pA[a2.nbA + i]; }
    for (int i = threadIdx.x; i < a2.nbB; i += 256) { q[0] += a2.partB[i]; q[1] += a2.partB[a2.nbB + i]; }
    block_sum<2>(q, lds);
    if (threadIdx.x == 0) {
        if (q[0] / q[1] < a.tol2) *a.done = 1; else { atomicAdd(a.sweeps, 1); atomicAdd(a.total, 1); }
    }
}

// THREE AND MORE COLOURS: the same idea as k_gs_color2 -- no residual SpMV of its own, one launch per colour and sweep.  After sweep
// k the rows of the LAST colour see only final neighbour values: their residuals are taken right after their update (POST, as in
// the two-colour scheme).  The rows of every EARLIER colour c are untouched until their kernel of sweep k+1, which takes their
// sweep-k residuals before it updates them -- but by then the rows of the colours before c have already moved on to sweep k+1, so
// every such kernel keeps the old value of its rows in `xb` before overwriting it, and colour c sums its residual rows with xb for
// neighbours of a colour < c and x for the others (sell_row_mixed: one more pass over the row; the first colour needs none).  The test
// of sweep k is settled by the LAST-colour kernel of sweep k+1; if the sweep had converged, the speculative updates of the earlier
// colours are rolled back from xb and the solve stops exactly where the reference stops (NodalMultiColorGS.hpp:136-140).
struct GsNArgs {
    GsArgs g;
    double *xb;                  // [3 nv] values of the rows of colours 0 .. C-2 before their update of the current sweep
    double *partL;               // [2 (sweep parity)][2][nbL] last-colour partials (|r|^2, |b|^2)
    double *partE;               // [2][nE] partials of the earlier colours' kernels, concatenated (block nbE_off + blockIdx.x)
    int nbL, nE;
    int s0_early, ns_early;      // slices of the colours 0 .. C-2 (contiguous in the colour-ordered SELL: roll-back, final pass)
    const unsigned char *low;    // per SELL entry: the column's colour is below the row's (its value of the previous sweep is in xb)
};
// cur = sum_k Ahat(row,k) x_col and old = sum_k Ahat(row,k) xo_col in ONE pass: xo = xb for the columns flagged `low` (their colour is
// below the row's: already updated in this sweep), x otherwise.  The flags stand next to the entries (same index as col / val), so
// the second sum adds gathers, not a dependent memory stage.
__device__ __forceinline__ void sell_row_both(const SellA &A, const unsigned char *__restrict__ low, int s, int lane, const double *__restrict__ x,
                                              const double *__restrict__ xb, double *cur, double *old) {
    const int w = A.w[s];
    const int *__restrict__ cp = A.col + A.ptr[s] + lane;
    const double *__restrict__ vp = A.val + A.ptr[s] + lane;
    const unsigned char *__restrict__ lp = low + A.ptr[s] + lane;
    cur[0] = cur[1] = cur[2] = 0.0; old[0] = old[1] = old[2] = 0.0;
    for (int k = 0; k < w; k += 4) {
        int col[4]; double a[4]; unsigned char lo[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { col[i] = cp[64 * (k + i)]; a[i] = vp[64 * (k + i)]; lo[i] = lp[64 * (k + i)]; }
        double g[12], h[12];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double *p = x + 3 * (size_t)col[i];
            g[3 * i] = p[0]; g[3 * i + 1] = p[1]; g[3 * i + 2] = p[2];
            h[3 * i] = g[3 * i]; h[3 * i + 1] = g[3 * i + 1]; h[3 * i + 2] = g[3 * i + 2];
            if (lo[i]) { const double *q = xb + 3 * (size_t)col[i]; h[3 * i] = q[0]; h[3 * i + 1] = q[1]; h[3 * i + 2] = q[2]; }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            cur[0] = fma(a[i], g[3 * i], cur[0]); cur[1] = fma(a[i], g[3 * i + 1], cur[1]); cur[2] = fma(a[i], g[3 * i + 2], cur[2]);
            old[0] = fma(a[i], h[3 * i], old[0]); old[1] = fma(a[i], h[3 * i + 1], old[1]); old[2] = fma(a[i], h[3 * i + 2], old[2]);
        }
    }
}
// ROLE 0: an earlier colour (my_color = 0 .. C-2), ROLE 2: the last colour.  RESID (ROLE 0): there is a previous sweep whose residual
// this kernel contributes to.  UPDATE = false: the residual pass that settles the LAST sweep (all earlier colours in one launch).
template <int ROLE, bool RESID, bool UPDATE>
__global__ __launch_bounds__(256) void k_gs_colorN(GsNArgs aN, int slice0, int nslices, Obstacles ob, int decide, int parity, int part_off,
                                                   int my_color) {
    const GsArgs &a = aN.g;
    __shared__ double lds[8];
    const int done_flag = *a.done;
    const int lane = threadIdx.x & 63;
    if (ROLE == 2 && decide) {   // last colour of sweep k+1 (decide = k+1, the stamp of this launch): was sweep k converged?
        if (done_flag && done_flag != decide) return;   // raised by an earlier launch (see k_gs_color2: a block must not skip its roll-back share)
        double q[2] = {0.0, 0.0};
        const double *pL = aN.partL + (size_t)(parity ^ 1) * 2 * aN.nbL;
        for (int i = threadIdx.x; i < aN.nbL; i += 256) { q[0] += pL[i]; q[1] += pL[aN.nbL + i]; }
        for (int i = threadIdx.x; i < aN.nE; i += 256) { q[0] += aN.partE[i]; q[1] += aN.partE[aN.nE + i]; }
        block_sum<2>(q, lds);
        const bool conv = q[0] / q[1] < a.tol2;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            if (conv) *a.done = decide; else { atomicAdd(a.sweeps, 1); atomicAdd(a.total, 1); }
        }
        if (conv) {   // undo the speculative updates of the earlier colours of this sweep
            for (int ws = (int)(blockIdx.x * 4 + (threadIdx.x >> 6)); ws < aN.ns_early; ws += (int)gridDim.x * 4) {
                const int v = a.slot_node[(size_t)64 * (aN.s0_early + ws) + lane];
                if (v >= 0) {
#pragma unroll
                    for (int q3 = 0; q3 < 3; ++q3) a.x[3 * (size_t)v + q3] = aN.xb[3 * (size_t)v + q3];
                }
            }
            return;
        }
    }
    const int ws = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    double rs[2] = {0.0, 0.0};
    if (ws < nslices) {
        const int s = slice0 + ws;
        const int v = a.slot_node[(size_t)64 * s + lane];
        double LUx[3], LUo[3];
        if (ROLE == 0 && RESID && UPDATE && my_color > 0) sell_row_both(a.S, aN.low, s, lane, a.x, aN.xb, LUx, LUo);
        else {
            sell_row(a.S, s, lane, a.x, LUx);       // (first colour: nothing has moved yet in this sweep; final pass: every neighbour is final)
            LUo[0] = LUx[0]; LUo[1] = LUx[1]; LUo[2] = LUx[2];
        }
        if (v >= 0) {
            const double ad = a.diag[(size_t)64 * s + lane];
            const int pflag = a.pin_flag ? a.pin_flag[v] : 0;
            const bool pinned = pflag != 0;
            double aii[3], cx[3], bi[3], nx[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                aii[q] = ad + a.m[3 * (size_t)v + q];
                cx[q] = a.x[3 * (size_t)v + q];
                bi[q] = a.b[3 * (size_t)v + q];
                nx[q] = cx[q];
            }
            if (ROLE == 0 && RESID) {
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const double r = bi[q] - fma(aii[q], cx[q], LUo[q]);
                    rs[0] = fma(r, r, rs[0]); rs[1] = fma(bi[q], bi[q], rs[1]);
                }
            }
            if (ROLE == 0 && UPDATE && !done_flag) {
#pragma unroll
                for (int q = 0; q < 3; ++q) aN.xb[3 * (size_t)v + q] = cx[q];
            }
            if (UPDATE && !done_flag) {
                const double iaii[3] = {1.0 / aii[0], 1.0 / aii[1], 1.0 / aii[2]};
                if (pinned) { // :111-117 (flag 2: slide pin)
                    gs_pin_value(pflag, a.pin_xyz + 3 * (size_t)v, pflag == 2 ? a.pin_nrm + 3 * (size_t)v : a.pin_xyz, bi, LUx, iaii, nx);
                } else if (gs_relax(ob, a.omega, bi, LUx, iaii, cx, nx) && a.proj) atomicAdd(a.proj, 1ull);
#pragma unroll
                for (int q = 0; q < 3; ++q) a.x[3 * (size_t)v + q] = nx[q];
            }
            if (ROLE == 2) {
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const double r = bi[q] - fma(aii[q], nx[q], LUx[q]);
                    rs[0] = fma(r, r, rs[0]); rs[1] = fma(bi[q], bi[q], rs[1]);
                }
            }
        }
    }
    if (ROLE == 2 || RESID) {
        block_sum<2>(rs, lds);
        if (threadIdx.x == 0 && !done_flag) {
            if (ROLE == 2) { double *pL = aN.partL + (size_t)parity * 2 * aN.nbL; pL[blockIdx.x] = rs[0]; pL[aN.nbL + blockIdx.x] = rs[1]; }
            else { aN.partE[part_off + blockIdx.x] = rs[0]; aN.partE[aN.nE + part_off + blockIdx.x] = rs[1]; }
        }
    }
}
// settles the LAST sweep of the scheme (the earlier colours' residuals come from one UPDATE = false pass over all of them)
__global__ __launch_bounds__(256) void k_gs_checkN(GsNArgs aN, int parity) {
    __shared__ double lds[8];
    const GsArgs &a = aN.g;
    if (*a.done) return;
    double q[2] = {0.0, 0.0};
    const double *pL = aN.partL + (size_t)parity * 2 * aN.nbL;
    for (int i = threadIdx.x; i < aN.nbL; i += 256) { q[0] += pL[i]; q[1] += pL[aN.nbL + i]; }
    for (int i = threadIdx.x; i < aN.nE; i += 256) { q[0] += aN.partE[i]; q[1] += aN.partE[aN.nE + i]; }
    block_sum<2>(q, lds);
    if (threadIdx.x == 0) {
        if (q[0] / q[1] < a.tol2) *a.done = 1; else { atomicAdd(a.sweeps, 1); atomicAdd(a.total, 1); }
    }
}

// residual test of one sweep (:136-140): partial sums of |b - A x|^2 and |b|^2
__global__ __launch_bounds__(256) void k_gs_resid(SellA A, const double *__restrict__ m, const double *__restrict__ b,
                                                  const double *__restrict__ x, double *__restrict__ part, int NB,
                                                  const int *__restrict__ done, const unsigned char *__restrict__ skip) {
    __shared__ double lds[8];
    const int done_flag = *done;   // consulted only before the partials are written
    const int lane = threadIdx.x & 63;
    const int s = wave_slice();
    double q[2] = {0.0, 0.0};
    if (s < A.n_slices) {
        const int row = s * 64 + lane;
        double acc[3];
        sell_row(A, s, lane, x, acc);
        if (row < A.n_rows) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const size_t i = 3 * (size_t)row + j;
                const double bi = b[i];
                const double ri = bi - fma(m[i], x[i], acc[j]);
                if (!(skip && skip[row])) q[0] = fma(ri, ri, q[0]);   // rows touched by dynamic hits: k_gs_touched_resid
                q[1] = fma(bi, bi, q[1]);
            }
        }
    }
    block_sum<2>(q, lds);
    if (threadIdx.x == 0 && !done_flag) { part[blockIdx.x] = q[0]; part[NB + blockIdx.x] = q[1]; }
}

// one block: finish the reduction, count the sweep like the reference's `iter`, raise `done`
__global__ __launch_bounds__(256) void k_gs_check(const double *__restrict__ part, int NB, double tol2, int *done,
                                                  int *sweeps, int *total, int check) {
    __shared__ double lds[8];
    if (*done) return;
    bool conv = false;
    if (check) {
        double q[2] = {0.0, 0.0};
        for (int i = threadIdx.x; i < NB; i += 256) { q[0] += part[i]; q[1] += part[NB + i]; }
        block_sum<2>(q, lds);
        conv = (q[0] / q[1] < tol2);
    }
    if (threadIdx.x == 0) {
        if (conv) *done = 1; else { atomicAdd(sweeps, 1); atomicAdd(total, 1); }
    }
}

// ---------------------------------------------------------------------------------------------------
// GLOBAL STEP (ADMM_LS_UZAWACG): Schur-complement CG for [A C^T; C 0] (src/UzawaCG.hpp:57-125) with the
// constraint rows of ConstraintSet::make_matrix (src/ConstraintSet.hpp:59-116) for passive hits found by
// Collider::detect (src/Collider.hpp:152-212).  A passive hit constrains ONE vertex (row = ck n^T at the
// vertex, rhs = ck n.p), so C is stored per vertex: cn[nv][3] = ck n (0 when not hit), cc[nv] = ck n.p.
// A dynamic hit (TetMeshCollision, dyn_collide.hpp) is also ONE row per vertex (ConstraintSet.hpp:96-99 keeps at
// most one row per vertex) that additionally touches the three vertices of the hit face: dface[nv][3] (-1 = none),
// dbary[nv][3]; row = cn^T (x_v - sum_j bary_j x_face_j), rhs 0.
// The inner A^-1 applications are the GPU PCG above.
struct UzScal { double denom, alpha, beta, rr; int stop; int iters; int nhits; int pad_; };

} // namespace admm_k
#include "dyn_collide.hpp"   // dynamic rows (TetMeshCollision): a row may also touch the three vertices of a face
namespace admm_k {

// Collider::detect for every vertex against the passive objects; builds the per-vertex constraint rows
// (mask != nullptr: only the vertices of Solver::surface_inds are candidates, Collider.hpp:157,163)
__global__ __launch_bounds__(256) void k_uz_detect(int nv, const double *__restrict__ x, Obstacles ob, double ck,
                                                   double *__restrict__ cn, double *__restrict__ cc, int *__restrict__ nhits,
                                                   const unsigned char *__restrict__ mask) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= nv) return;
    const double xv[3] = {x[3 * (size_t)v], x[3 * (size_t)v + 1], x[3 * (size_t)v + 2]};
    // Collider::detect: every object updates the payload when its distance is lower (no early exit)
    double best = 1.7976931348623157e308, n[3] = {0, 0, 0}, p[3] = {0, 0, 0};
    for (int j = 0; j < ob.n; ++j) obstacle_payload(ob, j, xv, best, n, p);
    const bool hit = best < 0.0 && (mask == nullptr || mask[v]);
#pragma unroll
    for (int c = 0; c < 3; ++c) cn[3 * (size_t)v + c] = hit ? ck * n[c] : 0.0;
    cc[v] = hit ? ck * dot3(n, p) : 0.0;
    if (hit) atomicAdd(nhits, 1);
}

// Look-ahead of the UzawaCG column cache: the vertices that are not in contact yet, would reach a passive object within `ahead`
// seconds at their current speed, and have no column of K^-1 (slot < 0).  Unordered (the host sorts the list).
__global__ __launch_bounds__(256) void k_uz_near(int nv, const double *__restrict__ x, const double *__restrict__ vel, Obstacles ob, double ahead,
                                                 const int *__restrict__ slot, const unsigned char *__restrict__ mask,
                                                 int *__restrict__ list, int *__restrict__ count) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= nv || slot[v] >= 0 || (mask != nullptr && !mask[v])) return;
    const double xv[3] = {x[3 * (size_t)v], x[3 * (size_t)v + 1], x[3 * (size_t)v + 2]};
    double best = 1.7976931348623157e308, n[3] = {0, 0, 0}, p[3] = {0, 0, 0};
    for (int j = 0; j < ob.n; ++j) obstacle_payload(ob, j, xv, best, n, p);
    const double w[3] = {vel[3 * (size_t)v], vel[3 * (size_t)v + 1], vel[3 * (size_t)v + 2]};
    if (best >= 0.0 && best < ahead * sqrt(dot3(w, w))) list[atomicAdd(count, 1)] = v;
}

// out = base - C^T y  (mode 0)   or   out = C^T y (mode 1)
__global__ __launch_bounds__(256) void k_uz_ct(int nv, int mode, const double *__restrict__ base, const double *__restrict__ cn,
                                               const double *__restrict__ y, double *__restrict__ out) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= nv) return;
    const double yv = y[v];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const size_t i = 3 * (size_t)v + c;
        out[i] = (mode == 0) ? base[i] - cn[i] * yv : cn[i] * yv;
    }
}

// r = C x - c ; d = r ; also clears y when the number of hits changed (UzawaCG.hpp:74)
__global__ __launch_bounds__(256) void k_uz_resid(int nv, const double *__restrict__ x, const double *__restrict__ cn,
                                                  const double *__restrict__ cc, double *__restrict__ r, double *__restrict__ d,
                                                  const int *__restrict__ dface, const double *__restrict__ dbary, UzScal *__restrict__ sc) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *sc = UzScal{};      // the scalars of the Schur CG this kernel opens (instead of a memset)
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= nv) return;
    const double rv = cn[3 * (size_t)v] * x[3 * (size_t)v] + cn[3 * (size_t)v + 1] * x[3 * (size_t)v + 1] +
                      cn[3 * (size_t)v + 2] * x[3 * (size_t)v + 2] - cc[v] + dyn_row_faces(v, cn, dface, dbary, x);
    r[v] = rv; d[v] = rv;
}

// q3 = C q2 ; partial sums of d.q3 and d.r
__global__ __launch_bounds__(256) void k_uz_dots(int nv, const double *__restrict__ q2, const double *__restrict__ cn,
                                                 const double *__restrict__ d, const double *__restrict__ r,
                                                 double *__restrict__ q3, double *__restrict__ part, int NBp,
                                                 const int *__restrict__ dface, const double *__restrict__ dbary) {
    __shared__ double lds[8];
    double q[2] = {0.0, 0.0};
    for (int v = blockIdx.x * 256 + threadIdx.x; v < nv; v += gridDim.x * 256) {
        const double t = cn[3 * (size_t)v] * q2[3 * (size_t)v] + cn[3 * (size_t)v + 1] * q2[3 * (size_t)v + 1] +
                         cn[3 * (size_t)v + 2] * q2[3 * (size_t)v + 2] + dyn_row_faces(v, cn, dface, dbary, q2);
        q3[v] = t;
        q[0] = fma(d[v], t, q[0]);
        q[1] = fma(d[v], r[v], q[1]);
    }
    block_sum<2>(q, lds);
    if (threadIdx.x == 0) { part[blockIdx.x] = q[0]; part[NBp + blockIdx.x] = q[1]; }
}

// alpha = d.r / d.q3 (stop when the denominator vanishes, UzawaCG.hpp:103-105)
__global__ __launch_bounds__(256) void k_uz_alpha(const double *__restrict__ part, int NBp, UzScal *sc) {
    __shared__ double lds[8];
    if (sc->stop) return;
    double q[2] = {0.0, 0.0};
    for (int i = threadIdx.x; i < NBp; i += 256) { q[0] += part[i]; q[1] += part[NBp + i]; }
    block_sum<2>(q, lds);
    if (threadIdx.x == 0) {
        sc->denom = q[0];
        if (fabs(q[0]) < 2.2250738585072014e-308) { sc->stop = 1; sc->alpha = 0.0; }
        else sc->alpha = q[1] / q[0];
    }
}

// x -= alpha q2 ; y += alpha d ; r -= alpha q3 ; partial sums of r.r and r.q3
__global__ __launch_bounds__(256) void k_uz_step(int nv, const UzScal *__restrict__ sc, double *__restrict__ x,
                                                 const double *__restrict__ q2, double *__restrict__ y,
                                                 const double *__restrict__ d, double *__restrict__ r,
                                                 const double *__restrict__ q3, double *__restrict__ part, int NBp) {
    __shared__ double lds[8];
    if (sc->stop) return;
    const double al = sc->alpha;
    double q[2] = {0.0, 0.0};
    for (int v = blockIdx.x * 256 + threadIdx.x; v < nv; v += gridDim.x * 256) {
#pragma unroll
        for (int c = 0; c < 3; ++c) x[3 * (size_t)v + c] -= al * q2[3 * (size_t)v + c];
        y[v] += al * d[v];
        const double rv = r[v] - al * q3[v];
        r[v] = rv;
        q[0] = fma(rv, rv, q[0]);
        q[1] = fma(rv, q3[v], q[1]);
    }
    block_sum<2>(q, lds);
    if (threadIdx.x == 0) { part[blockIdx.x] = q[0]; part[NBp + blockIdx.x] = q[1]; }
}

// residual test (UzawaCG.hpp:112-113), beta (UzawaCG.hpp:115-118); counts the iteration like the reference
__global__ __launch_bounds__(256) void k_uz_beta(const double *__restrict__ part, int NBp, double tol2, UzScal *sc) {
    __shared__ double lds[8];
    if (sc->stop) return;
    double q[2] = {0.0, 0.0};
    for (int i = threadIdx.x; i < NBp; i += 256) { q[0] += part[i]; q[1] += part[NBp + i]; }
    block_sum<2>(q, lds);
    if (threadIdx.x == 0) {
        sc->rr = q[0];
        if (q[0] < tol2) { sc->stop = 1; return; }
        sc->beta = q[1] / sc->denom;
        sc->iters += 1;
    }
}

// d = r - beta d
__global__ __launch_bounds__(256) void k_uz_dir(int nv, const UzScal *__restrict__ sc, const double *__restrict__ r, double *__restrict__ d) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= nv || sc->stop) return;
    d[v] = r[v] - sc->beta * d[v];
}


// ---- cached columns of K^-1 for the Schur-complement CG ------------------------------------------------------------
// Every Schur iteration of UzawaCG applies A^-1 to C^T d (src/UzawaCG.hpp:96-97; the reference back-substitutes through its LDLT
// factor).  A = K (x) I3 never changes after initialize (src/Solver.cpp:225-226) and C^T d is non-zero only at the vertices that
// carry a constraint row (and the face vertices of dynamic rows): A^-1 C^T d = sum over those vertices v of (K^-1 e_v) (C^T d)_v.
// The columns K^-1 e_v are solved for once (on-chip PCG, three columns per launch -- one per axis --, tight tolerance) when a
// vertex first becomes active, kept in HBM (nv doubles each: 157 KB at 105 k tets, 1.4 MB at 1 M), and a Schur iteration becomes
// one pass over the active columns (k_uz_cols_apply, HBM-bound: 8 n_active nv bytes) instead of a 50-iteration PCG solve.
// Contacts persist from iteration to iteration and frame to frame, so new columns are rare after the first touch.
__global__ __launch_bounds__(256) void k_uz_act_flags(int nv, const double *__restrict__ cn, const int *__restrict__ dface,
                                                      unsigned char *__restrict__ flag) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= nv) return;
    if (cn[3 * (size_t)v] == 0.0 && cn[3 * (size_t)v + 1] == 0.0 && cn[3 * (size_t)v + 2] == 0.0) return;
    flag[v] = 1;
    if (dface != nullptr && dface[3 * (size_t)v] >= 0)
        for (int j = 0; j < 3; ++j) flag[dface[3 * (size_t)v + j]] = 1;      // (several rows may share a face vertex: same value)
}
// One block: the flagged vertices in ascending order (the order of the sums in k_uz_cols_apply / k_uzc_matvec: deterministic),
// those of them that have no column yet, and the inverse map pos[v] = place of v in the list (-1: not active).
// info[0] = active vertices, info[1] = missing columns.  Four consecutive vertices per thread and pass.
// flag == nullptr (passive rows only): a vertex is active iff its row of C is not zero -- read from cn directly (no flag pass, no memset).
// nhits != nullptr: info[2] = *nhits, and the counter is cleared for the next detect (one read-back instead of two, no memset).
__global__ __launch_bounds__(1024) void k_uz_act_compact(int nv, const unsigned char *__restrict__ flag, const int *__restrict__ slot,
                                                         int *__restrict__ act, int *__restrict__ miss, int *__restrict__ pos, int *__restrict__ info,
                                                         const double *__restrict__ cn = nullptr, int *__restrict__ nhits = nullptr) {
    __shared__ int wsum[2][16], base[2];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < 2) base[tid] = 0;
    __syncthreads();
    for (int v0 = 0; v0 < nv; v0 += 4096) {
        const int vb = v0 + 4 * tid;
        bool a[4], m[4];
        int na = 0, nm = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (flag) a[k] = vb + k < nv && flag[vb + k] != 0;
            else a[k] = vb + k < nv && (cn[3 * (size_t)(vb + k)] != 0.0 || cn[3 * (size_t)(vb + k) + 1] != 0.0 || cn[3 * (size_t)(vb + k) + 2] != 0.0);
            m[k] = a[k] && slot[vb + k] < 0;
            na += a[k] ? 1 : 0; nm += m[k] ? 1 : 0;
        }
        // exclusive prefix over the wave (lanes in vertex order), then over the waves
        int pa = na, pm = nm;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int ta = __shfl_up(pa, o), tm = __shfl_up(pm, o);
            if (lane >= o) { pa += ta; pm += tm; }
        }
        if (lane == 63) { wsum[0][wv] = pa; wsum[1][wv] = pm; }
        __syncthreads();
        int oa = base[0] + pa - na, om = base[1] + pm - nm;
        for (int w = 0; w < wv; ++w) { oa += wsum[0][w]; om += wsum[1][w]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (vb + k < nv) pos[vb + k] = a[k] ? oa : -1;
            if (a[k]) act[oa++] = vb + k;
            if (m[k]) miss[om++] = vb + k;
        }
        __syncthreads();
        if (tid == 0) { int ta = 0, tm = 0; for (int w = 0; w < 16; ++w) { ta += wsum[0][w]; tm += wsum[1][w]; } base[0] += ta; base[1] += tm; }
        __syncthreads();
    }
    if (tid == 0) { info[0] = base[0]; info[1] = base[1]; if (nhits) { info[2] = *nhits; *nhits = 0; } }
}
// The same list from many blocks (scenes with more than 16 384 vertices: one block would walk nv / 4096 passes): block b of
// k_uz_act_count counts the active / missing vertices of its 4096, block b of k_uz_act_scatter adds the counts of the blocks before it
// (fixed order) and places its own -- ascending order as above.  The last block writes info.
__device__ __forceinline__ bool uz_vertex_active(const unsigned char *flag, const double *cn, int v) {
    return flag ? flag[v] != 0 : (cn[3 * (size_t)v] != 0.0 || cn[3 * (size_t)v + 1] != 0.0 || cn[3 * (size_t)v + 2] != 0.0);
}
__global__ __launch_bounds__(1024) void k_uz_act_count(int nv, const unsigned char *__restrict__ flag, const int *__restrict__ slot,
                                                       const double *__restrict__ cn, int *__restrict__ counts /* [2][gridDim.x] */) {
    __shared__ int wsum[2][16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, vb = (int)blockIdx.x * 4096 + 4 * tid;
    int na = 0, nm = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool a = vb + k < nv && uz_vertex_active(flag, cn, vb + k);
        na += a ? 1 : 0; nm += (a && slot[vb + k] < 0) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { na += __shfl_xor(na, o, 64); nm += __shfl_xor(nm, o, 64); }
    if (lane == 0) { wsum[0][wv] = na; wsum[1][wv] = nm; }
    __syncthreads();
    if (tid == 0) {
        int ta = 0, tm = 0;
        for (int w = 0; w < 16; ++w) { ta += wsum[0][w]; tm += wsum[1][w]; }
        counts[blockIdx.x] = ta; counts[gridDim.x + blockIdx.x] = tm;
    }
}
__global__ __launch_bounds__(1024) void k_uz_act_scatter(int nv, const unsigned char *__restrict__ flag, const int *__restrict__ slot,
                                                         const double *__restrict__ cn, const int *__restrict__ counts, int *__restrict__ act,
                                                         int *__restrict__ miss, int *__restrict__ pos, int *__restrict__ info, int *__restrict__ nhits) {
    __shared__ int wsum[2][16], base[2];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, vb = (int)blockIdx.x * 4096 + 4 * tid, nb = (int)gridDim.x;
    {   // counts of the blocks before this one
        int ba = 0, bm = 0;
        for (int b = tid; b < (int)blockIdx.x; b += 1024) { ba += counts[b]; bm += counts[nb + b]; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { ba += __shfl_xor(ba, o, 64); bm += __shfl_xor(bm, o, 64); }
        if (lane == 0) { wsum[0][wv] = ba; wsum[1][wv] = bm; }
        __syncthreads();
        if (tid == 0) { int ta = 0, tm = 0; for (int w = 0; w < 16; ++w) { ta += wsum[0][w]; tm += wsum[1][w]; } base[0] = ta; base[1] = tm; }
        __syncthreads();
    }
    bool a[4], m[4];
    int na = 0, nm = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        a[k] = vb + k < nv && uz_vertex_active(flag, cn, vb + k);
        m[k] = a[k] && slot[vb + k] < 0;
        na += a[k] ? 1 : 0; nm += m[k] ? 1 : 0;
    }
    int pa = na, pm = nm;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int ta = __shfl_up(pa, o), tm = __shfl_up(pm, o);
        if (lane >= o) { pa += ta; pm += tm; }
    }
    if (lane == 63) { wsum[0][wv] = pa; wsum[1][wv] = pm; }
    __syncthreads();
    int oa = base[0] + pa - na, om = base[1] + pm - nm;
    for (int w = 0; w < wv; ++w) { oa += wsum[0][w]; om += wsum[1][w]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (vb + k < nv) pos[vb + k] = a[k] ? oa : -1;
        if (a[k]) act[oa++] = vb + k;
        if (m[k]) miss[om++] = vb + k;
    }
    if ((int)blockIdx.x == nb - 1 && tid == 1023) {      // (the last thread of the last block holds the totals)
        info[0] = oa; info[1] = om;
        if (nhits) { info[2] = *nhits; *nhits = 0; }
    }
}
// unit right-hand sides of one column solve: axis j of the launch solves K g = e_(v_j)  (rhs zeroed by the caller; v_j < 0: none)
__global__ void k_uz_unit_rhs(int v0, int v1, int v2, double *__restrict__ rhs) {
    if (v0 >= 0) rhs[3 * (size_t)v0] = 1.0;
    if (v1 >= 0) rhs[3 * (size_t)v1 + 1] = 1.0;
    if (v2 >= 0) rhs[3 * (size_t)v2 + 2] = 1.0;
}
// the same right-hand side and the zero start vector in ONE launch (the side-stream batches are bound by the host's launch rate)
__global__ __launch_bounds__(256) void k_uz_unit_rhs_x0(int n3, int v0, int v1, int v2, double *__restrict__ rhs, double *__restrict__ x) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n3) return;
    rhs[i] = (i == 3 * v0 || i == 3 * v1 + 1 || i == 3 * v2 + 2) ? 1.0 : 0.0;      // (v < 0: 3 v + j < 0 never matches)
    x[i] = 0.0;
}
__global__ __launch_bounds__(256) void k_uz_store_cols(int nv, const double *__restrict__ sol, double *__restrict__ cols, int s0, int s1, int s2) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nv) return;
    if (s0 >= 0) cols[(size_t)s0 * nv + i] = sol[3 * (size_t)i];
    if (s1 >= 0) cols[(size_t)s1 * nv + i] = sol[3 * (size_t)i + 1];
    if (s2 >= 0) cols[(size_t)s2 * nv + i] = sol[3 * (size_t)i + 2];
}
// q2 = A^-1 q1 for a q1 supported on the active vertices: q2[i][:] = sum_a col_(slot(act_a))[i] q1[act_a][:].
// Block = 64 consecutive i (one 512-byte run per column and wave) x 4 waves, wave w takes the a = w, w + 4, ...; the four partial
// sums meet in LDS in a fixed order.  Eight column loads in flight per lane.
__global__ __launch_bounds__(256) void k_uz_cols_apply(int nv, int n_act, const int *__restrict__ act, const int *__restrict__ slot,
                                                       const double *__restrict__ cols, const double *__restrict__ q1,
                                                       double *__restrict__ q2, const int *__restrict__ stop) {
    if (stop && *stop) return;
    __shared__ double part[3][4][64];
    __shared__ double tq[3][256];
    __shared__ unsigned so[256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    const int ic = i < nv ? i : nv - 1;
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
    for (int a0 = 0; a0 < n_act; a0 += 256) {
        const int na = min(256, n_act - a0);
        __syncthreads();
        if ((int)threadIdx.x < na) {
            const int v = act[a0 + threadIdx.x];
            so[threadIdx.x] = (unsigned)slot[v];
            tq[0][threadIdx.x] = q1[3 * (size_t)v]; tq[1][threadIdx.x] = q1[3 * (size_t)v + 1]; tq[2][threadIdx.x] = q1[3 * (size_t)v + 2];
        }
        __syncthreads();
        int k = wv;
        for (; k + 28 < na; k += 32) {
            double g[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) g[u] = cols[(size_t)so[k + 4 * u] * nv + ic];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc0 = fma(g[u], tq[0][k + 4 * u], acc0); acc1 = fma(g[u], tq[1][k + 4 * u], acc1); acc2 = fma(g[u], tq[2][k + 4 * u], acc2);
            }
        }
        for (; k < na; k += 4) {
            const double g = cols[(size_t)so[k] * nv + ic];
            acc0 = fma(g, tq[0][k], acc0); acc1 = fma(g, tq[1][k], acc1); acc2 = fma(g, tq[2][k], acc2);
        }
    }
    part[0][wv][lane] = acc0; part[1][wv][lane] = acc1; part[2][wv][lane] = acc2;
    __syncthreads();
    if (threadIdx.x < 192) {
        const int c = threadIdx.x >> 6;
        const int ii = blockIdx.x * 64 + lane;
        if (ii < nv) q2[3 * (size_t)ii + c] = ((part[c][0][lane] + part[c][1][lane]) + part[c][2][lane]) + part[c][3][lane];
    }
}

// ---- the Schur iterations on the ACTIVE vertices only -----------------------------------------------------------------
// Inside the Schur CG only C A^-1 C^T d is needed (r -= alpha C q2), i.e. q2 = A^-1 C^T d at the active vertices: the active x
// active block of K^-1, extracted once per solve from the cached columns (k_uzc_extract: G[j][i] = (K^-1 e_(act_j))[act_i],
// n_act^2 doubles: 4 MB at 729 active vertices against 115 MB for the full-height columns).  An iteration is then two launches:
// k_uzc_matvec (g = G t with t = (C^T d) at the active vertices, segments of the j range in parallel) and k_uzc_rows (ONE block:
// sums the segments, q3 = C g, alpha, y += alpha d, r -= alpha q3, the stop test, beta, d = r - beta d: UzawaCG.hpp:96-118, the
// arithmetic of k_uz_dots / alpha / step / beta / dir on the active rows).  x is not touched inside the loop:
// x = x0 - A^-1 C^T (y - y0) is applied once after it through the full columns (k_uz_cols_apply).
// cn != nullptr (the persistent Schur kernel, uz_persist.hpp, passive rows only): S_ij = G_ij (n_i . n_j) instead, n = the rows of C.
__global__ __launch_bounds__(256) void k_uzc_extract(int nv, int n_act, int ld, const int *__restrict__ act, const int *__restrict__ slot,
                                                     const double *__restrict__ cols, double *__restrict__ G, const double *__restrict__ cn) {
    const int j = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_act) return;
    const int vj = act[j], vi = act[i];
    double g = cols[(size_t)slot[vj] * nv + vi];
    if (cn) g *= fma(cn[3 * (size_t)vi], cn[3 * (size_t)vj], fma(cn[3 * (size_t)vi + 1], cn[3 * (size_t)vj + 1], cn[3 * (size_t)vi + 2] * cn[3 * (size_t)vj + 2]));
    G[(size_t)j * ld + i] = g;
}
// The Schur matrix of the ROWS when rows couple several vertices (dynamic rows: the hit vertex with weight 1 and the three vertices of
// the face with weights -bary, ConstraintSet.hpp:92-110), from the active x active block G of K^-1:
//     S[k][i] = (n_i . n_k) sum_p sum_q w_ip w_kq G[a_kq][a_ip],    a = place of a vertex in the active list (pos), n = the row of C at
// the hit vertex.  rows = the vertices that carry a row, ascending.  Input of the persistent Schur kernel (uz_persist.hpp).
__global__ __launch_bounds__(256) void k_uzc_schur(int n_rows, int ldS, int ldG, const int *__restrict__ rows, const int *__restrict__ pos,
                                                   const int *__restrict__ dface, const double *__restrict__ dbary, const double *__restrict__ cn,
                                                   const double *__restrict__ G, double *__restrict__ S) {
    const int k = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_rows) return;
    const int vi = rows[i], vk = rows[k];
    int ai[4], ak[4], ni = 1, nk = 1; double wi[4], wk[4];
    ai[0] = pos[vi]; wi[0] = 1.0; ak[0] = pos[vk]; wk[0] = 1.0;
    if (dface != nullptr && dface[3 * (size_t)vi] >= 0) { for (int j = 0; j < 3; ++j) { ai[1 + j] = pos[dface[3 * (size_t)vi + j]]; wi[1 + j] = -dbary[3 * (size_t)vi + j]; } ni = 4; }
    if (dface != nullptr && dface[3 * (size_t)vk] >= 0) { for (int j = 0; j < 3; ++j) { ak[1 + j] = pos[dface[3 * (size_t)vk + j]]; wk[1 + j] = -dbary[3 * (size_t)vk + j]; } nk = 4; }
    double sm = 0.0;
    for (int q = 0; q < nk; ++q)
        for (int p = 0; p < ni; ++p) sm = fma(wi[p] * wk[q], G[(size_t)ak[q] * ldG + ai[p]], sm);
    const double nn = fma(cn[3 * (size_t)vi], cn[3 * (size_t)vk], fma(cn[3 * (size_t)vi + 1], cn[3 * (size_t)vk + 1], cn[3 * (size_t)vi + 2] * cn[3 * (size_t)vk + 2]));
    S[(size_t)k * ldS + i] = nn * sm;
}
// part[s][i][:] = sum over the j of segment s of G[j][i] t_j; t_j = q1[act_j] (q1 = C^T d formed by the dense kernels: scenes with
// dynamic rows) or cn[act_j] d[act_j] (passive rows only: q1 == nullptr).  Block = 64 i x 4 waves (wave w: j = w, w + 4, ...).
__global__ __launch_bounds__(256) void k_uzc_matvec(int n_act, int ld, int seg_len, const int *__restrict__ act, const double *__restrict__ G,
                                                    const double *__restrict__ cn, const double *__restrict__ d, const double *__restrict__ q1,
                                                    double *__restrict__ part, const int *__restrict__ stop) {
    if (stop && *stop) return;
    __shared__ double red[3][4][64];
    __shared__ double tq[3][256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane, ic = i < n_act ? i : n_act - 1;
    const int j0 = blockIdx.y * seg_len, j1 = min(n_act, j0 + seg_len);
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
    for (int a0 = j0; a0 < j1; a0 += 256) {
        const int na = min(256, j1 - a0);
        __syncthreads();
        if ((int)threadIdx.x < na) {
            const int v = act[a0 + threadIdx.x];
            if (q1) { tq[0][threadIdx.x] = q1[3 * (size_t)v]; tq[1][threadIdx.x] = q1[3 * (size_t)v + 1]; tq[2][threadIdx.x] = q1[3 * (size_t)v + 2]; }
            else { const double dv = d[v]; tq[0][threadIdx.x] = cn[3 * (size_t)v] * dv; tq[1][threadIdx.x] = cn[3 * (size_t)v + 1] * dv; tq[2][threadIdx.x] = cn[3 * (size_t)v + 2] * dv; }
        }
        __syncthreads();
        const double *Gp = G + (size_t)a0 * ld + ic;
        int k = wv;
        for (; k + 28 < na; k += 32) {
            double g[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) g[u] = Gp[(size_t)(k + 4 * u) * ld];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc0 = fma(g[u], tq[0][k + 4 * u], acc0); acc1 = fma(g[u], tq[1][k + 4 * u], acc1); acc2 = fma(g[u], tq[2][k + 4 * u], acc2);
            }
        }
        for (; k < na; k += 4) {
            const double g = Gp[(size_t)k * ld];
            acc0 = fma(g, tq[0][k], acc0); acc1 = fma(g, tq[1][k], acc1); acc2 = fma(g, tq[2][k], acc2);
        }
    }
    red[0][wv][lane] = acc0; red[1][wv][lane] = acc1; red[2][wv][lane] = acc2;
    __syncthreads();
    if (threadIdx.x < 192) {
        const int c = threadIdx.x >> 6;
        if (i < n_act) part[((size_t)blockIdx.y * n_act + i) * 3 + c] = ((red[c][0][lane] + red[c][1][lane]) + red[c][2][lane]) + red[c][3][lane];
    }
}
// sum of two quantities over a block of 1024 threads, fixed order; result in all threads
__device__ __forceinline__ void block_sum2_1024(double &a, double &b, double *lds /* [32] */) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const double sa = wave_sum(a), sb = wave_sum(b);
    __syncthreads();
    if (lane == 0) { lds[wv] = sa; lds[16 + wv] = sb; }
    __syncthreads();
    double ta = 0.0, tb = 0.0;
    for (int w = 0; w < 16; ++w) { ta += lds[w]; tb += lds[16 + w]; }
    a = ta; b = tb;
}
// ONE: n_act <= 1024, one active vertex per thread: everything it needs is loaded once (all loads in flight together) and stays
// in registers / LDS across the phases -- the kernel is a chain of dependent global latencies otherwise.
template <bool ONE>
__global__ __launch_bounds__(1024) void k_uzc_rows(int n_act, int nseg, const int *__restrict__ act, const int *__restrict__ pos,
                                                   const double *__restrict__ part, double *__restrict__ gq, const double *__restrict__ cn,
                                                   const int *__restrict__ dface, const double *__restrict__ dbary, double *__restrict__ d,
                                                   double *__restrict__ r, double *__restrict__ y, double *__restrict__ q3, double tol2,
                                                   UzScal *__restrict__ sc) {
    __shared__ double lds[32];
    __shared__ double s_alpha, s_beta;
    __shared__ int s_stop;
    __shared__ double sg[ONE ? 3 * 1024 : 3];
    if (sc->stop) return;
    const int tid = threadIdx.x;
    if (ONE) {
        const bool on = tid < n_act;
        const int v = on ? act[tid] : 0;
        double c0 = 0.0, c1 = 0.0, c2 = 0.0, dv = 0.0, rv = 0.0, yv = 0.0, g[3] = {0.0, 0.0, 0.0}, bw[3] = {0.0, 0.0, 0.0};
        int fp[3] = {-1, -1, -1};
        if (on) {
            c0 = cn[3 * (size_t)v]; c1 = cn[3 * (size_t)v + 1]; c2 = cn[3 * (size_t)v + 2];
            dv = d[v]; rv = r[v]; yv = y[v];
            if (dface != nullptr) {
                const int f0 = dface[3 * (size_t)v];
                if (f0 >= 0) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) { fp[j] = pos[dface[3 * (size_t)v + j]]; bw[j] = dbary[3 * (size_t)v + j]; }
                }
            }
            for (int s = 0; s < nseg; ++s)
#pragma unroll
                for (int c = 0; c < 3; ++c) g[c] += part[((size_t)s * n_act + tid) * 3 + c];
        }
        sg[3 * tid] = g[0]; sg[3 * tid + 1] = g[1]; sg[3 * tid + 2] = g[2];
        __syncthreads();
        double t = c0 * g[0] + c1 * g[1] + c2 * g[2];
        if (fp[0] >= 0 && (c0 != 0.0 || c1 != 0.0 || c2 != 0.0)) {
            double rr = 0.0;
#pragma unroll
            for (int j = 0; j < 3; ++j) rr -= bw[j] * (c0 * sg[3 * fp[j]] + c1 * sg[3 * fp[j] + 1] + c2 * sg[3 * fp[j] + 2]);
            t += rr;
        }
        double s0 = dv * t, s1 = dv * rv;
        block_sum2_1024(s0, s1, lds);
        const double denom = s0;
        if (fabs(denom) < 2.2250738585072014e-308) { if (tid == 0) { sc->denom = denom; sc->stop = 1; sc->alpha = 0.0; } return; }
        const double al = s1 / denom;
        yv += al * dv; rv -= al * t;
        s0 = rv * rv; s1 = rv * t;
        block_sum2_1024(s0, s1, lds);
        if (on) { y[v] = yv; r[v] = rv; q3[v] = t; }
        if (s0 < tol2) { if (tid == 0) { sc->denom = denom; sc->alpha = al; sc->rr = s0; sc->stop = 1; } return; }
        const double be = s1 / denom;
        if (on) d[v] = rv - be * dv;
        if (tid == 0) { sc->denom = denom; sc->alpha = al; sc->rr = s0; sc->beta = be; sc->iters += 1; }
        return;
    }
    for (int a = tid; a < n_act; a += 1024)                   // g = sum of the segments (fixed order)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double g = 0.0;
            for (int s = 0; s < nseg; ++s) g += part[((size_t)s * n_act + a) * 3 + c];
            gq[3 * (size_t)a + c] = g;
        }
    __syncthreads();
    double s0 = 0.0, s1 = 0.0;
    for (int a = tid; a < n_act; a += 1024) {                 // q3 = C g on the rows; d.q3, d.r   (k_uz_dots)
        const int v = act[a];
        const double c0 = cn[3 * (size_t)v], c1 = cn[3 * (size_t)v + 1], c2 = cn[3 * (size_t)v + 2];
        double t = c0 * gq[3 * (size_t)a] + c1 * gq[3 * (size_t)a + 1] + c2 * gq[3 * (size_t)a + 2];
        if (dface != nullptr && dface[3 * (size_t)v] >= 0 && (c0 != 0.0 || c1 != 0.0 || c2 != 0.0)) {
            double rr = 0.0;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int f = pos[dface[3 * (size_t)v + j]];
                rr -= dbary[3 * (size_t)v + j] * (c0 * gq[3 * (size_t)f] + c1 * gq[3 * (size_t)f + 1] + c2 * gq[3 * (size_t)f + 2]);
            }
            t += rr;
        }
        q3[v] = t;
        s0 = fma(d[v], t, s0);
        s1 = fma(d[v], r[v], s1);
    }
    block_sum2_1024(s0, s1, lds);
    if (tid == 0) {                                           // k_uz_alpha
        sc->denom = s0;
        s_stop = 0;
        if (fabs(s0) < 2.2250738585072014e-308) { sc->stop = 1; sc->alpha = 0.0; s_stop = 1; s_alpha = 0.0; }
        else { s_alpha = s1 / s0; sc->alpha = s_alpha; }
    }
    __syncthreads();
    if (s_stop) return;
    const double al = s_alpha;
    s0 = 0.0; s1 = 0.0;
    for (int a = tid; a < n_act; a += 1024) {                 // y += alpha d; r -= alpha q3; r.r, r.q3   (k_uz_step without x)
        const int v = act[a];
        y[v] += al * d[v];
        const double rv = r[v] - al * q3[v];
        r[v] = rv;
        s0 = fma(rv, rv, s0);
        s1 = fma(rv, q3[v], s1);
    }
    block_sum2_1024(s0, s1, lds);
    if (tid == 0) {                                           // k_uz_beta
        sc->rr = s0;
        if (s0 < tol2) { sc->stop = 1; s_stop = 1; }
        else { s_beta = s1 / sc->denom; sc->beta = s_beta; sc->iters += 1; }
    }
    __syncthreads();
    if (s_stop) return;
    const double be = s_beta;
    for (int a = tid; a < n_act; a += 1024) { const int v = act[a]; d[v] = r[v] - be * d[v]; }   // k_uz_dir
}
// w = y - y0 (the multiplier update of the whole Schur CG), and x -= q2
__global__ __launch_bounds__(256) void k_uzc_dy(int nv, const double *__restrict__ y, const double *__restrict__ y0, double *__restrict__ w) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v < nv) w[v] = y[v] - y0[v];
}
__global__ __launch_bounds__(256) void k_uzc_xsub(int n3, double *__restrict__ x, const double *__restrict__ q2) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n3) x[i] -= q2[i];
}

} // namespace admm_k
