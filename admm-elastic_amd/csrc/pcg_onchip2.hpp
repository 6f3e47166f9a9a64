// pcg_onchip2.hpp -- the global solve of the ADMM step: one persistent launch per solve for ANY mesh, two-level preconditioned
// pipelined CG, every matrix-vector product out of LDS.
//
// Replaces the prefactored LDLT solve of src/LinearSolver.hpp:87-90 (same system; stop rule on the TRUE residual,
// r . D^-1 r <= tol^2 b . D^-1 b, the rule of the launch-per-iteration path in kernels.hpp, which stays as the fallback for
// systems that do not fit the chip).  Why: at the BASELINE sizes one CG iteration of that path moves ~90 MB through L2 /
// Infinity Cache although the whole problem fits ON CHIP: 256 CUs x (160 KB LDS + 512 KB VGPRs).
//   * block = one CU = a COMPACT patch of the mesh in the plan's internal row order (oc_plan.cpp: recursive graph bisection;
//     79 % of the non-zeros of the unstructured 1 M-tet body are block-local), wave = one 64-row SELL slice, thread = one
//     vertex (3 dofs); the thread's matrix row lives in LDS for the whole solve, its vector entries in registers;
//   * a block keeps a LOCAL VECTOR in LDS: its own 64 spb entries plus its halo list -- the rows of other blocks its matrix
//     rows reference, each fetched ONCE per product.  The product itself runs entirely out of LDS: values (8 B) + 16-bit
//     local columns, four columns per 8-byte word;
//   * two-level preconditioner  M^-1 = S + P (P^T A P)^-1 P^T : P = kOcSub functions per block (the affine functions
//     {1, x, y, z} of the block's vertices, energy-orthonormalised; or indicator vectors of compact aggregates), <= 1024
//     coarse unknowns, dense inverse formed once on the host (the system matrix of a scene never changes,
//     src/Solver.cpp:225-226); S = a degree-2 Chebyshev polynomial of the block-diagonal part of A (all of it in this
//     block's LDS), applied behind the grid barrier's latency and carried by recurrence.  The coarse part needs P^T v of ALL
//     blocks: an all-to-all of 12 numbers per block.  It costs no extra synchronisation: m = M^-1 w is carried like w
//     itself -- with y_w = Ac^-1 P^T w and y_z = Ac^-1 P^T z (this block's 4 x 3 entries), n = A m gives
//     y_n = Ac^-1 P^T n after one all-to-all of the coarse sums of n, then y_z = y_n + beta y_z, y_w -= alpha y_z -- and that
//     all-to-all rides on the iteration's one grid barrier next to the partial dot products;
//   * ONE grid barrier per iteration: the recurrences are those of pipelined CG (Ghysels & Vanroose 2014, general form with r
//     and q = M^-1 s carried explicitly), whose dot products use vectors that exist BEFORE the product n = A m.  An
//     iteration is: publish m; neighbour hand-off (flags, no barrier: oc_sync.hpp); halo fetch; n = A m out of LDS; publish
//     {P^T n, dots}; ONE grid barrier (S n inside its latency); reduce; coarse rows; update;
//   * the recycled (Galerkin) warm start of the ADMM loop is the first phase of the same launch and the new (correction,
//     A correction) pair is written in its epilogue;
//   * pipelined CG carries w = A u by recurrence, so its recursive residual can drift from the true one.  When the recurrence
//     reports convergence the kernel recomputes r = b - A x, applies the stop rule to it, and starts the next PASS from that
//     true residual if the test fails (a pass is trusted for nine orders of magnitude, kOcPipeFloor); a first pass of
//     <= kOc2TrustIters iterations at a tolerance >= 1e-10 is not verified (its deviation is far below the tolerance).  After
//     four passes, or when a pass stagnates near the FP64 floor, the kernel continues SEAMLESSLY (same x, u, p, gamma) in the
//     CLASSIC Hestenes-Stiefel form (p = u + beta p, s = A p, alpha = gamma / (p . s): three synchronisations per iteration,
//     but p . A p is computed directly -- the Chronopoulos-Gear form's alpha is a difference of nearly equal numbers near
//     the floor and sent x to 1e254 on free nearly incompressible bodies).  A verification that fails twice without a 4x
//     improvement means the FP64 floor has been reached: the solve stops as converged;
//   * the three axes are independent systems (A = Ahat (x) I3) with their own b . M^-1 b; an axis whose right-hand side
//     vanishes (C^T d of a floor contact has no x / z part) is measured against the largest axis, b = 0 returns x = 0;
//   * sums that turn non-finite, or grow 1e8x above the best residual seen, send the solve back to the ENTRY x (still in
//     global memory: x is written once, in the epilogue) and on in the classic form; a second failure returns the entry x,
//     reported as unconverged -- never a non-finite vector;
//   * every block reduces the partial records in the same fixed order, so all blocks take the same decisions and the result
//     is deterministic run to run.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels.hpp"
#include "oc_sync.hpp"

namespace admm_k {

struct Oc2Args {
    int n_rows, n_slices;            // internal rows (= 64 n_slices), slices (= G spb)
    const int *ptr, *w;              // per slice: entry offset into val / col16, width (multiple of 4)
    const double *val;               // entry (s, k, lane) at ptr[s] + 64 k + lane
    const unsigned short *col16;     // its local column at ptr[s] + ((k / 4) 64 + lane) 4 + k % 4
    const int *lds_off, *wl_s;       // per slice: slab offset (columns) and columns held in LDS
    double sm_ab, sm_b;              // block-local smoother S v = D^-1 (sm_ab v - sm_b offdiag(A_bb) D^-1 v); sm_b = 0: S = D^-1
    double sm_c0, sm_k1, sm_k2;      // sm_k2 != 0: three Chebyshev steps instead of two (two products): with y = D^-1 v, N = D^-1 offdiag(A_bb):
                                     // v1 = sm_k1 y + sm_k2 N y,  S v = sm_c0 y + v1 + N v1
    int bcols;                       // slab columns per block
    const int *orig;                 // [n_rows] vertex | aggregate << 28 (-1 = dummy row)
    const int *halo_ptr, *halo_src;  // [G + 1]; internal rows of the halo entries (sorted per block)
    int vec_len;                     // entries per axis of the local vector: 64 spb own + halo capacity
    const double *mdiag, *dinv, *b;  // mass + diag(Ahat) by internal row; dinv, b by vertex
    double *x, *u_out;               // by vertex
    double *ubuf;                    // [2][n_rows][4] published vector (x, y, z, pad: one 32-byte sector per row, so a halo
                                     // entry is ONE request), double-buffered by phase parity
    double *part;                    // [2][8][G] per-block partial sums, double-buffered by barrier parity
    unsigned *bar;                   // barrier words (see oc_sync.hpp)
    const int *nbr; unsigned long long *flags;
    int *counters; CgScal *scal; int *sig;
    unsigned long long *prof; int prof_block;
    int spb, G, max_iters, seq;
    double tol2;
    int rc_on; RcBasis rc; double *rc_xs, *rc_r0, *rc_Eslot, *rc_Rslot, *rc_part;   // recycled warm start (internal rows)
    const double *ainv; double *cbuf; int nc, ncp;   // two-level: [nc][ncp] coarse inverse, [2][3][ncp] published aggregate sums
    const float *cwt;    // [n_rows][kOcSubK] row r of P: the row's weights in the coarse functions of its block (one-hot on its aggregate,
                         // or (1, x, y, z) per block: oc_plan.cpp)
    int trust_short;     // 1: a short first pass needs no verification of its residual (see kOc2TrustIters)
    const int *skip;     // optional: *skip != 0 (set by an earlier kernel of the stream, e.g. UzawaCG's stop flag) makes the
                         // launch a no-op -- lets the host enqueue outer iterations ahead without synchronising
    // END PROJECTION ON SOFT MODES (admm_hip_set_soft_modes; kernels.hpp: k_defl_* is the same step as separate launches): after a converged
    // solve x += Z (Z^T K Z)^-1 Z^T r on defl_k <= kOc2DeflMax smooth global vectors Z (internal row order, [defl_k][n_rows]).
    int defl_dbg;      // (experiments: bit 0 no mode loads in the dots, bit 1 no own-row loads, bit 2 no G^-1 staging)
    int defl_k; const float *defl_Z; const double *defl_Ginv; double *defl_rec;      // defl_Z: SINGLE precision (the step stays an exact Galerkin step: G is formed
                                                                                     // from the rounded vectors); defl_rec: [2][3 kOc2DeflMax][G] block sums, by solve parity
    // THE SOLVE SUMS ITS OWN RIGHT-HAND SIDE (g_inc != nullptr; the ADMM loop's contact-free solves on one GPU): what k_gather_rhs does as a
    // launch of its own (kernels.hpp; src/Solver.cpp:98) -- b = M x_bar + the records of the local step + the pin terms -- is done by the
    // thread that owns the row, in the shadow of the LDS fill: the record lists come in the plan's internal row order (one SELL slice per
    // wave, like the matrix), b is also written to g_b (= b: later readers, the recovery path).  Same lists, same order: the same bits.
    const int *g_ptr, *g_w, *g_inc; int g_pad; const double *g_rec, *g_Mxbar; double *g_b;      // g_pad: the all-zero record the lists are padded with
    const int *g_vert_pin; const double *g_pin_xyz; const int *g_pin_active; double *g_pin_u, *g_pin_z; double g_pin_sc; const double *g_pin_nrm;
};
constexpr int kOc2DeflMax = 32;

#ifndef ADMM_OC2_ATTR
#define ADMM_OC2_ATTR
#endif
#ifndef ADMM_OC2_LB
#define ADMM_OC2_LB(t) (t)          // (ISA experiments: another register budget)
#endif
constexpr int kOc2Scratch = 4096;
#ifndef ADMM_OC2_REC_CHUNK
#define ADMM_OC2_REC_CHUNK 1        // 1: a block's seven sums are ONE 64-byte chunk of the record buffer ([parity][block][8]: two sectors of its own), read
#endif                              // chunk-wise; 0 (rounds 2-5): [parity][sum][block] -- four blocks on four XCDs share every sector, a wave per sum reads it back
#ifndef ADMM_OC2_TRUST_SAMPLE
#define ADMM_OC2_TRUST_SAMPLE 1      // the trust rule checked on a sample of solves, revoked when a check fails
#endif
#ifndef ADMM_OC2_TRUST
#define ADMM_OC2_TRUST 1            // (0: compiled out -- same-box A/B of the code generation)
#endif
constexpr double kOc2TrustTol2 = 1e-20;   // ... at a tolerance >= 1e-10 (round 4: the bench tolerance moved from 1e-8 to below 1e-9)
constexpr int kOc2TrustIters = 40;  // pipelined iterations of a first pass whose recursive residual is believed without verification   // bytes of LDS scratch ahead of the local vector and the matrix slab
typedef __attribute__((address_space(3))) unsigned long long LdsU64;

// first half of oc_barrier: drain this block's stores and arrive; oc_barrier_wait (oc_sync.hpp) is the second half
__device__ __forceinline__ void oc2_barrier_arrive(unsigned *bar) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(bar + 16 * ((int)blockIdx.x & 7), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int MAXT>
__global__ __launch_bounds__(ADMM_OC2_LB(MAXT)) ADMM_OC2_ATTR void k_pcg2(Oc2Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *red = (double *)smem;                   // [16][24] wave totals of up to 24 quantities
    double *res24 = (double *)(smem + 3072);        // [24] their block totals
    double *bc = (double *)(smem + 3328);           // [8] reduced scalars of the current phase
    double *sc = (double *)(smem + 3392);           // [8] gamma_prev[3], alpha_prev[3]
    double *gbl = (double *)(smem + 3456);          // [4] b . D^-1 b per axis
    double *glast = (double *)(smem + 3488);        // [4] last gamma per axis (reporting)
    double *ctl = (double *)(smem + 3520);          // [0] best ratio, [1] ratio of the last failed verification; [2..4] alpha,
                                                    // [5..7] beta of this iteration; [8..10] 1 / (b . D^-1 b)
    int *ictl = (int *)(smem + 3616);               // [0] iterations since best, [1] failed verifications, [2] action
    int *ok_lds = (int *)(smem + 3632);
    double *ycur = (double *)(smem + 3648);         // [kOcSubK][3] result of the last coarse solve
    double *yw = ycur + 3 * kOcSubK, *yz = ycur + 9 * kOcSubK;   // coarse parts carried by the w ([2][3 kOcSubK], by parity) and z recurrences
    const int T = (int)blockDim.x, tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = T >> 6;
    // REGISTER DISCIPLINE of the iteration loop.  The loop carries ten recurrence vectors (60 VGPRs) inside a 168-VGPR budget
    // (three waves per SIMD).  Left alone, the compiler hoists every per-lane address, mask and index the helpers below derive
    // from the thread index out of the loop -- > 100 loop-invariant VGPRs -- and then pays for them by parking NINE DOUBLES OF THE
    // RECURRENCE STATE in scratch memory: stored once and reloaded twice per iteration, the stores draining in front of the next
    // exchange (measured: 5 us of a 21-us iteration, and any unrelated edit moved it by +-20 %).  So every helper the loop uses
    // derives what it needs from an OPAQUE copy of the thread index (a volatile move the optimiser cannot see through, hence
    // cannot hoist): a handful of integer instructions per use, live for a few lines.
    auto otid = [&]() -> int { int t; asm volatile("v_mov_b32_e32 %0, %1" : "=v"(t) : "v"(tid)); return t; };
    auto opq = [](int x) -> int { int t; asm volatile("v_mov_b32_e32 %0, %1" : "=v"(t) : "v"(x)); return t; };
    const bool prof = a.prof && (int)blockIdx.x == a.prof_block && tid == 0;
    if (prof) a.prof[63 * 8 + 0] = wall_clock64();
    const int NV = a.vec_len;
    LdsD *vec = (LdsD *)(smem + kOc2Scratch);                    // [3][NV]: own entries [0, T), halo entries [T, T + nh)
    LdsD *lv_all = vec + 3 * NV;                                 // slab values [bcols][64]
    LdsU64 *lc_all = (LdsU64 *)(lv_all + a.bcols * 64);          // slab columns [bcols / 4][64] x 4 x 16 bit

    const int s = __builtin_amdgcn_readfirstlane((int)blockIdx.x * a.spb + wv);
    const int row = s * 64 + lane;
    const int oa = a.orig[row];
    const bool live = oa >= 0;
    const int vi = live ? (oa & 0x0fffffff) : 0;       // the vertex this row belongs to
    const int myagg = live ? (oa >> 28) & 3 : 0;
    // (cvt_here: the float -> double conversion as a volatile instruction.  Left to the compiler it is hoisted out of the
    // iteration loop, single-precision constants then occupy twice the registers in a loop that has none to spare, and most of
    // them are spilled and come back from scratch memory every iteration.)
    auto cvt_here = [](float f) -> double { double d; asm volatile("v_cvt_f64_f32_e32 %0, %1" : "=v"(d) : "v"(f)); return d; };
    float cw[kOcSubK];   // this row of P (zero for dummy rows)
    {
        const float4 t = *(const float4 *)(a.cwt + 4 * (size_t)row);
        cw[0] = t.x; cw[1] = t.y; cw[2] = t.z; cw[3] = t.w;
    }
    static_assert(kOcSubK == 4, "four coarse functions per block");
    // (P y)_row for a [kOcSubK][3] coarse vector in LDS
    auto prolong = [&](const double *y, int j) -> double {
        return fma(cvt_here(cw[0]), y[j], fma(cvt_here(cw[1]), y[3 + j], fma(cvt_here(cw[2]), y[6 + j], cvt_here(cw[3]) * y[9 + j])));
    };
    const int w = __builtin_amdgcn_readfirstlane(a.w[s]);
    const int base = __builtin_amdgcn_readfirstlane(a.ptr[s]);
    const int wl_s = __builtin_amdgcn_readfirstlane(a.wl_s[s]);
    const int slab_off = __builtin_amdgcn_readfirstlane(a.lds_off[s]);
    const double *const vpg_w = a.val + base;                                             // (wave-uniform bases: SGPRs)
    const unsigned long long *const cpg_w = (const unsigned long long *)(a.col16 + base);
    LdsD *const lv_w = lv_all + slab_off * 64;
    LdsU64 *const lc_w = lc_all + (slab_off >> 2) * 64;
    // This row's right-hand side (see Oc2Args::g_inc): records of the local step + pin term + M x_bar, INTERLEAVED with the LDS fill -- as one
    // dependent chain in front of it (index -> record -> sum: three round trips on 12 waves per CU) the fill phase took 24 us instead of 8
    // (ADMM_HIP_OC_PROF, round 6), more than the launch it replaces.  Stage A: the first eight list entries; the slab's values; stage C: their
    // records; the slab's columns; stage E: the sums, in list order (the order of k_gather_rhs: the same bits), longer lists in the plain loop.
    const bool gon = a.g_inc != nullptr;
    int ge[8]; const int *ginc = nullptr; int gw = 0;
    if (gon) {
        gw = __builtin_amdgcn_readfirstlane(a.g_w[s]);
        ginc = a.g_inc + __builtin_amdgcn_readfirstlane(a.g_ptr[s]) + lane;
#pragma unroll
        for (int i = 0; i < 4; ++i) ge[i] = ginc[64 * i];
#pragma unroll
        for (int i = 4; i < 8; ++i) ge[i] = gw > 4 ? ginc[64 * i] : a.g_pad;
    }
    {   // the thread's matrix row -> LDS, once per solve: values
        LdsD *lvw = lv_w + lane;
        const double *vpg = vpg_w + lane;
        for (int k = 0; k < wl_s; ++k) lvw[64 * k] = vpg[64 * k];
    }
    union { double d[2]; bv4u v; } gr0[8]; union { double d; bv2u v; } gr1[8];
    if (gon) {
        const __amdgpu_buffer_rsrc_t rr = soa_rsrc(a.g_rec);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            gr0[i].v = __builtin_amdgcn_raw_buffer_load_b128(rr, ge[i] * 32, 0, 0);
            gr1[i].v = __builtin_amdgcn_raw_buffer_load_b64(rr, ge[i] * 32 + 16, 0, 0);
        }
    }
    {   // ... and its columns
        LdsU64 *lcw = lc_w + lane;
        const unsigned long long *cpg = cpg_w + lane;
        for (int k = 0; k < (wl_s >> 2); ++k) lcw[64 * k] = cpg[64 * k];
    }
    if (gon) {
        double acc[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < 8; ++i) { acc[0] += gr0[i].d[0]; acc[1] += gr0[i].d[1]; acc[2] += gr1[i].d; }
        if (gw > 8) gather_records(ginc + 64 * 8, gw - 8, a.g_rec, acc);
        if (live) {
            if (a.g_vert_pin) pin_term_update(a.g_vert_pin, a.g_pin_xyz, a.g_pin_active, a.g_pin_u, a.g_pin_z, a.g_pin_sc, a.g_pin_nrm, a.x, vi, true, acc);
#pragma unroll
            for (int j = 0; j < 3; ++j) a.g_b[3 * (size_t)vi + j] = acc[j] + a.g_Mxbar[3 * (size_t)vi + j];
        }
    }
    const int hp0 = a.halo_ptr[blockIdx.x], nh = a.halo_ptr[blockIdx.x + 1] - hp0;
    // the halo entries this thread fetches (two per thread cover nh <= 2 T; more are read from the list every time)
    const int hs0 = tid < nh ? a.halo_src[hp0 + tid] : 0, hs1 = tid + T < nh ? a.halo_src[hp0 + tid + T] : 0;
    if (prof) a.prof[63 * 8 + 1] = wall_clock64();
    const int ub = a.n_rows * 32;           // bytes of one published-vector buffer
    __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc((void *)a.ubuf, 0, 2 * ub, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc((void *)a.part, 0, 2 * 8 * a.G * 8, 0x00020000);
    const bool two_level = a.ainv != nullptr && a.nc <= 2 * T;
    __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc((void *)a.cbuf, 0, a.cbuf ? 2 * 3 * a.ncp * 8 : 0, 0x00020000);

    double rx[3], ru[3], rw[3], rp[3], rsv[3], rz[3], rq[3], rr[3];
    double sw[3] = {0.0, 0.0, 0.0}, sz[3] = {0.0, 0.0, 0.0};     // S w and S z (S = block-local part of M^-1), by recurrence.
    // (FP64 like everything the recurrences carry: in single precision -- measured, round 3 -- the absolute error of S w stays at
    // 6e-8 of its INITIAL size while w shrinks by five orders, the preconditioner turns to noise and the solves of the 1 M-tet
    // body take 29.7 instead of 8.8 iterations.)
    // The plan only exists for masses that are the same on the three axes of a vertex (oc_plan.cpp; the reference has no others:
    // m_masses[3 i + j] = mass of vertex i): the diagonal and its inverse are ONE value per row, not three -- the kernel sits
    // on the edge of its register budget, these are 8 VGPRs.
    const double rd0 = live ? a.dinv[3 * (size_t)vi] : 0.0, rm0 = live ? a.mdiag[3 * (size_t)row] : 0.0;
    const double rd[3] = {rd0, rd0, rd0}, rm[3] = {rm0, rm0, rm0};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        rx[j] = live ? a.x[3 * (size_t)vi + j] : 0.0;
        ru[j] = rw[j] = rp[j] = rsv[j] = rz[j] = rq[j] = rr[j] = 0.0;
    }
    unsigned *const bar = a.bar + 32 * 16 * (a.seq & 1);
    if (blockIdx.x == 0 && tid < 9) a.bar[32 * 16 * ((a.seq & 1) ^ 1) + 16 * (tid < 8 ? tid : 17)] = 0u;
    if (a.skip && *a.skip) return;    // (after the clearing above: the next launch counts on the set this one cleared)
    if (tid < 3 * kOcSubK) { yw[tid] = 0.0; yw[3 * kOcSubK + tid] = 0.0; yz[tid] = 0.0; ycur[tid] = 0.0; }
    if (tid == 0) ictl[3] = ADMM_OC2_TRUST_SAMPLE ? a.counters[76] : 0;      // trust revoked for this context (a SAMPLED verification of a short first pass failed: below)
    int ywp = 0;         // offset of the current y_w buffer (0 or 3 kOcSubK)
    unsigned ph = 0;     // publish phase of the vector: buffer parity = ph & 1, tag of the neighbour flags
    unsigned be = 0;     // grid-barrier epoch (arrivals of this block so far); record parity = be & 1
    int prof_n = 0;
#ifdef ADMM_OC2_MARKS      // (ISA inspection only: comment markers between the phases of the iteration)
#define OC2_MARK(slot) asm volatile("; OC2MARK " #slot)
#else
#define OC2_MARK(slot)
#endif
#define OC2_STAMP(slot) do { OC2_MARK(slot); if (prof && prof_n < 62) a.prof[prof_n * 8 + (slot)] = wall_clock64(); } while (0)

    // own entries -> local vector and -> this wave's 64 sectors of ubuf: two 16-byte write-through stores per lane, each
    // instruction covering 1 KB of whole sectors (lane pairs write the two halves of a row's sector: half-written sectors
    // from a row-per-lane layout measured 3x slower to drain); transposed through the local vector
    // layout of the local vector: component j of entry c.  ADMM_OC2_AOS: the three components of an entry side by side (24-byte
    // stride: the row loop's three reads per matrix entry become one ds_read2_b64 + one ds_read_b64); 0: per-axis arrays
#ifndef ADMM_OC2_AOS
#define ADMM_OC2_AOS 0
#endif
#if ADMM_OC2_AOS
#define OC2_VX(c, j) (3 * (c) + (j))
#else
#define OC2_VX(c, j) ((j) * NV + (c))
#endif
    auto publish = [&](const double *v) {
        const int tid = otid(), lane = tid & 63;
        const int wb = tid & ~63;
        vec[OC2_VX(wb + lane, 0)] = v[0]; vec[OC2_VX(wb + lane, 1)] = v[1]; vec[OC2_VX(wb + lane, 2)] = v[2];
        const int r0 = lane >> 1, hi = lane & 1;
        const int bo = (int)(ph & 1u) * ub + s * 2048 + lane * 16;
        const double a0 = vec[OC2_VX(wb + r0, hi ? 2 : 0)], a1 = hi ? 0.0 : vec[OC2_VX(wb + r0, 1)];
        const double b0 = vec[OC2_VX(wb + 32 + r0, hi ? 2 : 0)], b1 = hi ? 0.0 : vec[OC2_VX(wb + 32 + r0, 1)];
        oc_store_sc1(rs_u, bo, a0, a1);
        oc_store_sc1(rs_u, bo + 1024, b0, b1);
    };
    // this thread's row (off-diagonal part) times the local vector
    auto row_times_local_vector = [&](double *acc) {
        const int lane = otid() & 63;
        const LdsD *lv = lv_w + lane;
        const LdsU64 *lc = lc_w + lane;
        const double *vpg = vpg_w + lane;
        const unsigned long long *cpg = cpg_w + lane;
        for (int k = 0; k < w; k += 4) {
            unsigned long long cc; double vv[4];
            if (k < wl_s) {
                cc = lc[64 * (k >> 2)];
#pragma unroll
                for (int i = 0; i < 4; ++i) vv[i] = lv[64 * (k + i)];
            } else {
                cc = cpg[64 * (k >> 2)];
#pragma unroll
                for (int i = 0; i < 4; ++i) vv[i] = vpg[64 * (k + i)];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = (int)((cc >> (16 * i)) & 0xffffull);
                acc[0] = fma(vv[i], vec[OC2_VX(c, 0)], acc[0]); acc[1] = fma(vv[i], vec[OC2_VX(c, 1)], acc[1]); acc[2] = fma(vv[i], vec[OC2_VX(c, 2)], acc[2]);
            }
        }
    };
    // after the synchronisation of phase ph: halo entries -> local vector, then out = A v from LDS
    auto halo_and_rows = [&](const double *self, double *out) {
        const int vb = (int)(ph & 1u) * ub;
        const int tid = otid();
        for (int h = tid, it = 0; h < nh; h += T, ++it) {
            const int src = it == 0 ? hs0 : it == 1 ? hs1 : a.halo_src[hp0 + h];
            union { double d[2]; v4u v; } g0, g1;
            g0.v = __builtin_amdgcn_raw_buffer_load_b128(rs_u, vb + src * 32, 0, 16);
            g1.v = __builtin_amdgcn_raw_buffer_load_b128(rs_u, vb + src * 32 + 16, 0, 16);
            vec[OC2_VX(T + h, 0)] = g0.d[0]; vec[OC2_VX(T + h, 1)] = g0.d[1]; vec[OC2_VX(T + h, 2)] = g1.d[0];
        }
        __syncthreads();
        double acc[3] = {0.0, 0.0, 0.0};
        row_times_local_vector(acc);
#pragma unroll
        for (int j = 0; j < 3; ++j) out[j] = fma(rm[j], self[j], acc[j]);
    };
    auto halo_and_rows_self_from_vec = [&](double *out) {
        const int tid = otid();
        const double self[3] = {vec[OC2_VX(tid, 0)], vec[OC2_VX(tid, 1)], vec[OC2_VX(tid, 2)]};    // written by this thread in publish()
        halo_and_rows(self, out);
    };
    // The block-local part of the preconditioner: a degree-2 Chebyshev polynomial in D^-1 A_bb, A_bb = the entries of A whose
    // row AND column sit in this block -- data the block holds in LDS, no exchange (experiments/block_cheb_proto.py: 114 ->
    // 81 iterations next to the coarse space on the 1 M-tet body; the exact block solve would give 63).  In closed form
    // S v = D^-1 (sm_ab v - sm_b offdiag(A_bb) D^-1 v): D^-1 v goes into the local vector with the halo part zeroed, the
    // ordinary row loop does the rest.  S is symmetric positive definite as long as lambda_max(D^-1 A_bb) stays below the
    // bound the host derived the coefficients from (oc_plan.cpp: power iteration + margin).
    // (counters[75]: set for good by a solve whose pipelined pass broke off with non-finite or negative sums -- the sign of a
    // preconditioner that is not positive definite, e.g. a bound of the smoother's interval that was too low; later solves of the
    // context then run with S = D^-1)
    const bool smoothing = a.sm_b != 0.0 && __builtin_amdgcn_readfirstlane(a.counters[75]) == 0;
    auto smooth = [&](const double *v, double *out) {
        if (!smoothing) {
#pragma unroll
            for (int j = 0; j < 3; ++j) out[j] = rd[j] * v[j];
            return;
        }
        const int tid = otid();
#pragma unroll
        for (int j = 0; j < 3; ++j) vec[OC2_VX(tid, j)] = rd[j] * v[j];
        for (int h = tid; h < nh; h += T) { vec[OC2_VX(T + h, 0)] = 0.0; vec[OC2_VX(T + h, 1)] = 0.0; vec[OC2_VX(T + h, 2)] = 0.0; }
        __syncthreads();
        double acc[3] = {0.0, 0.0, 0.0};
        row_times_local_vector(acc);
        if (a.sm_k2 != 0.0) {     // (uniform) second product: the halo part of the local vector is still zero
            __syncthreads();
            const int tid = otid();
#pragma unroll
            for (int j = 0; j < 3; ++j) vec[OC2_VX(tid, j)] = rd[j] * fma(a.sm_k2, acc[j], a.sm_k1 * v[j]);
            __syncthreads();
            double acc2[3] = {0.0, 0.0, 0.0};
            row_times_local_vector(acc2);
            const int t2 = otid();
#pragma unroll
            for (int j = 0; j < 3; ++j) out[j] = fma(rd[j], fma(a.sm_c0, v[j], acc2[j]), vec[OC2_VX(t2, j)]);
            __syncthreads();
            return;
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) out[j] = rd[j] * fma(-a.sm_b, acc[j], a.sm_ab * v[j]);
        __syncthreads();   // the local vector is rewritten by the next publish
    };
    // block totals of 8 NG quantities -> res24 (valid after the call for all threads); fixed order -> deterministic.
    // The quantities are asked for EIGHT AT A TIME (gen(h, q8) fills group h) with a scheduling barrier between the groups: handed
    // over as one array, all 24 were formed before the first reduction started -- 48 registers on top of the ten recurrence
    // vectors of the iteration, which the allocator paid for by keeping part of THEM in scratch memory.
    auto block_sums_gen = [&](auto gen, auto ng_tag) {
        constexpr int NG = decltype(ng_tag)::value;
#pragma unroll
        for (int h = 0; h < NG; ++h) {
            double q8[8], b0, b1;
            gen(h, q8);
            row_sum8(q8, b0, b1);
            b0 += __shfl_xor(b0, 16, 64); b1 += __shfl_xor(b1, 16, 64);
            b0 += __shfl_xor(b0, 32, 64); b1 += __shfl_xor(b1, 32, 64);
            const int tid = otid(), lane = tid & 63;
            if (lane < 4) {
                double *dst = red + (tid >> 6) * 24 + 8 * h + 4 * (lane & 1) + (lane & 2);
                dst[0] = b0; dst[1] = b1;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        const int tid = otid();
        if (tid < 8 * NG) {
            double sm = 0.0;
            for (int k = 0; k < nw; ++k) sm += red[k * 24 + tid];
            res24[tid] = sm;
        }
        __syncthreads();
    };
    auto block_sums = [&](const double *q24, auto ng_tag) {
        block_sums_gen([&](int h, double *q8) {
#pragma unroll
            for (int i = 0; i < 8; ++i) q8[i] = q24[8 * h + i];
        }, ng_tag);
    };
    auto block_sums24 = [&](const double *q24) { block_sums(q24, std::integral_constant<int, 3>()); };
    // this block's record: q7[0..6] -> part (parity par) and, two-level, P^T v -> cbuf (parity par)
    // (with_v is a flag, not "v or nullptr": an array whose address is selected against nullptr stays in scratch memory -- the
    // n of every iteration did, and came back through twelve conditional scratch loads)
    const double zero3[3] = {0.0, 0.0, 0.0};
    auto publish_record = [&](const double *q7, const double *v, bool with_v, int par) {
        block_sums_gen([&](int h, double *q8) {     // groups: [q7, 0] [aggregates 0, 1 and x, y of 2] [z of 2, aggregate 3, 0 ...]
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int f = 8 * h + i;                       // compile-time after unrolling
                if (f < 8) q8[i] = f < 7 ? q7[f < 7 ? f : 0] : 0.0;
                else if (f < 8 + 3 * kOcSubK) {
                    const int ag = (f - 8) / 3, j = (f - 8) % 3;
                    q8[i] = with_v ? cvt_here(cw[ag]) * v[j] : 0.0;                 // (P^T v)_(ag, j): this row's share
                } else q8[i] = 0.0;
            }
        }, std::integral_constant<int, 3>());
        const int tid = otid();
#if ADMM_OC2_REC_CHUNK
        if (tid < 8) oc_store_sc1(rs_p, ((par * a.G + (int)blockIdx.x) * 8 + tid) * 8, res24[tid]);      // (res24[7] = 0: the chunk is written whole)
#else
        if (tid < 7) oc_store_sc1(rs_p, ((par * 8 + tid) * a.G + (int)blockIdx.x) * 8, res24[tid]);
#endif
        else if (with_v && tid >= 8 && tid < 8 + 3 * kOcSubK) {
            const int ag = (tid - 8) / 3, j = (tid - 8) - 3 * ag;
            oc_store_sc1(rs_c, ((par * 3 + j) * a.ncp + (int)blockIdx.x * kOcSubK + ag) * 8, res24[tid]);
        }
    };
    // after the grid barrier: bc[0..nsum) = the global sums of the records of parity par
#if ADMM_OC2_REC_CHUNK
    // The records of all blocks, chunk-wise (round 6): the buffer of one parity is 4 G units of 16 bytes (unit o = sums 2 (o & 3), 2 (o & 3) + 1 of block
    // o >> 2); thread tid takes the units tid, tid + T, ... -- T is a multiple of four, so all of them carry the SAME pair of sums -- one or two
    // 16-byte loads per thread instead of four 8-byte loads per lane of seven waves, every sector asked for once per block.  rec_reduce then adds
    // up the pairs: over the lanes of equal lane & 3 inside each row of sixteen (two DPP shifts), over the four rows (two lane exchanges), over the
    // waves (LDS, the idle local vector).  The same order in every block: the same bits, the same decisions.
    union RecUnit { double d[2]; v4u v; };
    auto rec_issue = [&](int par, RecUnit &g0, RecUnit &g1) {
        const int tid = otid(), n16 = 4 * a.G;
        g0.d[0] = 0.0; g0.d[1] = 0.0; g1.d[0] = 0.0; g1.d[1] = 0.0;
        if (tid < n16) g0.v = __builtin_amdgcn_raw_buffer_load_b128(rs_p, par * a.G * 64 + tid * 16, 0, 16);
        if (tid + T < n16) g1.v = __builtin_amdgcn_raw_buffer_load_b128(rs_p, par * a.G * 64 + (tid + T) * 16, 0, 16);
    };
    auto rec_reduce = [&](int par, int nsum, const RecUnit &g0, const RecUnit &g1) {      // -> bc[0..nsum), valid after the NEXT block barrier
        double sa = g0.d[0] + g1.d[0], sb = g0.d[1] + g1.d[1];
        const int tid = otid(), n16 = 4 * a.G;
        for (int o = tid + 2 * T; o < n16; o += T) {      // (blocks of few waves)
            RecUnit g; g.v = __builtin_amdgcn_raw_buffer_load_b128(rs_p, par * a.G * 64 + o * 16, 0, 16);
            sa += g.d[0]; sb += g.d[1];
        }
        sa += dpp_f64<0x114>(sa); sb += dpp_f64<0x114>(sb);      // row_shr:4 (lanes without a source add 0)
        sa += dpp_f64<0x118>(sa); sb += dpp_f64<0x118>(sb);      // row_shr:8 -> lanes 12..15 of a row: the row's total of their pair
        sa += __shfl_xor(sa, 16, 64); sb += __shfl_xor(sb, 16, 64);
        sa += __shfl_xor(sa, 32, 64); sb += __shfl_xor(sb, 32, 64);
        double *red2 = (double *)(smem + kOc2Scratch);      // [waves][8], in the idle local vector
        const int lane = tid & 63;
        if (lane >= 12 && lane < 16) { red2[(tid >> 6) * 8 + 2 * (lane & 3)] = sa; red2[(tid >> 6) * 8 + 2 * (lane & 3) + 1] = sb; }
        __syncthreads();
        if (tid < nsum) {
            double sm = 0.0;
            for (int k = 0; k < nw; ++k) sm += red2[k * 8 + tid];
            bc[tid] = sm;
        }
    };
#endif
    auto reduce_records = [&](int par, int nsum) {
#if ADMM_OC2_REC_CHUNK
        RecUnit g0, g1;
        rec_issue(par, g0, g1);
        rec_reduce(par, nsum, g0, g1);
        __syncthreads();
        return;
#endif
        const int tid = otid(), lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        for (int k = wv; k < nsum; k += nw) {
            double rec[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { const int g = lane + 64 * i; rec[i] = g < a.G ? oc_load_sc1_f64(rs_p, ((par * 8 + k) * a.G + g) * 8) : 0.0; }
            double sm = (rec[0] + rec[1]) + (rec[2] + rec[3]);
            for (int g = lane + 256; g < a.G; g += 64) sm += oc_load_sc1_f64(rs_p, ((par * 8 + k) * a.G + g) * 8);
            sm = wave_sum(sm);
            if (lane == 0) bc[k] = sm;
        }
        __syncthreads();
    };
    // The rows of Ac^-1 of this block's aggregates, columns tid and tid + T: constant over the solve, fetched (L2) ahead
    // of the grid barrier so that their latency hides behind it
    struct AinvRows { float v[2][kOcSubK]; };   // (a preconditioner: single precision, applied the same way every time, is exact enough)
    auto ainv_prefetch = [&](AinvRows &ar) {
        const int tid = otid();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int c = tid + it * T;
#pragma unroll
            for (int ag = 0; ag < kOcSubK; ++ag) ar.v[it][ag] = c < a.nc ? (float)a.ainv[(size_t)((int)blockIdx.x * kOcSubK + ag) * a.ncp + c] : 0.0f;
        }
    };
    // after the grid barrier: bc[0..nsum) = global sums of the records, ycur = (rows of Ac^-1 of this block's aggregates) x
    // (published coarse vector), all of parity par.  Every global load is issued before the first use.
    auto reduce_and_coarse = [&](int par, int nsum, const AinvRows &ar) {
        double rec[4] = {0.0, 0.0, 0.0, 0.0}, cn[2][3];
        const int tid = otid(), lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
#if ADMM_OC2_REC_CHUNK
        RecUnit g0, g1;
        if (nsum > 0) rec_issue(par, g0, g1);
#else
        if (wv < nsum) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { const int g = lane + 64 * i; rec[i] = g < a.G ? oc_load_sc1_f64(rs_p, ((par * 8 + wv) * a.G + g) * 8) : 0.0; }
        }
#endif
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int c = tid + it * T;
#pragma unroll
            for (int j = 0; j < 3; ++j) cn[it][j] = c < a.nc ? oc_load_sc1_f64(rs_c, ((par * 3 + j) * a.ncp + c) * 8) : 0.0;
        }
#if ADMM_OC2_REC_CHUNK
        if (nsum > 0) rec_reduce(par, nsum, g0, g1);      // (bc is read behind the block barriers of the coarse rows below)
        (void)rec; (void)lane; (void)wv;
#else
        if (wv < nsum) {
            double sm = (rec[0] + rec[1]) + (rec[2] + rec[3]);
            for (int g = lane + 256; g < a.G; g += 64) sm += oc_load_sc1_f64(rs_p, ((par * 8 + wv) * a.G + g) * 8);
            sm = wave_sum(sm);
            if (lane == 0) bc[wv] = sm;
        }
        for (int k = wv + nw; k < nsum; k += nw) {   // blocks with fewer waves than sums
            double sm = 0.0;
            for (int g = lane; g < a.G; g += 64) sm += oc_load_sc1_f64(rs_p, ((par * 8 + k) * a.G + g) * 8);
            sm = wave_sum(sm);
            if (lane == 0) bc[k] = sm;
        }
#endif
        __builtin_amdgcn_sched_barrier(0);     // (the record sums are done and their registers free before the coarse rows start)
        block_sums_gen([&](int h, double *q8) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int f = 8 * h + i, ag = f / 3, j = f % 3;          // compile-time after unrolling
                if (f < 3 * kOcSubK) q8[i] = fma(cvt_here(ar.v[0][ag]), cn[0][j], cvt_here(ar.v[1][ag]) * cn[1][j]);
                else q8[i] = 0.0;
            }
        }, std::integral_constant<int, 2>());
        if (otid() < 3 * kOcSubK) { const int t = otid(); ycur[t] = res24[t]; }
        __syncthreads();
    };
    // y = (P Ac^-1 P^T v) on this thread's row: one all-to-all (its own grid barrier)
    auto coarse_apply = [&](const double *v, double *y) -> bool {
        ++be;
        const int par = (int)(be & 1u);
        const double z7[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        publish_record(z7, v, true, par);
        AinvRows ar;
        ainv_prefetch(ar);
        if (!oc_barrier(bar, be, a.G, ok_lds, a.sig)) return false;
        reduce_and_coarse(par, 0, ar);
#pragma unroll
        for (int j = 0; j < 3; ++j) y[j] = prolong(ycur, j);                     // (start of a pass only)
        return true;
    };
    // ... with the block smoother applied to the same vector BEHIND the barrier's latency (round 6: at the start of a solve S r sat in front of
    // the coarse all-to-all, ~3 us on the critical path of every solve since the smoother is alive in every context)
    auto coarse_apply_smooth = [&](const double *v, double *y, double *sv) -> bool {
        ++be;
        const int par = (int)(be & 1u);
        const double z7[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        publish_record(z7, v, true, par);
        AinvRows ar;
        ainv_prefetch(ar);
        oc2_barrier_arrive(bar);
        smooth(v, sv);
        if (!oc_barrier_wait(bar, be, a.G, ok_lds, a.sig)) return false;
        reduce_and_coarse(par, 0, ar);
#pragma unroll
        for (int j = 0; j < 3; ++j) y[j] = prolong(ycur, j);
        return true;
    };
    int iters = 0, pipe_iters = 0;
    bool conv = false, aborted = false;
    auto action = [&]() -> int { __syncthreads(); return __builtin_amdgcn_readfirstlane(ictl[2]); };
    // u = D^-1 (b - A x) from the x held in registers (x goes through the published copy: the columns are local indices);
    // leaves r . D^-1 r (and optionally b . D^-1 b) in q[0..5]
    auto true_residual = [&](bool with_bnorm, double *q, double *ri_out) -> bool {
        double ax[3];
        ++ph; publish(rx);
        if (a.nbr) { if (!oc_announce_and_wait_neighbours<false>(bar, a.flags, a.nbr, (unsigned)a.seq, ph, ok_lds, a.sig)) return false; }
        else { ++be; if (!oc_barrier(bar, be, a.G, ok_lds, a.sig)) return false; }
        halo_and_rows(rx, ax);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double bj = live ? a.b[3 * (size_t)vi + j] : 0.0;
            const double ri = bj - ax[j];
            if (ri_out) ri_out[j] = ri;
            ru[j] = rd[j] * ri;
            q[j] = ru[j] * ri;
            q[3 + j] = with_bnorm ? bj * rd[j] * bj : 0.0;
        }
        __syncthreads();   // the local vector is rewritten by the next publish
        return true;
    };
    do {
        // ---- start: TRUE residual of the warm start (after the recycled projection), stop test, w = A u ---------------
        {
            double q[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            if (!a.rc_on) { if (!true_residual(true, q, nullptr)) { aborted = true; break; } }
            else {
                // recycled warm start (see k_rc_* in kernels.hpp): A-orthogonal projection of the initial
                // error on the stored exact pairs (E_j, R_j = A E_j): per axis G c = g, x += E c, r0 -= R c
                double ri[3], bj[3];
                const int cnt = a.rc.cnt;
                double e[kRc][3], r[kRc][3];
#pragma unroll
                for (int jj = 0; jj < kRc; ++jj)
#pragma unroll
                    for (int ax = 0; ax < 3; ++ax) {
                        const bool on = live && jj < cnt;
                        e[jj][ax] = on ? a.rc.E[jj][3 * (size_t)row + ax] : 0.0;
                        r[jj][ax] = on ? a.rc.R[jj][3 * (size_t)row + ax] : 0.0;
                    }
                if (!true_residual(true, q, ri)) { aborted = true; break; }
                if (prof) a.prof[62 * 8 + 0] = wall_clock64();
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    bj[j] = live ? a.b[3 * (size_t)vi + j] : 0.0;
                    if (live) { a.rc_xs[3 * (size_t)row + j] = rx[j]; a.rc_r0[3 * (size_t)row + j] = ri[j]; }
                }
                if (cnt > 0) {
                    __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc((void *)a.rc_part, 0, 72 * a.G * 8, 0x00020000);
                    // Block totals of the products, 24 at a time.  Round 6: the Gram matrix E_i . R_j = E_i . A E_j is symmetric (the pairs are exact
                    // to rounding, and rc_cholesky takes the symmetric part anyway), so only i <= j is summed: per axis 10 + kRc + 2 = 16 sums,
                    // 48 in all -- two rounds of 24 instead of three, 48 instead of 66 sums to add up behind the barrier.
                    constexpr int kRcS = kRc * (kRc + 1) / 2 + kRc + 2;      // 16: [0, 10) G_ij (i <= j, row by row), [10, 14) E_i . r0, 14 r0 . D^-1 r0, 15 b . D^-1 b
                    static_assert(3 * kRcS == 48, "two rounds of 24 sums");
#pragma unroll
                    for (int g24 = 0; g24 < 2; ++g24) {
                        double q24[24];
#pragma unroll
                        for (int i = 0; i < 24; ++i) {
                            const int f = 24 * g24 + i, ax = f / kRcS, qi = f % kRcS;     // compile-time after unrolling
                            int gi = 0, gj = 0;
                            { int k = qi; for (int ii = 0; ii < kRc; ++ii) { if (k < kRc - ii) { gi = ii; gj = ii + k; break; } k -= kRc - ii; } }
                            constexpr int NG_ = kRc * (kRc + 1) / 2;
                            q24[i] = (qi < NG_) ? e[gi][ax] * r[gj][ax]
                                   : (qi < NG_ + kRc) ? e[qi - NG_][ax] * ri[ax]
                                   : (qi == NG_ + kRc) ? ri[ax] * rd[ax] * ri[ax]
                                   : bj[ax] * rd[ax] * bj[ax];
                        }
                        block_sums24(q24);
                        if (tid < 24) oc_store_sc1(rs_r, ((24 * g24 + tid) * a.G + (int)blockIdx.x) * 8, res24[tid]);
                    }
                    if (prof) a.prof[62 * 8 + 1] = wall_clock64();
                    ++be;
                    if (!oc_barrier(bar, be, a.G, ok_lds, a.sig)) { aborted = true; break; }
                    if (prof) a.prof[62 * 8 + 2] = wall_clock64();
                    double *sums = (double *)(smem + kOc2Scratch);   // [3 kRcS + 3 kRc] in the (idle) local vector
                    {   // every block adds the G partials of every sum in the same order; wave wv takes sums wv, wv + nw, ...:
                        // all loads first, one round trip
                        double v[4][4];
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const int k = wv + nw * t;
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int g = lane + 64 * j;
                                v[t][j] = (k < 3 * kRcS && g < a.G) ? oc_load_sc1_f64(rs_r, (k * a.G + g) * 8) : 0.0;
                            }
                        }
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const int k = wv + nw * t;
                            if (k < 3 * kRcS) {
                                double sm = ((v[t][0] + v[t][1]) + v[t][2]) + v[t][3];
                                for (int g = lane + 256; g < a.G; g += 64) sm += oc_load_sc1_f64(rs_r, (k * a.G + g) * 8);
                                sm = wave_sum(sm);
                                if (lane == 0) sums[k] = sm;
                            }
                        }
                        for (int k = wv + 4 * nw; k < 3 * kRcS; k += nw) {   // blocks with fewer than 12 waves
                            double sm = 0.0;
                            for (int g = lane; g < a.G; g += 64) sm += oc_load_sc1_f64(rs_r, (k * a.G + g) * 8);
                            sm = wave_sum(sm);
                            if (lane == 0) sums[k] = sm;
                        }
                    }
                    __syncthreads();
                    double *coefL = sums + 3 * kRcS;   // [3][kRc]
                    if (tid < 3) {
                        bool skip = true;    // r0 already meets the tolerance on every axis: the pairs must not perturb x
                        for (int ax = 0; ax < 3; ++ax) skip = skip && (sums[ax * kRcS + 14] <= a.tol2 * sums[ax * kRcS + 15] + 1e-300);
                        // the layout rc_cholesky reads (G_ij at i kRc + j, E_i . r0 at kRc^2 + i), filled from the triangle
                        double S[kRc * kRc + kRc];
                        const double *sa = sums + kRcS * tid;
                        {
                            int k = 0;
#pragma unroll
                            for (int ii = 0; ii < kRc; ++ii)
#pragma unroll
                                for (int jj = ii; jj < kRc; ++jj) { S[ii * kRc + jj] = sa[k]; S[jj * kRc + ii] = sa[k]; ++k; }
#pragma unroll
                            for (int ii = 0; ii < kRc; ++ii) S[kRc * kRc + ii] = sa[kRc * (kRc + 1) / 2 + ii];
                        }
                        double c[kRc];
                        rc_cholesky(S, cnt, skip, c);
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
                    __syncthreads();   // the local vector is rewritten by the next publish
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    ru[j] = rd[j] * ri[j];
                    q[j] = ru[j] * ri[j];
                    q[3 + j] = bj[j] * rd[j] * bj[j];
                }
            }
            // r explicitly; u = M^-1 r (q stays the Jacobi norm)
            if (prof) a.prof[62 * 8 + 3] = wall_clock64();
#pragma unroll
            for (int j = 0; j < 3; ++j) rr[j] = live ? ru[j] * fast_rcp(rd[j]) : 0.0;
            if (two_level) {
                double y[3];
                if (!coarse_apply_smooth(rr, y, ru)) { aborted = true; break; }
#pragma unroll
                for (int j = 0; j < 3; ++j) ru[j] += y[j];
            } else smooth(rr, ru);
            // u to the neighbours (hand-off, no barrier), w = A u, then ONE record: the stop-test sums and P^T w
            if (prof) a.prof[62 * 8 + 4] = wall_clock64();
            ++ph; publish(ru);
            if (a.nbr) { if (!oc_announce_and_wait_neighbours<false>(bar, a.flags, a.nbr, (unsigned)a.seq, ph, ok_lds, a.sig)) { aborted = true; break; } }
            else { ++be; if (!oc_barrier(bar, be, a.G, ok_lds, a.sig)) { aborted = true; break; } }
            halo_and_rows(ru, rw);                   // w = A u
            if (prof) a.prof[62 * 8 + 5] = wall_clock64();
            ++be;
            publish_record(q, rw, two_level, (int)(be & 1u));
            if (prof) a.prof[62 * 8 + 6] = wall_clock64();
        }
        {
            AinvRows ar0;
            if (two_level) ainv_prefetch(ar0);
            // (S w for the first pass behind this barrier's latency, like S n in the iterations: start_pass(have_uw = true) finds it done)
            oc2_barrier_arrive(bar);
            smooth(rw, sw);
            if (!oc_barrier_wait(bar, be, a.G, ok_lds, a.sig)) { aborted = true; break; }
            if (two_level) {
                reduce_and_coarse((int)(be & 1u), 6, ar0);      // ... and y_w = Ac^-1 P^T w for the first pass
                if (tid < 3 * kOcSubK) { yw[ywp + tid] = ycur[tid]; yz[tid] = 0.0; }
            } else reduce_records((int)(be & 1u), 6);
        }
        if (tid == 0) {
            // The three axes are independent systems with their own b . D^-1 b.  An axis whose right-hand side vanishes or is
            // > 15 orders below the largest one is measured against the largest one.  b = 0 altogether: the solution is x = 0.
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
            if (act0 == 1) {   // the epilogue expects u = D^-1 r
#pragma unroll
                for (int j = 0; j < 3; ++j) ru[j] = rd[j] * rr[j];
                conv = true; break;
            }
        }
        bool fresh = true;
        int restarts = 0;
        if (prof) a.prof[63 * 8 + 2] = wall_clock64();
        // TRUE residual at the current x (into u = D^-1 r).  1: it meets the tolerance, or the FP64 floor is reached (a failed
        // verification that did not improve on the previous one by 4x); 0: it does not -- CG restarts from it; -1: aborted
        auto verify = [&]() -> int {
            double q[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            if (!true_residual(false, q, nullptr)) return -1;
            ++be;
            publish_record(q, zero3, false, (int)(be & 1u));
            if (!oc_barrier(bar, be, a.G, ok_lds, a.sig)) return -1;
            reduce_records((int)(be & 1u), 3);
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
                ctl[0] = tr; ictl[0] = 0;
                ictl[2] = done ? 1 : 0;
            }
            return action() == 1 ? 1 : 0;
        };
        bool entry_restart = false, go_classic = false;
        AinvRows ar;   // this block's rows of Ac^-1 stay in registers for the whole loop
        if (two_level) ainv_prefetch(ar);
        // ---- pipelined CG (general recurrences of Ghysels & Vanroose: r and q = M^-1 s carried explicitly) with the
        //      two-level (or, without a coarse space, the Jacobi) preconditioner.  The recurrences are trusted for nine orders
        //      of magnitude per PASS (kOcPipeFloor): a pass ends when the recursive residual reports the tolerance -- or
        //      1e-9 of the pass's starting residual, or stagnates -- and is followed by a verification on the TRUE residual;
        //      if that fails the next pass restarts from the true residual (residual replacement), so tolerances below 1e-9
        //      (UzawaCG's inner solves) still run in this form.  Anything irregular hands over to the classic form below ----
        {
            // u = M^-1 r, w = A u, y_w from the residual in rr / ru = D^-1 r (the state a pass starts from)
            auto start_pass = [&](bool have_uw) -> bool {   // (have_uw: the start phase left u, w and y_w behind)
                if (!have_uw) {
                    double y[3];
#pragma unroll
                    for (int j = 0; j < 3; ++j) rr[j] = live ? ru[j] * fast_rcp(rd[j]) : 0.0;
                    smooth(rr, ru);
                    if (two_level) {
                        if (!coarse_apply(rr, y)) return false;
#pragma unroll
                        for (int j = 0; j < 3; ++j) ru[j] += y[j];
                    }
                    ++ph; ++be; publish(ru);
                    if (!oc_barrier(bar, be, a.G, ok_lds, a.sig)) return false;
                    halo_and_rows(ru, rw);
                    __syncthreads();
                }
                if (two_level && !have_uw) {
                    double y[3];
                    if (!coarse_apply(rw, y)) return false;
                    if (tid < 3 * kOcSubK) { yw[ywp + tid] = ycur[tid]; yz[tid] = 0.0; }
                    __syncthreads();
                }
                if (!have_uw) smooth(rw, sw);      // (have_uw: formed behind the start phase's last barrier)
#pragma unroll
                for (int j = 0; j < 3; ++j) sz[j] = 0.0;
                return true;
            };
            int passes = 0;
            double pass_start = 0.0;     // sum over the axes of r . D^-1 r / b . D^-1 b at the start of the pass
#pragma unroll
            for (int j = 0; j < 3; ++j) pass_start = fma(bc[j], ctl[8 + j], pass_start);
            bool have_uw = true;         // the start phase above left u = M^-1 r and w = A u
            while (!aborted && !conv && !go_classic && iters < a.max_iters) {
                if (!start_pass(have_uw)) { aborted = true; break; }
                have_uw = false;
                const double target = fmax(kOcTrig * a.tol2, kOcPipeFloor * pass_start);
                // A FIRST pass that starts from the true residual and reports the tolerance within kOc2TrustIters iterations is
                // believed without the verification exchange (~25 us per solve): the gap between the recursive and the true
                // residual of pipelined CG grows with the local rounding errors, ~ iterations x eps x |A| |x| -- after <= 40
                // iterations seven orders below a tolerance >= 1e-10 (tests: the bench-tolerance solves against exact solves,
                // test_short_pass_needs_no_verification).  Tighter tolerances, later passes (they start after a FAILED verification),
                // floor-limited targets and ADMM_HIP_OC_VERIFY=1 verify as before.
                // The rule is an estimate, so it is CHECKED: the host withholds the trust from every 16th solve (and a context's first 40); if such
                // a solve's short first pass then fails its verification, block 0 revokes the trust for the context (counters[76]) and every later
                // solve verifies.  (A 1 k-vertex body at pcg_tol 1e-10: unverified 5e-6 from the 1e-13 trajectory after eight frames, verified 8e-9
                // -- experiments/r05_small_body_accuracy.py; the 1 M-tet bench body never fails one: its trajectory is bit-identical either way.)
                const bool trusted = ADMM_OC2_TRUST && a.trust_short && ictl[3] == 0 && passes == 0 && a.tol2 >= kOc2TrustTol2 && target == kOcTrig * a.tol2;
                const int pass_it0 = iters;
                double rho_best = 1e300;
                int since = 0;
                bool next_pass = false;
                while (iters < a.max_iters) {
                    OC2_STAMP(0);
                    double rn[3], sn[3];
                    {
                        double mm[3];
                        const int oa_ = opq(oa);
#pragma unroll
                        for (int j = 0; j < 3; ++j) mm[j] = oa_ >= 0 ? sw[j] + (two_level ? prolong(yw + ywp, j) : 0.0) : 0.0;   // m = M^-1 w = S w + P Ac^-1 P^T w
                        ++ph; publish(mm);
                    }
                    OC2_STAMP(1);
                    if (a.nbr) {
                        if (!oc_announce_and_wait_neighbours<false>(bar, a.flags, a.nbr, (unsigned)a.seq, ph, ok_lds, a.sig)) { aborted = true; break; }
                    } else {   // more than 64 neighbour blocks somewhere: a grid barrier orders the exchange
                        ++be;
                        if (!oc_barrier(bar, be, a.G, ok_lds, a.sig)) { aborted = true; break; }
                    }
                    OC2_STAMP(2);
                    halo_and_rows_self_from_vec(rn);                                             // n = A m (m's own entry is in the local vector)
                    OC2_STAMP(3);
                    ++be;
                    const int par = (int)(be & 1u);
                    {   // (the sums just before their record, not across the exchange and the row loop: 14 VGPRs)
                        double q[7];
                        q[6] = 0.0;
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            q[j] = rr[j] * ru[j];                                                // gamma = r . u
                            q[3 + j] = rw[j] * ru[j];                                            // delta = w . u
                            q[6] = fma(rr[j] * rd[j] * rr[j], ctl[8 + j], q[6]);                 // Jacobi-norm residual (stop test)
                        }
                        publish_record(q, rn, two_level, par);
                    }
                    OC2_STAMP(4);
                    // S n (the block-local part of M^-1 n, data of this block only) behind the latency of the grid barrier:
                    // S w is then carried by the recurrences below like w itself, no smoothing on the critical path
                    oc2_barrier_arrive(bar);
                    smooth(rn, sn);
                    if (!oc_barrier_wait(bar, be, a.G, ok_lds, a.sig)) { aborted = true; break; }
                    OC2_STAMP(5);
                    if (two_level) reduce_and_coarse(par, 7, ar);                                // the sums, and ycur = Ac^-1 P^T n
                    else reduce_records(par, 7);
                    OC2_STAMP(6);
                    if (otid() < 64) {
                        const int lane = otid();
                        const int j = lane < 3 ? lane : 0;
                        const double g = bc[j], d = bc[3 + j], rs = bc[6];
                        const unsigned long long m3 = 7ull;
                        const bool finite = (__ballot(g < 1e290 && g >= 0.0 && d < 1e290) & m3) == m3 && rs < 1e290 && !(rs > 1e16 * rho_best);
                        since = rs < rho_best ? 0 : since + 1;
                        int act = 0;
                        if (!finite) act = 2;
                        else if (rs <= target) act = (trusted && iters - pass_it0 <= kOc2TrustIters) ? 4 : 1;
                        else if (since >= kOcStagnation) act = 1;
                        else if (lane < 3) {
                            double alpha, beta;
                            if (fresh) { beta = 0.0; alpha = (d > 0.0) ? g / d : 0.0; }
                            else {
                                const double gp = sc[j], ap = sc[3 + j];
                                beta = (gp > 0.0) ? g / gp : 0.0;
                                const double den = (ap != 0.0) ? d - beta * g / ap : d;
                                alpha = (den > 0.0) ? g / den : 0.0;
                            }
                            sc[j] = g; sc[3 + j] = alpha; glast[j] = g;
                            ctl[2 + j] = alpha; ctl[5 + j] = beta;
                        }
                        rho_best = fmin(rho_best, rs);
                        if (lane == 0) ictl[2] = act;
                    }
                    const int act = action();
                    if (act == 2) {
                        if (blockIdx.x == 0 && otid() == 0 && smoothing) a.counters[75] = 1;
                        entry_restart = true; go_classic = true; break;
                    }
                    if (act == 4) {                  // converged by the recursive residual of a short first pass: no verification
#pragma unroll
                        for (int j = 0; j < 3; ++j) ru[j] = rd[j] * rr[j];
                        conv = true; break;
                    }
                    if (act == 1) {
                        const int v = verify();      // leaves u = D^-1 (true residual)
                        if (v < 0) { aborted = true; break; }
                        if (v == 1) { conv = true; break; }
                        if (ADMM_OC2_TRUST_SAMPLE && passes == 0 && a.tol2 >= kOc2TrustTol2 && iters - pass_it0 <= kOc2TrustIters && blockIdx.x == 0 && otid() == 0) { a.counters[76] = 1; atomicAdd(a.counters + 77, 1); }
                        fresh = true;                // the true residual replaces the recursive one: beta = 0
                        if (++passes >= 4) { go_classic = true; break; }
                        pass_start = 3.0 * ctl[1];   // (the largest axis ratio of the verification, as a bound of the sum)
                        next_pass = true;
                        break;
                    }
                    const int oa_ = opq(oa);
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const double alpha = ctl[2 + j], beta = ctl[5 + j];
                        // (m is formed again from S w and y_w, bit for bit what was published, instead of being held across the
                        // exchange, the row loops and the reductions: 6 VGPRs; y_w is double-buffered by iteration parity so that
                        // its update by threads 0..11 below cannot overtake these reads)
                        const double mmj = oa_ >= 0 ? sw[j] + (two_level ? prolong(yw + ywp, j) : 0.0) : 0.0;
                        rz[j] = fma(beta, rz[j], rn[j]);
                        sz[j] = fma(beta, sz[j], sn[j]);
                        sw[j] = fma(-alpha, sz[j], sw[j]);
                        rq[j] = fma(beta, rq[j], mmj);
                        rsv[j] = fma(beta, rsv[j], rw[j]);
                        rp[j] = fma(beta, rp[j], ru[j]);
                        rx[j] = fma(alpha, rp[j], rx[j]);
                        rr[j] = fma(-alpha, rsv[j], rr[j]);
                        ru[j] = fma(-alpha, rq[j], ru[j]);
                        rw[j] = fma(-alpha, rz[j], rw[j]);
                    }
                    if (two_level && otid() < 3 * kOcSubK) {
                        const int tid = otid();
                        const int j = tid % 3;
                        const double zz = fma(ctl[5 + j], yz[tid], ycur[tid]);
                        yz[tid] = zz;
                        yw[(ywp ^ (3 * kOcSubK)) + tid] = fma(-ctl[2 + j], zz, yw[ywp + tid]);
                    }
                    ywp ^= 3 * kOcSubK;
                    __syncthreads();   // yw is read, ctl / bc / ycur / the local vector are rewritten by the next iteration
                    ++iters; ++pipe_iters; fresh = false;
                    OC2_STAMP(7);
                    if (prof) ++prof_n;
                    OC2_MARK(8);
                }
                if (!next_pass) break;
            }
            if (!conv && !go_classic && !aborted) {   // iteration cap: leave u = D^-1 r behind (epilogue, recycled pair)
#pragma unroll
                for (int j = 0; j < 3; ++j) ru[j] = rd[j] * rr[j];
            }
            if (aborted || conv || !go_classic) break;
        }
        // ---- classic (Hestenes-Stiefel) CG with the Jacobi preconditioner, three synchronisations per iteration: gamma =
        // r . u, then p, s = A p and delta = p . s computed directly (stable end game on ill-conditioned systems) ----
        while (iters < a.max_iters) {
            if (entry_restart) {
                // Non-finite or runaway sums: back to the entry x (still in global memory) and its true residual.  A second
                // failure gives up: the solve is reported as unconverged and hands back the entry x.
                entry_restart = false;
#pragma unroll
                for (int j = 0; j < 3; ++j) rx[j] = live ? a.x[3 * (size_t)vi + j] : 0.0;
                double q[7];
                if (!true_residual(false, q, nullptr)) { aborted = true; break; }
                if (tid == 0) { ctl[0] = 1e300; ictl[0] = 0; }
                if (++restarts > 1) break;
                fresh = true;
            }
            double q[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int j = 0; j < 3; ++j) q[j] = (live ? ru[j] * ru[j] * fast_rcp(rd[j]) : 0.0);
            ++be;
            publish_record(q, zero3, false, (int)(be & 1u));
            if (!oc_barrier(bar, be, a.G, ok_lds, a.sig)) { aborted = true; break; }
            reduce_records((int)(be & 1u), 3);
            if (wv == 0) {   // lanes 0..2 = one axis each
                const int j = lane < 3 ? lane : 0;
                const double g = bc[j], gbj = gbl[j];
                const double ratio = g * ctl[8 + j];
                const unsigned long long m3 = 7ull;
                const bool finite = (__ballot(g < 1e290 && ratio < 1e290 && !(ratio > 1e16 * ctl[0])) & m3) == m3;
                const bool below_tol = (__ballot(g <= a.tol2 * gbj + 1e-300) & m3) == m3;
                double rmax = fmax(ratio, __shfl(ratio, 1, 64));
                rmax = fmax(rmax, __shfl(ratio, 2, 64));
                int act = 0;
                if (!finite) act = 2;
                else if (below_tol) act = 1;
                else {
                    if (lane == 0 && rmax < ctl[0]) ctl[0] = rmax;
                    if (lane < 3) {
                        const double gp = sc[j];
                        const double beta = (!fresh && gp > 0.0) ? g * fast_rcp(gp) : 0.0;
                        sc[j] = g; sc[3 + j] = 0.0; glast[j] = g;
                        ctl[5 + j] = beta;
                    }
                }
                if (lane == 0) ictl[2] = act;
            }
            const int act = action();
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
            ++ph; ++be; publish(rp);
            if (!oc_barrier(bar, be, a.G, ok_lds, a.sig)) { aborted = true; break; }
            halo_and_rows(rp, rsv);                                                // s = A p
#pragma unroll
            for (int j = 0; j < 3; ++j) { q[j] = 0.0; q[3 + j] = rp[j] * rsv[j]; }
            ++be;
            publish_record(q, zero3, false, (int)(be & 1u));
            if (!oc_barrier(bar, be, a.G, ok_lds, a.sig)) { aborted = true; break; }
            reduce_records((int)(be & 1u), 6);
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
#undef OC2_STAMP
    // this solve's pair: e = x - x_entry, A e = r_entry - r_final (exact: u carries the true residual).  Formed BEFORE the end projection below,
    // which moves x without updating r; STORED behind it: on this hardware a wave's loads return in order with its stores, so six stores per row
    // in front of the projection's loads cost it 10 us (measured: ADMM_HIP_OC_PROF)
    // ALL global loads of the epilogue are issued here, before the first use of any (a wave's loads return in order): the pair's inputs, the
    // wave's modes for the dots (up to three of them), the row's own entries of every mode, G^-1.  As four dependent stages -- pair, G^-1 -> LDS,
    // first mode, second mode -- the same loads cost four round trips to HBM (ADMM_HIP_OC_PROF: 18.8 us for "pair stores + dots").
    constexpr int SPBMAX = MAXT / 64;
    constexpr int MPW = (kOc2DeflMax + (MAXT / 64) - 1) / (MAXT / 64) + 1;      // modes per wave: 32 modes over 12 (16) waves, +1 when fewer waves run
    const bool proj = a.defl_k > 0 && conv && !aborted;
    const int K = a.defl_k;
    double xs_in[3] = {0.0, 0.0, 0.0}, r0_in[3] = {0.0, 0.0, 0.0};
    if (live && a.rc_on) {
#pragma unroll
        for (int j = 0; j < 3; ++j) { const size_t i = 3 * (size_t)row + j; xs_in[j] = a.rc_xs[i]; r0_in[j] = a.rc_r0[i]; }
    }
    float zw[MPW][SPBMAX], zmine[kOc2DeflMax];
    double gv[2] = {0.0, 0.0};
    if (proj) {
        const size_t zrow0 = (size_t)blockIdx.x * (size_t)T;
#pragma unroll
        for (int u = 0; u < MPW; ++u) {
            const int q = wv + u * nw;
            const float *zq = a.defl_Z + (size_t)(q < K ? q : 0) * a.n_rows + zrow0 + lane;
#pragma unroll
            for (int i = 0; i < SPBMAX; ++i) zw[u][i] = (q < K && i < a.spb && !(a.defl_dbg & 1)) ? zq[64 * i] : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < kOc2DeflMax; ++q) zmine[q] = (live && q < K && !(a.defl_dbg & 2)) ? a.defl_Z[(size_t)q * a.n_rows + row] : 0.0f;
        if (!(a.defl_dbg & 4)) { if (tid < K * K) gv[0] = a.defl_Ginv[tid]; if (tid + T < K * K) gv[1] = a.defl_Ginv[tid + T]; }
    }
    double pe[3] = {0.0, 0.0, 0.0}, pr[3] = {0.0, 0.0, 0.0};
    if (live && a.rc_on) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            pe[j] = rx[j] - xs_in[j];
            pr[j] = r0_in[j] - ru[j] * fast_rcp(rd[j]);
        }
    }
    if (proj) {
        // ---- end projection on the soft modes: x += Z G^-1 Z^T r (one more all-to-all) ----
        // r of the block's rows -> the (idle) local vector; wave w takes the modes w, w + nw, ...: lanes over the block's rows, ONE wave sum
        // per mode and axis (every thread summing every mode's product would cost 6 x 96 lane exchanges per wave: measured on k_big_vec).
        const int dpar = a.seq & 1;
        {
            const int t = otid();
#pragma unroll
            for (int j = 0; j < 3; ++j) vec[OC2_VX(t, j)] = live ? ru[j] * fast_rcp(rd[j]) : 0.0;
        }
        // (G^-1 into LDS: the slab's first K K doubles are not needed any more)
        LdsD *ginv_l = lv_all;
        if (!(a.defl_dbg & 4)) {
            if (tid < K * K) ginv_l[tid] = gv[0];
            if (tid + T < K * K) ginv_l[tid + T] = gv[1];
            for (int o = tid + 2 * T; o < K * K; o += T) ginv_l[o] = a.defl_Ginv[o];      // (small blocks, many modes)
        }
        __syncthreads();
        __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc((void *)a.defl_rec, 0, 2 * 3 * kOc2DeflMax * a.G * 8, 0x00020000);
        // modes: [mode][internal row], single precision.  (A [block][mode][row] layout, one contiguous 74-KB slice per block, was SLOWER: 44 us.)
#pragma unroll
        for (int u = 0; u < MPW; ++u) {
            const int q = wv + u * nw;
            if (q < K) {
                double acc[3] = {0.0, 0.0, 0.0};
#pragma unroll
                for (int i = 0; i < SPBMAX; ++i)
                    if (i < a.spb) {
                        const int rl = lane + 64 * i;
                        const double zd = (double)zw[u][i];
                        acc[0] = fma(zd, vec[OC2_VX(rl, 0)], acc[0]); acc[1] = fma(zd, vec[OC2_VX(rl, 1)], acc[1]); acc[2] = fma(zd, vec[OC2_VX(rl, 2)], acc[2]);
                    }
                acc[0] = wave_sum(acc[0]); acc[1] = wave_sum(acc[1]); acc[2] = wave_sum(acc[2]);
                if (lane < 3) oc_store_sc1(rs_d, ((dpar * 3 * kOc2DeflMax + 3 * q + lane) * a.G + (int)blockIdx.x) * 8, lane == 0 ? acc[0] : lane == 1 ? acc[1] : acc[2]);
            }
        }
        for (int q = wv + MPW * nw; q < K; q += nw) {      // (blocks of fewer waves than the instance allows: the remaining modes, plainly)
            const float *zq = a.defl_Z + (size_t)q * a.n_rows + (size_t)blockIdx.x * (size_t)T + lane;
            double acc[3] = {0.0, 0.0, 0.0};
            for (int i = 0; i < a.spb; ++i) {
                const int rl = lane + 64 * i;
                const double zd = (a.defl_dbg & 1) ? 0.0 : (double)zq[64 * i];
                acc[0] = fma(zd, vec[OC2_VX(rl, 0)], acc[0]); acc[1] = fma(zd, vec[OC2_VX(rl, 1)], acc[1]); acc[2] = fma(zd, vec[OC2_VX(rl, 2)], acc[2]);
            }
            acc[0] = wave_sum(acc[0]); acc[1] = wave_sum(acc[1]); acc[2] = wave_sum(acc[2]);
            if (lane < 3) oc_store_sc1(rs_d, ((dpar * 3 * kOc2DeflMax + 3 * q + lane) * a.G + (int)blockIdx.x) * 8, lane == 0 ? acc[0] : lane == 1 ? acc[1] : acc[2]);
        }
        // (this row's entries of Z for the update below stay in registers across the grid barrier)
        if (prof) a.prof[63 * 8 + 5] = wall_clock64();
        ++be;
        if (!oc_barrier(bar, be, a.G, ok_lds, a.sig)) aborted = true;
        else {
            if (prof) a.prof[63 * 8 + 6] = wall_clock64();
            // the blocks' sums, every block in the same order: wave w the quantities w, w + nw, ... (<= 8 of 96 with 12 waves), four blocks per lane
            constexpr int QW = (3 * kOc2DeflMax + 11) / 12 + 1;
            double part[QW][4];
#pragma unroll
            for (int u = 0; u < QW; ++u) {
                const int k = wv + u * nw;
#pragma unroll
                for (int i = 0; i < 4; ++i) { const int g = lane + 64 * i; part[u][i] = (k < 3 * K && g < a.G) ? oc_load_sc1_f64(rs_d, ((dpar * 3 * kOc2DeflMax + k) * a.G + g) * 8) : 0.0; }
            }
#pragma unroll
            for (int u = 0; u < QW; ++u) {
                const int k = wv + u * nw;
                if (k < 3 * K) {
                    double sm = (part[u][0] + part[u][1]) + (part[u][2] + part[u][3]);
                    for (int g = lane + 256; g < a.G; g += 64) sm += oc_load_sc1_f64(rs_d, ((dpar * 3 * kOc2DeflMax + k) * a.G + g) * 8);
                    sm = wave_sum(sm);
                    if (lane == 0) red[k] = sm;
                }
            }
            for (int k = wv + QW * nw; k < 3 * K; k += nw) {      // (fewer than 12 waves: the rest, plainly)
                double sm = 0.0;
                for (int g = lane; g < a.G; g += 64) sm += oc_load_sc1_f64(rs_d, ((dpar * 3 * kOc2DeflMax + k) * a.G + g) * 8);
                sm = wave_sum(sm);
                if (lane == 0) red[k] = sm;
            }
            __syncthreads();
            if (prof) a.prof[63 * 8 + 7] = wall_clock64();
            for (int o = tid; o < 3 * K; o += T) {       // y = G^-1 d, from LDS
                const int q = o / 3, ax = o - 3 * q;
                double acc = 0.0;
                for (int pp = 0; pp < K; ++pp) acc = fma(ginv_l[q * K + pp], red[3 * pp + ax], acc);
                red[3 * kOc2DeflMax + o] = acc;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < kOc2DeflMax; ++q) {
                if (q < K) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) rx[j] = fma((double)zmine[q], red[3 * kOc2DeflMax + 3 * q + j], rx[j]);
                    if (a.defl_dbg & 8) {      // the pair carries the soft step too: e += Z y, A e += K Z y ~ Z (G y) = Z d (Z: Ritz vectors of K)
#pragma unroll
                        for (int j = 0; j < 3; ++j) { pe[j] = fma((double)zmine[q], red[3 * kOc2DeflMax + 3 * q + j], pe[j]); pr[j] = fma((double)zmine[q], red[3 * q + j], pr[j]); }
                    }
                }
            }
        }
    }
    if (live) {
#pragma unroll
        for (int j = 0; j < 3; ++j) { a.x[3 * (size_t)vi + j] = rx[j]; a.u_out[3 * (size_t)vi + j] = ru[j]; }
        if (a.rc_on) {
#pragma unroll
            for (int j = 0; j < 3; ++j) { const size_t i = 3 * (size_t)row + j; a.rc_Eslot[i] = pe[j]; a.rc_Rslot[i] = pr[j]; }
        }
    }
    if (blockIdx.x == 0 && tid == 0) {
        CgScal o;
#pragma unroll
        for (int j = 0; j < 3; ++j) { o.gamma[j] = glast[j]; o.alpha[j] = 0.0; o.gamma_b[j] = gbl[j]; }
        o.alpha[0] = (double)ictl[1]; o.alpha[1] = (double)be; o.alpha[2] = ctl[1] / a.tol2;
        o.converged = (conv && !aborted) ? 1 : 0; o.iters = iters; o.seq = a.seq; o.pad_ = pipe_iters;
        a.scal[0] = o;
        atomicAdd(a.counters, iters);
        if (o.converged) {
            atomicAdd(a.counters + 4, 1);
            atomicMax(a.counters + 3, iters);
            a.counters[8 + (a.seq & 63)] = iters;
        }
        // totals since admm_hip_create (never reset: admm_hip_solve_totals)
        atomicAdd(a.counters + 72, 1); atomicAdd(a.counters + 73, o.converged); atomicAdd(a.counters + 74, iters);
    }
    if (prof) a.prof[63 * 8 + 4] = wall_clock64();     // (the profiled block's own epilogue)
}

// Latency floor of the two synchronisations an iteration of k_pcg2 is made of, measured on the same grid with the same
// primitives and payloads but no arithmetic (admm_hip_probe_sync, bench.py "roofline_global"):
//   mode 0: n x { 19 block sums -> record -> grid barrier -> read all records }          (the all-to-all)
//   mode 1: n x { publish 32 B per row -> neighbour flags -> halo fetch into LDS }        (the vector exchange)
template <int MAXT>
__global__ __launch_bounds__(MAXT) void k_sync_probe(Oc2Args a, int n, int mode, double *sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *ok_lds = (int *)(smem + 3632);
    double *bc = (double *)(smem + 3328);
    const int T = (int)blockDim.x, tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = T >> 6;
    const int NV = a.vec_len;
    LdsD *vec = (LdsD *)(smem + kOc2Scratch);
    const int s = (int)blockIdx.x * a.spb + wv, row = s * 64 + lane;
    const int hp0 = a.halo_ptr[blockIdx.x], nh = a.halo_ptr[blockIdx.x + 1] - hp0;
    const int ub = a.n_rows * 32;
    __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc((void *)a.ubuf, 0, 2 * ub, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc((void *)a.part, 0, 2 * 8 * a.G * 8, 0x00020000);
    unsigned *const bar = a.bar + 32 * 16 * (a.seq & 1);
    if (blockIdx.x == 0 && tid < 9) a.bar[32 * 16 * ((a.seq & 1) ^ 1) + 16 * (tid < 8 ? tid : 17)] = 0u;
    double acc = (double)row;
    unsigned ph = 0, be = 0;
    for (int it = 0; it < n; ++it) {
        if (mode == 0) {
            ++be;
            const int par = (int)(be & 1u);
#if ADMM_OC2_REC_CHUNK      // (as k_pcg2: publish_record, rec_issue, rec_reduce)
            if (tid < 8) oc_store_sc1(rs_p, ((par * a.G + (int)blockIdx.x) * 8 + tid) * 8, acc);
            if (!oc_barrier(bar, be, a.G, ok_lds, a.sig)) break;
            {
                double sa = 0.0, sb = 0.0;
                for (int o = tid; o < 4 * a.G; o += T) {
                    union { double d[2]; v4u v; } g;
                    g.v = __builtin_amdgcn_raw_buffer_load_b128(rs_p, par * a.G * 64 + o * 16, 0, 16);
                    sa += g.d[0]; sb += g.d[1];
                }
                sa += dpp_f64<0x114>(sa); sb += dpp_f64<0x114>(sb);
                sa += dpp_f64<0x118>(sa); sb += dpp_f64<0x118>(sb);
                sa += __shfl_xor(sa, 16, 64); sb += __shfl_xor(sb, 16, 64);
                sa += __shfl_xor(sa, 32, 64); sb += __shfl_xor(sb, 32, 64);
                double *red2 = (double *)(smem + kOc2Scratch);
                if (lane >= 12 && lane < 16) { red2[wv * 8 + 2 * (lane & 3)] = sa; red2[wv * 8 + 2 * (lane & 3) + 1] = sb; }
                __syncthreads();
                if (tid < 7) { double sm = 0.0; for (int k = 0; k < nw; ++k) sm += red2[k * 8 + tid]; bc[tid] = sm; }
            }
#else
            if (tid < 7) oc_store_sc1(rs_p, ((par * 8 + tid) * a.G + (int)blockIdx.x) * 8, acc);
            if (!oc_barrier(bar, be, a.G, ok_lds, a.sig)) break;
            if (wv < 7) {
                double sm = 0.0;
                for (int g = lane; g < a.G; g += 64) sm += oc_load_sc1_f64(rs_p, ((par * 8 + wv) * a.G + g) * 8);
                sm = wave_sum(sm);
                if (lane == 0) bc[wv] = sm;
            }
#endif
            __syncthreads();
            acc = acc * 0.5 + bc[0] * 1e-300;
        } else if (mode >= 2) {
            // (experiments, ADMM_HIP_PROBE_A2A_MODE: the block's 7 sums as ONE 64-byte chunk [parity][block][8] -- whole sectors of its own -- instead of
            // [parity][sum][block], where four blocks share a sector.  2: read sum by sum (64-byte stride); 3: read chunk-wise, one 16-byte load per thread;
            // 4: as 3, stored with four 16-byte stores)
            ++be;
            const int par = (int)(be & 1u);
            if (mode == 4) { if (tid < 4) oc_store_sc1(rs_p, ((par * a.G + (int)blockIdx.x) * 8 + 2 * tid) * 8, acc, acc); }
            else if (tid < 8) oc_store_sc1(rs_p, ((par * a.G + (int)blockIdx.x) * 8 + tid) * 8, acc);
            if (!oc_barrier(bar, be, a.G, ok_lds, a.sig)) break;
            if (mode == 2) {
                if (wv < 7) {
                    double sm = 0.0;
                    for (int g = lane; g < a.G; g += 64) sm += oc_load_sc1_f64(rs_p, ((par * a.G + g) * 8 + wv) * 8);
                    sm = wave_sum(sm);
                    if (lane == 0) bc[wv] = sm;
                }
            } else {
                double sm = 0.0;
                for (int o = tid; o < 4 * a.G; o += T) {
                    union { double d[2]; v4u v; } g;
                    g.v = __builtin_amdgcn_raw_buffer_load_b128(rs_p, par * a.G * 64 + o * 16, 0, 16);
                    sm += g.d[0] + g.d[1];
                }
                sm = wave_sum(sm);
                if (lane == 0 && wv < 7) bc[wv] = sm;
            }
            __syncthreads();
            acc = acc * 0.5 + bc[0] * 1e-300;
        } else {
            ++ph;
            LdsD *o = vec + wv * 64;
            o[lane] = acc; o[NV + lane] = acc; o[2 * NV + lane] = acc;
            const int bo = (int)(ph & 1u) * ub + s * 2048 + lane * 16;
            oc_store_sc1(rs_u, bo, o[lane >> 1], o[NV + (lane >> 1)]);
            oc_store_sc1(rs_u, bo + 1024, o[32 + (lane >> 1)], o[NV + 32 + (lane >> 1)]);
            if (!oc_announce_and_wait_neighbours<false>(bar, a.flags, a.nbr, (unsigned)a.seq, ph, ok_lds, a.sig)) break;
            for (int h = tid; h < nh; h += T) {
                const int src = a.halo_src[hp0 + h];
                union { double d[2]; v4u v; } g0, g1;
                g0.v = __builtin_amdgcn_raw_buffer_load_b128(rs_u, (int)(ph & 1u) * ub + src * 32, 0, 16);
                g1.v = __builtin_amdgcn_raw_buffer_load_b128(rs_u, (int)(ph & 1u) * ub + src * 32 + 16, 0, 16);
                vec[T + h] = g0.d[0]; vec[NV + T + h] = g0.d[1]; vec[2 * NV + T + h] = g1.d[0];
            }
            __syncthreads();
            acc = acc * 0.5 + vec[T + (tid % (nh > 0 ? nh : 1))] * 1e-300;
            __syncthreads();
        }
    }
    if (sink && acc == 12345.678) sink[0] = acc;   // keeps the loop alive
    (void)nw;
}

} // namespace admm_k
