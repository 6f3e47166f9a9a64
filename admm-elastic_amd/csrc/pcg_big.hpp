// pcg_big.hpp -- the LAUNCH-PATH two-level PCG: the global solve (src/LinearSolver.hpp:87-90, the prefactored LDLT the north-star replaces
// by a GPU sparse solve) of systems that do not fit the chip's LDS -- more than 262 144 vertices, where k_pcg2 (pcg_onchip2.hpp) cannot hold
// matrix and vectors on chip -- and the fall-back of k_pcg2 after a barrier time-out.  Rounds 1-4 served those systems with Jacobi-PCG
// (k_cg_*: 7x the iterations); this is the SAME preconditioner as the on-chip solver's,
//     M^-1 = D^-1 + P (P^T A P)^-1 P^T,   P = {1, x, y, z} per compact aggregate of ~768 vertices, energy-orthonormalised (oc_plan.cpp: build_big_plan),
// in kernels that stream the matrix (SELL-64, internal row order = aggregate by aggregate) -- bound by HBM / L2 bandwidth, three launches
// per iteration:
//     k_big_spmv   w = A u, partial sums of gamma = r.u and delta = u.w per thread block                      (12 B per non-zero + 72 B per row)
//     k_big_vec    alpha, beta from the partials (every block re-reduces them: fixed order, no atomics); p = u + beta p, s = w + beta s,
//                  x += alpha p, r -= alpha s; c = P^T r and rho = r.D^-1 r of the block's aggregate         (10 vector passes: 240 B per row)
//     k_big_coarse y = (P^T A P)^-1 c (the aggregate's four rows of the dense single-precision inverse, accumulated in FP64); stop test on
//                  rho (the Jacobi-scaled residual norm, the documented meaning of pcg_tol); u = D^-1 r + P y   (4 nc floats + 24 nc B per block)
// Conjugate gradients in the Chronopoulos-Gear form (one reduction point per iteration), the three axes as three systems sharing Ahat
// (own alpha / beta per axis), like k_cg_*.  Vectors live in the internal order; b is gathered and x scattered through `orig` once per
// solve.  Deterministic: every sum has a fixed order.
#pragma once
#include "kernels.hpp"

namespace admm_k {

struct BigArgs {
    SellA A;                                      // Ahat, internal order, global (internal) columns
    const double *mass, *dinv, *cwt; const float *ainv;
    const int *orig;
    int n_rows, NBt, G, ra, nc, ncp;
    const double *b_api; double *x_api, *u_api;   // API order: right-hand side, solution (in/out), M_jacobi^-1 r_final for the recycled pairs
    const double *dinv_api;
    double *xi, *r, *u, *w, *p, *s;               // internal order [3 n_rows]
    double *part;                                 // [6][NBt]: gamma / delta partials of k_big_spmv; [3][NBt] of the entry residual's b.D^-1 b
    double *dots; int *tick;                      // [6]: their sums, formed by the LAST block of the kernel that wrote the partials (ticket counter)
    double *cvec;                                 // [3][ncp]: c = P^T r
    double *rho;                                  // [3][G]: r.D^-1 r per aggregate
    CgScal *scal;                                 // two slots, alternating by iteration parity
    int *counters; int *sig;
    double tol2; int seq;
    int row_lo, row_hi;                           // distributed solve: the internal rows this rank owns (aggregate-aligned)
};

constexpr int kBigVecT = 256;      // rocprofv3 at 2 M tets (profiles/r05_*_blob2m_launch_path_*.csv): 1024 threads whose EVERY thread formed every final sum
                                   // 70 us; 256 threads with three rows each 43 us; 768 threads, one row each, 54 us (k_big_coarse 21 -> 34 us) -- the kernel
                                   // streams 344 B per row from HBM (ten vectors of 8.5 MB), more waves per block only add barrier time
// block-wide sum of NQ quantities over any number of waves (<= 16): wave sums to LDS, the first NQ threads add them up, everybody reads
template <int NQ>
__device__ __forceinline__ void block_sum_wide(double *q, double *lds /* [16 NQ + NQ] */) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (int)(blockDim.x >> 6);
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const double s = wave_sum(q[i]);
        if (lane == 0) lds[wv * NQ + i] = s;
    }
    __syncthreads();
    if ((int)threadIdx.x < NQ) { double s = 0.0; for (int k = 0; k < nw; ++k) s += lds[k * NQ + threadIdx.x]; lds[16 * NQ + threadIdx.x] = s; }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NQ; ++i) q[i] = lds[16 * NQ + i];
    __syncthreads();
}


// The partial sums of a kernel's blocks, added up by whichever block finishes LAST (ticket counter), in the fixed order of the block
// index: one reduction per launch instead of one per consumer block (every block of k_big_vec re-reduced 6 NBt doubles -- 133 KB at
// 4 M tets).  Partials and ticket travel at agent scope (the XCDs' L2s are not coherent for plain accesses inside a kernel).
template <int NQ>
__device__ __forceinline__ void big_last_block_sums(const BigArgs &a, const double *q, double *lds /* [4 NQ] */) {
    __shared__ int last;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) __hip_atomic_store(a.part + (size_t)i * a.NBt + blockIdx.x, q[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (no release / acquire: at agent scope they write back and invalidate the whole L2 -- k_big_spmv 27 -> 77 us, measured.  The partials
        // are write-through stores, drained before the ticket is taken; the last block reads them with loads that bypass its L2.)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int t = __hip_atomic_fetch_add(a.tick, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = (t == (int)gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!last) return;
    double t[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) t[i] = 0.0;
    for (int b0 = threadIdx.x; b0 < a.NBt; b0 += 2 * 256) {      // 2 NQ loads in flight (more: registers the streaming part of k_big_spmv would pay for)
        double v[2][NQ];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int b = b0 + 256 * u;
#pragma unroll
            for (int i = 0; i < NQ; ++i) v[u][i] = b < a.NBt ? __hip_atomic_load(a.part + (size_t)i * a.NBt + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int i = 0; i < NQ; ++i) t[i] += v[u][i];
        }
    }
    block_sum<NQ>(t, lds);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) a.dots[i] = t[i];
        __hip_atomic_store(a.tick, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// x, b into the internal order; p = s = 0
__global__ __launch_bounds__(256) void k_big_gather(BigArgs a) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= a.n_rows) return;
    const int v = a.orig[row];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const size_t i = 3 * (size_t)row + j;
        a.xi[i] = v >= 0 ? a.x_api[3 * (size_t)v + j] : 0.0;
        a.p[i] = 0.0; a.s[i] = 0.0;
    }
}

// r = b - A x (b gathered through orig), partial sums of b.D^-1 b
__global__ __launch_bounds__(256) void k_big_resid(BigArgs a) {
    __shared__ double lds[12];
    if (blockIdx.x == 0 && threadIdx.x == 0) { a.scal[0].converged = 0; a.scal[0].iters = 0; a.scal[0].seq = a.seq; a.scal[0].pad_ = 0; }
    const int lane = threadIdx.x & 63;
    const int s = wave_slice();
    double q[3] = {0.0, 0.0, 0.0};
    if (s < a.A.n_slices) {
        const int row = s * 64 + lane;
        const bool own = row >= a.row_lo && row < a.row_hi;
        double acc[3] = {0.0, 0.0, 0.0};
        if (__any(own)) sell_row(a.A, s, lane, a.xi, acc);
        const int v = row < a.n_rows ? a.orig[row] : -1;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const size_t i = 3 * (size_t)row + j;
            if (row < a.n_rows) {
                const double bi = (v >= 0 && own) ? a.b_api[3 * (size_t)v + j] : 0.0;
                a.r[i] = (v >= 0 && own) ? bi - fma(a.mass[i], a.xi[i], acc[j]) : 0.0;
                q[j] = fma(bi * a.dinv[i], bi, q[j]);
            }
        }
    }
    block_sum<3>(q, lds);
    big_last_block_sums<3>(a, q, lds);
}

// w = A u ; partials gamma = r.u, delta = u.w
// (Measured and dropped: one block per aggregate with the aggregate's own entries of u staged in LDS -- four of five gathers become LDS reads --
// 768 / 1024 threads: 32.9 -> 39.0 us at 2 M tets, 76 -> 81 us at 4 M.  The product is not bound by its gathers: every SpMV-shaped kernel of
// the library takes ~80 us at 4 M tets whatever its row order -- ~2.8 TB/s of matrix + vectors once the working set leaves the 256-MB MALL,
// against 5.9 TB/s at 1 M tets where it fits.)
__global__ __launch_bounds__(256) void k_big_spmv(BigArgs a, int it) {
    __shared__ double lds[24];
    if (a.scal[it & 1].converged) return;
    const int lane = threadIdx.x & 63;
    const int s = wave_slice();
    double q[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (s < a.A.n_slices) {
        const int row = s * 64 + lane;
        const bool own = row >= a.row_lo && row < a.row_hi;
        if (__any(own)) {
            double acc[3];
            sell_row(a.A, s, lane, a.u, acc);
            if (row < a.n_rows && own) {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const size_t i = 3 * (size_t)row + j;
                    const double ui = a.u[i], wi = fma(a.mass[i], ui, acc[j]);
                    a.w[i] = wi;
                    q[j] = fma(a.r[i], ui, q[j]);
                    q[3 + j] = fma(wi, ui, q[3 + j]);
                }
            }
        }
    }
    block_sum<6>(q, lds);
    big_last_block_sums<6>(a, q, lds);
}

// (Measured and dropped: the rows' loads at the very top of k_big_vec / k_big_coarse, ahead of the scalars -- 222 / 165 VGPRs, one block per CU
// fewer resident: 44 -> 51 and 36 -> 57 us at 4 M tets, unchanged at 2 M.)
// it < 0: the entry pass (no update: c = P^T r, rho, and gamma_b = b.D^-1 b from the entry residual's partials).
// it >= 0: alpha / beta, the vector updates, then c and rho.  One block = one aggregate.
__global__ __launch_bounds__(kBigVecT) void k_big_vec(BigArgs a, int it, int mark_here) {
    __shared__ double lds[17 * 15];
    const bool entry = it < 0;
    const CgScal pv = a.scal[entry ? 0 : (it & 1)];
    CgScal *next = a.scal + (entry ? 0 : ((it + 1) & 1));
    if (mark_here && blockIdx.x == 0 && threadIdx.x == 0) {      // progress mark for the host: the chunk this kernel closes has (almost) drained
        const int m = atomicAdd(a.counters + 5, 1) + 1;
        __hip_atomic_store(a.sig + 1, m, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (!entry && pv.converged) {
        if (blockIdx.x == 0 && threadIdx.x == 0) *next = pv;
        return;
    }
    const int g = (int)blockIdx.x;
    double alpha[3] = {0.0, 0.0, 0.0}, beta[3] = {0.0, 0.0, 0.0};
    {   // the sums of the partials (k_big_resid / k_big_spmv's last block; distributed: all-reduced over the ranks): identical scalars everywhere
        double q[6];
#pragma unroll
        for (int kk = 0; kk < 6; ++kk) q[kk] = a.dots[kk];
        if (entry) {
            if (g == 0 && threadIdx.x == 0) {
                CgScal o = pv;
                for (int j = 0; j < 3; ++j) { o.gamma_b[j] = q[j]; o.gamma[j] = 0.0; o.alpha[j] = 0.0; }
                o.converged = 0; o.iters = 0; o.seq = a.seq;
                *next = o;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double gm = q[j], d = q[3 + j];
                if (it == 0) { beta[j] = 0.0; alpha[j] = (d > 0.0) ? gm / d : 0.0; }
                else {
                    beta[j] = (pv.gamma[j] > 0.0) ? gm / pv.gamma[j] : 0.0;
                    const double den = (pv.alpha[j] != 0.0) ? d - beta[j] * gm / pv.alpha[j] : d;
                    alpha[j] = (den > 0.0) ? gm / den : 0.0;
                }
            }
            if (g == 0 && threadIdx.x == 0) {
                CgScal o = pv;
                for (int j = 0; j < 3; ++j) { o.gamma[j] = q[j]; o.alpha[j] = alpha[j]; }
                o.converged = 0; o.iters = pv.iters + 1;
                *next = o;
                atomicAdd(a.counters, 1);
            }
        }
    }
    double q[15];
#pragma unroll
    for (int i = 0; i < 15; ++i) q[i] = 0.0;
    const int r0 = g * a.ra;
    if (r0 >= a.row_lo && r0 < a.row_hi) {
        // three rows per thread, ALL their loads issued before the first store (the compiler cannot move a load of u above a store to p:
        // as a plain loop the rows went one after the other, ~2 waves per SIMD hiding nothing)
        constexpr int RPT = 3;
        for (int base = r0 + (int)threadIdx.x; base < r0 + a.ra; base += RPT * kBigVecT) {
            double P[RPT][3], U[RPT][3], S[RPT][3], W[RPT][3], X[RPT][3], R[RPT][3], DI[RPT][3], CW[RPT][4];
            bool ok[RPT];
#pragma unroll
            for (int t = 0; t < RPT; ++t) {
                const int row = base + t * kBigVecT;
                ok[t] = row < r0 + a.ra;
                const size_t i0 = 3 * (size_t)(ok[t] ? row : r0);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    R[t][j] = a.r[i0 + j]; DI[t][j] = a.dinv[i0 + j];
                    if (!entry) { P[t][j] = a.p[i0 + j]; U[t][j] = a.u[i0 + j]; S[t][j] = a.s[i0 + j]; W[t][j] = a.w[i0 + j]; X[t][j] = a.xi[i0 + j]; }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) CW[t][k] = a.cwt[4 * (size_t)(ok[t] ? row : r0) + k];
            }
#pragma unroll
            for (int t = 0; t < RPT; ++t) {
                if (!ok[t]) continue;
                const size_t i0 = 3 * (size_t)(base + t * kBigVecT);
                double rr[3];
                if (entry) { rr[0] = R[t][0]; rr[1] = R[t][1]; rr[2] = R[t][2]; }
                else {
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const double pi = fma(beta[j], P[t][j], U[t][j]);
                        const double si = fma(beta[j], S[t][j], W[t][j]);
                        a.p[i0 + j] = pi; a.s[i0 + j] = si;
                        a.xi[i0 + j] = fma(alpha[j], pi, X[t][j]);
                        rr[j] = fma(-alpha[j], si, R[t][j]);
                        a.r[i0 + j] = rr[j];
                    }
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    q[j] = fma(CW[t][0], rr[j], q[j]); q[3 + j] = fma(CW[t][1], rr[j], q[3 + j]); q[6 + j] = fma(CW[t][2], rr[j], q[6 + j]); q[9 + j] = fma(CW[t][3], rr[j], q[9 + j]);
                    q[12 + j] = fma(rr[j] * DI[t][j], rr[j], q[12 + j]);
                }
            }
        }
    }
    block_sum_wide<15>(q, lds);
    if (threadIdx.x < 12) a.cvec[(threadIdx.x % 3) * a.ncp + 4 * g + threadIdx.x / 3] = q[threadIdx.x];      // c[axis][4 g + k] = q[3 k + axis]
    else if (threadIdx.x < 15) a.rho[(threadIdx.x - 12) * a.G + g] = q[threadIdx.x];
}

// y_g = rows 4 g .. 4 g + 3 of (P^T A P)^-1 times c; the stop test; u = D^-1 r + P y on the aggregate's rows
__global__ __launch_bounds__(kBigVecT) void k_big_coarse(BigArgs a, int it) {
    __shared__ double lds[17 * 15];
    CgScal *cur = a.scal + ((it + 1) & 1);        // the slot k_big_vec (it) wrote (entry pass: it = -1 -> slot 0)
    if (cur->converged) return;
    const int g = (int)blockIdx.x;
    double q[15];
#pragma unroll
    for (int i = 0; i < 15; ++i) q[i] = 0.0;
    // four columns per thread and trip, all 28 loads in flight before the first product (the rows of the inverse come from HBM exactly once)
    const int r0c = g * a.ra;
    const bool mine_c = r0c >= a.row_lo && r0c < a.row_hi;      // (distributed solve: the dense rows of the rank's own aggregates only)
    for (int j0 = threadIdx.x; j0 < (mine_c ? a.nc : 0); j0 += 4 * kBigVecT) {
        double cc[4][3]; float mm[4][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int j = j0 + t * kBigVecT, jj = j < a.nc ? j : 0;
            cc[t][0] = a.cvec[jj]; cc[t][1] = a.cvec[a.ncp + jj]; cc[t][2] = a.cvec[2 * a.ncp + jj];
#pragma unroll
            for (int k = 0; k < 4; ++k) mm[t][k] = j < a.nc ? a.ainv[(size_t)(4 * g + k) * a.ncp + jj] : 0.0f;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double m = (double)mm[t][k];
                q[3 * k] = fma(m, cc[t][0], q[3 * k]); q[3 * k + 1] = fma(m, cc[t][1], q[3 * k + 1]); q[3 * k + 2] = fma(m, cc[t][2], q[3 * k + 2]);
            }
        }
    }
    for (int j = threadIdx.x; j < a.G; j += kBigVecT) { q[12] += a.rho[j]; q[13] += a.rho[a.G + j]; q[14] += a.rho[2 * a.G + j]; }
    block_sum_wide<15>(q, lds);
    const CgScal sc = *cur;
    bool conv = true;
#pragma unroll
    for (int j = 0; j < 3; ++j) conv = conv && (q[12 + j] <= a.tol2 * sc.gamma_b[j] + 1e-300);
    __syncthreads();      // (everybody has read *cur before block 0 changes it)
    if (conv) {
        if (g == 0 && threadIdx.x == 0) {
            cur->converged = 1;
            atomicAdd(a.counters + 4, 1);
            atomicMax(a.counters + 3, sc.iters);
            a.counters[8 + (sc.seq & 63)] = sc.iters;
            __hip_atomic_store(a.sig + 3, sc.iters, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);      // (the iteration it converged at: the distributed solve's ranks stop at the same chunk)
            atomicAdd(a.counters + 73, 1);      // converged solves since create (admm_hip_solve_totals; solves and iterations: k_big_scatter)
            __hip_atomic_store(a.sig, sc.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    const int r0 = g * a.ra;
    if (r0 < a.row_lo || r0 >= a.row_hi) return;
    constexpr int RPT = 3;
    for (int base = r0 + (int)threadIdx.x; base < r0 + a.ra; base += RPT * kBigVecT) {      // (loads of all three rows first: see k_big_vec)
        double R[RPT][3], DI[RPT][3], CW[RPT][4];
#pragma unroll
        for (int t = 0; t < RPT; ++t) {
            const int row = base + t * kBigVecT, rw = row < r0 + a.ra ? row : r0;
#pragma unroll
            for (int j = 0; j < 3; ++j) { R[t][j] = a.r[3 * (size_t)rw + j]; DI[t][j] = a.dinv[3 * (size_t)rw + j]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) CW[t][k] = a.cwt[4 * (size_t)rw + k];
        }
#pragma unroll
        for (int t = 0; t < RPT; ++t) {
            const int row = base + t * kBigVecT;
            if (row >= r0 + a.ra) continue;
#pragma unroll
            for (int j = 0; j < 3; ++j)
                a.u[3 * (size_t)row + j] = fma(DI[t][j], R[t][j], fma(CW[t][0], q[j], fma(CW[t][1], q[3 + j], fma(CW[t][2], q[6 + j], CW[t][3] * q[9 + j]))));
        }
    }
}

// DISTRIBUTED solve (launch_pcg_dist_big): the interface rows of u -- rows a rank owns that another rank's matrix rows reference -- travel in one
// compact buffer: every rank packs the rows it owns (zeros elsewhere), a sum all-reduce assembles the buffer, every rank unpacks the
// rows it does not own.  A contiguous range of aggregates of the recursive bisection is a compact subdomain: the interface is its surface.
__global__ __launch_bounds__(256) void k_big_if_pack(BigArgs a, const int *__restrict__ if_rows, int nif, double *__restrict__ buf) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nif) return;
    const int row = if_rows[i];
    const bool own = row >= a.row_lo && row < a.row_hi;
#pragma unroll
    for (int j = 0; j < 3; ++j) buf[3 * (size_t)i + j] = own ? a.u[3 * (size_t)row + j] : 0.0;
}
__global__ __launch_bounds__(256) void k_big_if_unpack(BigArgs a, const int *__restrict__ if_rows, int nif, const double *__restrict__ buf) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nif) return;
    const int row = if_rows[i];
    if (row >= a.row_lo && row < a.row_hi) return;
#pragma unroll
    for (int j = 0; j < 3; ++j) a.u[3 * (size_t)row + j] = buf[3 * (size_t)i + j];
}

// x back to the API order; u_api = D^-1 r_final (what k_rc_record expects of the Jacobi path: r = u / dinv)
__global__ __launch_bounds__(256) void k_big_scatter(BigArgs a, int final_slot) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {      // totals since create; a solve that ran out of iterations is logged here (a converged one: k_big_coarse)
        const CgScal sc = a.scal[final_slot];
        atomicAdd(a.counters + 72, 1); atomicAdd(a.counters + 74, sc.iters);
        if (!sc.converged) { atomicMax(a.counters + 3, sc.iters); a.counters[8 + (sc.seq & 63)] = sc.iters; }
    }
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= a.n_rows) return;
    const int v = a.orig[row];
    if (v < 0) return;
    if (row < a.row_lo || row >= a.row_hi) {
#pragma unroll
        for (int j = 0; j < 3; ++j) { a.x_api[3 * (size_t)v + j] = 0.0; if (a.u_api) a.u_api[3 * (size_t)v + j] = 0.0; }
        return;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        a.x_api[3 * (size_t)v + j] = a.xi[3 * (size_t)row + j];
        if (a.u_api) a.u_api[3 * (size_t)v + j] = a.dinv_api[3 * (size_t)v + j] * a.r[3 * (size_t)row + j];
    }
}

} // namespace admm_k
