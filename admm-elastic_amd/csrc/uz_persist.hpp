// uz_persist.hpp -- the Schur-complement CG of UzawaCG::solve (src/UzawaCG.hpp:92-120) on the active rows as ONE persistent launch.
//
// With the columns of K^-1 cached (admm_hip.hip: uz_ensure_columns) a Schur iteration touches the active vertices only: with one
// passive row per vertex, q3 = C A^-1 C^T d is the product of d with the dense symmetric matrix
//     S_ij = (K^-1)_(act_i, act_j) (n_i . n_j),      n_i = the row of C at vertex act_i (ck x the obstacle's normal),
// and the rest of an iteration is two dot-product rounds and three vector updates on n_act <= 1024 numbers.  As kernels that was two
// launches per iteration (k_uzc_matvec + k_uzc_rows: 4.8 + 7.9 us and two dispatch gaps, ~19 us; 18 iterations per solve, 20 solves
// per frame).  Here every block keeps R = 16 (8) rows of S in LDS for the whole solve and an iteration costs two hand-offs by tagged
// granules (gs_persist.hpp: 16 bytes = value + stamp, one write-through store, no drain, no flag, no grid barrier):
//     d (all blocks' rows)  ->  q3 = S d on the block's rows  ->  five partial sums per block  ->  alpha, beta, stop  ->  y, r, d
//   * every block adds the same partial sums in the same order: identical alpha / beta / verdicts everywhere, deterministic;
//   * r.r and r.q3 of the UPDATED r (the stop test :112 and beta :115) come from the same round of sums as alpha:
//         r' = r - alpha q3   =>   r'.q3 = r.q3 - alpha q3.q3,   r'.r' = r.r - 2 alpha r.q3 + alpha^2 q3.q3
//     (one hand-off instead of two; a typical Schur iteration shrinks |r|^2 by ~10x, so the cancellation costs one digit of sixteen;
//     a step that shrinks it by more than 1e3 -- small active sets -- forms both from the updated r in a second round instead);
//   * x is not touched: x = x0 - A^-1 C^T (y - y0) is applied once after the loop (launch_uzawa), as with the two-launch iterations;
//   * every poll is bounded: a hand-off that cannot complete raises the abort word and sig[2], the host takes the recovery path;
//   * the host does not wait for the launch: the stop verdict stays on the device, the iteration count goes to a device counter.
// Rows that couple several vertices (dynamic rows: hit vertex + the three vertices of a face) run the same kernel: the host lists the
// ROW vertices and k_uzc_schur (kernels.hpp) forms S on them from the active x active block of K^-1.  At most 1024 rows.
#pragma once
#include <hip/hip_runtime.h>
#include "gs_persist.hpp"

namespace admm_k {

constexpr int kUzpT = 256, kUzpMaxAct = 1024, kUzpMaxBlocks = 128;

struct UzpArgs {
    int n_act, ld, R, NB, max_iters;      // R rows per block, NB = ceil(n_act / R) blocks
    const int *act;                        // [n_act] active vertices (ascending)
    const double *G;                       // [n_act][ld]: S[j][i] = (K^-1 e_(act_j))[act_i] (n_i . n_j)   (k_uzc_extract with the rows of C)
    double *d, *r, *y, *q3;                // [nv] Schur-CG vectors by vertex (read at entry, written back at exit)
    double tol2; UzScal *sc;
    v4u *dbox;                             // [2][kUzpMaxAct] granules: d of an iteration, by parity
    v4u *sbox;                             // [2][kUzpMaxBlocks][8] granules: the blocks' five partial sums, by parity
    unsigned stamp0;                       // launch number x 1024: stamps of this launch are stamp0 + 4 k + {0: d, 1: sums, 2: exact r.r / r.q3}
    unsigned *abort_word; int *sig;
    int *iters_step, *applies_total;       // device-side statistics (the host does not wait for the launch): Schur iterations of this step; products S d since create
};

inline int uzp_rows_per_block(int n_act) { return n_act <= 800 ? 16 : 8; }
constexpr int kUzpScratch = 1024;     // bytes of LDS ahead of the S rows: sums, control words
inline size_t uzp_lds_bytes(int n_act, int R) { return kUzpScratch + (size_t)n_act * (size_t)(R + 1) * 8 + (size_t)n_act * 8; }

__global__ __launch_bounds__(kUzpT) void k_uz_persist(UzpArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    LdsD *red = (LdsD *)smem;                         // [5][16] the rows' terms of the five sums
    LdsD *tot = (LdsD *)(smem + 640);                 // [5] their totals over all blocks
    LdsI32 *ctl = (LdsI32 *)(smem + 704);             // [0] a poll of this block failed
    const int n = a.n_act, R = a.R, RS = R + 1;       // (rows of S are stored with a stride of R + 1 doubles: conflict-free reads)
    LdsD *Sl = (LdsD *)(smem + kUzpScratch);          // S(i0 + r, j) at j RS + r
    LdsD *dl = Sl + (size_t)n * RS;                   // d of the current iteration, all active vertices
    const int t = (int)threadIdx.x, b = (int)blockIdx.x;
    const int TPR = kUzpT / R, r_ = t / TPR, l = t % TPR;     // row of the block, lane of the row
    const int i = b * R + r_;
    const bool row_on = i < n, leader = row_on && l == 0;
    if (t == 0) {
        ctl[0] = __hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1 : 0;
        if (ctl[0]) __hip_atomic_store(a.sig + 2, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    if (ctl[0] || a.sc->stop) return;
    const __amdgpu_buffer_rsrc_t rd = soa_rsrc(a.dbox), rs = soa_rsrc(a.sbox);
    // ---- fill: the block's rows of S (k_uzc_extract has formed S_ij = G_ij (n_i . n_j)); row index fastest: the R entries of a column j
    //      are one contiguous run, eight loads in flight per thread ----
    const int vi = row_on ? a.act[i] : 0;
    {
        const int total = n * R;
        for (int e0 = t; e0 < total; e0 += 8 * kUzpT) {
            double g[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * kUzpT, rr_ = e % R, j = e / R, ii = b * R + rr_;
                g[u] = (e < total && ii < n) ? a.G[(size_t)j * a.ld + ii] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * kUzpT, rr_ = e % R, j = e / R;
                if (e < total) Sl[j * RS + rr_] = g[u];
            }
        }
    }
    for (int j = t; j < n; j += kUzpT) dl[j] = a.d[a.act[j]];
    double di = leader ? a.d[vi] : 0.0, ri = leader ? a.r[vi] : 0.0, yi = leader ? a.y[vi] : 0.0, qi = 0.0;
    __syncthreads();

    auto poll_failed = [&](unsigned &spins) -> bool {
        if (++spins > kGspSpin || ((spins & 127u) == 0u && __hip_atomic_load(a.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
            if (!ctl[0]) {
                ctl[0] = 1;
                __hip_atomic_store(a.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(a.sig + 2, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            return true;
        }
        __builtin_amdgcn_s_sleep(1);
        return false;
    };
    int iters = 0, stop = 0;
    double denom = 0.0, alpha = 0.0, beta = 0.0, rr_new = 0.0;
    for (int k = 0; k < a.max_iters; ++k) {
        const int par = k & 1;
        if (k > 0) {      // d of this iteration from every block
            const unsigned want = a.stamp0 + 4u * (unsigned)k;
            for (int j = t; j < n; j += kUzpT) {
                v4u g; unsigned spins = 0;
                while (true) { g = gsp_load(rd, (par * kUzpMaxAct + j) * 16); if (gsp_ok(g, want) || poll_failed(spins)) break; }
                dl[j] = gsp_val(g);
            }
            __syncthreads();
            if (ctl[0]) return;
        }
        // q3 = S d on the block's rows
        double acc = 0.0;
        {   // (four entries' LDS reads in flight per step; the sum keeps the order of j)
            int j = l;
            for (; j + 3 * TPR < n; j += 4 * TPR) {
                const double s0 = Sl[j * RS + r_], s1 = Sl[(j + TPR) * RS + r_], s2 = Sl[(j + 2 * TPR) * RS + r_], s3 = Sl[(j + 3 * TPR) * RS + r_];
                const double d0 = dl[j], d1 = dl[j + TPR], d2 = dl[j + 2 * TPR], d3 = dl[j + 3 * TPR];
                acc = fma(s3, d3, fma(s2, d2, fma(s1, d1, fma(s0, d0, acc))));
            }
            for (; j < n; j += TPR) acc = fma(Sl[j * RS + r_], dl[j], acc);
        }
        for (int o = 1; o < TPR; o <<= 1) acc += __shfl_xor(acc, o, 64);
        if (leader) {
            qi = acc;
            red[0 * 16 + r_] = di * qi; red[1 * 16 + r_] = di * ri; red[2 * 16 + r_] = ri * qi; red[3 * 16 + r_] = qi * qi; red[4 * 16 + r_] = ri * ri;
        } else if (l == 0) {
            red[0 * 16 + r_] = 0.0; red[1 * 16 + r_] = 0.0; red[2 * 16 + r_] = 0.0; red[3 * 16 + r_] = 0.0; red[4 * 16 + r_] = 0.0;
        }
        __syncthreads();
        const unsigned sw = a.stamp0 + 4u * (unsigned)k + 1u;
        if (t < 5) {
            double s = 0.0;
            for (int q = 0; q < R; ++q) s += red[t * 16 + q];
            gsp_store(rs, ((par * kUzpMaxBlocks + b) * 8 + t) * 16, gsp_pack(s, sw));
        }
        // the five sums over all blocks: wave w takes sum w (wave 0 also the fifth), lane = block (and block + 64)
        for (int q = t >> 6; q < 5; q += 4) {
            const int lane = t & 63;
            double v = 0.0;
            for (int bb = lane; bb < a.NB; bb += 64) {
                v4u g; unsigned spins = 0;
                while (true) { g = gsp_load(rs, ((par * kUzpMaxBlocks + bb) * 8 + q) * 16); if (gsp_ok(g, sw) || poll_failed(spins)) break; }
                v += gsp_val(g);
            }
            v = wave_sum(v);
            if (lane == 0) tot[q] = v;
        }
        __syncthreads();
        if (ctl[0]) return;
        const double dq = tot[0], dr = tot[1], rq = tot[2], qq = tot[3], rr = tot[4];
        denom = dq;
        if (fabs(denom) < 2.2250738585072014e-308) { alpha = 0.0; stop = 1; break; }       // :100-103 (nothing moves)
        alpha = dr / denom;                                                               // :104
        if (leader) { yi = fma(alpha, di, yi); ri = fma(-alpha, qi, ri); }                // :106-107 (x: after the loop)
        rr_new = fma(alpha * alpha, qq, fma(-2.0 * alpha, rq, rr));
        double rq_new = rq - alpha * qq;
        if (!(rr_new > 1e-3 * rr)) {
            // A step that takes |r|^2 down by more than three orders (small active sets converge superlinearly) leaves the expansions
            // above with few correct digits -- or none, and a stop that is not one.  Then r.r and r.q3 are formed from the updated r
            // itself, in a second round of sums (every block sees the same numbers and takes the same branch).
            __syncthreads();      // (tot has been read by everybody)
            if (l == 0) { red[0 * 16 + r_] = leader ? ri * ri : 0.0; red[1 * 16 + r_] = leader ? ri * qi : 0.0; }
            __syncthreads();
            const unsigned s2 = a.stamp0 + 4u * (unsigned)k + 2u;
            if (t < 2) {
                double sm = 0.0;
                for (int q = 0; q < R; ++q) sm += red[t * 16 + q];
                gsp_store(rs, ((par * kUzpMaxBlocks + b) * 8 + 5 + t) * 16, gsp_pack(sm, s2));
            }
            if (t < 128) {
                const int q = t >> 6, lane = t & 63;
                double v = 0.0;
                for (int bb = lane; bb < a.NB; bb += 64) {
                    v4u g; unsigned spins = 0;
                    while (true) { g = gsp_load(rs, ((par * kUzpMaxBlocks + bb) * 8 + 5 + q) * 16); if (gsp_ok(g, s2) || poll_failed(spins)) break; }
                    v += gsp_val(g);
                }
                v = wave_sum(v);
                if (lane == 0) tot[q] = v;
            }
            __syncthreads();
            if (ctl[0]) return;
            rr_new = tot[0]; rq_new = tot[1];
        }
        if (rr_new < a.tol2) { stop = 1; break; }                                         // :112
        beta = rq_new / denom;                                                            // :115
        if (leader) {
            di = fma(-beta, di, ri);                                                      // :117
            if (k + 1 < a.max_iters) gsp_store(rd, (((k + 1) & 1) * kUzpMaxAct + i) * 16, gsp_pack(di, a.stamp0 + 4u * (unsigned)(k + 1)));
        }
        ++iters;
        __syncthreads();      // red / tot are rewritten by the next iteration
    }
    if (leader) { a.y[vi] = yi; a.r[vi] = ri; a.d[vi] = di; a.q3[vi] = qi; }
    if (b == 0 && t == 0) {
        a.sc->denom = denom; a.sc->alpha = alpha; a.sc->beta = beta; a.sc->rr = rr_new; a.sc->stop = stop; a.sc->iters += iters;
        *a.iters_step += iters; *a.applies_total += iters + stop;
    }
}

} // namespace admm_k
