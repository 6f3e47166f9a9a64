// oc_plan.cpp -- host-side plan of the on-chip PCG (pcg_onchip.hpp) for GENERAL meshes.  No GPU calls.
//
// The persistent PCG kernel gives every CU one block of rows.  Which rows, and in which order, decides (i) how much of a
// matrix-vector product stays inside a block, (ii) how many other blocks a block waits for, (iii) how well the rows fill
// the LDS slab and (iv) what a block-local / two-level preconditioner can do.  The caller's vertex numbering is kept at
// the API (Solver::m_x, src/Solver.hpp:66); INSIDE the solve the rows are renumbered:
//   * vertices -> G compact blocks by recursive graph bisection (level structures from a pseudo-peripheral vertex,
//     Simon 1991): on the 1 M-tet unstructured body 81 % of the non-zeros are block-local, against 46 % for the index
//     strips of a reverse Cuthill-McKee numbering;
//   * every block -> kOcSub compact aggregates (whole wavefronts each): the coarse space of the two-level
//     preconditioner  M^-1 = D^-1 + P (P^T A P)^-1 P^T  (P = aggregate indicator vectors; the dense inverse of the
//     <= 1024 x 1024 coarse matrix is formed here once, the system matrix of a scene never changes, src/Solver.cpp:225-226);
//   * rows of an aggregate sorted by length, so that the 64 rows of a wavefront (one SELL slice) have similar lengths and
//     the slab is not spent on padding; the diagonal is not stored (it is 1 / dinv - m).
#include "host_setup.hpp"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>

namespace admm_host {

namespace {

// Vertex graph of the non-zero off-diagonal couplings
struct Graph {
    std::vector<int32_t> ptr, adj;
};
Graph coupling_graph(const Csr &A) {
    Graph g;
    g.ptr.assign(A.n + 1, 0);
    for (int32_t r = 0; r < A.n; ++r) {
        int32_t c = 0;
        for (int32_t k = A.rowptr[r]; k < A.rowptr[r + 1]; ++k) c += (A.col[k] != r && A.val[k] != 0.0) ? 1 : 0;
        g.ptr[r + 1] = g.ptr[r] + c;
    }
    g.adj.resize(g.ptr[A.n]);
    for (int32_t r = 0; r < A.n; ++r) {
        int32_t o = g.ptr[r];
        for (int32_t k = A.rowptr[r]; k < A.rowptr[r + 1]; ++k)
            if (A.col[k] != r && A.val[k] != 0.0) g.adj[o++] = A.col[k];
    }
    return g;
}

// Breadth-first order of the members of one part (mark[v] == id), starting at `start`; unreachable members are
// appended component by component.  Returns the order; `seen` is scratch (all zero on entry and on exit).
void bfs_order(const Graph &g, const std::vector<int32_t> &members, const std::vector<int32_t> &mark, int32_t id, int32_t start,
               std::vector<char> &seen, std::vector<int32_t> &order) {
    order.clear();
    order.reserve(members.size());
    size_t next_seed = 0;
    int32_t seed = start;
    while (order.size() < members.size()) {
        if (seed < 0) {
            while (seen[members[next_seed]]) ++next_seed;
            seed = members[next_seed];
        }
        size_t head = order.size();
        order.push_back(seed); seen[seed] = 1;
        while (head < order.size()) {
            const int32_t v = order[head++];
            for (int32_t k = g.ptr[v]; k < g.ptr[v + 1]; ++k) {
                const int32_t w = g.adj[k];
                if (mark[w] == id && !seen[w]) { seen[w] = 1; order.push_back(w); }
            }
        }
        seed = -1;
    }
    for (int32_t v : order) seen[v] = 0;
}

// Split `members` (all marked `id`) into parts of the given sizes, recursively by halves of the size list: the members
// are ordered by a level structure rooted at a pseudo-peripheral vertex (two sweeps) and cut where the first half of the
// sizes ends.  part_of[v] = first_part + index of the part v lands in.
void bisect(const Graph &g, std::vector<int32_t> &members, const int32_t *sizes, int n_parts, int32_t first_part,
            std::vector<int32_t> &mark, int32_t &next_id, std::vector<char> &seen, std::vector<int32_t> &part_of) {
    if (members.empty()) return;
    if (n_parts == 1) {
        for (int32_t v : members) part_of[v] = first_part;
        return;
    }
    const int32_t id = mark[members[0]];
    std::vector<int32_t> order;
    bfs_order(g, members, mark, id, members[0], seen, order);
    const int32_t far1 = order.back();
    bfs_order(g, members, mark, id, far1, seen, order);
    const int32_t far2 = order.back();
    bfs_order(g, members, mark, id, far2, seen, order);
    const int h = n_parts / 2;
    size_t n0 = 0;
    for (int i = 0; i < h; ++i) n0 += (size_t)sizes[i];
    std::vector<int32_t> left(order.begin(), order.begin() + n0), right(order.begin() + n0, order.end());
    members.clear(); members.shrink_to_fit();
    const int32_t idl = next_id++, idr = next_id++;
    for (int32_t v : left) mark[v] = idl;
    for (int32_t v : right) mark[v] = idr;
    bisect(g, left, sizes, h, first_part, mark, next_id, seen, part_of);
    bisect(g, right, sizes + h, n_parts - h, first_part + h, mark, next_id, seen, part_of);
}

// In-place inverse of a dense symmetric positive definite matrix (row-major n x n) by Cholesky; false if not SPD.
bool spd_inverse(int n, std::vector<double> &a) {
    // A = L L^T (lower, in place)
    for (int j = 0; j < n; ++j) {
        double d = a[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= a[(size_t)j * n + k] * a[(size_t)j * n + k];
        if (!(d > 0.0)) return false;
        d = std::sqrt(d);
        a[(size_t)j * n + j] = d;
        const double inv = 1.0 / d;
        for (int i = j + 1; i < n; ++i) {
            double s = a[(size_t)i * n + j];
            const double *ri = &a[(size_t)i * n], *rj = &a[(size_t)j * n];
            for (int k = 0; k < j; ++k) s -= ri[k] * rj[k];
            a[(size_t)i * n + j] = s * inv;
        }
    }
    // L^-1 (lower, in place): column by column
    for (int j = 0; j < n; ++j) {
        a[(size_t)j * n + j] = 1.0 / a[(size_t)j * n + j];
        for (int i = j + 1; i < n; ++i) {
            double s = 0.0;
            for (int k = j; k < i; ++k) s -= a[(size_t)i * n + k] * a[(size_t)k * n + j];
            a[(size_t)i * n + j] = s / a[(size_t)i * n + i];
        }
    }
    // A^-1 = L^-T L^-1: (i, j) = sum_{k >= max(i, j)} Linv(k, i) Linv(k, j); work on the transposed copy for unit stride
    std::vector<double> lt((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) lt[(size_t)j * n + i] = a[(size_t)i * n + j];   // lt(j, i) = Linv(i, j): row j holds column j of Linv
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            const double *ci = &lt[(size_t)i * n], *cj = &lt[(size_t)j * n];
            double s = 0.0;
            for (int k = i; k < n; ++k) s += ci[k] * cj[k];
            a[(size_t)i * n + j] = s; a[(size_t)j * n + i] = s;
        }
    return true;
}

} // namespace

OcPlan build_oc_plan(const Csr &A, const double *mass3, int G, int spb, int lds_cols, bool want_coarse) {
    OcPlan P;
    const int32_t nv = A.n;
    P.G = G; P.spb = spb; P.sub = kOcSub;
    P.n_rows = G * spb * 64;
    const Graph g = coupling_graph(A);
    // ---- blocks, then aggregates inside every block ----
    std::vector<int32_t> part_of(nv, 0), mark(nv, 0);
    std::vector<char> seen(nv, 0);
    int32_t next_id = 1;
    {
        std::vector<int32_t> sizes(G), members(nv);
        for (int b = 0; b < G; ++b) sizes[b] = (int32_t)(((int64_t)nv * (b + 1)) / G - ((int64_t)nv * b) / G);
        std::iota(members.begin(), members.end(), 0);
        bisect(g, members, sizes.data(), G, 0, mark, next_id, seen, part_of);
    }
    std::vector<std::vector<int32_t> > blocks(G);
    for (int32_t v = 0; v < nv; ++v) blocks[part_of[v]].push_back(v);
    // waves of aggregate a of a block: [wave0[a], wave0[a + 1])
    int wave0[kOcSub + 1];
    for (int a = 0; a <= kOcSub; ++a) wave0[a] = (spb * a) / kOcSub;
    P.agg_of_slice.assign((size_t)G * spb, 0);
    for (int b = 0; b < G; ++b)
        for (int a = 0; a < kOcSub; ++a)
            for (int w = wave0[a]; w < wave0[a + 1]; ++w) P.agg_of_slice[(size_t)b * spb + w] = (signed char)a;
    P.orig.assign(P.n_rows, -1);
    P.pos.assign(nv, -1);
    std::vector<int32_t> len(nv, 0);
    for (int32_t r = 0; r < nv; ++r) len[r] = g.ptr[r + 1] - g.ptr[r];
    std::vector<int32_t> agg_part(nv, 0);
    for (int b = 0; b < G; ++b) {
        std::vector<int32_t> &mem = blocks[b];
        const int32_t nb = (int32_t)mem.size();
        if (nb > 64 * spb) { P.ok = false; return P; }
        // aggregate sizes proportional to their wave counts (an aggregate with no wave gets nothing)
        int32_t sizes[kOcSub];
        int32_t given = 0;
        for (int a = 0; a < kOcSub; ++a) {
            const int64_t hi = ((int64_t)nb * wave0[a + 1] + spb - 1) / spb;
            sizes[a] = (int32_t)std::min<int64_t>(hi, nb) - given;
            sizes[a] = std::min(sizes[a], 64 * (wave0[a + 1] - wave0[a]));
            given += sizes[a];
        }
        // rounding may leave a few members over: hand them to aggregates with room
        for (int a = 0; a < kOcSub && given < nb; ++a) {
            const int32_t room = 64 * (wave0[a + 1] - wave0[a]) - sizes[a];
            const int32_t add = std::min(room, nb - given);
            sizes[a] += add; given += add;
        }
        if (nb > 0) {
            const int32_t idb = next_id++;
            for (int32_t v : mem) mark[v] = idb;
            std::vector<int32_t> copy(mem);
            // empty parts are legal for bisect only at the ends of the list; compact the non-empty ones
            int32_t nz_sizes[kOcSub]; int nz_map[kOcSub]; int nnz = 0;
            for (int a = 0; a < kOcSub; ++a) if (sizes[a] > 0) { nz_sizes[nnz] = sizes[a]; nz_map[nnz] = a; ++nnz; }
            bisect(g, copy, nz_sizes, nnz, 0, mark, next_id, seen, agg_part);
            for (int32_t v : mem) agg_part[v] = nz_map[agg_part[v]];
        }
        // rows of an aggregate: longest first (ties: vertex index), laid into the aggregate's waves
        for (int a = 0; a < kOcSub; ++a) {
            std::vector<int32_t> rows;
            for (int32_t v : mem) if (agg_part[v] == a) rows.push_back(v);
            std::stable_sort(rows.begin(), rows.end(), [&](int32_t x, int32_t y) { return len[x] != len[y] ? len[x] > len[y] : x < y; });
            int32_t slot = (b * spb + wave0[a]) * 64;
            for (int32_t v : rows) { P.orig[slot] = v; P.pos[v] = slot; ++slot; }
        }
    }
    // ---- SELL of the off-diagonal non-zeros, internal numbering ----
    Sell &S = P.A;
    S.n_rows = P.n_rows; S.n_slices = G * spb;
    S.slice_ptr.assign(S.n_slices + 1, 0); S.slice_width.assign(S.n_slices, 0);
    for (int32_t s = 0; s < S.n_slices; ++s) {
        int32_t w = 0;
        for (int l = 0; l < 64; ++l) { const int32_t v = P.orig[64 * (size_t)s + l]; if (v >= 0) w = std::max(w, len[v]); }
        w = std::max(4, (w + 3) / 4 * 4);
        S.slice_width[s] = w;
        S.slice_ptr[s + 1] = S.slice_ptr[s] + 64 * w;
    }
    S.idx.assign(S.slice_ptr[S.n_slices], 0); S.val.assign(S.slice_ptr[S.n_slices], 0.0);
    P.mdiag.assign(3 * (size_t)P.n_rows, 0.0);
    for (int32_t s = 0; s < S.n_slices; ++s)
        for (int l = 0; l < 64; ++l) {
            const int32_t r = 64 * s + l, v = P.orig[r];
            int32_t k = 0;
            if (v >= 0) {
                // columns in increasing INTERNAL order (neighbouring lanes then tend to read neighbouring entries)
                std::vector<std::pair<int32_t, double> > ent;
                double diag = 0.0;
                for (int32_t q = A.rowptr[v]; q < A.rowptr[v + 1]; ++q) {
                    if (A.col[q] == v) { diag = A.val[q]; continue; }
                    if (A.val[q] != 0.0) ent.emplace_back(P.pos[A.col[q]], A.val[q]);
                }
                std::sort(ent.begin(), ent.end());
                for (auto &e : ent) { const size_t o = (size_t)S.slice_ptr[s] + 64 * k + l; S.idx[o] = e.first; S.val[o] = e.second; ++k; }
                for (int j = 0; j < 3; ++j) P.mdiag[3 * (size_t)r + j] = mass3[3 * (size_t)v + j] + diag;
            }
            for (; k < S.slice_width[s]; ++k) { const size_t o = (size_t)S.slice_ptr[s] + 64 * k + l; S.idx[o] = r; S.val[o] = 0.0; }
        }
    // ---- LDS slab: columns of every slice held on chip ----
    P.wl_s.assign(S.n_slices, 0); P.lds_off.assign(S.n_slices, 0);
    P.bcols = 0;
    for (int b = 0; b < G; ++b) {
        int cap = 0;
        for (int w = 0; w < spb; ++w) cap = std::max(cap, S.slice_width[b * spb + w]);
        auto total = [&](int c) { int t = 0; for (int w = 0; w < spb; ++w) t += std::min(S.slice_width[b * spb + w], c); return t; };
        while (cap > 4 && total(cap) > lds_cols) cap -= 4;
        int off = 0;
        for (int w = 0; w < spb; ++w) {
            const int s = b * spb + w;
            P.wl_s[s] = std::min(std::min(S.slice_width[s], cap), std::max(0, lds_cols - off) / 4 * 4);
            P.lds_off[s] = off;
            off += P.wl_s[s];
        }
        P.bcols = std::max(P.bcols, off);
    }
    {   // statistics
        int64_t stored = 0, onchip = 0, local = 0, nnz = 0;
        for (int32_t s = 0; s < S.n_slices; ++s) { stored += 64 * (int64_t)S.slice_width[s]; onchip += 64 * (int64_t)P.wl_s[s]; }
        for (int32_t v = 0; v < nv; ++v)
            for (int32_t k = g.ptr[v]; k < g.ptr[v + 1]; ++k) { ++nnz; local += (part_of[g.adj[k]] == part_of[v]) ? 1 : 0; }
        P.stat_nnz = nnz; P.stat_stored = stored; P.stat_onchip = onchip; P.stat_local = local;
    }
    // ---- neighbour blocks (hand-off lists of the pipelined iteration) ----
    P.nbr.assign((size_t)G * 64, -1);
    P.nbr_ok = true; P.nbr_max = 0;
    for (int b = 0; b < G && P.nbr_ok; ++b) {
        std::vector<char> sb(G, 0);
        int n = 0;
        for (int32_t v : blocks[b])
            for (int32_t k = g.ptr[v]; k < g.ptr[v + 1]; ++k) {
                const int bj = part_of[g.adj[k]];
                if (bj == b || sb[bj]) continue;
                sb[bj] = 1;
                if (n == 64) { P.nbr_ok = false; break; }
                P.nbr[(size_t)b * 64 + n++] = bj;
            }
        P.nbr_max = std::max(P.nbr_max, n);
    }
    // ---- coarse space: aggregate indicators; (P^T A P)^-1 dense ----
    P.nc = G * kOcSub; P.ncp = (P.nc + 63) / 64 * 64;
    P.coarse_ok = false;
    bool uniform_mass = true;
    for (int32_t v = 0; v < nv && uniform_mass; ++v)
        uniform_mass = mass3[3 * (size_t)v] == mass3[3 * (size_t)v + 1] && mass3[3 * (size_t)v] == mass3[3 * (size_t)v + 2];
    if (want_coarse && uniform_mass && P.nc <= 2048) {
        const int nc = P.nc;
        std::vector<int32_t> agg(nv);
        for (int32_t v = 0; v < nv; ++v) agg[v] = part_of[v] * kOcSub + agg_part[v];
        std::vector<double> Ac((size_t)nc * nc, 0.0);
        for (int32_t v = 0; v < nv; ++v) {
            const int ci = agg[v];
            Ac[(size_t)ci * nc + ci] += mass3[3 * (size_t)v];
            for (int32_t q = A.rowptr[v]; q < A.rowptr[v + 1]; ++q) Ac[(size_t)ci * nc + agg[A.col[q]]] += A.val[q];
        }
        // empty aggregates (blocks smaller than their slots): unit diagonal, they never receive a residual
        for (int c = 0; c < nc; ++c) if (Ac[(size_t)c * nc + c] == 0.0) Ac[(size_t)c * nc + c] = 1.0;
        for (int i = 0; i < nc; ++i)   // symmetrise the round-off
            for (int j = 0; j < i; ++j) { const double s = 0.5 * (Ac[(size_t)i * nc + j] + Ac[(size_t)j * nc + i]); Ac[(size_t)i * nc + j] = s; Ac[(size_t)j * nc + i] = s; }
        if (spd_inverse(nc, Ac)) {
            P.ainv.assign((size_t)nc * P.ncp, 0.0);
            for (int i = 0; i < nc; ++i) std::memcpy(&P.ainv[(size_t)i * P.ncp], &Ac[(size_t)i * nc], nc * sizeof(double));
            P.coarse_ok = true;
        }
    }
    P.ok = true;
    return P;
}

} // namespace admm_host
