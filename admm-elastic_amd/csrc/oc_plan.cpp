// oc_plan.cpp -- host-side plan of the on-chip PCG (pcg_onchip2.hpp).  No GPU calls.
//
// The persistent PCG kernel gives every CU one block of rows.  Which rows, and in which order, decides (i) how much of a
// matrix-vector product stays inside a block, (ii) how many other blocks a block waits for, (iii) how well the rows fill
// the LDS slab and (iv) what a block-local / two-level preconditioner can do.  The caller's vertex numbering is kept at
// the API (Solver::m_x, src/Solver.hpp:66); INSIDE the solve the rows are renumbered:
//   * vertices -> G compact blocks by recursive graph bisection (level structures from a pseudo-peripheral vertex,
//     Simon 1991): on the 1 M-tet unstructured body 81 % of the non-zeros are block-local, against 46 % for the index
//     strips of a reverse Cuthill-McKee numbering;
//   * every block -> kOcSub compact aggregates: the coarse space of the two-level preconditioner
//     M^-1 = D^-1 + P (P^T A P)^-1 P^T  (P = aggregate indicator vectors; the dense inverse of the <= 1024 x 1024 coarse
//     matrix is formed here once, the system matrix of a scene never changes, src/Solver.cpp:225-226);
//   * rows of a block sorted by length, so that the 64 rows of a wavefront (one SELL slice) have similar lengths and the
//     slab is not spent on padding; the diagonal is not stored (it is 1 / dinv - m);
//   * columns as 16-bit indices into the block's LOCAL vector: own rows first, then the block's halo list (the rows of
//     other blocks its matrix rows reference, each fetched once per product).
#include "host_setup.hpp"
#include <algorithm>
#include <cmath>
#include <cstdlib>
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

// Hierarchical block order of mesh vertices (mesh preprocessing, admm_host_block_order): the vertices are split into compact
// leaves of ~`leaf` vertices by the same recursive graph bisection the on-chip PCG uses for its blocks, leaves numbered in the
// order of the recursion tree (neighbouring leaves are siblings), vertices inside a leaf breadth-first.  Against reverse
// Cuthill-McKee the "active window" of a gather -- the span of indices the neighbours of a run of consecutive vertices touch
// -- shrinks from a level-set front (thousands of vertices on an unstructured 1 M-tet body) to a few leaves.
void block_order(int32_t nv, int32_t n_elems, int32_t corners, const int32_t *idx, int32_t leaf, int32_t *new_id) {
    Graph g;
    {
        std::vector<std::vector<int32_t> > adj(nv);
        for (int e = 0; e < n_elems; ++e)
            for (int a = 0; a < corners; ++a)
                for (int b = 0; b < corners; ++b)
                    if (a != b) adj[idx[(size_t)corners * e + a]].push_back(idx[(size_t)corners * e + b]);
        g.ptr.assign(nv + 1, 0);
        for (int32_t v = 0; v < nv; ++v) {
            std::sort(adj[v].begin(), adj[v].end());
            adj[v].erase(std::unique(adj[v].begin(), adj[v].end()), adj[v].end());
            g.ptr[v + 1] = g.ptr[v] + (int32_t)adj[v].size();
        }
        g.adj.reserve(g.ptr[nv]);
        for (int32_t v = 0; v < nv; ++v) g.adj.insert(g.adj.end(), adj[v].begin(), adj[v].end());
    }
    const int L = std::max(1, (nv + std::max(leaf, 1) - 1) / std::max(leaf, 1));
    std::vector<int32_t> sizes(L), members(nv), part_of(nv, 0), mark(nv, 0);
    for (int b = 0; b < L; ++b) sizes[b] = (int32_t)(((int64_t)nv * (b + 1)) / L - ((int64_t)nv * b) / L);
    std::iota(members.begin(), members.end(), 0);
    std::vector<char> seen(nv, 0);
    int32_t next_id = 1;
    bisect(g, members, sizes.data(), L, 0, mark, next_id, seen, part_of);
    std::vector<std::vector<int32_t> > leaves(L);
    for (int32_t v = 0; v < nv; ++v) leaves[part_of[v]].push_back(v);
    int32_t next = 0;
    std::vector<int32_t> order;
    for (int b = 0; b < L; ++b) {
        if (leaves[b].empty()) continue;
        bfs_order(g, leaves[b], mark, mark[leaves[b][0]], leaves[b][0], seen, order);
        for (int32_t v : order) new_id[v] = next++;
    }
}

OcPlan build_oc_plan(const Csr &A, const double *mass3, int G, int spb, int lds_bytes, bool want_coarse, const double *xyz) {
    OcPlan P;
    const int32_t nv = A.n;
    const int T = 64 * spb;
    P.G = G; P.spb = spb; P.sub = kOcSub;
    P.n_rows = G * T;
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
    P.orig.assign(P.n_rows, -1);
    P.pos.assign(nv, -1);
    std::vector<int32_t> len(nv, 0);
    for (int32_t r = 0; r < nv; ++r) len[r] = g.ptr[r + 1] - g.ptr[r];
    std::vector<int32_t> agg_part(nv, 0), bfs_rank(nv, 0);
    for (int b = 0; b < G; ++b) {
        std::vector<int32_t> &mem = blocks[b];
        const int32_t nb = (int32_t)mem.size();
        if (nb > T) { P.ok = false; return P; }
        if (nb > 0) {
            int32_t sizes[kOcSub]; int nz_map[kOcSub]; int nnz = 0;
            for (int a = 0; a < kOcSub; ++a) {
                const int32_t sz = (int32_t)(((int64_t)nb * (a + 1)) / kOcSub - ((int64_t)nb * a) / kOcSub);
                if (sz > 0) { sizes[nnz] = sz; nz_map[nnz] = a; ++nnz; }
            }
            const int32_t idb = next_id++;
            for (int32_t v : mem) mark[v] = idb;
            std::vector<int32_t> copy(mem);
            bisect(g, copy, sizes, nnz, 0, mark, next_id, seen, agg_part);
            for (int32_t v : mem) agg_part[v] = nz_map[agg_part[v]];
        }
        // rows of the block: longest first -> the 64 rows of a wavefront have similar lengths; rows of equal length in
        // breadth-first order of the block's graph, so that neighbouring lanes read neighbouring entries of the local vector
        std::vector<int32_t> rows(mem);
        if (nb > 0) {
            const int32_t idb2 = next_id++;
            for (int32_t v : mem) mark[v] = idb2;
            std::vector<int32_t> bfs;
            bfs_order(g, mem, mark, idb2, mem[0], seen, bfs);
            for (int32_t i = 0; i < nb; ++i) bfs_rank[bfs[i]] = i;
        }
        std::stable_sort(rows.begin(), rows.end(), [&](int32_t x, int32_t y) {
            if (len[x] != len[y]) return len[x] > len[y];
            return bfs_rank[x] < bfs_rank[y];
        });
        int32_t slot = b * T;
        for (int32_t v : rows) { P.orig[slot] = v; P.pos[v] = slot; ++slot; }
    }
    P.row_agg.assign(P.n_rows, 0);
    for (int32_t r = 0; r < P.n_rows; ++r) if (P.orig[r] >= 0) P.row_agg[r] = (signed char)agg_part[P.orig[r]];
    // ---- halo lists: the rows of other blocks a block's matrix rows reference, sorted ----
    P.halo_ptr.assign(G + 1, 0);
    std::vector<std::vector<int32_t> > halo(G);
    for (int b = 0; b < G; ++b) {
        std::vector<int32_t> &h = halo[b];
        for (int32_t v : blocks[b])
            for (int32_t k = g.ptr[v]; k < g.ptr[v + 1]; ++k)
                if (part_of[g.adj[k]] != b) h.push_back(P.pos[g.adj[k]]);
        std::sort(h.begin(), h.end());
        h.erase(std::unique(h.begin(), h.end()), h.end());
        P.halo_ptr[b + 1] = P.halo_ptr[b] + (int32_t)h.size();
        P.nh_max = std::max(P.nh_max, (int32_t)h.size());
    }
    P.halo_src.reserve(P.halo_ptr[G]);
    for (int b = 0; b < G; ++b) P.halo_src.insert(P.halo_src.end(), halo[b].begin(), halo[b].end());
    P.nh_cap = (P.nh_max + 63) / 64 * 64;
    if (T + P.nh_cap > 65535) { P.ok = false; return P; }
    // ---- SELL of the off-diagonal non-zeros: values + 16-bit local columns (own row = internal row - block base, halo
    //      entry h = T + h), four columns of a lane packed into one 8-byte word ----
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
    S.val.assign(S.slice_ptr[S.n_slices], 0.0);
    P.col16.assign(S.slice_ptr[S.n_slices], 0);
    P.mdiag.assign(3 * (size_t)P.n_rows, 0.0);
    // Entries of a row may stand in any order: they are placed so that the 32 lanes of a half wavefront -- the group one
    // ds_read_b64 serves per LDS cycle -- read DISTINCT bank pairs of the local vector in every column (bank pair of entry c =
    // c mod 32: the vector starts on a 256-byte boundary).  Placed by increasing column index, a column had ~3.5 lanes on its
    // busiest bank pair, i.e. the random reads of the vector -- 12 of the 17 LDS instructions of four entries -- ran at less
    // than a third of the LDS rate.  Greedy, column by column: rows with the most entries left choose first, each the entry
    // whose bank pair is least used in this column so far; rows without entries left pad with a zero times an own entry of
    // the least used bank pair.
    int64_t conflict_sorted = 0, conflict_placed = 0, columns_total = 0;
    const bool place_by_bank = [] { const char *e = getenv("ADMM_HIP_OC_BANKS"); return !(e && e[0] == '0'); }();
    for (int32_t s = 0; s < S.n_slices; ++s) {
        const int b = s / spb;
        const std::vector<int32_t> &h = halo[b];
        const int32_t w = S.slice_width[s];
        std::vector<std::pair<int32_t, double> > ent[64];   // (local index, value), increasing local index
        for (int l = 0; l < 64; ++l) {
            const int32_t r = 64 * s + l, v = P.orig[r];
            if (v < 0) continue;
            double diag = 0.0;
            for (int32_t q = A.rowptr[v]; q < A.rowptr[v + 1]; ++q) {
                const int32_t cv = A.col[q];
                if (cv == v) { diag = A.val[q]; continue; }
                if (A.val[q] == 0.0) continue;
                const int32_t pr = P.pos[cv];
                int32_t lc;
                if (part_of[cv] == b) lc = pr - b * T;
                else lc = T + (int32_t)(std::lower_bound(h.begin(), h.end(), pr) - h.begin());
                ent[l].emplace_back(lc, A.val[q]);
            }
            std::sort(ent[l].begin(), ent[l].end());
            for (int j = 0; j < 3; ++j) P.mdiag[3 * (size_t)r + j] = mass3[3 * (size_t)v + j] + diag;
        }
        auto put = [&](int l, int32_t k, uint16_t c, double x) {
            S.val[(size_t)S.slice_ptr[s] + 64 * k + l] = x;
            P.col16[(size_t)S.slice_ptr[s] + ((size_t)(k >> 2) * 64 + l) * 4 + (k & 3)] = c;
        };
        for (int half = 0; half < 2; ++half) {
            const int l0 = 32 * half;
            {   // what the order by column index would cost (statistics)
                for (int32_t k = 0; k < w; ++k) {
                    int load[32] = {0}, mx = 0;
                    for (int l = l0; l < l0 + 32; ++l) {
                        const int32_t c = k < (int32_t)ent[l].size() ? ent[l][k].first : (int32_t)(64 * s + l - b * T);
                        mx = std::max(mx, ++load[c & 31]);
                    }
                    conflict_sorted += mx; ++columns_total;
                }
            }
            if (!place_by_bank) {
                for (int l = l0; l < l0 + 32; ++l)
                    for (int32_t k = 0; k < w; ++k) {
                        if (k < (int32_t)ent[l].size()) put(l, k, (uint16_t)ent[l][k].first, ent[l][k].second);
                        else put(l, k, (uint16_t)(64 * s + l - b * T), 0.0);
                    }
                continue;
            }
            std::vector<char> used[32];
            for (int l = l0; l < l0 + 32; ++l) used[l - l0].assign(ent[l].size(), 0);
            int rem[32];
            for (int l = l0; l < l0 + 32; ++l) rem[l - l0] = (int)ent[l].size();
            for (int32_t k = 0; k < w; ++k) {
                int load[32] = {0};
                int order[32];
                std::iota(order, order + 32, 0);
                std::stable_sort(order, order + 32, [&](int x, int y) { return rem[x] > rem[y]; });
                int mx = 0;
                for (int oi = 0; oi < 32; ++oi) {
                    const int i = order[oi], l = l0 + i;
                    if (rem[i] > 0) {
                        int best = -1, best_load = 1 << 30;
                        for (int e = 0; e < (int)ent[l].size(); ++e)
                            if (!used[i][e]) {
                                const int ld = load[ent[l][e].first & 31];
                                if (ld < best_load) { best_load = ld; best = e; }
                            }
                        used[i][best] = 1; --rem[i];
                        mx = std::max(mx, ++load[ent[l][best].first & 31]);
                        put(l, k, (uint16_t)ent[l][best].first, ent[l][best].second);
                    } else {
                        int bc = 0;
                        for (int c = 1; c < 32; ++c) if (load[c] < load[bc]) bc = c;
                        mx = std::max(mx, ++load[bc]);
                        put(l, k, (uint16_t)bc, 0.0);      // zero times own entry bc of the block (T >= 64: it exists and is finite)
                    }
                }
                conflict_placed += mx;
            }
        }
    }
    P.stat_bank_sorted = columns_total ? (double)conflict_sorted / (double)columns_total : 0.0;
    P.stat_bank_placed = columns_total ? (double)(place_by_bank ? conflict_placed : conflict_sorted) / (double)columns_total : 0.0;
    // ---- LDS: the block's local vector (3 axes x (T + halo)) and the slab (10 bytes per entry) ----
    P.vec_len = T + P.nh_cap;
    const int lds_cols = std::max(0, (lds_bytes - 3 * 8 * P.vec_len) / (64 * 10)) / 4 * 4;
    P.wl_s.assign(S.n_slices, 0); P.lds_off.assign(S.n_slices, 0);
    P.bcols = 0;
    for (int b = 0; b < G; ++b) {
        int cap = 0;
        for (int w = 0; w < spb; ++w) cap = std::max(cap, S.slice_width[b * spb + w]);
        auto total = [&](int c) { int t = 0; for (int w = 0; w < spb; ++w) t += std::min(S.slice_width[b * spb + w], c); return t; };
        while (cap > 0 && total(cap) > lds_cols) cap -= 4;
        int off = 0;
        for (int w = 0; w < spb; ++w) {
            const int s = b * spb + w;
            P.wl_s[s] = std::min(S.slice_width[s], cap);
            P.lds_off[s] = off;
            off += P.wl_s[s];
        }
        // columns left over go to the widest slices, four at a time
        for (bool more = true; more && off + 4 <= lds_cols;) {
            more = false;
            for (int w = 0; w < spb && off + 4 <= lds_cols; ++w) {
                const int s = b * spb + w;
                if (P.wl_s[s] < S.slice_width[s]) { P.wl_s[s] += 4; off += 4; more = true; }
            }
        }
        off = 0;
        for (int w = 0; w < spb; ++w) { P.lds_off[b * spb + w] = off; off += P.wl_s[b * spb + w]; }
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
        for (int32_t pr : halo[b]) {
            const int bj = pr / T;
            if (sb[bj]) continue;
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
    if (!uniform_mass) { P.ok = false; return P; }   // k_pcg2 keeps ONE diagonal value per row (the launch path serves such a system)
    // Coarse functions of a block, as weights per row (row r of P).  Without coordinates: the indicator vectors of its kOcSub
    // compact aggregates.  With coordinates: {1, x, y, z}, centred on the block and scaled to its half-extent -- the same four
    // unknowns per block, but smooth error modes (the low-energy modes of the Laplacian-like Ahat, which carry most of the
    // POSITION error of an iterate) are represented to second order instead of to first.  CPU prototype on the bench scenes
    // (experiments/stop_rule_study.py): unstructured body, same iterations, 3.4-4.8x smaller position error at the stop;
    // Kuhn cube 16.6 -> 12.6 iterations per solve.  A direction in which a block has no extent (flat cloth, tiny blocks) gets a
    // zero weight column: its coarse unknown is empty (unit diagonal below), as an empty aggregate is.
    P.cwt.assign(4 * (size_t)P.n_rows, 0.0f);
    P.affine = xyz != nullptr && !(getenv("ADMM_HIP_OC_AFFINE") && getenv("ADMM_HIP_OC_AFFINE")[0] == '0');
    static_assert(kOcSub == 4, "four coarse functions per block");
    for (int b = 0; b < G; ++b) {
        const std::vector<int32_t> &mem = blocks[b];
        if (mem.empty()) continue;
        if (!P.affine) {
            for (int32_t v : mem) P.cwt[4 * (size_t)P.pos[v] + agg_part[v]] = 1.0f;
            continue;
        }
        double c[3] = {0.0, 0.0, 0.0}, h[3] = {0.0, 0.0, 0.0}, hmax = 0.0;
        for (int32_t v : mem) for (int k = 0; k < 3; ++k) c[k] += xyz[3 * (size_t)v + k];
        for (int k = 0; k < 3; ++k) c[k] /= (double)mem.size();
        for (int32_t v : mem) for (int k = 0; k < 3; ++k) h[k] = std::max(h[k], std::fabs(xyz[3 * (size_t)v + k] - c[k]));
        for (int k = 0; k < 3; ++k) hmax = std::max(hmax, h[k]);
        for (int32_t v : mem) {
            float *wt = &P.cwt[4 * (size_t)P.pos[v]];
            wt[0] = 1.0f;
            for (int k = 0; k < 3; ++k)       // (a direction with < 1e-6 of the block's extent, or fewer than 4 vertices: no function)
                wt[1 + k] = (mem.size() >= 4 && h[k] > 1e-6 * hmax && hmax > 0.0) ? (float)((xyz[3 * (size_t)v + k] - c[k]) / h[k]) : 0.0f;
        }
    }
    if (P.affine) {
        // Energy-orthonormal functions per block.  {1, x, y, z} as they stand make P^T A P badly conditioned wherever a block holds
        // a vertex with a huge diagonal entry (a SpringPin: dt^2 w^2 ~ 1e7 against ~1e1): its 4 x 4 diagonal block is a rank-one
        // giant plus a small rest, the inverse lives on cancellation, and the kernel keeps that inverse in SINGLE precision --
        // on the pinned cloth the rounded preconditioner was indefinite (the pipelined pass broke off after 38 iterations and the
        // solve finished in the Jacobi end game, 148 iterations instead of 39).  So the four functions of a block are replaced by
        // combinations of themselves that are orthonormal in the energy of the block's own part of A (Cholesky of the 4 x 4 Gram
        // matrix G_b = P_b^T A_bb P_b, P_b <- P_b L^-T): same span, same preconditioner in exact arithmetic, the diagonal blocks of
        // P^T A P become identities.  A function whose pivot vanishes (no extent in that direction) drops out.
        std::vector<double> Gb((size_t)G * 16, 0.0);
        for (int32_t v = 0; v < nv; ++v) {
            const float *wv = &P.cwt[4 * (size_t)P.pos[v]];
            const int b = part_of[v];
            bool diag_seen = false;
            for (int32_t q = A.rowptr[v]; q < A.rowptr[v + 1]; ++q) {
                const int32_t u = A.col[q];
                if (part_of[u] != b) continue;
                if (u == v) diag_seen = true;
                const double a = A.val[q] + (u == v ? mass3[3 * (size_t)v] : 0.0);
                const float *wu = &P.cwt[4 * (size_t)P.pos[u]];
                for (int k = 0; k < 4; ++k) for (int l = 0; l < 4; ++l) Gb[(size_t)b * 16 + 4 * k + l] += (double)wv[k] * a * (double)wu[l];
            }
            if (!diag_seen) for (int k = 0; k < 4; ++k) for (int l = 0; l < 4; ++l) Gb[(size_t)b * 16 + 4 * k + l] += (double)wv[k] * mass3[3 * (size_t)v] * (double)wv[l];
        }
        for (int b = 0; b < G; ++b) {
            double *g = &Gb[(size_t)b * 16], L[16] = {0.0};
            bool keep[4];
            double dmax = 0.0;
            for (int k = 0; k < 4; ++k) dmax = std::max(dmax, g[5 * k]);
            for (int j = 0; j < 4; ++j) {     // Cholesky with dropped pivots: L L^T = G on the kept functions
                double d = g[5 * j];
                for (int k = 0; k < j; ++k) if (keep[k]) d -= L[4 * j + k] * L[4 * j + k];
                keep[j] = d > 1e-12 * dmax && dmax > 0.0;
                if (!keep[j]) continue;
                L[5 * j] = std::sqrt(d);
                for (int i = j + 1; i < 4; ++i) {
                    double sm = g[4 * i + j];
                    for (int k = 0; k < j; ++k) if (keep[k]) sm -= L[4 * i + k] * L[4 * j + k];
                    L[4 * i + j] = sm / L[5 * j];
                }
            }
            for (int32_t v : blocks[b]) {       // new weights y = L^-1 w (forward substitution over the kept functions)
                float *wt = &P.cwt[4 * (size_t)P.pos[v]];
                double y[4];
                for (int j = 0; j < 4; ++j) {
                    if (!keep[j]) { y[j] = 0.0; continue; }
                    double sm = (double)wt[j];
                    for (int k = 0; k < j; ++k) if (keep[k]) sm -= L[4 * j + k] * y[k];
                    y[j] = sm / L[5 * j];
                }
                for (int j = 0; j < 4; ++j) wt[j] = (float)y[j];
            }
        }
    }
    if (want_coarse && P.nc <= 2048) {
        const int nc = P.nc;
        std::vector<double> Ac((size_t)nc * nc, 0.0);
        for (int32_t v = 0; v < nv; ++v) {
            const float *wv = &P.cwt[4 * (size_t)P.pos[v]];
            const int bv = part_of[v] * kOcSub;
            for (int32_t q = A.rowptr[v]; q < A.rowptr[v + 1]; ++q) {
                const int32_t u = A.col[q];
                const double a = A.val[q] + (u == v ? mass3[3 * (size_t)v] : 0.0);
                if (a == 0.0) continue;
                const float *wu = &P.cwt[4 * (size_t)P.pos[u]];
                const int bu = part_of[u] * kOcSub;
                for (int k = 0; k < kOcSub; ++k) {
                    if (wv[k] == 0.0f) continue;
                    double *row = &Ac[(size_t)(bv + k) * nc + bu];
                    for (int l = 0; l < kOcSub; ++l) row[l] += (double)wv[k] * a * (double)wu[l];
                }
            }
        }
        {   // (a vertex without a stored diagonal entry still has its mass)
            std::vector<char> has_diag(nv, 0);
            for (int32_t v = 0; v < nv; ++v)
                for (int32_t q = A.rowptr[v]; q < A.rowptr[v + 1]; ++q) if (A.col[q] == v) has_diag[v] = 1;
            for (int32_t v = 0; v < nv; ++v)
                if (!has_diag[v]) {
                    const float *wv = &P.cwt[4 * (size_t)P.pos[v]];
                    const int bv = part_of[v] * kOcSub;
                    for (int k = 0; k < kOcSub; ++k) for (int l = 0; l < kOcSub; ++l) Ac[(size_t)(bv + k) * nc + bv + l] += (double)wv[k] * mass3[3 * (size_t)v] * (double)wv[l];
                }
        }
        // empty coarse unknowns (empty aggregates, directions without extent): unit diagonal, they never receive a residual
        for (int c = 0; c < nc; ++c) if (Ac[(size_t)c * nc + c] == 0.0) Ac[(size_t)c * nc + c] = 1.0;
        for (int i = 0; i < nc; ++i)   // symmetrise the round-off
            for (int j = 0; j < i; ++j) { const double sm = 0.5 * (Ac[(size_t)i * nc + j] + Ac[(size_t)j * nc + i]); Ac[(size_t)i * nc + j] = sm; Ac[(size_t)j * nc + i] = sm; }
        if (spd_inverse(nc, Ac)) {
            P.ainv.assign((size_t)nc * P.ncp, 0.0);
            for (int i = 0; i < nc; ++i) std::memcpy(&P.ainv[(size_t)i * P.ncp], &Ac[(size_t)i * nc], nc * sizeof(double));
            P.coarse_ok = true;
        }
    }
    P.ok = true;
    // ---- largest eigenvalue of D^-1 A_bb (A_bb = entries of M + Ahat inside one block, D = its diagonal; the smallest of the
    // three axes' masses makes the bound safe for all of them): the block-local Chebyshev smoother of k_pcg2 needs an upper
    // bound.  Power iteration on the symmetrised operator, deterministic start, 40 steps (the estimate approaches from below;
    // the caller adds its margin).
    {
        std::vector<double> dis(nv);
        for (int32_t v = 0; v < nv; ++v) {
            double aii = 0.0;
            for (int32_t k = A.rowptr[v]; k < A.rowptr[v + 1]; ++k) if (A.col[k] == v) aii += A.val[k];
            const double mm = std::min(mass3[3 * (size_t)v], std::min(mass3[3 * (size_t)v + 1], mass3[3 * (size_t)v + 2]));
            dis[v] = 1.0 / std::sqrt(mm + aii);
        }
        // Power iteration on the symmetrised operator.  The estimate approaches lambda_max from BELOW, and a top eigenvector that
        // is localised (a few stiff or sliver elements, heterogeneous materials) is found late from a smooth start: three
        // deterministic starts (smooth, alternating, pseudo-random), each run until its Rayleigh quotient stagnates (relative
        // change < 1e-5 over ten steps, at most 400 steps); the largest wins.  The caller caps it with the Gershgorin bound below.
        std::vector<double> x(nv), y(nv);
        double lam = 0.0;
        for (int start = 0; start < 3; ++start) {
            for (int32_t v = 0; v < nv; ++v) {
                const uint32_t hsh = (uint32_t)v * 2654435761u;
                x[v] = start == 0 ? 1.0 + 0.5 * std::sin(0.7 * v + 0.3) : start == 1 ? ((v & 1) ? -1.0 : 1.0) * (1.0 + 0.1 * std::sin(0.3 * v)) : (double)(hsh >> 8) / 8388608.0 - 1.0;
            }
            double rq_prev10 = 0.0, rq = 0.0;
            for (int it = 0; it < 400; ++it) {
                double nrm = 0.0;
                for (int32_t v = 0; v < nv; ++v) nrm += x[v] * x[v];
                nrm = 1.0 / std::sqrt(nrm);
                for (int32_t v = 0; v < nv; ++v) x[v] *= nrm;
                rq = 0.0;
                for (int32_t v = 0; v < nv; ++v) {
                    double acc = 0.0;
                    const int32_t bv = part_of[v];
                    for (int32_t k = A.rowptr[v]; k < A.rowptr[v + 1]; ++k) {
                        const int32_t c = A.col[k];
                        if (c != v && part_of[c] == bv) acc += A.val[k] * dis[c] * x[c];
                    }
                    y[v] = x[v] + dis[v] * acc;        // (unit diagonal of the scaled operator)
                    rq += x[v] * y[v];
                }
                x.swap(y);
                if (it % 10 == 9) {
                    if (it >= 39 && std::fabs(rq - rq_prev10) <= 1e-5 * rq) break;
                    rq_prev10 = rq;
                }
            }
            lam = std::max(lam, rq);
        }
        {   // Gershgorin: 1 + max_v sum_{c != v, same block} |a_vc| / sqrt(a_vv a_cc)
            double g = 1.0;
            for (int32_t v = 0; v < nv; ++v) {
                double row = 1.0;
                const int32_t bv = part_of[v];
                for (int32_t k = A.rowptr[v]; k < A.rowptr[v + 1]; ++k) {
                    const int32_t c = A.col[k];
                    if (c != v && part_of[c] == bv) row += std::fabs(A.val[k]) * dis[v] * dis[c];
                }
                g = std::max(g, row);
            }
            P.lam_bb_gersh = g;
        }
        P.lam_bb = lam;
    }
    return P;
}


// ---- plan of the launch-path two-level PCG (pcg_big.hpp) -------------------------------------------------------------------------
BigPlan build_big_plan(const Csr &A, const double *mass3, const double *xyz, int max_aggregates) {
    BigPlan P;
    const int32_t nv = A.n;
    for (int32_t v = 0; v < nv; ++v)
        if (!(mass3[3 * (size_t)v] == mass3[3 * (size_t)v + 1] && mass3[3 * (size_t)v] == mass3[3 * (size_t)v + 2])) return P;   // one coarse operator for the three axes
    // aggregates of ~768 rows (the block size the on-chip solver's coarse space was tuned on), more only to keep the dense coarse inverse
    // at <= 4 max_aggregates unknowns
    int ra = 768;
    while ((nv + ra - 1) / ra > max_aggregates) ra += 256;
    const int G = std::max(1, (nv + ra - 1) / ra);
    P.G = G; P.ra = ra; P.n_rows = G * ra;
    const Graph g = coupling_graph(A);
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
    P.orig.assign(P.n_rows, -1); P.pos.assign(nv, -1);
    for (int b = 0; b < G; ++b) {      // rows of an aggregate breadth-first: neighbouring lanes gather neighbouring entries
        std::vector<int32_t> &mem = blocks[b];
        if ((int)mem.size() > ra) return P;
        if (mem.empty()) continue;
        const int32_t idb = next_id++;
        for (int32_t v : mem) mark[v] = idb;
        std::vector<int32_t> bfs;
        bfs_order(g, mem, mark, idb, mem[0], seen, bfs);
        // longest rows first (the 64 rows of a SELL slice then have similar lengths: less padding to stream), rows of equal length breadth-first
        std::vector<int32_t> rank_of(bfs.size());
        std::vector<std::pair<int32_t, int32_t> > key(bfs.size());
        for (size_t i = 0; i < bfs.size(); ++i) key[i] = std::make_pair(-(g.ptr[bfs[i] + 1] - g.ptr[bfs[i]]), (int32_t)i);
        std::sort(key.begin(), key.end());
        int32_t slot = b * ra;
        for (const auto &kv : key) { const int32_t v = bfs[kv.second]; P.orig[slot] = v; P.pos[v] = slot; ++slot; }
    }
    // ---- A in the internal order (rows of dummy slots: empty) ----
    {
        Csr B; B.n = P.n_rows; B.rowptr.assign(P.n_rows + 1, 0);
        std::vector<std::pair<int32_t, double> > row;
        for (int32_t r = 0; r < P.n_rows; ++r) {
            const int32_t v = P.orig[r];
            if (v >= 0) {
                row.clear();
                for (int32_t k = A.rowptr[v]; k < A.rowptr[v + 1]; ++k) if (A.val[k] != 0.0 || A.col[k] == v) row.emplace_back(P.pos[A.col[k]], A.val[k]);
                std::sort(row.begin(), row.end());
                for (auto &e : row) { B.col.push_back(e.first); B.val.push_back(e.second); }
            }
            B.rowptr[r + 1] = (int32_t)B.col.size();
        }
        P.A = csr_to_sell(B);
    }
    P.mass.assign(3 * (size_t)P.n_rows, 1.0); P.dinv.assign(3 * (size_t)P.n_rows, 0.0);
    for (int32_t r = 0; r < P.n_rows; ++r) {
        const int32_t v = P.orig[r];
        if (v < 0) continue;
        double aii = 0.0;
        for (int32_t k = A.rowptr[v]; k < A.rowptr[v + 1]; ++k) if (A.col[k] == v) aii += A.val[k];
        for (int j = 0; j < 3; ++j) { P.mass[3 * (size_t)r + j] = mass3[3 * (size_t)v + j]; P.dinv[3 * (size_t)r + j] = 1.0 / (mass3[3 * (size_t)v + j] + aii); }
    }
    // ---- coarse space: {1, x, y, z} per aggregate (constants only without coordinates), energy-orthonormalised per aggregate (as in
    //      build_oc_plan: the diagonal blocks of P^T A P become identities, the dense inverse stays well conditioned in single precision) ----
    P.nc = 4 * G; P.ncp = (P.nc + 63) / 64 * 64;
    P.cwt.assign(4 * (size_t)P.n_rows, 0.0);
    for (int b = 0; b < G; ++b) {
        const std::vector<int32_t> &mem = blocks[b];
        if (mem.empty()) continue;
        double c[3] = {0.0, 0.0, 0.0}, h[3] = {0.0, 0.0, 0.0}, hmax = 0.0;
        if (xyz) {
            for (int32_t v : mem) for (int k = 0; k < 3; ++k) c[k] += xyz[3 * (size_t)v + k];
            for (int k = 0; k < 3; ++k) c[k] /= (double)mem.size();
            for (int32_t v : mem) for (int k = 0; k < 3; ++k) h[k] = std::max(h[k], std::fabs(xyz[3 * (size_t)v + k] - c[k]));
            for (int k = 0; k < 3; ++k) hmax = std::max(hmax, h[k]);
        }
        for (int32_t v : mem) {
            double *wt = &P.cwt[4 * (size_t)P.pos[v]];
            wt[0] = 1.0;
            for (int k = 0; k < 3; ++k) wt[1 + k] = (xyz && mem.size() >= 4 && h[k] > 1e-6 * hmax && hmax > 0.0) ? (xyz[3 * (size_t)v + k] - c[k]) / h[k] : 0.0;
        }
    }
    {
        std::vector<double> Gb((size_t)G * 16, 0.0);
        for (int32_t v = 0; v < nv; ++v) {
            const double *wv = &P.cwt[4 * (size_t)P.pos[v]];
            const int b = part_of[v];
            bool diag_seen = false;
            for (int32_t q = A.rowptr[v]; q < A.rowptr[v + 1]; ++q) {
                const int32_t u = A.col[q];
                if (part_of[u] != b) continue;
                if (u == v) diag_seen = true;
                const double a = A.val[q] + (u == v ? mass3[3 * (size_t)v] : 0.0);
                const double *wu = &P.cwt[4 * (size_t)P.pos[u]];
                for (int k = 0; k < 4; ++k) for (int l = 0; l < 4; ++l) Gb[(size_t)b * 16 + 4 * k + l] += wv[k] * a * wu[l];
            }
            if (!diag_seen) for (int k = 0; k < 4; ++k) for (int l = 0; l < 4; ++l) Gb[(size_t)b * 16 + 4 * k + l] += wv[k] * mass3[3 * (size_t)v] * wv[l];
        }
        for (int b = 0; b < G; ++b) {
            double *gm = &Gb[(size_t)b * 16], L[16] = {0.0};
            bool keep[4];
            double dmax = 0.0;
            for (int k = 0; k < 4; ++k) dmax = std::max(dmax, gm[5 * k]);
            for (int j = 0; j < 4; ++j) {
                double d = gm[5 * j];
                for (int k = 0; k < j; ++k) if (keep[k]) d -= L[4 * j + k] * L[4 * j + k];
                keep[j] = d > 1e-12 * dmax && dmax > 0.0;
                if (!keep[j]) continue;
                L[5 * j] = std::sqrt(d);
                for (int i = j + 1; i < 4; ++i) {
                    double sm = gm[4 * i + j];
                    for (int k = 0; k < j; ++k) if (keep[k]) sm -= L[4 * i + k] * L[4 * j + k];
                    L[4 * i + j] = sm / L[5 * j];
                }
            }
            for (int32_t v : blocks[b]) {
                double *wt = &P.cwt[4 * (size_t)P.pos[v]], y[4];
                for (int j = 0; j < 4; ++j) {
                    if (!keep[j]) { y[j] = 0.0; continue; }
                    double sm = wt[j];
                    for (int k = 0; k < j; ++k) if (keep[k]) sm -= L[4 * j + k] * y[k];
                    y[j] = sm / L[5 * j];
                }
                for (int j = 0; j < 4; ++j) wt[j] = y[j];
            }
        }
    }
    {
        const int nc = P.nc;
        std::vector<double> Ac((size_t)nc * nc, 0.0);
        std::vector<char> has_diag(nv, 0);
        for (int32_t v = 0; v < nv; ++v) {
            const double *wv = &P.cwt[4 * (size_t)P.pos[v]];
            const int bv = part_of[v] * 4;
            for (int32_t q = A.rowptr[v]; q < A.rowptr[v + 1]; ++q) {
                const int32_t u = A.col[q];
                if (u == v) has_diag[v] = 1;
                const double a = A.val[q] + (u == v ? mass3[3 * (size_t)v] : 0.0);
                if (a == 0.0) continue;
                const double *wu = &P.cwt[4 * (size_t)P.pos[u]];
                const int bu = part_of[u] * 4;
                for (int k = 0; k < 4; ++k) {
                    if (wv[k] == 0.0) continue;
                    double *row = &Ac[(size_t)(bv + k) * nc + bu];
                    for (int l = 0; l < 4; ++l) row[l] += wv[k] * a * wu[l];
                }
            }
        }
        for (int32_t v = 0; v < nv; ++v)
            if (!has_diag[v]) {
                const double *wv = &P.cwt[4 * (size_t)P.pos[v]];
                const int bv = part_of[v] * 4;
                for (int k = 0; k < 4; ++k) for (int l = 0; l < 4; ++l) Ac[(size_t)(bv + k) * nc + bv + l] += wv[k] * mass3[3 * (size_t)v] * wv[l];
            }
        for (int c = 0; c < nc; ++c) if (Ac[(size_t)c * nc + c] == 0.0) Ac[(size_t)c * nc + c] = 1.0;
        for (int i = 0; i < nc; ++i)
            for (int j = 0; j < i; ++j) { const double sm = 0.5 * (Ac[(size_t)i * nc + j] + Ac[(size_t)j * nc + i]); Ac[(size_t)i * nc + j] = sm; Ac[(size_t)j * nc + i] = sm; }
        if (!spd_inverse(nc, Ac)) return P;
        P.ainv.assign((size_t)nc * P.ncp, 0.0f);
        for (int i = 0; i < nc; ++i) for (int j = 0; j < nc; ++j) P.ainv[(size_t)i * P.ncp + j] = (float)Ac[(size_t)i * nc + j];
    }
    P.ok = true;
    return P;
}

// ---- plan of the persistent multi-colour GS kernel (host_setup.hpp: GsPlan) -------------------------------------------------------
GsPlan build_gs_plan(const Csr &A, int n_colors, const int32_t *color, int max_blocks, int rows_target, int lds_limit) {
    GsPlan P;
    const int32_t nv = A.n;
    const int C = n_colors;
    if (C < 1 || C > kGspMaxC || nv < 1 || max_blocks < 1) return P;
    const int G = std::max(1, std::min(max_blocks, (nv + std::max(rows_target, 1) - 1) / std::max(rows_target, 1)));
    P.G = G; P.C = C;
    const Graph g = coupling_graph(A);
    std::vector<int32_t> part_of(nv, 0), mark(nv, 0);
    {
        std::vector<char> seen(nv, 0);
        std::vector<int32_t> sizes(G), members(nv);
        for (int b = 0; b < G; ++b) sizes[b] = (int32_t)(((int64_t)nv * (b + 1)) / G - ((int64_t)nv * b) / G);
        std::iota(members.begin(), members.end(), 0);
        int32_t next_id = 1;
        bisect(g, members, sizes.data(), G, 0, mark, next_id, seen, part_of);
    }
    // boundary rows: referenced by a row of another block
    std::vector<char> boundary(nv, 0);
    for (int32_t v = 0; v < nv; ++v)
        for (int32_t k = g.ptr[v]; k < g.ptr[v + 1]; ++k)
            if (part_of[g.adj[k]] != part_of[v]) { boundary[v] = 1; break; }
    // own rows per block: by colour, boundary rows first, then by vertex index
    std::vector<std::vector<int32_t> > rows(G);
    for (int32_t v = 0; v < nv; ++v) rows[part_of[v]].push_back(v);
    std::vector<int32_t> local(nv, -1), ob_of(nv, -1);
    P.hdr.assign((size_t)G * kGspHdr, 0);
    int32_t row_base = 0, ob_base = 0;
    for (int b = 0; b < G; ++b) {
        std::vector<int32_t> &r = rows[b];
        std::stable_sort(r.begin(), r.end(), [&](int32_t x, int32_t y) {
            if (color[x] != color[y]) return color[x] < color[y];
            if (boundary[x] != boundary[y]) return boundary[x] > boundary[y];
            return x < y;
        });
        int32_t *h = &P.hdr[(size_t)b * kGspHdr];
        h[0] = (int32_t)r.size(); h[2] = row_base; h[5] = ob_base;
        int32_t n_out = 0;
        for (size_t i = 0; i < r.size(); ++i) {
            local[r[i]] = (int32_t)i;
            h[8 + color[r[i]] + 1] += 1;
            if (boundary[r[i]]) ob_of[r[i]] = n_out++;
        }
        for (int c = 0; c < C; ++c) h[8 + c + 1] += h[8 + c];
        h[6] = n_out;
        for (int32_t v : r) { P.orig.push_back(v); P.out_idx.push_back(ob_of[v]); }
        row_base += (int32_t)r.size(); ob_base += n_out;
        P.max_rows = std::max(P.max_rows, (int32_t)r.size());
    }
    P.ob_total = ob_base;
    P.diag.assign(P.orig.size(), 0.0);
    // halo lists per block: by colour, then by (source block, source row)
    int32_t halo_base = 0; int64_t ent_base = 0;
    for (int b = 0; b < G; ++b) {
        int32_t *h = &P.hdr[(size_t)b * kGspHdr];
        std::vector<int32_t> halo;
        for (int32_t v : rows[b])
            for (int32_t k = g.ptr[v]; k < g.ptr[v + 1]; ++k)
                if (part_of[g.adj[k]] != b) halo.push_back(g.adj[k]);
        std::sort(halo.begin(), halo.end());
        halo.erase(std::unique(halo.begin(), halo.end()), halo.end());
        std::stable_sort(halo.begin(), halo.end(), [&](int32_t x, int32_t y) {
            if (color[x] != color[y]) return color[x] < color[y];
            if (part_of[x] != part_of[y]) return part_of[x] < part_of[y];
            return local[x] < local[y];
        });
        const int32_t n_own = h[0], n_halo = (int32_t)halo.size();
        if (n_own + n_halo > 32767) return P;       // (bit 15 of a column flags "the neighbour's colour is below the row's")
        h[1] = n_halo; h[3] = halo_base;
        std::vector<int32_t> hloc(n_halo);
        {
            std::vector<int32_t> nb;
            for (int32_t i = 0; i < n_halo; ++i) {
                const int32_t v = halo[i];
                h[21 + color[v] + 1] += 1;
                P.halo_box.push_back(P.hdr[(size_t)part_of[v] * kGspHdr + 5] + ob_of[v]);
                P.halo_orig.push_back(v);
                nb.push_back(part_of[v]);
            }
            std::sort(nb.begin(), nb.end()); nb.erase(std::unique(nb.begin(), nb.end()), nb.end());
            P.max_nbr = std::max(P.max_nbr, (int32_t)nb.size());
        }
        for (int c = 0; c < C; ++c) h[21 + c + 1] += h[21 + c];
        P.max_halo = std::max(P.max_halo, n_halo);
        // local index of a halo vertex: binary search would need the sort key; a map through a scratch array instead
        std::vector<std::pair<int32_t, int32_t> > hmap(n_halo);
        for (int32_t i = 0; i < n_halo; ++i) hmap[i] = std::make_pair(halo[i], n_own + i);
        std::sort(hmap.begin(), hmap.end());
        auto loc_of = [&](int32_t v) -> int32_t {
            if (part_of[v] == b) return local[v];
            return std::lower_bound(hmap.begin(), hmap.end(), std::make_pair(v, (int32_t)-1))->second;
        };
        // entries: per colour an ELL of the colour's rows
        h[4] = (int32_t)ent_base;
        int32_t eoff = 0;
        for (int c = 0; c < C; ++c) {
            const int32_t r0 = h[8 + c], n_c = h[8 + c + 1] - r0;
            int32_t W = 0;
            for (int32_t i = 0; i < n_c; ++i) {
                const int32_t v = rows[b][r0 + i];
                int32_t len = 0;
                for (int32_t q = A.rowptr[v]; q < A.rowptr[v + 1]; ++q) len += (A.col[q] != v && A.val[q] != 0.0) ? 1 : 0;
                W = std::max(W, len);
            }
            W = (W + 3) & ~3;      // whole groups of four entries: the kernel's row loop has no remainder loop (padding = 0 x own value)
            h[34 + c] = W; h[46 + c] = eoff;
            const size_t base = P.vals.size();
            P.vals.resize(base + (size_t)W * n_c, 0.0);
            P.cols.resize(base + (size_t)W * n_c, 0);
            for (int32_t i = 0; i < n_c; ++i) {
                const int32_t v = rows[b][r0 + i];
                int32_t k = 0;
                for (int32_t q = A.rowptr[v]; q < A.rowptr[v + 1]; ++q) {
                    if (A.col[q] == v) { P.diag[(size_t)h[2] + r0 + i] = A.val[q]; continue; }
                    if (A.val[q] == 0.0) continue;
                    P.vals[base + (size_t)k * n_c + i] = A.val[q];
                    P.cols[base + (size_t)k * n_c + i] = (uint16_t)(loc_of(A.col[q]) | (color[A.col[q]] < color[v] ? 0x8000 : 0));
                    ++k;
                }
                for (; k < W; ++k) P.cols[base + (size_t)k * n_c + i] = (uint16_t)(r0 + i);      // padding: 0 x own value
            }
            eoff += W * n_c;
        }
        h[7] = eoff;
        ent_base += eoff; halo_base += n_halo;
        if (ent_base > (int64_t)1 << 30) return P;
        P.lds_bytes = std::max(P.lds_bytes, gsp_lds_bytes(n_own, n_halo, eoff));
    }
    if (P.vals.empty()) { P.vals.push_back(0.0); P.cols.push_back(0); }
    if (P.halo_box.empty()) { P.halo_box.push_back(0); P.halo_orig.push_back(0); }
    P.ok = P.lds_bytes <= lds_limit;
    return P;
}

} // namespace admm_host
