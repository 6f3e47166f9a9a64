// host_setup.cpp -- host-side set-up arithmetic of Solver::initialize (reference src/Solver.cpp:167-261)
// re-designed for the GPU data layout.  No GPU calls.
#include "host_setup.hpp"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <utility>
#include <numeric>
#include <set>
#include <tuple>

namespace admm_host {

namespace {
struct Entry { int32_t col; double val; };
} // namespace

// A = M + dt^2 D^T W^2 D (src/Solver.cpp:225-226).  Every term writes the same scalar on the x, y and
// z rows (src/TetEnergyTerm.cpp:66-68, src/TriEnergyTerm.cpp:65-68, src/SpringEnergyTerm.hpp:56-58),
// so A = M + Ahat (x) I3 and only the scalar n_verts x n_verts Ahat is assembled.
Csr assemble_Ahat(int32_t n_verts, double dt,
                  int32_t n_tets, const int32_t *tet_idx, const double *tet_Binv, const double *tet_w,
                  int32_t n_tris, const int32_t *tri_idx, const double *tri_rest, const double *tri_w,
                  int32_t n_pins, const int32_t *pin_vert, double pin_w) {
    const double dt2 = dt * dt;
    // pass 1: count entries per row
    std::vector<int64_t> cnt(n_verts + 1, 0);
    for (int32_t t = 0; t < n_tets; ++t)
        for (int a = 0; a < 4; ++a) cnt[tet_idx[4 * t + a] + 1] += 4;
    for (int32_t t = 0; t < n_tris; ++t)
        for (int a = 0; a < 3; ++a) cnt[tri_idx[3 * t + a] + 1] += 3;
    for (int32_t p = 0; p < n_pins; ++p) cnt[pin_vert[p] + 1] += 1;
    for (int32_t i = 0; i < n_verts; ++i) cnt[i + 1] += cnt[i];
    std::vector<Entry> ent(cnt[n_verts]);
    std::vector<int64_t> pos(cnt.begin(), cnt.end() - 1);

    for (int32_t t = 0; t < n_tets; ++t) {
        const int32_t *id = tet_idx + 4 * t;
        const double *Bi = tet_Binv + 9 * t; // column-major: Bi[c*3+m] = Binv(m,c)
        const double w2 = tet_w[t] * tet_w[t] * dt2;
        double d[3][4]; // d[r][corner]
        for (int r = 0; r < 3; ++r) {
            d[r][0] = -(Bi[r * 3 + 0] + Bi[r * 3 + 1] + Bi[r * 3 + 2]);
            d[r][1] = Bi[r * 3 + 0]; d[r][2] = Bi[r * 3 + 1]; d[r][3] = Bi[r * 3 + 2];
        }
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b) {
                double v = 0.0;
                for (int r = 0; r < 3; ++r) v += d[r][a] * d[r][b];
                ent[pos[id[a]]++] = {id[b], w2 * v};
            }
    }
    for (int32_t t = 0; t < n_tris; ++t) {
        const int32_t *id = tri_idx + 3 * t;
        const double *R = tri_rest + 4 * t; // column-major 2x2: R[c*2+m] = rest(m,c)
        const double w2 = tri_w[t] * tri_w[t] * dt2;
        double d[2][3];
        for (int c = 0; c < 2; ++c) {
            d[c][0] = -(R[c * 2 + 0] + R[c * 2 + 1]);
            d[c][1] = R[c * 2 + 0]; d[c][2] = R[c * 2 + 1];
        }
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b)
                ent[pos[id[a]]++] = {id[b], w2 * (d[0][a] * d[0][b] + d[1][a] * d[1][b])};
    }
    for (int32_t p = 0; p < n_pins; ++p) ent[pos[pin_vert[p]]++] = {pin_vert[p], pin_w * pin_w * dt2};

    Csr A;
    A.n = n_verts;
    A.rowptr.assign(n_verts + 1, 0);
    A.col.reserve(ent.size() / 3);
    A.val.reserve(ent.size() / 3);
    for (int32_t i = 0; i < n_verts; ++i) {
        Entry *b = ent.data() + cnt[i], *e = ent.data() + cnt[i + 1];
        std::stable_sort(b, e, [](const Entry &x, const Entry &y) { return x.col < y.col; });
        bool has_diag = false;
        for (Entry *q = b; q < e;) {
            int32_t c = q->col;
            double s = 0.0;
            for (; q < e && q->col == c; ++q) s += q->val;
            if (c == i) has_diag = true;
            A.col.push_back(c);
            A.val.push_back(s);
        }
        if (!has_diag) { // isolated vertex: keep an explicit (zero) diagonal so every row has one
            // insert keeping columns sorted
            size_t start = A.rowptr[i], end = A.col.size();
            size_t ins = start;
            while (ins < end && A.col[ins] < i) ++ins;
            A.col.insert(A.col.begin() + ins, i);
            A.val.insert(A.val.begin() + ins, 0.0);
        }
        A.rowptr[i + 1] = (int32_t)A.col.size();
    }
    return A;
}

// Entries that are exactly 0.0 (structural cancellation, e.g. the face/body diagonals of a Kuhn
// triangulation: 8 of 15 entries per row) are not stored; the diagonal always is.
Sell csr_to_sell(const Csr &A) {
    Sell S;
    S.n_rows = A.n;
    S.n_slices = (A.n + 63) / 64;
    S.slice_ptr.assign(S.n_slices + 1, 0);
    S.slice_width.assign(S.n_slices, 0);
    auto keep = [&](int32_t r, int32_t k) { return A.val[k] != 0.0 || A.col[k] == r; };
    std::vector<int32_t> len(A.n, 0);
    for (int32_t r = 0; r < A.n; ++r)
        for (int32_t k = A.rowptr[r]; k < A.rowptr[r + 1]; ++k) len[r] += keep(r, k) ? 1 : 0;
    for (int32_t s = 0; s < S.n_slices; ++s) {
        int32_t w = 0;
        for (int32_t r = 64 * s; r < std::min(A.n, 64 * s + 64); ++r) w = std::max(w, len[r]);
        w = std::max(4, (w + 3) / 4 * 4); // kernels consume 4 entries per software-pipelined round
        S.slice_width[s] = w;
        S.slice_ptr[s + 1] = S.slice_ptr[s] + 64 * w;
    }
    S.idx.assign(S.slice_ptr[S.n_slices], 0);
    S.val.assign(S.slice_ptr[S.n_slices], 0.0);
    for (int32_t s = 0; s < S.n_slices; ++s)
        for (int32_t l = 0; l < 64; ++l) {
            const int32_t r = 64 * s + l;
            const int32_t rr = std::min(r, A.n - 1);
            int32_t k = 0;
            if (r < A.n)
                for (int32_t q = A.rowptr[r]; q < A.rowptr[r + 1]; ++q) {
                    if (!keep(r, q)) continue;
                    const size_t o = (size_t)S.slice_ptr[s] + 64 * k + l;
                    S.idx[o] = A.col[q]; S.val[o] = A.val[q];
                    ++k;
                }
            for (; k < S.slice_width[s]; ++k) { // padding: harmless self reference, zero value
                const size_t o = (size_t)S.slice_ptr[s] + 64 * k + l;
                S.idx[o] = rr; S.val[o] = 0.0;
            }
        }
    return S;
}

Sell incidence_sell(int32_t n_verts, int32_t n_elems, int32_t corners, const int32_t *idx, int32_t pad_code, const int32_t *row_vertex) {
    std::vector<int32_t> cnt(n_verts + 1, 0);
    for (int64_t i = 0; i < (int64_t)n_elems * corners; ++i) cnt[idx[i] + 1]++;
    for (int32_t i = 0; i < n_verts; ++i) cnt[i + 1] += cnt[i];
    std::vector<int32_t> lst(cnt[n_verts]);
    std::vector<int32_t> pos(cnt.begin(), cnt.end() - 1);
    for (int32_t e = 0; e < n_elems; ++e)
        for (int32_t c = 0; c < corners; ++c) lst[pos[idx[(int64_t)e * corners + c]]++] = e * 4 + c;
    Sell S;
    S.n_rows = n_verts;
    S.n_slices = (n_verts + 63) / 64;
    S.slice_ptr.assign(S.n_slices + 1, 0);
    S.slice_width.assign(S.n_slices, 0);
    auto vert = [&](int32_t r) { return row_vertex ? row_vertex[r] : r; };   // row r of the SELL gathers for this vertex
    for (int32_t s = 0; s < S.n_slices; ++s) {
        int32_t w = 0;
        for (int32_t r = 64 * s; r < std::min(n_verts, 64 * s + 64); ++r) w = std::max(w, cnt[vert(r) + 1] - cnt[vert(r)]);
        w = std::max(8, (w + 7) / 8 * 8); // the gather kernel consumes 8 incidences per pipelined round
        S.slice_width[s] = w;
        S.slice_ptr[s + 1] = S.slice_ptr[s] + 64 * w;
    }
    S.idx.assign(S.slice_ptr[S.n_slices], pad_code);
    for (int32_t r = 0; r < n_verts; ++r) {
        const int32_t s = r / 64, l = r % 64, v = vert(r);
        for (int32_t k = 0; k < cnt[v + 1] - cnt[v]; ++k) S.idx[(size_t)S.slice_ptr[s] + 64 * k + l] = lst[cnt[v] + k];
    }
    return S;
}

TetChunks tet_chunks(int32_t n_tets, const int32_t *tet_idx, const int32_t kind_begin[6]) {
    TetChunks C;
    C.group_base.push_back(0); C.rec_base.push_back(0);
    std::vector<std::pair<int32_t, int32_t>> vc;      // (vertex, tl * 4 + corner) of one chunk
    for (int k = 0; k < 5; ++k) {
        for (int32_t t0 = kind_begin[k]; t0 < kind_begin[k + 1]; t0 += 256) {
            const int32_t t1 = std::min(kind_begin[k + 1], t0 + 256);
            vc.clear();
            for (int32_t t = t0; t < t1; ++t)
                for (int c = 0; c < 4; ++c) vc.emplace_back(tet_idx[4 * (size_t)t + c], (t - t0) * 4 + c);
            std::sort(vc.begin(), vc.end());
            // records: runs of one vertex, cut after kChunkFan entries
            const size_t g0 = C.ent.size();
            int32_t nrec = 0;
            for (size_t i = 0; i < vc.size();) {
                size_t j = i;
                while (j < vc.size() && vc[j].first == vc[i].first && j - i < (size_t)kChunkFan) ++j;
                for (size_t e = i; e < i + kChunkFan; ++e) {
                    uint16_t off = kChunkPad;
                    if (e < j) { const int32_t tl = vc[e].second >> 2, c = vc[e].second & 3; off = (uint16_t)(((3 * c) * kChunkLd + tl) * 8); }
                    C.ent.push_back(off);
                }
                C.rec_vertex.push_back(vc[i].first);
                ++nrec;
                i = j;
            }
            const int32_t groups = std::max(1, (nrec + 255) / 256);
            C.ent.resize(g0 + (size_t)groups * 256 * kChunkFan, kChunkPad);
            C.group_base.push_back(C.group_base.back() + groups);
            C.rec_base.push_back(C.rec_base.back() + nrec);
            C.n_chunks += 1;
        }
    }
    C.n_rec = C.rec_base.back();
    (void)n_tets;
    return C;
}

Sell record_incidence(int32_t n_verts, int32_t n_rec, const int32_t *rec_vertex, int32_t pad_code, const int32_t *row_vertex, int32_t n_rows) {
    // n_rows >= 0: the lists of n_rows rows, row r gathering for vertex row_vertex[r] (< 0: a dummy row, empty list) -- the on-chip
    // solver's internal row order (k_pcg2 sums the right-hand side of its own rows); n_rows < 0: one row per vertex
    if (n_rows >= 0) {
        std::vector<int32_t> cntv(n_verts + 1, 0);
        for (int32_t i = 0; i < n_rec; ++i) cntv[rec_vertex[i] + 1]++;
        for (int32_t i = 0; i < n_verts; ++i) cntv[i + 1] += cntv[i];
        std::vector<int32_t> lst(cntv[n_verts]);
        std::vector<int32_t> pos(cntv.begin(), cntv.end() - 1);
        for (int32_t e = 0; e < n_rec; ++e) lst[pos[rec_vertex[e]]++] = e;
        Sell S;
        S.n_rows = n_rows;
        S.n_slices = (n_rows + 63) / 64;
        S.slice_ptr.assign(S.n_slices + 1, 0);
        S.slice_width.assign(S.n_slices, 0);
        auto len = [&](int32_t r) { const int32_t v = row_vertex[r]; return v < 0 ? 0 : cntv[v + 1] - cntv[v]; };
        for (int32_t s = 0; s < S.n_slices; ++s) {
            int32_t w = 0;
            for (int32_t r = 64 * s; r < std::min(n_rows, 64 * s + 64); ++r) w = std::max(w, len(r));
            w = std::max(4, (w + 3) / 4 * 4);
            S.slice_width[s] = w;
            S.slice_ptr[s + 1] = S.slice_ptr[s] + 64 * w;
        }
        S.idx.assign(S.slice_ptr[S.n_slices], pad_code);
        for (int32_t r = 0; r < n_rows; ++r) {
            const int32_t s = r / 64, l = r % 64, v = row_vertex[r];
            for (int32_t k = 0; k < len(r); ++k) S.idx[(size_t)S.slice_ptr[s] + 64 * k + l] = lst[cntv[v] + k];
        }
        return S;
    }
    std::vector<int32_t> cnt(n_verts + 1, 0);
    for (int32_t i = 0; i < n_rec; ++i) cnt[rec_vertex[i] + 1]++;
    for (int32_t i = 0; i < n_verts; ++i) cnt[i + 1] += cnt[i];
    std::vector<int32_t> lst(cnt[n_verts]);
    std::vector<int32_t> pos(cnt.begin(), cnt.end() - 1);
    for (int32_t e = 0; e < n_rec; ++e) lst[pos[rec_vertex[e]]++] = e;
    Sell S;
    S.n_rows = n_verts;
    S.n_slices = (n_verts + 63) / 64;
    S.slice_ptr.assign(S.n_slices + 1, 0);
    S.slice_width.assign(S.n_slices, 0);
    auto vert = [&](int32_t r) { return row_vertex ? row_vertex[r] : r; };
    for (int32_t s = 0; s < S.n_slices; ++s) {
        int32_t w = 0;
        for (int32_t r = 64 * s; r < std::min(n_verts, 64 * s + 64); ++r) w = std::max(w, cnt[vert(r) + 1] - cnt[vert(r)]);
        w = std::max(4, (w + 3) / 4 * 4);   // the gather consumes 4 records per pipelined round
        S.slice_width[s] = w;
        S.slice_ptr[s + 1] = S.slice_ptr[s] + 64 * w;
    }
    S.idx.assign(S.slice_ptr[S.n_slices], pad_code);
    for (int32_t r = 0; r < n_verts; ++r) {
        const int32_t s = r / 64, l = r % 64, v = vert(r);
        for (int32_t k = 0; k < cnt[v + 1] - cnt[v]; ++k) S.idx[(size_t)S.slice_ptr[s] + 64 * k + l] = lst[cnt[v] + k];
    }
    return S;
}

int32_t component_partition(int32_t nv, int32_t n_tets, const int32_t *tet_idx, int32_t n_tris, const int32_t *tri_idx, int world, int32_t *vertex_rank,
                            int32_t n_bends, const int32_t *bend_idx) {
    std::vector<int32_t> parent(nv);
    std::iota(parent.begin(), parent.end(), 0);
    auto find = [&](int32_t a) { while (parent[a] != a) { parent[a] = parent[parent[a]]; a = parent[a]; } return a; };
    auto unite = [&](int32_t a, int32_t b) { a = find(a); b = find(b); if (a != b) parent[std::max(a, b)] = std::min(a, b); };   // root = lowest vertex
    for (int32_t t = 0; t < n_tets; ++t) for (int k = 1; k < 4; ++k) unite(tet_idx[4 * (size_t)t], tet_idx[4 * (size_t)t + k]);
    for (int32_t t = 0; t < n_tris; ++t) for (int k = 1; k < 3; ++k) unite(tri_idx[3 * (size_t)t], tri_idx[3 * (size_t)t + k]);
    for (int32_t t = 0; t < n_bends; ++t) for (int k = 1; k < 4; ++k) unite(bend_idx[4 * (size_t)t], bend_idx[4 * (size_t)t + k]);   // (a hinge couples its four vertices)
    std::vector<int64_t> load_of_root(nv, 0);
    for (int32_t t = 0; t < n_tets; ++t) load_of_root[find(tet_idx[4 * (size_t)t])] += 1;
    for (int32_t t = 0; t < n_tris; ++t) load_of_root[find(tri_idx[3 * (size_t)t])] += 1;
    std::vector<int32_t> roots;
    for (int32_t v = 0; v < nv; ++v) if (find(v) == v) roots.push_back(v);
    std::stable_sort(roots.begin(), roots.end(), [&](int32_t a, int32_t b) { return load_of_root[a] != load_of_root[b] ? load_of_root[a] > load_of_root[b] : a < b; });
    std::vector<int64_t> load(std::max(world, 1), 0);
    std::vector<int32_t> rank_of_root(nv, 0);
    int32_t bodies = 0;      // components that own at least one element: loose vertices (no tet, no triangle) are not bodies --
    for (int32_t r : roots) {   // they go to the least loaded rank after the bodies (sorted last) and weigh nothing
        int best = 0;
        for (int k = 1; k < world; ++k) if (load[k] < load[best]) best = k;
        rank_of_root[r] = best;
        load[best] += load_of_root[r];
        bodies += load_of_root[r] > 0 ? 1 : 0;
    }
    for (int32_t v = 0; v < nv; ++v) vertex_rank[v] = rank_of_root[find(v)];
    return bodies;
}

// ---- 4-vertex stencil terms (bending hinges) ------------------------------------------------------------------------------------
Csr add_stencil_terms(const Csr &A, double dt, int32_t n, const int32_t *idx4, const double *coef4, const double *w) {
    if (n <= 0) return A;
    const double dt2 = dt * dt;
    const int32_t nv = A.n;
    std::vector<int64_t> cnt(nv + 1, 0);
    for (int32_t h = 0; h < n; ++h) for (int a = 0; a < 4; ++a) cnt[idx4[4 * (size_t)h + a] + 1] += 4;
    for (int32_t i = 0; i < nv; ++i) cnt[i + 1] += cnt[i];
    std::vector<Entry> ent(cnt[nv]);
    std::vector<int64_t> pos(cnt.begin(), cnt.end() - 1);
    for (int32_t h = 0; h < n; ++h) {
        const int32_t *id = idx4 + 4 * (size_t)h; const double *c = coef4 + 4 * (size_t)h;
        const double w2 = w[h] * w[h] * dt2;
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) ent[pos[id[a]]++] = {id[b], w2 * c[a] * c[b]};
    }
    Csr B; B.n = nv; B.rowptr.assign(nv + 1, 0);
    B.col.reserve(A.col.size() + ent.size() / 4); B.val.reserve(A.col.size() + ent.size() / 4);
    for (int32_t i = 0; i < nv; ++i) {
        Entry *b = ent.data() + cnt[i], *e = ent.data() + cnt[i + 1];
        std::stable_sort(b, e, [](const Entry &x, const Entry &y) { return x.col < y.col; });
        int32_t k = A.rowptr[i]; const int32_t ke = A.rowptr[i + 1];
        Entry *q = b;
        while (k < ke || q < e) {       // merge of two sorted lists; equal columns add (the stencil's entries after A's: fixed order)
            const int32_t ca = k < ke ? A.col[k] : 0x7fffffff, cb = q < e ? q->col : 0x7fffffff, cmin = std::min(ca, cb);
            double sum = 0.0;
            if (ca == cmin) sum = A.val[k++];
            for (; q < e && q->col == cmin; ++q) sum += q->val;
            B.col.push_back(cmin); B.val.push_back(sum);
        }
        B.rowptr[i + 1] = (int32_t)B.col.size();
    }
    return B;
}

int32_t bend_hinges(int32_t n_verts, int32_t n_tris, const int32_t *tris, const double *verts, int32_t cap, int32_t *hinge_idx, double *coef, double *area) {
    struct EdgeUse { int32_t a, b, tri, opp; };
    std::vector<EdgeUse> use; use.reserve(3 * (size_t)std::max(n_tris, 0));
    for (int32_t t = 0; t < n_tris; ++t)
        for (int e = 0; e < 3; ++e) {
            const int32_t p = tris[3 * (size_t)t + e], q = tris[3 * (size_t)t + (e + 1) % 3], o = tris[3 * (size_t)t + (e + 2) % 3];
            if (p < 0 || q < 0 || o < 0 || p >= n_verts || q >= n_verts || o >= n_verts) return -1;
            use.push_back({std::min(p, q), std::max(p, q), t, o});
        }
    std::stable_sort(use.begin(), use.end(), [](const EdgeUse &x, const EdgeUse &y) { return x.a != y.a ? x.a < y.a : x.b != y.b ? x.b < y.b : x.tri < y.tri; });
    auto cot_at = [&](int32_t p, int32_t q, int32_t r) {      // cotangent of the angle at p between (q - p) and (r - p)
        double u[3], v[3], cr[3];
        for (int j = 0; j < 3; ++j) { u[j] = verts[3 * (size_t)q + j] - verts[3 * (size_t)p + j]; v[j] = verts[3 * (size_t)r + j] - verts[3 * (size_t)p + j]; }
        cr[0] = u[1] * v[2] - u[2] * v[1]; cr[1] = u[2] * v[0] - u[0] * v[2]; cr[2] = u[0] * v[1] - u[1] * v[0];
        return (u[0] * v[0] + u[1] * v[1] + u[2] * v[2]) / std::sqrt(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]);
    };
    auto tri_area = [&](int32_t p, int32_t q, int32_t r) {
        double u[3], v[3], cr[3];
        for (int j = 0; j < 3; ++j) { u[j] = verts[3 * (size_t)q + j] - verts[3 * (size_t)p + j]; v[j] = verts[3 * (size_t)r + j] - verts[3 * (size_t)p + j]; }
        cr[0] = u[1] * v[2] - u[2] * v[1]; cr[1] = u[2] * v[0] - u[0] * v[2]; cr[2] = u[0] * v[1] - u[1] * v[0];
        return 0.5 * std::sqrt(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]);
    };
    int32_t n = 0;
    for (size_t i = 0; i < use.size();) {
        size_t j = i;
        while (j < use.size() && use[j].a == use[i].a && use[j].b == use[i].b) ++j;
        if (j - i == 2 && use[i].opp != use[i + 1].opp) {      // an interior, manifold edge
            const int32_t v0 = use[i].a, v1 = use[i].b, v2 = use[i].opp, v3 = use[i + 1].opp;
            if (n < cap && hinge_idx && coef && area) {
                const double c01 = cot_at(v0, v1, v2), c02 = cot_at(v0, v1, v3), c03 = cot_at(v1, v0, v2), c04 = cot_at(v1, v0, v3);
                hinge_idx[4 * (size_t)n] = v0; hinge_idx[4 * (size_t)n + 1] = v1; hinge_idx[4 * (size_t)n + 2] = v2; hinge_idx[4 * (size_t)n + 3] = v3;
                coef[4 * (size_t)n] = c03 + c04; coef[4 * (size_t)n + 1] = c01 + c02; coef[4 * (size_t)n + 2] = -(c01 + c03); coef[4 * (size_t)n + 3] = -(c02 + c04);
                area[n] = tri_area(v0, v1, v2) + tri_area(v0, v1, v3);
            }
            ++n;
        }
        i = j;
    }
    return n;
}

// ---- tabulated user splines --------------------------------------------------------------------------------------------------
int tabulate_spline(spline_fn fn, void *user, double s_min, double s_max, double *out) {
    if (!fn || !out || !(s_min > 0.0) || !(s_max > s_min) || !(s_max < 1e100)) return -1;
    for (int which = 0; which < 3; ++which) {       // f on stretches, g on products of two, h on products of three
        const double lo = std::pow(s_min, which + 1), hi = std::pow(s_max, which + 1);
        double *tab = out + (size_t)which * kSplineFnDoublesH;
        const double t0 = std::log(lo), dt = (std::log(hi) - t0) / (kSplineNodesH - 1);
        tab[0] = t0; tab[1] = dt; tab[2] = 1.0 / dt; tab[3] = (double)kSplineNodesH;
        for (int i = 0; i < kSplineNodesH; ++i) {
            const double x = std::exp(t0 + dt * i);
            const double F = fn(user, which, x), d1 = fn(user, which + 3, x);
            const double e = 1e-5;                  // second derivative: central difference of the spline's own first derivative
            const double d2 = (fn(user, which + 3, x * (1.0 + e)) - fn(user, which + 3, x * (1.0 - e))) / (2.0 * e * x);
            if (!std::isfinite(F) || !std::isfinite(d1) || !std::isfinite(d2)) return -2;
            tab[4 + 3 * i] = F;                     // F(e^t)
            tab[4 + 3 * i + 1] = x * d1;            // dF/dt   = x F'
            tab[4 + 3 * i + 2] = x * x * d2 + x * d1;   // d2F/dt2 = x^2 F'' + x F'
        }
    }
    return 0;
}
void spline_table_eval(const double *table, int which, double x, double *out3) {
    const double *tab = table + (size_t)which * kSplineFnDoublesH;
    const double t0 = tab[0], dt = tab[1], idt = tab[2];
    const int n = (int)tab[3];
    const double t = std::log(std::max(x, 1e-300));
    const double r = (t - t0) * idt;
    int i = (int)std::floor(r);
    i = i < 0 ? 0 : (i > n - 2 ? n - 2 : i);
    const double u = r - (double)i;
    const double *a = tab + 4 + 3 * i;
    double p, pt, ptt;
    if (u < 0.0 || u > 1.0) {      // outside the table: the Taylor quadratic in x of the nearest end node
        const double *e = u < 0.0 ? a : a + 3;
        const double xe = std::exp(u < 0.0 ? t0 : t0 + dt * (n - 1));
        const double d1 = e[1] / xe, d2 = (e[2] - e[1]) / (xe * xe), dx = x - xe;
        out3[0] = e[0] + dx * (d1 + 0.5 * dx * d2); out3[1] = d1 + dx * d2; out3[2] = d2;
        return;
    } else {
        const double F0 = a[0], G0 = a[1] * dt, H0 = a[2] * dt * dt, F1 = a[3], G1 = a[4] * dt, H1 = a[5] * dt * dt, dF = F1 - F0;
        const double c0 = F0, c1 = G0, c2 = 0.5 * H0;
        const double c3 = 10.0 * dF - 6.0 * G0 - 4.0 * G1 - 1.5 * H0 + 0.5 * H1;
        const double c4 = -15.0 * dF + 8.0 * G0 + 7.0 * G1 + 1.5 * H0 - H1;
        const double c5 = 6.0 * dF - 3.0 * (G0 + G1) - 0.5 * H0 + 0.5 * H1;
        p = c0 + u * (c1 + u * (c2 + u * (c3 + u * (c4 + u * c5))));
        pt = (c1 + u * (2.0 * c2 + u * (3.0 * c3 + u * (4.0 * c4 + u * 5.0 * c5)))) * idt;
        ptt = (2.0 * c2 + u * (6.0 * c3 + u * (12.0 * c4 + u * 20.0 * c5))) * idt * idt;
    }
    const double xx = std::max(x, 1e-300);
    out3[0] = p; out3[1] = pt / xx; out3[2] = (ptt - pt) / (xx * xx);
}

// Row order of the incidence lists: inside every window of 512 consecutive vertices the vertices with the most incident
// elements come first, so the 64 rows of a slice have similar lengths (unstructured 1 M-tet body: 2.17x -> 1.24x stored
// per real incidence) while a slice still gathers from one neighbourhood of the mesh.
std::vector<int32_t> incidence_row_order(int32_t n_verts, int32_t n_tets, const int32_t *tet_idx, int32_t n_tris, const int32_t *tri_idx, int32_t window) {
    std::vector<int32_t> cnt(n_verts, 0), order(n_verts);
    for (int64_t i = 0; i < (int64_t)4 * n_tets; ++i) cnt[tet_idx[i]]++;
    for (int64_t i = 0; i < (int64_t)3 * n_tris; ++i) cnt[tri_idx[i]]++;
    std::iota(order.begin(), order.end(), 0);
    if (window < 2) return order;
    for (int32_t b = 0; b < n_verts; b += window)
        std::stable_sort(order.begin() + b, order.begin() + std::min(n_verts, b + window), [&](int32_t x, int32_t y) { return cnt[x] > cnt[y]; });
    return order;
}

static int first_fit_natural(int32_t n, const int32_t *rowptr, const int32_t *col, int32_t *color) {
    int ncol = 0;
    std::vector<int32_t> mark;
    for (int32_t i = 0; i < n; ++i) color[i] = -1;
    for (int32_t i = 0; i < n; ++i) {
        mark.assign(ncol + 1, 0);
        for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
            const int32_t j = col[k];
            if (j != i && color[j] >= 0) mark[color[j]] = 1;
        }
        int c = 0;
        while (c < ncol && mark[c]) ++c;
        color[i] = c;
        if (c == ncol) ++ncol;
    }
    return ncol;
}

static int dsatur(int32_t n, const int32_t *rowptr, const int32_t *col, int32_t *color) {
    // neighbour-colour sets as 64-bit masks: gives up (returns a huge count) beyond 64 colours
    std::vector<uint64_t> used(n, 0);
    std::vector<int32_t> sat(n, 0), deg(n, 0);
    for (int32_t i = 0; i < n; ++i) {
        color[i] = -1;
        for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) deg[i] += col[k] != i;
    }
    typedef std::tuple<int32_t, int32_t, int32_t> Key;   // (-saturation, -degree, index): begin() = next vertex
    std::set<Key> queue;
    for (int32_t i = 0; i < n; ++i) queue.insert(Key(0, -deg[i], i));
    int ncol = 0;
    while (!queue.empty()) {
        const int32_t i = std::get<2>(*queue.begin());
        queue.erase(queue.begin());
        int c = 0;
        while (c < 64 && ((used[i] >> c) & 1ull)) ++c;
        if (c >= 64) return 1 << 30;
        color[i] = c;
        ncol = std::max(ncol, c + 1);
        for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
            const int32_t j = col[k];
            if (j == i || color[j] >= 0 || ((used[j] >> c) & 1ull)) continue;
            queue.erase(Key(-sat[j], -deg[j], j));
            used[j] |= 1ull << c; sat[j] += 1;
            queue.insert(Key(-sat[j], -deg[j], j));
        }
    }
    return ncol;
}

int greedy_coloring(int32_t n, const int32_t *rowptr, const int32_t *col, int32_t *color) {
    std::vector<int32_t> alt(std::max<int32_t>(n, 1));
    const int na = first_fit_natural(n, rowptr, col, color);
    const int nb = dsatur(n, rowptr, col, alt.data());
    if (nb < na) { std::copy(alt.begin(), alt.begin() + n, color); return nb; }
    return na;
}

GsSell build_gs_sell(const Csr &A, int n_colors, const std::vector<int32_t> &color) {
    GsSell g;
    std::vector<std::vector<int32_t> > by_color(n_colors);
    for (int32_t i = 0; i < A.n; ++i) by_color[color[i]].push_back(i);
    g.color_slice.assign(n_colors + 1, 0);
    for (int c = 0; c < n_colors; ++c) {
        const int32_t nsl = ((int32_t)by_color[c].size() + 63) / 64;
        g.color_slice[c + 1] = g.color_slice[c] + nsl;
        for (int32_t k = 0; k < nsl * 64; ++k) g.slot_node.push_back(k < (int32_t)by_color[c].size() ? by_color[c][k] : -1);
    }
    const int32_t n_slices = g.color_slice[n_colors];
    Sell &S = g.sell;
    S.n_rows = 64 * n_slices; S.n_slices = n_slices;
    S.slice_ptr.assign(n_slices + 1, 0); S.slice_width.assign(n_slices, 0);
    g.diag.assign((size_t)64 * n_slices, 1.0);
    auto keep = [&](int32_t r, int32_t k) { return A.val[k] != 0.0 && A.col[k] != r; }; // off-diagonal non-zeros
    for (int32_t s = 0; s < n_slices; ++s) {
        int32_t w = 0;
        for (int32_t l = 0; l < 64; ++l) {
            const int32_t r = g.slot_node[(size_t)64 * s + l];
            if (r < 0) continue;
            int32_t len = 0;
            for (int32_t k = A.rowptr[r]; k < A.rowptr[r + 1]; ++k) len += keep(r, k) ? 1 : 0;
            w = std::max(w, len);
        }
        w = std::max(4, (w + 3) / 4 * 4);
        S.slice_width[s] = w;
        S.slice_ptr[s + 1] = S.slice_ptr[s] + 64 * w;
    }
    S.idx.assign(S.slice_ptr[n_slices], 0);
    S.val.assign(S.slice_ptr[n_slices], 0.0);
    for (int32_t s = 0; s < n_slices; ++s)
        for (int32_t l = 0; l < 64; ++l) {
            const int32_t r = g.slot_node[(size_t)64 * s + l];
            int32_t k = 0;
            if (r >= 0)
                for (int32_t q = A.rowptr[r]; q < A.rowptr[r + 1]; ++q) {
                    if (A.col[q] == r) { g.diag[(size_t)64 * s + l] = A.val[q]; continue; }
                    if (!keep(r, q)) continue;
                    const size_t o = (size_t)S.slice_ptr[s] + 64 * k + l;
                    S.idx[o] = A.col[q]; S.val[o] = A.val[q];
                    ++k;
                }
            for (; k < S.slice_width[s]; ++k) { // padding: any valid column, zero value
                const size_t o = (size_t)S.slice_ptr[s] + 64 * k + l;
                S.idx[o] = r >= 0 ? r : 0; S.val[o] = 0.0;
            }
        }
    return g;
}

// src/TetEnergyTerm.cpp:31-48
int tet_rest(int32_t n, const int32_t *idx, const double *verts, double *Binv, double *vol) {
    for (int32_t t = 0; t < n; ++t) {
        const double *v0 = verts + 3 * idx[4 * t], *v1 = verts + 3 * idx[4 * t + 1];
        const double *v2 = verts + 3 * idx[4 * t + 2], *v3 = verts + 3 * idx[4 * t + 3];
        double B[3][3]; // B[r][c]
        for (int r = 0; r < 3; ++r) { B[r][0] = v1[r] - v0[r]; B[r][1] = v2[r] - v0[r]; B[r][2] = v3[r] - v0[r]; }
        const double det = B[0][0] * (B[1][1] * B[2][2] - B[1][2] * B[2][1])
                         - B[0][1] * (B[1][0] * B[2][2] - B[1][2] * B[2][0])
                         + B[0][2] * (B[1][0] * B[2][1] - B[1][1] * B[2][0]);
        vol[t] = det / 6.0;
        if (vol[t] < 0) return -(t + 1);
        const double id = 1.0 / det;
        double *o = Binv + 9 * t; // column-major o[c*3+r]
        o[0 * 3 + 0] =  (B[1][1] * B[2][2] - B[1][2] * B[2][1]) * id;
        o[1 * 3 + 0] = -(B[0][1] * B[2][2] - B[0][2] * B[2][1]) * id;
        o[2 * 3 + 0] =  (B[0][1] * B[1][2] - B[0][2] * B[1][1]) * id;
        o[0 * 3 + 1] = -(B[1][0] * B[2][2] - B[1][2] * B[2][0]) * id;
        o[1 * 3 + 1] =  (B[0][0] * B[2][2] - B[0][2] * B[2][0]) * id;
        o[2 * 3 + 1] = -(B[0][0] * B[1][2] - B[0][2] * B[1][0]) * id;
        o[0 * 3 + 2] =  (B[1][0] * B[2][1] - B[1][1] * B[2][0]) * id;
        o[1 * 3 + 2] = -(B[0][0] * B[2][1] - B[0][1] * B[2][0]) * id;
        o[2 * 3 + 2] =  (B[0][0] * B[1][1] - B[0][1] * B[1][0]) * id;
    }
    return 0;
}

// Rest POSITIONS behind the tets' edges_inv (TetEnergyTerm's constructor inverts the rest edge matrix of the vertex positions
// it is given, src/TetEnergyTerm.cpp:31-48; all tets of a mesh are built from ONE set of positions).  With them the local step
// recomputes Binv from 4 gathered positions (vertex data: cache-resident) instead of streaming 72 bytes per tet and launch.
// Candidates: `cand` (the caller's coordinates: exact when the solver is initialised in its rest state), else positions
// propagated from tet to tet through inv(Binv) (defined up to a translation per connected component, which no kernel sees).
// A candidate is accepted only if EVERY tet's recomputed Binv equals the given one to 1e-11 of its largest entry (the kernels then
// reproduce the streamed-Binv results to ~1e-13); 0 = the tets do not come from one set of positions: Binv stays streamed.
static bool rest_positions_match(int32_t nt, const int32_t *idx, const double *Binv, const double *x) {
    for (int32_t t = 0; t < nt; ++t) {
        double B[9], vol;
        if (tet_rest(1, idx + 4 * (size_t)t, x, B, &vol) != 0) return false;
        double big = 0.0, dif = 0.0;
        for (int k = 0; k < 9; ++k) {
            big = std::max(big, std::fabs(Binv[9 * (size_t)t + k]));
            dif = std::max(dif, std::fabs(Binv[9 * (size_t)t + k] - B[k]));
        }
        if (!(dif <= 1e-11 * big)) return false;      // (also catches NaN)
    }
    return true;
}
int tet_rest_positions(int32_t nv, int32_t nt, const int32_t *idx, const double *Binv, const double *cand, double *x0) {
    if (nt <= 0) return 0;
    for (size_t i = 0; i < (size_t)4 * nt; ++i) if (idx[i] < 0 || idx[i] >= nv) return 0;
    if (cand && rest_positions_match(nt, idx, Binv, cand)) { std::memcpy(x0, cand, sizeof(double) * 3 * (size_t)nv); return 1; }
    // vertex -> tets
    std::vector<int32_t> ptr((size_t)nv + 1, 0), inc((size_t)4 * nt);
    for (size_t i = 0; i < (size_t)4 * nt; ++i) ptr[idx[i] + 1]++;
    for (int32_t v = 0; v < nv; ++v) ptr[v + 1] += ptr[v];
    { std::vector<int32_t> fill(ptr.begin(), ptr.end() - 1); for (int32_t t = 0; t < nt; ++t) for (int k = 0; k < 4; ++k) inc[fill[idx[4 * (size_t)t + k]]++] = t; }
    std::vector<char> known(nv, 0), seen(nt, 0);
    std::fill(x0, x0 + 3 * (size_t)nv, 0.0);
    std::vector<int32_t> queue; queue.reserve(nt);
    for (int32_t seed = 0; seed < nt; ++seed) {
        if (seen[seed]) continue;
        known[idx[4 * (size_t)seed]] = 1;               // the component's translation: this vertex at the origin
        seen[seed] = 1; queue.clear(); queue.push_back(seed);
        for (size_t h = 0; h < queue.size(); ++h) {
            const int32_t t = queue[h];
            const int32_t *id = idx + 4 * (size_t)t;
            const double *b = Binv + 9 * (size_t)t;     // column-major b[c * 3 + r]
            // E = Binv^-1 (columns = edges x1 - x0, x2 - x0, x3 - x0), in long double
            long double m[3][3], e[3][3];
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) m[r][c] = b[c * 3 + r];
            const long double det = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0])
                                  + m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
            if (!(det > 0.0L) && !(det < 0.0L)) return 0;
            e[0][0] =  (m[1][1] * m[2][2] - m[1][2] * m[2][1]) / det; e[0][1] = -(m[0][1] * m[2][2] - m[0][2] * m[2][1]) / det; e[0][2] =  (m[0][1] * m[1][2] - m[0][2] * m[1][1]) / det;
            e[1][0] = -(m[1][0] * m[2][2] - m[1][2] * m[2][0]) / det; e[1][1] =  (m[0][0] * m[2][2] - m[0][2] * m[2][0]) / det; e[1][2] = -(m[0][0] * m[1][2] - m[0][2] * m[1][0]) / det;
            e[2][0] =  (m[1][0] * m[2][1] - m[1][1] * m[2][0]) / det; e[2][1] = -(m[0][0] * m[2][1] - m[0][1] * m[2][0]) / det; e[2][2] =  (m[0][0] * m[1][1] - m[0][1] * m[1][0]) / det;
            int a = -1;
            for (int k = 0; k < 4; ++k) if (known[id[k]]) { a = k; break; }
            if (a < 0) return 0;                        // (cannot happen: a tet is queued through a known vertex)
            long double rel[4][3];                      // corner k relative to corner 0
            for (int j = 0; j < 3; ++j) { rel[0][j] = 0.0L; for (int k = 1; k < 4; ++k) rel[k][j] = e[j][k - 1]; }
            for (int k = 0; k < 4; ++k) {
                if (known[id[k]]) continue;
                for (int j = 0; j < 3; ++j) x0[3 * (size_t)id[k] + j] = (double)((long double)x0[3 * (size_t)id[a] + j] + (rel[k][j] - rel[a][j]));
                known[id[k]] = 1;
            }
            for (int k = 0; k < 4; ++k)
                for (int32_t q = ptr[id[k]]; q < ptr[id[k] + 1]; ++q) if (!seen[inc[q]]) { seen[inc[q]] = 1; queue.push_back(inc[q]); }
        }
    }
    return rest_positions_match(nt, idx, Binv, x0) ? 2 : 0;
}

// src/TriEnergyTerm.cpp:29-52
int tri_rest(int32_t n, const int32_t *idx, const double *verts, double *rest, double *area) {
    for (int32_t t = 0; t < n; ++t) {
        const double *v0 = verts + 3 * idx[3 * t], *v1 = verts + 3 * idx[3 * t + 1], *v2 = verts + 3 * idx[3 * t + 2];
        double e12[3], e13[3], n1[3], n2[3];
        for (int r = 0; r < 3; ++r) { e12[r] = v1[r] - v0[r]; e13[r] = v2[r] - v0[r]; }
        double l = std::sqrt(e12[0] * e12[0] + e12[1] * e12[1] + e12[2] * e12[2]);
        for (int r = 0; r < 3; ++r) n1[r] = e12[r] / l;
        const double dp = e13[0] * n1[0] + e13[1] * n1[1] + e13[2] * n1[2];
        for (int r = 0; r < 3; ++r) n2[r] = e13[r] - dp * n1[r];
        l = std::sqrt(n2[0] * n2[0] + n2[1] * n2[1] + n2[2] * n2[2]);
        for (int r = 0; r < 3; ++r) n2[r] /= l;
        const double m00 = n1[0] * e12[0] + n1[1] * e12[1] + n1[2] * e12[2];
        const double m01 = n1[0] * e13[0] + n1[1] * e13[1] + n1[2] * e13[2];
        const double m10 = n2[0] * e12[0] + n2[1] * e12[1] + n2[2] * e12[2];
        const double m11 = n2[0] * e13[0] + n2[1] * e13[1] + n2[2] * e13[2];
        const double det = m00 * m11 - m01 * m10;
        area[t] = det / 2.0;
        if (area[t] < 0) return -(t + 1);
        double *o = rest + 4 * t; // column-major 2x2
        o[0] = m11 / det; o[1] = -m10 / det; o[2] = -m01 / det; o[3] = m00 / det;
    }
    return 0;
}

// src/EnergyTerm.hpp:34-59
void lame(double youngs, double poisson, double *mu, double *lambda, double *bulk) {
    *mu = youngs / (2.0 * (1.0 + poisson));
    *lambda = youngs * poisson / ((1.0 + poisson) * (1.0 - 2.0 * poisson));
    *bulk = *lambda + (2.0 / 3.0) * (*mu);
}

void partition(int32_t n_items, int world_size, int rank, int32_t *begin, int32_t *end) {
    const int64_t n = n_items;
    *begin = (int32_t)(n * rank / world_size);
    *end = (int32_t)(n * (rank + 1) / world_size);
}

} // namespace admm_host

// ---------------------------------------------------------------------------------------------------
// Implicit 8-ary tree over Morton-sorted primitives (dynamic self-collision)
namespace admm_host {

static inline uint64_t spread21(uint64_t v) { // 21 bits -> every third bit
    v &= 0x1fffffULL;
    v = (v | v << 32) & 0x1f00000000ffffULL;
    v = (v | v << 16) & 0x1f0000ff0000ffULL;
    v = (v | v << 8) & 0x100f00f00f00f00fULL;
    v = (v | v << 4) & 0x10c30c30c30c30c3ULL;
    v = (v | v << 2) & 0x1249249249249249ULL;
    return v;
}

OctTree build_octtree(int32_t n, const double *cen) {
    OctTree T;
    T.n_prims = n;
    T.n_padded = std::max(8, (n + 7) / 8 * 8);
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i)
        for (int c = 0; c < 3; ++c) {
            const double v = cen[3 * (size_t)i + c];
            if (i == 0 || v < lo[c]) lo[c] = v;
            if (i == 0 || v > hi[c]) hi[c] = v;
        }
    std::vector<std::pair<uint64_t, int32_t> > key(n);
    for (int i = 0; i < n; ++i) {
        uint64_t code = 0;
        for (int c = 0; c < 3; ++c) {
            const double ext = hi[c] - lo[c];
            double t = ext > 0 ? (cen[3 * (size_t)i + c] - lo[c]) / ext : 0.0;
            t = std::min(std::max(t, 0.0), 1.0);
            code |= spread21((uint64_t)(t * 2097151.0)) << c;
        }
        key[i] = std::make_pair(code, i);
    }
    std::sort(key.begin(), key.end());
    T.order.assign(T.n_padded, -1);
    for (int i = 0; i < n; ++i) T.order[i] = key[i].second;
    int cnt = T.n_padded / 8, off = 0;
    for (;;) {
        T.level_off.push_back(off); T.level_n.push_back(cnt);
        off += cnt;
        if (cnt == 1) break;
        cnt = (cnt + 7) / 8;
    }
    T.level_off.push_back(off);
    T.n_levels = (int)T.level_n.size();
    return T;
}

std::vector<double> octtree_boxes_tris(const OctTree &T, const int32_t *faces, const double *verts) {
    const double inf = std::numeric_limits<double>::infinity();
    std::vector<double> box(6 * (size_t)T.level_off[T.n_levels]);
    for (size_t i = 0; i < box.size(); i += 6) { box[i] = box[i + 1] = box[i + 2] = inf; box[i + 3] = box[i + 4] = box[i + 5] = -inf; }
    for (int i = 0; i < T.level_n[0]; ++i) {
        double *b = &box[6 * (size_t)i];
        for (int k = 0; k < 8; ++k) {
            const int f = T.order[8 * (size_t)i + k];
            if (f < 0) continue;
            for (int c = 0; c < 3; ++c) {
                const double *p = verts + 3 * (size_t)faces[3 * (size_t)f + c];
                for (int a = 0; a < 3; ++a) { b[a] = std::min(b[a], p[a]); b[3 + a] = std::max(b[3 + a], p[a]); }
            }
        }
    }
    for (int l = 1; l < T.n_levels; ++l)
        for (int i = 0; i < T.level_n[l]; ++i) {
            double *b = &box[6 * (size_t)(T.level_off[l] + i)];
            for (int k = 0; k < 8 && 8 * i + k < T.level_n[l - 1]; ++k) {
                const double *ch = &box[6 * (size_t)(T.level_off[l - 1] + 8 * i + k)];
                for (int a = 0; a < 3; ++a) { b[a] = std::min(b[a], ch[a]); b[3 + a] = std::max(b[3 + a], ch[3 + a]); }
            }
        }
    return box;
}

} // namespace admm_host

// ---------------------------------------------------------------------------------------------------
// Reverse Cuthill-McKee vertex ordering (mesh preprocessing; see admm_host_locality_order in admm_hip.h)
namespace admm_host {

static double mean_edge_span(int32_t n_elems, int32_t corners, const int32_t *idx, const int32_t *id) {
    double s = 0.0; long long cnt = 0;
    for (int e = 0; e < n_elems; ++e)
        for (int a = 0; a < corners; ++a)
            for (int b = a + 1; b < corners; ++b) {
                const int i = idx[(size_t)corners * e + a], j = idx[(size_t)corners * e + b];
                s += std::fabs((double)(id ? id[i] : i) - (double)(id ? id[j] : j)); ++cnt;
            }
    return cnt ? s / (double)cnt : 0.0;
}

void locality_order(int32_t nv, int32_t n_elems, int32_t corners, const int32_t *idx, int32_t *new_id, double *span_before,
                    double *span_after) {
    // adjacency (CSR, duplicates removed)
    std::vector<std::vector<int32_t> > adj(nv);
    for (int e = 0; e < n_elems; ++e)
        for (int a = 0; a < corners; ++a)
            for (int b = 0; b < corners; ++b)
                if (a != b) adj[idx[(size_t)corners * e + a]].push_back(idx[(size_t)corners * e + b]);
    for (auto &l : adj) { std::sort(l.begin(), l.end()); l.erase(std::unique(l.begin(), l.end()), l.end()); }
    std::vector<int32_t> order; order.reserve(nv);
    std::vector<char> seen(nv, 0);
    // start vertices: lowest degree first (a cheap stand-in for a pseudo-peripheral search), one per component
    std::vector<int32_t> by_degree(nv);
    std::iota(by_degree.begin(), by_degree.end(), 0);
    std::stable_sort(by_degree.begin(), by_degree.end(), [&](int a, int b) { return adj[a].size() < adj[b].size(); });
    for (int32_t start : by_degree) {
        if (seen[start]) continue;
        size_t head = order.size();
        order.push_back(start); seen[start] = 1;
        while (head < order.size()) {
            const int32_t v = order[head++];
            std::vector<int32_t> next;
            for (int32_t w : adj[v]) if (!seen[w]) { seen[w] = 1; next.push_back(w); }
            std::stable_sort(next.begin(), next.end(), [&](int a, int b) { return adj[a].size() < adj[b].size(); });
            order.insert(order.end(), next.begin(), next.end());
        }
    }
    for (int32_t i = 0; i < nv; ++i) new_id[order[nv - 1 - i]] = i;     // reversed
    if (span_before) *span_before = mean_edge_span(n_elems, corners, idx, nullptr);
    if (span_after) *span_after = mean_edge_span(n_elems, corners, idx, new_id);
}

} // namespace admm_host
