// dyn_collide.hpp -- dynamic (self-)collision on the GPU: TetMeshCollision (src/DynamicObject.hpp:31-121) as queried by
// Collider::detect (src/Collider.hpp:152-212), feeding the dynamic rows of ConstraintSet::make_matrix
// (src/ConstraintSet.hpp:92-110).  SURVEY 8(f) item 2.
//
// The reference rebuilds an mclscene AABB tree over the deformed tets at every detect (DynamicObject.hpp:66-69) and
// keeps a static tree over the REST surface triangles (:56-58).  Here both are implicit 8-ary trees over primitives
// sorted ONCE along a Morton curve of the rest centroids (host_setup.cpp: build_octtree): node i of level 0 covers
// sorted primitives [8i, 8i+8), node i of level l covers nodes [8i, 8i+8) of level l-1.  No pointers, no per-detect
// sort, no atomics: a detect REFITS the boxes of the tet tree bottom-up (one thread per node, one launch per level --
// 7 launches at 1 M tets) and queries it with eight lanes per candidate vertex (see k_dyn_query).  The order stays good under deformation
// because neighbours at rest stay neighbours.
//
// What the absent traversal order would decide is decided by index (so the result does not depend on the traversal):
// lowest tet index among the tets containing the vertex, lowest face index among equally near rest triangles.  FP64.
#pragma once
#include <hip/hip_runtime.h>

namespace admm_k {

constexpr int kOctMaxLevels = 12;

struct OctLevels { int n_levels; int off[kOctMaxLevels + 1]; int n[kOctMaxLevels]; };

struct DynMesh {
    int vert_offset, n_verts;
    OctLevels tt;               // tets tree
    const int4 *tet;            // [padded] GLOBAL vertex ids in sorted order (x = -1: padding)
    const int *tet_id;          // [padded] original tet index
    double *t_box;              // [6 * nodes] lo xyz, hi xyz -- refitted
    OctLevels ft;               // rest-surface faces tree
    const int *face;            // [3 * padded] LOCAL vertex ids in sorted order (-1: padding)
    const int *face_id;         // [padded] original face index
    const double *f_box;        // [6 * nodes] static
    const double *rest;         // [3 * n_verts]
};

__device__ __forceinline__ void box_reset(double *b) {
    b[0] = b[1] = b[2] = __builtin_inf(); b[3] = b[4] = b[5] = -__builtin_inf();
}
__device__ __forceinline__ void box_grow(double *b, const double *p) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { b[a] = fmin(b[a], p[a]); b[3 + a] = fmax(b[3 + a], p[a]); }
}

// level 0: box of 8 consecutive (sorted) tets at the current positions
__global__ __launch_bounds__(256) void k_dyn_refit0(DynMesh M, const double *__restrict__ x) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M.tt.n[0]) return;
    double b[6]; box_reset(b);
    for (int k = 0; k < 8; ++k) {
        const int4 t = M.tet[8 * (size_t)i + k];
        if (t.x < 0) continue;
        const int id[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) { const double p[3] = {x[3 * (size_t)id[c]], x[3 * (size_t)id[c] + 1], x[3 * (size_t)id[c] + 2]}; box_grow(b, p); }
    }
    double *o = M.t_box + 6 * (size_t)i;
#pragma unroll
    for (int a = 0; a < 6; ++a) o[a] = b[a];
}

// level l >= 1: union of the (up to 8) child boxes
__global__ __launch_bounds__(256) void k_dyn_refit_up(DynMesh M, int l) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M.tt.n[l]) return;
    double b[6]; box_reset(b);
    for (int k = 0; k < 8 && 8 * i + k < M.tt.n[l - 1]; ++k) {
        const double *ch = M.t_box + 6 * (size_t)(M.tt.off[l - 1] + 8 * i + k);
#pragma unroll
        for (int a = 0; a < 3; ++a) { b[a] = fmin(b[a], ch[a]); b[3 + a] = fmax(b[3 + a], ch[3 + a]); }
    }
    double *o = M.t_box + 6 * (size_t)(M.tt.off[l] + i);
#pragma unroll
    for (int a = 0; a < 6; ++a) o[a] = b[a];
}

// all levels from l0 up (each with at most 256 nodes: one per thread) in ONE block: the top of the tree is a handful of tiny levels,
// and one launch per level would cost more than the work (5 us each)
__global__ __launch_bounds__(256) void k_dyn_refit_top(DynMesh M, int l0) {
    for (int l = l0; l < M.tt.n_levels; ++l) {
        for (int i = (int)threadIdx.x; i < M.tt.n[l]; i += 256) {
            double b[6]; box_reset(b);
            for (int k = 0; k < 8 && 8 * i + k < M.tt.n[l - 1]; ++k) {
                const double *ch = M.t_box + 6 * (size_t)(M.tt.off[l - 1] + 8 * i + k);
#pragma unroll
                for (int a = 0; a < 3; ++a) { b[a] = fmin(b[a], ch[a]); b[3 + a] = fmax(b[3 + a], ch[3 + a]); }
            }
            double *o = M.t_box + 6 * (size_t)(M.tt.off[l] + i);
#pragma unroll
            for (int a = 0; a < 6; ++a) o[a] = b[a];
        }
        __threadfence_block();
        __syncthreads();
    }
}

__device__ __forceinline__ bool tet_barycentric(const double *x, const double *p0, const double *p1, const double *p2, const double *p3, double *b) {
    double e1[3], e2[3], e3[3], r[3], cf[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { e1[c] = p1[c] - p0[c]; e2[c] = p2[c] - p0[c]; e3[c] = p3[c] - p0[c]; r[c] = x[c] - p0[c]; }
    cross3(e2, e3, cf);
    const double det = dot3(e1, cf);
    if (det == 0.0 || !(det == det)) return false;
    b[1] = dot3(r, cf) / det;
    cross3(e3, e1, cf); b[2] = dot3(r, cf) / det;
    cross3(e1, e2, cf); b[3] = dot3(r, cf) / det;
    b[0] = 1.0 - b[1] - b[2] - b[3];
    return true;
}

// Ericson, Real-Time Collision Detection 5.1.5: closest point of triangle (a,b,c) to p; returns the squared distance,
// bc = barycentrics of the closest point
__device__ __forceinline__ double closest_on_triangle(const double *p, const double *a, const double *b, const double *c, double *bc) {
    double ab[3], ac[3], ap[3], bp[3], cp[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { ab[i] = b[i] - a[i]; ac[i] = c[i] - a[i]; ap[i] = p[i] - a[i]; bp[i] = p[i] - b[i]; cp[i] = p[i] - c[i]; }
    const double d1 = dot3(ab, ap), d2 = dot3(ac, ap), d3 = dot3(ab, bp), d4 = dot3(ac, bp), d5 = dot3(ab, cp), d6 = dot3(ac, cp);
    const double vc = d1 * d4 - d3 * d2, vb = d5 * d2 - d1 * d6, va = d3 * d6 - d5 * d4;
    double u, v, w;
    if (d1 <= 0.0 && d2 <= 0.0) { u = 1; v = 0; w = 0; }
    else if (d3 >= 0.0 && d4 <= d3) { u = 0; v = 1; w = 0; }
    else if (vc <= 0.0 && d1 >= 0.0 && d3 <= 0.0) { v = d1 / (d1 - d3); u = 1 - v; w = 0; }
    else if (d6 >= 0.0 && d5 <= d6) { u = 0; v = 0; w = 1; }
    else if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) { w = d2 / (d2 - d6); u = 1 - w; v = 0; }
    else if (va <= 0.0 && (d4 - d3) >= 0.0 && (d5 - d6) >= 0.0) { w = (d4 - d3) / ((d4 - d3) + (d5 - d6)); v = 1 - w; u = 0; }
    else { const double den = 1.0 / (va + vb + vc); v = vb * den; w = vc * den; u = 1.0 - v - w; }
    double d = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) { const double q = p[i] - (u * a[i] + v * b[i] + w * c[i]); d += q * q; }
    bc[0] = u; bc[1] = v; bc[2] = w;
    return d;
}

__device__ __forceinline__ bool box_contains(const double *b, const double *p) {
    return p[0] >= b[0] && p[1] >= b[1] && p[2] >= b[2] && p[0] <= b[3] && p[1] <= b[4] && p[2] <= b[5];
}
__device__ __forceinline__ double box_dist2(const double *b, const double *p) {
    double d = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) { const double e = fmax(fmax(b[a] - p[a], p[a] - b[3 + a]), 0.0); d += e * e; }
    return d;
}

// TetMeshCollision::signed_distance for every candidate vertex (query == nullptr: vertices 0..nq-1).  A vertex that
// already holds a dynamic payload (face >= 0, written by an earlier object) is skipped (DynamicObject.hpp:73).
//
// EIGHT LANES PER CANDIDATE (8 candidates per wave, 32 per block): a tree node has 8 children and a leaf 8 primitives,
// so every step of the traversal is one parallel test by the candidate's 8 lanes -- child boxes / tets / triangles are
// loaded as one 384-byte (or 8 x int4) access, the verdicts are combined with a ballot or a 3-step butterfly, and
// the traversal stack (identical for the 8 lanes) lives in LDS.  A lane-per-candidate version of the same traversal
// spent 585 us per mesh at 1 M tets / 23 k candidates: a few hundred dependent loads per lane and only 360 waves on
// the whole chip to hide them behind.
__global__ __launch_bounds__(256) void k_dyn_query(DynMesh M, int nq, const int *__restrict__ query, const double *__restrict__ x,
                                                   int *__restrict__ face_out, double *__restrict__ bary_out,
                                                   double *__restrict__ n_out, double *__restrict__ dx_out) {
    constexpr int kStack = 8 * kOctMaxLevels;
    __shared__ int s_node[32][kStack];
    __shared__ double s_dist[32][kStack];
    const int tid = (int)threadIdx.x, grp = tid >> 3, sub = tid & 7, gsh = (tid & 63) & ~7;   // gsh: the group's first lane in its wave
    const int q = (int)blockIdx.x * 32 + grp;
    const int vg = q < nq ? (query ? query[q] : q) : 0;
    const bool on = q < nq && face_out[3 * (size_t)vg] < 0;       // uniform over the group
    volatile int *st = s_node[grp];
    volatile double *sd = s_dist[grp];
    const double px[3] = {x[3 * (size_t)vg], x[3 * (size_t)vg + 1], x[3 * (size_t)vg + 2]};
    auto group_bits = [&](bool pred) -> unsigned { return (unsigned)((__ballot(pred) >> gsh) & 0xffull); };
    // ---- point in tet (:76-79): lowest original index among the tets that contain the vertex and do not touch it
    int found = 0x7fffffff, found_pos = -1;
    double fb[4] = {0, 0, 0, 0};
    int sp = 0;
    if (on) { if (sub == 0) st[0] = ((M.tt.n_levels - 1) << 27); sp = 1; }
    while (__any(sp > 0)) {
        if (sp > 0) {
            __builtin_amdgcn_wave_barrier();
            const int e = st[--sp], l = e >> 27, i = e & 0x7ffffff;
            __builtin_amdgcn_wave_barrier();
            if (l > 0) {
                const int j = 8 * i + sub;
                const bool in = j < M.tt.n[l - 1] && box_contains(M.t_box + 6 * (size_t)(M.tt.off[l - 1] + j), px);
                const unsigned m = group_bits(in);
                if (in) st[sp + __popc(m & ((1u << sub) - 1u))] = ((l - 1) << 27) | j;
                sp += __popc(m);
            } else {
                const int pos = 8 * i + sub;
                const int4 t = M.tet[pos];
                int cand = 0x7fffffff;
                double b[4] = {0, 0, 0, 0};
                if (t.x >= 0 && t.x != vg && t.y != vg && t.z != vg && t.w != vg) {
                    double p[4][3];
                    const int id[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) { p[c][0] = x[3 * (size_t)id[c]]; p[c][1] = x[3 * (size_t)id[c] + 1]; p[c][2] = x[3 * (size_t)id[c] + 2]; }
                    if (tet_barycentric(px, p[0], p[1], p[2], p[3], b) && b[0] >= 0.0 && b[1] >= 0.0 && b[2] >= 0.0 && b[3] >= 0.0)
                        cand = M.tet_id[pos];
                }
                int mn = cand;
                mn = min(mn, __shfl_xor(mn, 1, 64)); mn = min(mn, __shfl_xor(mn, 2, 64)); mn = min(mn, __shfl_xor(mn, 4, 64));
                const unsigned own = group_bits(cand == mn && mn != 0x7fffffff);     // tet ids are unique: one lane
                if (mn < found) {
                    const int src = gsh + __ffs(own) - 1;
                    found = mn; found_pos = __shfl(pos, src, 64);
#pragma unroll
                    for (int c = 0; c < 4; ++c) fb[c] = __shfl(b[c], src, 64);
                }
            }
        }
    }
    const bool hit_tet = on && found_pos >= 0;
    // ---- the same combination of the rest vertices (:92-97)
    double rx[3] = {0, 0, 0};
    if (hit_tet) {
        const int4 t = M.tet[found_pos];
        const int id[4] = {t.x - M.vert_offset, t.y - M.vert_offset, t.z - M.vert_offset, t.w - M.vert_offset};
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int a = 0; a < 3; ++a) rx[a] += fb[c] * M.rest[3 * (size_t)id[c] + a];
    }
    // ---- nearest rest-surface triangle that does not touch the vertex (:99-104): depth-first, NEAREST child first
    // (children are pushed far-to-near with their box distance and re-tested against the current best when popped),
    // so the first leaf reached is close to the query and almost everything else is pruned
    const int vl = vg - M.vert_offset;
    double best = __builtin_inf(), bbc[3] = {0, 0, 0};
    int best_id = 0x7fffffff, best_pos = -1;
    sp = 0;
    if (hit_tet) { if (sub == 0) { st[0] = ((M.ft.n_levels - 1) << 27); sd[0] = 0.0; } sp = 1; }
    while (__any(sp > 0)) {
        if (sp > 0) {
            __builtin_amdgcn_wave_barrier();
            --sp;
            const int e = st[sp];
            const double de = sd[sp];
            __builtin_amdgcn_wave_barrier();
            const int l = e >> 27, i = e & 0x7ffffff;
            if (!(de > best)) {
                if (l > 0) {
                    const int j = 8 * i + sub;
                    const bool valid = j < M.ft.n[l - 1];
                    const double d = valid ? box_dist2(M.f_box + 6 * (size_t)(M.ft.off[l - 1] + j), rx) : __builtin_inf();
                    const bool keep = valid && !(d > best);
                    int rank = 0;     // position among the kept children, farthest first
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const double dk = __shfl(d, gsh + k, 64);
                        const int kk = __shfl((int)keep, gsh + k, 64);
                        rank += (kk && (dk > d || (dk == d && k < sub))) ? 1 : 0;
                    }
                    if (keep) { st[sp + rank] = ((l - 1) << 27) | j; sd[sp + rank] = d; }
                    sp += __popc(group_bits(keep));
                } else {
                    int pos = 8 * i + sub;
                    const int f0 = M.face[3 * (size_t)pos], f1 = M.face[3 * (size_t)pos + 1], f2 = M.face[3 * (size_t)pos + 2];
                    double d = __builtin_inf(), bc[3] = {0, 0, 0};
                    int fid = 0x7fffffff;
                    if (f0 >= 0 && f0 != vl && f1 != vl && f2 != vl) {
                        d = closest_on_triangle(rx, M.rest + 3 * (size_t)f0, M.rest + 3 * (size_t)f1, M.rest + 3 * (size_t)f2, bc);
                        fid = M.face_id[pos];
                    }
#pragma unroll
                    for (int o = 1; o < 8; o <<= 1) {     // butterfly arg-min on (distance, face index)
                        const double od = __shfl_xor(d, o, 64);
                        const int oid = __shfl_xor(fid, o, 64), opos = __shfl_xor(pos, o, 64);
                        const double o0 = __shfl_xor(bc[0], o, 64), o1 = __shfl_xor(bc[1], o, 64), o2 = __shfl_xor(bc[2], o, 64);
                        if (od < d || (od == d && oid < fid)) { d = od; fid = oid; pos = opos; bc[0] = o0; bc[1] = o1; bc[2] = o2; }
                    }
                    if (fid != 0x7fffffff && (d < best || (d == best && fid < best_id))) {
                        best = d; best_id = fid; best_pos = pos; bbc[0] = bc[0]; bbc[1] = bc[1]; bbc[2] = bc[2];
                    }
                }
            }
        }
    }
    if (!hit_tet || best_pos < 0 || sub != 0) return;
    const double dx = -sqrt(best);                 // :112
    if (!(dx < 0.0)) return;                       // Collider.hpp:203
    const int f[3] = {M.face[3 * (size_t)best_pos], M.face[3 * (size_t)best_pos + 1], M.face[3 * (size_t)best_pos + 2]};
    double e1[3], e2[3], nr[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        e1[a] = M.rest[3 * (size_t)f[1] + a] - M.rest[3 * (size_t)f[0] + a];
        e2[a] = M.rest[3 * (size_t)f[2] + a] - M.rest[3 * (size_t)f[0] + a];
    }
    cross3(e1, e2, nr);
    const double il = 1.0 / sqrt(dot3(nr, nr));
    dx_out[vg] = dx;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        face_out[3 * (size_t)vg + a] = f[a] + M.vert_offset;    // :113
        bary_out[3 * (size_t)vg + a] = bbc[a];                  // :114
        n_out[3 * (size_t)vg + a] = nr[a] * il;                 // :110-111, :115
    }
}

// ConstraintSet::make_matrix, dynamic rows (ConstraintSet.hpp:92-110) in the per-vertex row storage of the Uzawa
// kernels: a candidate with a dynamic payload counts as a row; if the vertex already holds a passive row the
// dynamic row stays empty (`constrained`, :96-99), otherwise row = ck n^T (x_v - sum_j bary_j x_face_j), rhs 0.
__global__ __launch_bounds__(256) void k_dyn_rows(int nq, const int *__restrict__ query, double ck, double *__restrict__ cn,
                                                  double *__restrict__ cc, int *__restrict__ dface, const double *__restrict__ dn,
                                                  int *__restrict__ nhits) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    const int v = query ? query[q] : q;
    if (dface[3 * (size_t)v] < 0) return;
    atomicAdd(nhits, 1);
    if (cn[3 * (size_t)v] != 0.0 || cn[3 * (size_t)v + 1] != 0.0 || cn[3 * (size_t)v + 2] != 0.0) { dface[3 * (size_t)v] = -1; return; }
#pragma unroll
    for (int a = 0; a < 3; ++a) cn[3 * (size_t)v + a] = ck * dn[3 * (size_t)v + a];
    cc[v] = 0.0;
}

// The face-vertex part of C^T y: out += s * ck n bary_j y_v at face vertex j, s = +1 for (base - C^T y), -1 for C^T y.
// Several rows may share a face vertex.  No FP64 atomics (their order would make contact runs differ in the last bits from run
// to run): the contributions are summed as 64-bit INTEGERS in a fixed-point format chosen from their largest magnitude --
// integer addition is associative, so the sum does not depend on the order, and with 2^-50 of the largest contribution as the
// unit it is as accurate as the FP64 sum.  Three small launches: (1) the largest magnitude (atomicMax on the bit pattern of a
// non-negative double is order-independent too), (2) the scatter with integer atomic adds, (3) out += sum / scale, clearing the
// accumulator behind itself.  dmax[0] must be 0 on entry of (1); (3) leaves it 0 again.
__device__ __forceinline__ double uz_dyn_scale(double m) {      // 2^(50 - e) for m = f 2^e, f in [0.5, 1)
    const int e = (int)((__double2hiint(m) >> 20) & 0x7ff) - 1022;
    return __hiloint2double((1023 + 50 - e) << 20, 0);
}
__global__ __launch_bounds__(256) void k_uz_ct_dyn_max(int nq, const int *__restrict__ query, const double *__restrict__ cn,
                                                       const double *__restrict__ y, const int *__restrict__ dface,
                                                       const double *__restrict__ dbary, double *__restrict__ dmax) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    const int v = query ? query[q] : q;
    if (dface[3 * (size_t)v] < 0) return;
    const double c = fmax(fabs(cn[3 * (size_t)v]), fmax(fabs(cn[3 * (size_t)v + 1]), fabs(cn[3 * (size_t)v + 2])));
    const double b = fmax(fabs(dbary[3 * (size_t)v]), fmax(fabs(dbary[3 * (size_t)v + 1]), fabs(dbary[3 * (size_t)v + 2])));
    const double m = fabs(y[v]) * b * c;
    // (below 1e-280 the contributions are zero for every purpose, and 2^(50 - e) would leave the exponent range of a double)
    if (m > 1e-280 && m < 1e300) atomicMax((unsigned long long *)dmax, (unsigned long long)__double_as_longlong(m));
}
__global__ __launch_bounds__(256) void k_uz_ct_dyn(int nq, const int *__restrict__ query, int mode, const double *__restrict__ cn,
                                                   const double *__restrict__ y, const int *__restrict__ dface,
                                                   const double *__restrict__ dbary, const double *__restrict__ dmax,
                                                   long long *__restrict__ acc) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    const int v = query ? query[q] : q;
    if (dface[3 * (size_t)v] < 0) return;
    const double m = dmax[0];
    if (!(m > 0.0)) return;
    const double scale = uz_dyn_scale(m);
    const double s = (mode == 0 ? 1.0 : -1.0) * y[v];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int f = dface[3 * (size_t)v + j];
        const double w = s * dbary[3 * (size_t)v + j];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const long long k = __double2ll_rn(w * cn[3 * (size_t)v + a] * scale);
            if (k != 0) atomicAdd((unsigned long long *)(acc + 3 * (size_t)f + a), (unsigned long long)k);
        }
    }
}
__global__ __launch_bounds__(256) void k_uz_ct_dyn_apply(int n3, double *__restrict__ dmax, long long *__restrict__ acc, double *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const double m = dmax[0];
    if (i < n3 && m > 0.0) {
        const long long k = acc[i];
        if (k != 0) { out[i] += (double)k / uz_dyn_scale(m); acc[i] = 0; }
    }
    // (every thread has read dmax above; the last block to pass clears it for the next use)
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd((unsigned int *)(dmax + 1), 1u) == gridDim.x - 1) { dmax[0] = 0.0; ((unsigned int *)(dmax + 1))[0] = 0u; }
    }
}

// the face-vertex part of one row of C applied to a node vector: - sum_j bary_j cn . vec_face_j (0 for other rows)
__device__ __forceinline__ double dyn_row_faces(int v, const double *__restrict__ cn, const int *__restrict__ dface,
                                                const double *__restrict__ dbary, const double *__restrict__ vec) {
    if (dface == nullptr || dface[3 * (size_t)v] < 0) return 0.0;
    double r = 0.0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int f = dface[3 * (size_t)v + j];
        r -= dbary[3 * (size_t)v + j] * (cn[3 * (size_t)v] * vec[3 * (size_t)f] + cn[3 * (size_t)v + 1] * vec[3 * (size_t)f + 1] +
                                         cn[3 * (size_t)v + 2] * vec[3 * (size_t)f + 2]);
    }
    return r;
}

// ---------------------------------------------------------------------------------------------------------------------
// Dynamic hits in the multi-colour Gauss-Seidel (src/NodalMultiColorGS.hpp:80-86): the reference adds the penalty
// C^T C to A and re-colours.  C^T C couples only the four nodes of a hit (vertex + face vertices) and breaks the
// Ahat (x) I3 structure only in THEIR rows, so the matrix is never formed: every other node keeps its row of A, its
// colour and the SELL kernels (k_gs_color with the `skip` mask); the touched nodes are re-coloured on the host over
// new colours that follow the old ones (first fit in node order; conflicts = shared hit or non-zero of Ahat -- the
// colouring library is absent, the rule is shared with the oracle) and swept here, one lane per node, from a compact
// copy of their Ahat rows plus their hits:
//   (C^T C x)_(a,s) = ck^2 c_a n_s sum_k c_k n . x_k     (c = 1 at the vertex, -bary_j at face vertex j)
struct GsDyn {
    int n_touched, n_hits;
    const int *node;                 // [n_touched] node ids, grouped by new colour
    const int *rptr;                 // [n_touched + 1] the node's row of Ahat: rcol / rval (diagonal included)
    const int *rcol; const double *rval;
    const int *hptr;                 // [n_touched + 1] the node's hits: hidx (hit), hcoef (c_a)
    const int *hidx; const double *hcoef;
    const int4 *hnode;               // [n_hits] vertex, face vertices
    const double *hc;                // [n_hits][4]
    const double *hn;                // [n_hits][3]
    double ck2;
};

// off-diagonal row sums LUx and diagonals aii of the three rows of touched node t in M = A + C^T C, exactly as
// segment_update reads them (:180-215): exact zeros of Ahat skipped, same-node cross-axis terms of C^T C included
__device__ __forceinline__ void gsd_rows(const GsDyn &d, int t, int a, const double *__restrict__ m, const double *__restrict__ x,
                                         double *LUx, double *aii) {
    double ad = 0.0;
    LUx[0] = LUx[1] = LUx[2] = 0.0;
    for (int k = d.rptr[t]; k < d.rptr[t + 1]; ++k) {
        const double v = d.rval[k];
        const int j = d.rcol[k];
        if (j == a) { ad = v; continue; }
        if (v == 0.0) continue;
#pragma unroll
        for (int s = 0; s < 3; ++s) LUx[s] = fma(v, x[3 * (size_t)j + s], LUx[s]);
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) aii[s] = ad + m[3 * (size_t)a + s];
    const double xa[3] = {x[3 * (size_t)a], x[3 * (size_t)a + 1], x[3 * (size_t)a + 2]};
    for (int k = d.hptr[t]; k < d.hptr[t + 1]; ++k) {
        const int h = d.hidx[k];
        const double ca = d.hcoef[k];
        const int4 nd = d.hnode[h];
        const int id[4] = {nd.x, nd.y, nd.z, nd.w};
        const double n[3] = {d.hn[3 * (size_t)h], d.hn[3 * (size_t)h + 1], d.hn[3 * (size_t)h + 2]};
        double sh = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            sh += d.hc[4 * (size_t)h + q] * (n[0] * x[3 * (size_t)id[q]] + n[1] * x[3 * (size_t)id[q] + 1] + n[2] * x[3 * (size_t)id[q] + 2]);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const double w = d.ck2 * ca * n[s];
            aii[s] += w * ca * n[s];
            LUx[s] += w * (sh - ca * n[s] * xa[s]);
        }
    }
}

// one NEW colour of one sweep: touched nodes [t0, t1)
__global__ __launch_bounds__(64) void k_gs_touched(GsArgs a, GsDyn d, int t0, int t1, Obstacles ob) {
    const int t = t0 + (int)(blockIdx.x * 64 + threadIdx.x);
    if (t >= t1 || *a.done) return;
    const int v = d.node[t];
    if (a.pin_flag && a.pin_flag[v]) {
#pragma unroll
        for (int q = 0; q < 3; ++q) a.x[3 * (size_t)v + q] = a.pin_xyz[3 * (size_t)v + q];
        return;
    }
    double LUx[3], aii[3], jac[3], nx[3];
    gsd_rows(d, t, v, a.m, a.x, LUx, aii);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        jac[q] = (a.b[3 * (size_t)v + q] - LUx[q]) * (1.0 / aii[q]);      // (the same two roundings as gs_relax: kernels.hpp)
        nx[q] = (1.0 - a.omega) * a.x[3 * (size_t)v + q] + a.omega * jac[q];
    }
    double n[3], p[3];
    if (ob.n > 0 && passive_hit(ob, nx, n, p)) {
        double dx[3] = {jac[0] - p[0], jac[1] - p[1], jac[2] - p[2]};
        double nn[3] = {0.0, 0.0, 0.0}, uu[3], vv[3];
        if (n[0] > 0.999) nn[2] = 1.0; else nn[0] = 1.0;
        cross3(nn, n, uu);
        double il = 1.0 / sqrt(dot3(uu, uu));
#pragma unroll
        for (int q = 0; q < 3; ++q) uu[q] *= il;
        cross3(n, uu, vv);
        il = 1.0 / sqrt(dot3(vv, vv));
#pragma unroll
        for (int q = 0; q < 3; ++q) vv[q] *= il;
        const double t0d = dot3(uu, dx), t1d = dot3(vv, dx);
#pragma unroll
        for (int q = 0; q < 3; ++q) nx[q] = uu[q] * t0d + vv[q] * t1d + p[q];
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) a.x[3 * (size_t)v + q] = nx[q];
}

// |b - M x|^2 over the touched rows (one block) -> the extra slot `slot` of the residual partials (:136-140)
__global__ __launch_bounds__(256) void k_gs_touched_resid(GsArgs a, GsDyn d, double *__restrict__ part, int NBp, int slot) {
    __shared__ double lds[8];
    const int done_flag = *a.done;
    double q[2] = {0.0, 0.0};
    for (int t = threadIdx.x; t < d.n_touched; t += 256) {
        const int v = d.node[t];
        double LUx[3], aii[3];
        gsd_rows(d, t, v, a.m, a.x, LUx, aii);
#pragma unroll
        for (int s = 0; s < 3; ++s) { const double r = a.b[3 * (size_t)v + s] - LUx[s] - aii[s] * a.x[3 * (size_t)v + s]; q[0] = fma(r, r, q[0]); }
    }
    block_sum<2>(q, lds);
    if (threadIdx.x == 0 && !done_flag) { part[slot] = q[0]; part[NBp + slot] = 0.0; }
}

// compaction of the dynamic payloads into a hit list (order arbitrary; the host sorts by vertex)
struct DynHit { int v, f0, f1, f2; double b[3]; double n[3]; };
__global__ __launch_bounds__(256) void k_dyn_compact(int nq, const int *__restrict__ query, const int *__restrict__ dface,
                                                     const double *__restrict__ dbary, const double *__restrict__ dn,
                                                     DynHit *__restrict__ out, int cap, int *__restrict__ count) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    const int v = query ? query[q] : q;
    if (dface[3 * (size_t)v] < 0) return;
    const int i = atomicAdd(count, 1);
    if (i >= cap) return;
    DynHit h;
    h.v = v; h.f0 = dface[3 * (size_t)v]; h.f1 = dface[3 * (size_t)v + 1]; h.f2 = dface[3 * (size_t)v + 2];
#pragma unroll
    for (int s = 0; s < 3; ++s) { h.b[s] = dbary[3 * (size_t)v + s]; h.n[s] = dn[3 * (size_t)v + s]; }
    out[i] = h;
}

} // namespace admm_k
