// host_setup.hpp -- host-side set-up structures shared by the C ABI (no GPU code here).
#pragma once
#include <cstdint>
#include <vector>
#include <string>

namespace admm_host {

struct Csr {
    int32_t n = 0;
    std::vector<int32_t> rowptr, col;
    std::vector<double> val;
};

// Sliced ELL with slice height 64 (= one gfx950 wavefront): slice s holds rows [64 s, 64 s + 64),
// all padded to the slice's widest row; element (row 64 s + l, k) lives at slice_ptr[s] + 64 k + l,
// so a wave reads 64 consecutive entries per k (coalesced).
struct Sell {
    int32_t n_rows = 0, n_slices = 0;
    std::vector<int32_t> slice_ptr;   // [n_slices + 1], element offsets
    std::vector<int32_t> slice_width; // [n_slices]
    std::vector<int32_t> idx;         // column index / incidence code
    std::vector<double> val;          // empty for pure index lists
};

// One scalar reduction row of D-hat: coefficient `c` on vertex `v` (TetEnergyTerm.cpp:52-70 etc.)
struct TermRows {
    // For every energy term: its weight^2 and its rows as (vertex, coefficient) lists.
    // Assembled directly into Ahat = dt^2 sum_terms w^2 sum_rows d d^T  (Solver.cpp:225-226).
};

// Ahat (n_verts x n_verts, scalar; the mass diagonal is NOT included) from the flattened terms.
Csr assemble_Ahat(int32_t n_verts, double dt,
                  int32_t n_tets, const int32_t *tet_idx, const double *tet_Binv, const double *tet_w,
                  int32_t n_tris, const int32_t *tri_idx, const double *tri_rest, const double *tri_w,
                  int32_t n_pins, const int32_t *pin_vert, double pin_w);

// A + dt^2 sum_h w_h^2 c_h c_h^T for 4-vertex stencil terms (bending hinges: D-block = (c0..c3) (x) I3), merged into the sorted CSR
Csr add_stencil_terms(const Csr &A, double dt, int32_t n, const int32_t *idx4, const double *coef4, const double *w);
// interior edges of a triangle mesh and their cotangent stencils (include/admm_hip.h: admm_host_bend_hinges)
int32_t bend_hinges(int32_t n_verts, int32_t n_tris, const int32_t *tris, const double *verts, int32_t cap, int32_t *hinge_idx, double *coef, double *area);

Sell csr_to_sell(const Csr &A);

// vertex -> list of (element, corner) codes, code = elem * stride + corner; padding = pad_code
// row_vertex (optional, a permutation of the vertices): row r of the SELL holds the list of vertex row_vertex[r]
Sell incidence_sell(int32_t n_verts, int32_t n_elems, int32_t corners, const int32_t *idx, int32_t pad_code, const int32_t *row_vertex = nullptr);
// Block-level reduction of the tet corner forces (kernels.hpp: tet_compute_store).  A CHUNK = the 256 consecutive tets one
// block of the local step works on (chunks never straddle two constitutive models).  The block parks its 1024 corner forces in
// LDS and sums them per vertex into RECORDS of at most kChunkFan corner forces each (a vertex with more incident corners in the
// chunk gets several records), one record per thread and pass: the right-hand-side gather then reads ~3-4 records per vertex
// instead of ~24 corner forces, and the local step writes ~25 instead of 96 bytes per tet.
constexpr int kChunkFan = 8;            // corner forces per record = 16-byte entry list per thread
constexpr int kChunkLd = 257;           // doubles between two rows of the LDS block (column 256 = the all-zero padding column)
constexpr uint16_t kChunkPad = 256 * 8; // LDS byte offset of the padding entry
struct TetChunks {
    int32_t n_chunks = 0, n_rec = 0;
    std::vector<uint16_t> ent;          // [n_groups][256][kChunkFan]: LDS byte offsets (x component) of the corner forces of record (group, thread)
    std::vector<int32_t> group_base;    // [n_chunks + 1]: first 256-record group of a chunk (one group per chunk for tets in a vertex-coherent order)
    std::vector<int32_t> rec_base;      // [n_chunks + 1]: first record of a chunk
    std::vector<int32_t> rec_vertex;    // [n_rec]
};
TetChunks tet_chunks(int32_t n_tets, const int32_t *tet_idx, const int32_t kind_begin[6]);
// vertex -> records incidence lists (row r gathers for vertex row_vertex[r]); widths are multiples of 4, padding = pad_code
Sell record_incidence(int32_t n_verts, int32_t n_rec, const int32_t *rec_vertex, int32_t pad_code, const int32_t *row_vertex = nullptr, int32_t n_rows = -1);
std::vector<int32_t> incidence_row_order(int32_t n_verts, int32_t n_tets, const int32_t *tet_idx, int32_t n_tris, const int32_t *tri_idx, int32_t window);

int greedy_coloring(int32_t n, const int32_t *rowptr, const int32_t *col, int32_t *color);

// Colour-ordered SELL for the multi-colour Gauss-Seidel sweeps: every colour's node list is padded to whole
// 64-lane slices (slot_node = -1 for padding lanes); the SELL holds the OFF-diagonal non-zeros of each node's
// row in column order, the diagonal goes to `diag`.  Colour c owns slices [color_slice[c], color_slice[c+1]).
struct GsSell {
    Sell sell;
    std::vector<int32_t> slot_node;   // [64 * n_slices]
    std::vector<double> diag;         // [64 * n_slices]
    std::vector<int32_t> color_slice; // [n_colors + 1]
};
GsSell build_gs_sell(const Csr &A, int n_colors, const std::vector<int32_t> &color);

// Implicit 8-ary bounding-volume tree over primitives sorted along a Morton curve (dynamic self-collision,
// src/DynamicObject.hpp: the reference builds mclscene AABB trees; here the ORDER is fixed once from the rest
// centroids and only the boxes are refitted).  Level 0 node i covers sorted primitives [8 i, 8 i + 8), level l
// node i covers level l-1 nodes [8 i, 8 i + 8); the last level has one node.  Boxes of all levels live in one
// array, level l starting at node offset level_off[l].
struct OctTree {
    int32_t n_prims = 0, n_padded = 0, n_levels = 0;
    std::vector<int32_t> order;       // [n_padded] sorted position -> original primitive index (-1 = padding)
    std::vector<int32_t> level_off;   // [n_levels + 1] node offsets, level_off[n_levels] = total node count
    std::vector<int32_t> level_n;     // [n_levels] nodes per level
};
// centroids [3 * n]; ties in the Morton code keep the original order
OctTree build_octtree(int32_t n, const double *centroids);
// boxes (lo xyz, hi xyz per node) of every level for triangles `faces` (local ids, sorted order given by tree.order)
std::vector<double> octtree_boxes_tris(const OctTree &tree, const int32_t *faces, const double *verts);

// Locality ordering of mesh vertices: reverse Cuthill-McKee on the vertex graph of the elements (corners = 3 or 4).
// new_id[v] = position of vertex v in the ordering.  span_before / span_after = mean |i - j| over element edges.
void locality_order(int32_t n_verts, int32_t n_elems, int32_t corners, const int32_t *idx, int32_t *new_id,
                    double *span_before, double *span_after);

// Plan of the on-chip PCG for general meshes (oc_plan.cpp): internal row order (compact blocks of `spb` wavefronts, each
// split into kOcSub compact aggregates), the system matrix in that order (off-diagonal non-zeros, SELL-64, 16-bit columns
// into the block's local vector = own rows + halo list), the part of every slice that fits the LDS slab, the
// neighbour-block lists and the dense inverse of the aggregate coarse matrix.
constexpr int kOcSub = 4;
struct OcPlan {
    bool ok = false;
    int G = 0, spb = 0, sub = kOcSub;
    int32_t n_rows = 0;                 // G * spb * 64 internal rows (dummy rows: orig = -1)
    std::vector<int32_t> orig;          // [n_rows] internal row -> vertex
    std::vector<int32_t> pos;           // [n_verts] vertex -> internal row
    std::vector<signed char> row_agg;   // [n_rows] aggregate (0..kOcSub-1) of the row inside its block
    Sell A;                             // widths / offsets / values (idx unused: see col16)
    std::vector<uint16_t> col16;        // local column of entry (slice s, column k, lane l) at slice_ptr[s] + ((k / 4) 64 + l) 4 + k % 4
    std::vector<double> mdiag;          // [3 n_rows] mass + diagonal of Ahat (0 for dummy rows)
    std::vector<int32_t> halo_ptr, halo_src;   // [G + 1], internal rows of the halo entries of every block (sorted)
    int32_t nh_max = 0, nh_cap = 0, vec_len = 0;   // largest halo, rounded up to 64, entries per axis of the local vector
    std::vector<int32_t> wl_s, lds_off; // [n_slices] columns of the slice kept in LDS, their offset (columns) in the block's slab
    int bcols = 0;                      // slab columns of the fullest block
    std::vector<int32_t> nbr;           // [G][64] blocks a block's rows reference (-1 = none)
    bool nbr_ok = false; int nbr_max = 0;
    int nc = 0, ncp = 0;                // coarse unknowns (G * kOcSub), padded row length of ainv
    bool coarse_ok = false;
    std::vector<double> ainv;           // [nc][ncp] (P^T A P)^-1
    std::vector<float> cwt;             // [4 n_rows] row r of P: weights of the row in the kOcSub coarse functions of its block
    bool affine = false;                // coarse functions {1, x, y, z} per block (coordinates given) instead of kOcSub constants
    double stat_bank_sorted = 0.0, stat_bank_placed = 0.0;   // lanes on the busiest LDS bank pair per (half wavefront, column): entries by index / as placed
    double lam_bb = 0.0;                // estimate of lambda_max(D^-1 A_bb), A_bb = the block-diagonal part of M + Ahat (power iteration)
    double lam_bb_gersh = 0.0;          // rigorous upper bound of the same (Gershgorin on D^-1/2 A_bb D^-1/2)
    int64_t stat_nnz = 0, stat_stored = 0, stat_onchip = 0, stat_local = 0;
};
// A = Ahat (mass not included), mass3 [3 n]; lds_bytes = LDS one block may spend on its local vector and matrix slab
// xyz (optional, [3 n]): smooth vertex coordinates for the affine coarse space
OcPlan build_oc_plan(const Csr &A, const double *mass3, int G, int spb, int lds_bytes, bool want_coarse, const double *xyz = nullptr);

// Plan of the LAUNCH-PATH two-level PCG (oc_plan.cpp: build_big_plan; kernels: pcg_big.hpp) -- the global solve of bodies that do not fit
// the chip's LDS (more than 262 144 vertices) and the fall-back of the on-chip kernel: the same preconditioner M^-1 = D^-1 + P (P^T A P)^-1 P^T
// with the affine coarse space, in kernels that stream the matrix.  Internal row order: compact aggregates of `ra` rows (a multiple of 256,
// dummy rows orig = -1), each carrying {1, x, y, z} energy-orthonormalised; A in that order as SELL-64 with global (internal) columns.
struct BigPlan {
    bool ok = false;
    int G = 0, ra = 0;                  // aggregates, rows per aggregate
    int32_t n_rows = 0;                 // G * ra
    std::vector<int32_t> orig, pos;     // internal row -> vertex (-1 = dummy), vertex -> internal row
    Sell A;                             // Ahat (with its diagonal) in internal order
    std::vector<double> mass, dinv;     // [3 n_rows]
    std::vector<double> cwt;            // [4 n_rows] row r of P
    int nc = 0, ncp = 0;
    std::vector<float> ainv;            // [nc][ncp] (P^T A P)^-1, single precision (applied to FP64 vectors, accumulated in FP64)
};
BigPlan build_big_plan(const Csr &A, const double *mass3, const double *xyz, int max_aggregates);

// Plan of the PERSISTENT multi-colour Gauss-Seidel kernel (oc_plan.cpp: build_gs_plan; kernels: gs_persist.hpp).  The vertices are
// split into G compact blocks (one per CU, the recursive bisection of the on-chip PCG); a block keeps its rows' matrix entries,
// its part of x and the values of the rows of other blocks it references (its halo) in LDS for the whole solve, and a colour phase
// only exchanges the boundary rows of that colour with the neighbouring blocks (tagged granules in an outbox per block).
//   own rows of a block: by colour, inside a colour boundary rows (referenced by another block) first
//   local column index : own row -> its position, halo entry h -> n_own + h (halo entries by colour as well)
//   matrix entries     : per (block, colour) an ELL of that colour's rows, width W = longest row, entry (k, row i) at
//                        ent_base + eoff[c] + k * n_c + i; the entries of a row keep their CSR order (ascending original column), so
//                        the row sums are bit-identical to the colour kernels' (padding: value 0 on the row's own index); bit 15 of a
//                        column: the neighbour's colour is below the row's (it has already moved when the row's residual of the
//                        previous sweep is taken: the kernel then reads its parked value)
constexpr int kGspMaxC = 12;       // colours a plan supports
constexpr int kGspHdr = 64;        // ints per block header:
//   0 n_own, 1 n_halo, 2 row_base (into orig / out_idx / diag), 3 halo_base (into halo_box / halo_orig), 4 ent_base (into vals / cols),
//   5 ob_base (first outbox node of the block), 6 n_out, 7 ent_count, 8.. cs[C + 1] (own rows of colour c: [cs[c], cs[c + 1])),
//   21.. hs[C + 1] (halo entries of colour c), 34.. W[C], 46.. eoff[C]
struct GsPlan {
    bool ok = false;
    int G = 0, C = 0;
    std::vector<int32_t> hdr;         // [G][kGspHdr]
    std::vector<int32_t> orig;        // [rows]  vertex of every own row
    std::vector<int32_t> out_idx;     // [rows]  outbox node of the row inside its block, -1 = no other block reads it
    std::vector<double> diag;         // [rows]  Ahat(v, v)
    std::vector<int32_t> halo_box;    // [halo]  global outbox node the entry is read from
    std::vector<int32_t> halo_orig;   // [halo]  its vertex (initial value)
    std::vector<double> vals; std::vector<uint16_t> cols;
    int32_t ob_total = 0;             // outbox nodes of all blocks
    int32_t lds_bytes = 0;            // LDS the fullest block needs
    int32_t max_nbr = 0, max_halo = 0, max_rows = 0;
};
GsPlan build_gs_plan(const Csr &A, int n_colors, const int32_t *color, int max_blocks, int rows_target, int lds_limit);
// bytes of LDS a block with these counts needs (shared with the kernel: gs_persist.hpp computes the same offsets)
inline int32_t gsp_lds_bytes(int32_t n_own, int32_t n_halo, int32_t ent_count) {
    const int32_t L = n_own + n_halo;
    int32_t o = 1024;                         // scratch: reductions, control words
    o += 48 * L;                              // x [L][3], x of the previous sweep [L][3]
    o += 48 * n_own;                          // b, a_ii [n_own][3]
    o += 8 * ent_count;                       // values
    o += (2 * ent_count + 7) / 8 * 8;         // 16-bit local columns
    o += (4 * n_own + 7) / 8 * 8;             // outbox node of every own row
    o += (4 * n_halo + 7) / 8 * 8;            // source of every halo entry
    o += (n_own + 7) / 8 * 8;                 // pin flags
    o += 24 * n_own;                          // 1 / a_ii [n_own][3]
    return o;
}

// Hierarchical block order of mesh vertices (oc_plan.cpp): compact leaves of ~leaf vertices from a recursive graph
// bisection, leaves in recursion-tree order, breadth-first inside a leaf.  new_id[v] = position of vertex v.
void block_order(int32_t n_verts, int32_t n_elems, int32_t corners, const int32_t *idx, int32_t leaf, int32_t *new_id);

// Connected components of the scene (vertices joined by tets / triangles) and their assignment to `world` ranks: components by
// decreasing element count (ties: lowest vertex first), each to the least loaded rank so far (ties: lowest rank).  Returns the
// number of components; vertex_rank[v] = owning rank.  The multi-GPU partition that needs NO exchange inside a step.
int32_t component_partition(int32_t n_verts, int32_t n_tets, const int32_t *tet_idx, int32_t n_tris, const int32_t *tri_idx, int world,
                            int32_t *vertex_rank, int32_t n_bends = 0, const int32_t *bend_idx = nullptr);

// tabulated user splines (device_math.hpp: spline_table_eval; layout: 3 functions x {t0, dt, 1/dt, n, n x (F, dF/dt, d2F/dt2)})
constexpr int kSplineNodesH = 1024, kSplineFnDoublesH = 4 + 3 * kSplineNodesH, kSplineTableDoublesH = 3 * kSplineFnDoublesH;
typedef double (*spline_fn)(void *user, int which, double x);
int tabulate_spline(spline_fn fn, void *user, double s_min, double s_max, double *table_out);
void spline_table_eval(const double *table, int which, double x, double *out3);

int tet_rest(int32_t n, const int32_t *idx, const double *verts, double *Binv, double *vol);
// rest positions [3 nv] behind the tets' Binv: 1 = `cand` reproduces every Binv, 2 = propagated from tet to tet, 0 = none (host_setup.cpp)
int tet_rest_positions(int32_t nv, int32_t nt, const int32_t *idx, const double *Binv, const double *cand, double *x0);
int tri_rest(int32_t n, const int32_t *idx, const double *verts, double *rest, double *area);
void lame(double youngs, double poisson, double *mu, double *lambda, double *bulk);
void partition(int32_t n_items, int world_size, int rank, int32_t *begin, int32_t *end);

} // namespace admm_host
