// Meshes.hpp -- the scene/mesh layer the reference's samples take from mclscene (an absent, un-vendored
// submodule): tet / triangle mesh containers, the synthetic generators the samples use, lumped masses, a TetGen
// .node/.ele reader and OBJ / .node writers for headless runs.  SURVEY 8(f) item 1.  Own types in namespace
// admm (NOT stand-ins for the mcl:: headers: nothing of the reference is compiled against them).
//   reference call sites: samples/utils/AddMeshes.hpp:97-235 (weighted_masses, surface_inds, bounds, apply_xform),
//   samples/sca2016/beams.cpp:43-90 (make_tet_blocks + centre/scale/translate), trianglestrain.cpp (plane mesh).
#ifndef ADMM_MESHES_HPP
#define ADMM_MESHES_HPP 1

#include <algorithm>
#include <cstdint>
#include <array>
#include <cstdio>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>
#include "MiniLinAlg.hpp"

// include/admm_hip.h (libadmm_hip.so): reverse Cuthill-McKee vertex ordering, host only
extern "C" void admm_host_locality_order(int32_t n_verts, int32_t n_elems, int32_t corners, const int32_t *idx, int32_t *new_id,
                                         double *span_before, double *span_after);

namespace admm {

struct TetMesh {
    std::vector<Vec3> vertices;
    std::vector<Vec4i> tets;
    std::vector<Vec3i> faces;   // surface triangles, filled by need_faces() (mcl::TetMesh::faces)
    int flags = 0;   // binding::MeshFlags
    void need_faces() { if (faces.empty()) surface_faces(faces); }

    void bounds(Vec3 &lo, Vec3 &hi) const {
        if (vertices.empty()) throw std::runtime_error("TetMesh::bounds: empty mesh");
        lo = hi = vertices[0];
        for (const Vec3 &p : vertices)
            for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); }
    }
    // p -> s * (p + t)   (the samples' "scale * center" transform, beams.cpp:60-66)
    void translate(const Vec3 &t) { for (Vec3 &p : vertices) p += t; }
    void scale(double sx, double sy, double sz) { for (Vec3 &p : vertices) { p[0] *= sx; p[1] *= sy; p[2] *= sz; } }
    double signed_volume(int t) const {
        const Vec3 &a = vertices[tets[t][0]];
        return (vertices[tets[t][1]] - a).cross(vertices[tets[t][2]] - a).dot(vertices[tets[t][3]] - a) / 6.0;
    }
    // lumped masses: density * volume / 4 to every corner (AddMeshes.hpp:113-122 uses 1522 kg/m^3)
    void weighted_masses(std::vector<double> &m, double density) const {
        m.assign(vertices.size(), 0.0);
        for (size_t t = 0; t < tets.size(); ++t) {
            const double w = density * signed_volume((int)t) / 4.0;
            for (int c = 0; c < 4; ++c) m[tets[t][c]] += w;
        }
    }
    // Mesh preprocessing for the GPU: give the vertices a numbering with locality (admm_host_locality_order; the kernels
    // gather by vertex index).  Renumbers vertices, tets and faces when the mean edge span shrinks by more than 2x (or
    // force); returns new_id (identity when nothing was done) for everything else the caller indexes by vertex.
    std::vector<int> renumber_for_locality(bool force = false) {
        const int nv = (int)vertices.size(), nt = (int)tets.size();
        std::vector<int32_t> idx(4 * (size_t)nt), new_id(nv);
        for (int t = 0; t < nt; ++t) for (int c = 0; c < 4; ++c) idx[4 * (size_t)t + c] = tets[t][c];
        double before = 0, after = 0;
        admm_host_locality_order(nv, nt, 4, idx.data(), new_id.data(), &before, &after);
        std::vector<int> out(nv);
        if (!force && !(after < 0.5 * before)) { for (int i = 0; i < nv; ++i) out[i] = i; return out; }
        std::vector<Vec3> nvtx(nv);
        for (int i = 0; i < nv; ++i) { nvtx[new_id[i]] = vertices[i]; out[i] = new_id[i]; }
        vertices.swap(nvtx);
        for (Vec4i &t : tets) for (int c = 0; c < 4; ++c) t[c] = new_id[t[c]];
        for (Vec3i &f : faces) for (int c = 0; c < 3; ++c) f[c] = new_id[f[c]];
        return out;
    }
    // faces that belong to exactly one tet, outward orientation
    void surface_faces(std::vector<Vec3i> &faces) const {
        static const int F[4][3] = {{0, 2, 1}, {0, 1, 3}, {1, 2, 3}, {0, 3, 2}};
        std::map<std::array<int, 3>, std::pair<int, Vec3i> > seen;
        for (const Vec4i &t : tets)
            for (int f = 0; f < 4; ++f) {
                Vec3i tri(t[F[f][0]], t[F[f][1]], t[F[f][2]]);
                std::array<int, 3> key = {tri[0], tri[1], tri[2]};
                std::sort(key.begin(), key.end());
                auto it = seen.find(key);
                if (it == seen.end()) seen.emplace(key, std::make_pair(1, tri)); else it->second.first += 1;
            }
        faces.clear();
        for (const auto &kv : seen) if (kv.second.first == 1) faces.push_back(kv.second.second);
    }
    void surface_inds(std::vector<int> &inds) const {
        std::vector<Vec3i> faces; surface_faces(faces);
        std::vector<char> on(vertices.size(), 0);
        for (const Vec3i &f : faces) for (int c = 0; c < 3; ++c) on[f[c]] = 1;
        inds.clear();
        for (size_t i = 0; i < on.size(); ++i) if (on[i]) inds.push_back((int)i);
    }
};

struct TriangleMesh {
    std::vector<Vec3> vertices;
    std::vector<Vec3i> faces;
    int flags = 0;
    void translate(const Vec3 &t) { for (Vec3 &p : vertices) p += t; }
    void scale(double sx, double sy, double sz) { for (Vec3 &p : vertices) { p[0] *= sx; p[1] *= sy; p[2] *= sz; } }
    // density * area / 3 to every corner (AddMeshes.hpp:187 uses density 1)
    void weighted_masses(std::vector<double> &m, double density) const {
        m.assign(vertices.size(), 0.0);
        for (const Vec3i &f : faces) {
            const Vec3 &a = vertices[f[0]];
            const double w = density * 0.5 * (vertices[f[1]] - a).cross(vertices[f[2]] - a).norm() / 3.0;
            for (int c = 0; c < 3; ++c) m[f[c]] += w;
        }
    }
};

namespace factory {

// nx x ny x nz unit cells, 6 positively oriented tets per cell (Kuhn triangulation: every cell is cut along its
// main diagonal, so neighbouring cells conform).  The 6 tets of a cell are contiguous (cell-major order).
inline std::shared_ptr<TetMesh> make_tet_blocks(int nx, int ny, int nz) {
    if (nx < 1 || ny < 1 || nz < 1) throw std::runtime_error("make_tet_blocks: need at least one cell per axis");
    auto mesh = std::make_shared<TetMesh>();
    auto vid = [&](int i, int j, int k) { return (i * (ny + 1) + j) * (nz + 1) + k; };
    for (int i = 0; i <= nx; ++i) for (int j = 0; j <= ny; ++j) for (int k = 0; k <= nz; ++k) mesh->vertices.push_back(Vec3(i, j, k));
    static const int perms[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
    static const bool odd[6] = {false, true, true, false, false, true};
    for (int i = 0; i < nx; ++i) for (int j = 0; j < ny; ++j) for (int k = 0; k < nz; ++k)
        for (int p = 0; p < 6; ++p) {
            int c[3] = {i, j, k};
            Vec4i t;
            t[0] = vid(c[0], c[1], c[2]);
            for (int s = 0; s < 3; ++s) { c[perms[p][s]] += 1; t[s + 1] = vid(c[0], c[1], c[2]); }
            if (odd[p]) std::swap(t[2], t[3]);   // orientation = parity of the axis permutation
            mesh->tets.push_back(t);
        }
    return mesh;
}

// The cells of an n^3 Kuhn lattice over [lo, hi]^3 whose centre satisfies inside(p): a stand-in for the reference's
// samples/data sphere / torus TetGen meshes (any closed implicit body; unused lattice vertices are dropped).
template <class Inside>
inline std::shared_ptr<TetMesh> make_implicit_body(int n, double lo, double hi, Inside inside) {
    auto full = make_tet_blocks(n, n, n);
    const double h = (hi - lo) / n;
    for (Vec3 &p : full->vertices) p = Vec3(lo + h * p[0], lo + h * p[1], lo + h * p[2]);
    auto mesh = std::make_shared<TetMesh>();
    std::vector<int> id(full->vertices.size(), -1);
    for (size_t t = 0; t < full->tets.size(); t += 6) {     // the 6 tets of a cell are contiguous: keep or drop the cell
        Vec3 c(0, 0, 0);
        for (int q = 0; q < 6; ++q) for (int k = 0; k < 4; ++k) c += full->vertices[full->tets[t + q][k]];
        if (!inside(c * (1.0 / 24.0))) continue;
        for (int q = 0; q < 6; ++q) {
            Vec4i tt = full->tets[t + q];
            for (int k = 0; k < 4; ++k) {
                if (id[tt[k]] < 0) { id[tt[k]] = (int)mesh->vertices.size(); mesh->vertices.push_back(full->vertices[tt[k]]); }
                tt[k] = id[tt[k]];
            }
            mesh->tets.push_back(tt);
        }
    }
    if (mesh->tets.empty()) throw std::runtime_error("make_implicit_body: no cell inside");
    return mesh;
}
inline std::shared_ptr<TetMesh> make_ball(int n, double radius = 1.0) {
    return make_implicit_body(n, -radius, radius, [radius](const Vec3 &p) { return p.dot(p) <= radius * radius; });
}
// torus around the y axis: ring radius R, tube radius r
inline std::shared_ptr<TetMesh> make_torus(int n, double R = 1.0, double r = 0.4) {
    return make_implicit_body(n, -(R + r), R + r, [R, r](const Vec3 &p) {
        const double q = std::sqrt(p[0] * p[0] + p[2] * p[2]) - R;
        return q * q + p[1] * p[1] <= r * r;
    });
}

// m x m cells in the xz-plane at height y, extent `size`, two triangles (a,b,c), (a,c,d) per cell
inline std::shared_ptr<TriangleMesh> make_plane(int m, double size = 1.0, double y = 0.0) {
    if (m < 1) throw std::runtime_error("make_plane: need at least one cell");
    auto mesh = std::make_shared<TriangleMesh>();
    for (int i = 0; i <= m; ++i) for (int k = 0; k <= m; ++k) mesh->vertices.push_back(Vec3(i * (size / m), y, k * (size / m)));
    for (int i = 0; i < m; ++i) for (int k = 0; k < m; ++k) {
        const int a = i * (m + 1) + k, b = (i + 1) * (m + 1) + k, c = (i + 1) * (m + 1) + k + 1, d = i * (m + 1) + k + 1;
        mesh->faces.push_back(Vec3i(a, b, c)); mesh->faces.push_back(Vec3i(a, c, d));
    }
    return mesh;
}

} // namespace factory

namespace meshio {

inline bool next_data_line(std::istream &in, std::istringstream &ls) {
    std::string line;
    while (std::getline(in, line)) {
        const size_t h = line.find('#');
        if (h != std::string::npos) line.erase(h);
        if (line.find_first_not_of(" \t\r\n") == std::string::npos) continue;
        ls.clear(); ls.str(line);
        return true;
    }
    return false;
}

// TetGen <prefix>.node + <prefix>.ele (the format of the reference's samples/data/*).  Indices may be 0- or
// 1-based (detected from the first vertex index); negatively oriented tets are flipped, because the energy
// terms reject inverted rest poses (TetEnergyTerm.cpp:42-44).
inline std::shared_ptr<TetMesh> load_tetgen(const std::string &prefix) {
    auto mesh = std::make_shared<TetMesh>();
    std::ifstream node(prefix + ".node"), ele(prefix + ".ele");
    if (!node || !ele) throw std::runtime_error("load_tetgen: cannot open " + prefix + ".node/.ele");
    std::istringstream ls;
    int nv = 0, dim = 0, first = 0;
    if (!next_data_line(node, ls) || !(ls >> nv >> dim) || dim != 3) throw std::runtime_error("load_tetgen: bad .node header");
    mesh->vertices.resize(nv);
    for (int i = 0; i < nv; ++i) {
        int id; double x, y, z;
        if (!next_data_line(node, ls) || !(ls >> id >> x >> y >> z)) throw std::runtime_error("load_tetgen: truncated .node");
        if (i == 0) first = id;
        if (id - first < 0 || id - first >= nv) throw std::runtime_error("load_tetgen: vertex index out of range");
        mesh->vertices[id - first] = Vec3(x, y, z);
    }
    int nt = 0, per = 0;
    if (!next_data_line(ele, ls) || !(ls >> nt >> per) || per < 4) throw std::runtime_error("load_tetgen: bad .ele header");
    mesh->tets.resize(nt);
    for (int i = 0; i < nt; ++i) {
        int id, a, b, c, d;
        if (!next_data_line(ele, ls) || !(ls >> id >> a >> b >> c >> d)) throw std::runtime_error("load_tetgen: truncated .ele");
        Vec4i t(a - first, b - first, c - first, d - first);
        for (int k = 0; k < 4; ++k) if (t[k] < 0 || t[k] >= nv) throw std::runtime_error("load_tetgen: tet index out of range");
        mesh->tets[i] = t;
        if (mesh->signed_volume(i) < 0.0) std::swap(mesh->tets[i][2], mesh->tets[i][3]);
    }
    return mesh;
}

inline void save_tetgen(const std::string &prefix, const TetMesh &mesh) {
    std::ofstream node(prefix + ".node"), ele(prefix + ".ele");
    if (!node || !ele) throw std::runtime_error("save_tetgen: cannot open " + prefix);
    node.precision(17); node << mesh.vertices.size() << " 3 0 0\n";
    for (size_t i = 0; i < mesh.vertices.size(); ++i) node << i << " " << mesh.vertices[i][0] << " " << mesh.vertices[i][1] << " " << mesh.vertices[i][2] << "\n";
    ele << mesh.tets.size() << " 4 0\n";
    for (size_t i = 0; i < mesh.tets.size(); ++i) ele << i << " " << mesh.tets[i][0] << " " << mesh.tets[i][1] << " " << mesh.tets[i][2] << " " << mesh.tets[i][3] << "\n";
}

// Wavefront OBJ of a triangle list over the solver's node vector (x = 3 doubles per node)
inline void save_obj(const std::string &path, const VecX &x, const std::vector<Vec3i> &faces, int vertex_offset = 0) {
    std::ofstream out(path);
    if (!out) throw std::runtime_error("save_obj: cannot open " + path);
    out.precision(9);
    for (int i = 0; i < x.size() / 3; ++i) out << "v " << x[3 * i] << " " << x[3 * i + 1] << " " << x[3 * i + 2] << "\n";
    for (const Vec3i &f : faces) out << "f " << f[0] + vertex_offset + 1 << " " << f[1] + vertex_offset + 1 << " " << f[2] + vertex_offset + 1 << "\n";
}

// positions of all nodes, 17 significant digits, one node per line (offline comparison of trajectories)
inline void save_positions(const std::string &path, const VecX &x) {
    std::ofstream out(path);
    if (!out) throw std::runtime_error("save_positions: cannot open " + path);
    out.precision(17);
    for (int i = 0; i < x.size() / 3; ++i) out << x[3 * i] << " " << x[3 * i + 1] << " " << x[3 * i + 2] << "\n";
}

} // namespace meshio
} // namespace admm
#endif
