// curtain: a headless sample for the three terms the reference's README lists as TODOs and never shipped (README.md:23-28), built on the
// same class API as the reference's own samples (samples/sca2016/trianglestrain.cpp is the model):
//   * a cloth sheet with a BENDING term (create_bends_from_mesh -> BendEnergyTerm per interior edge, next to the TriEnergyTerm stretch terms),
//   * hung on a curtain rail: the vertices of its top edge carry SLIDE constraints (Solver::set_slide_pins) -- each may move along the rail
//     direction only (two planes per vertex would fix a line; here one plane normal to z keeps the edge in the rail's vertical plane, and
//     the two end vertices are pinned),
//   * next to it a block of STABLE NEO-HOOKEAN tets (StableNeoHookeanTet) started from a crushed (inverted) state, which an ordinary
//     Neo-Hookean block cannot be (log J).
//   usage: curtain [Settings flags] [--frames N] [--cells M] [--bend K] [--out prefix]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include "AddMeshes.hpp"

using namespace admm;

int main(int argc, char **argv) {
    Solver::Settings settings;
    int frames = 24, cells = 10;
    double k_bend = 0.05;
    std::string out;
    std::vector<char *> rest = {argv[0]};
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--frames") && i + 1 < argc) frames = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--cells") && i + 1 < argc) cells = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--bend") && i + 1 < argc) k_bend = atof(argv[++i]);
        else if (!strcmp(argv[i], "--out") && i + 1 < argc) out = argv[++i];
        else rest.push_back(argv[i]);
    }
    if (settings.parse_args((int)rest.size(), rest.data())) return EXIT_SUCCESS;

    Solver solver;
    // ---- the curtain: a horizontal sheet (y = 0.5) of 2 x 2 m whose max-z edge hangs on the rail ----
    std::shared_ptr<TriangleMesh> sheet = factory::make_plane(cells, 2.0, 0.5);
    sheet->flags = binding::NOSELFCOLLISION | binding::LINEAR;
    sheet->translate(Vec3(-1.0, 0.0, -1.0));
    Lame cloth(100, 0.1);
    binding::add_trimesh(&solver, sheet, cloth, settings.verbose > 0);
    const int nv_sheet = (int)sheet->vertices.size();
    std::vector<double> xs(3 * (size_t)nv_sheet);
    std::vector<int> tri(3 * sheet->faces.size());
    for (int i = 0; i < nv_sheet; ++i) for (int a = 0; a < 3; ++a) xs[3 * i + a] = sheet->vertices[i][a];
    for (size_t t = 0; t < sheet->faces.size(); ++t) for (int c = 0; c < 3; ++c) tri[3 * t + c] = sheet->faces[t][c];
    const int n_hinges = create_bends_from_mesh<double>(solver.energyterms, xs.data(), nv_sheet, tri.data(), (int)sheet->faces.size(), k_bend, 0);
    double zmax = -1e300, xmin = 1e300, xmax = -1e300;
    for (const Vec3 &v : sheet->vertices) { zmax = std::max(zmax, v[2]); xmin = std::min(xmin, v[0]); xmax = std::max(xmax, v[0]); }
    std::vector<int> pins, sliders; std::vector<Vec3> slide_pts, slide_nrm;
    for (int i = 0; i < nv_sheet; ++i) {
        const Vec3 &v = sheet->vertices[i];
        if (v[2] < zmax - 1e-9) continue;
        if (v[0] < xmin + 1e-9 || v[0] > xmax - 1e-9) pins.push_back(i);
        else { sliders.push_back(i); slide_pts.push_back(v); slide_nrm.push_back(Vec3(0.0, 1.0, 0.0)); }   // stays at rail height, free along the rail
    }
    // ---- the block: stable Neo-Hookean, started mirrored through its mid-plane (every tet inverted) ----
    std::shared_ptr<TetMesh> block = factory::make_tet_blocks(3, 3, 3);
    block->flags = binding::NOSELFCOLLISION;
    const int off = solver.m_x.rows() / 3;
    {
        std::vector<double> masses;
        block->weighted_masses(masses, 1100.0);
        const int nb = (int)block->vertices.size();
        std::vector<double> x(3 * (size_t)nb), m3(3 * (size_t)nb);
        for (int i = 0; i < nb; ++i) for (int a = 0; a < 3; ++a) { x[3 * i + a] = block->vertices[i][a] + (a == 0 ? 3.0 : 0.0); m3[3 * i + a] = masses[i]; }
        solver.add_nodes(x.data(), m3.data(), nb);
        std::vector<int> inds(4 * block->tets.size());
        for (size_t t = 0; t < block->tets.size(); ++t) for (int c = 0; c < 4; ++c) inds[4 * t + c] = block->tets[t][c];
        create_tets_from_mesh<double, StableNeoHookeanTet>(solver.energyterms, x.data(), inds.data(), (int)block->tets.size(), Lame::very_soft_rubber(), off);
        double ymin = 1e300, ymax = -1e300;
        for (int i = 0; i < nb; ++i) { ymin = std::min(ymin, x[3 * i + 1]); ymax = std::max(ymax, x[3 * i + 1]); }
        for (int i = 0; i < nb; ++i) {
            if (x[3 * i + 1] < ymin + 1e-9) pins.push_back(off + i);                         // its bottom face is glued down
            else solver.m_x[3 * (off + i) + 1] = ymin - 0.5 * (x[3 * i + 1] - ymin);          // the rest starts folded through the bottom face
        }
    }
    std::vector<Vec3> pin_pts;
    for (int v : pins) pin_pts.push_back(v < off ? Vec3(solver.m_x.segment<3>(3 * v)) : Vec3(solver.m_x[3 * v], solver.m_x[3 * v + 1], solver.m_x[3 * v + 2]));
    solver.set_pins(pins, pin_pts);
    solver.set_slide_pins(sliders, slide_pts, slide_nrm);
    try {
        if (!solver.initialize(settings)) return EXIT_FAILURE;
        for (int f = 0; f < frames; ++f) solver.step();
    } catch (const std::exception &e) {
        std::cerr << "curtain: " << e.what() << std::endl;
        return EXIT_FAILURE;
    }
    double rail_off = 0.0, slid = 0.0;
    for (size_t i = 0; i < sliders.size(); ++i) {
        rail_off = std::max(rail_off, std::fabs(solver.m_x[3 * sliders[i] + 1] - slide_pts[i][1]));
        slid = std::max(slid, std::fabs(solver.m_x[3 * sliders[i]] - slide_pts[i][0]) + std::fabs(solver.m_x[3 * sliders[i] + 2] - slide_pts[i][2]));
    }
    printf("curtain: %d hinges, %d slide constraints, %d frames: sliders off the rail plane by %.3e, moved %.3e inside it\n", n_hinges, (int)sliders.size(), frames, rail_off, slid);
    if (!out.empty()) meshio::save_positions(out + ".xyz", solver.m_x);
    return EXIT_SUCCESS;
}
