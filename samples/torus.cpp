// Headless version of the reference's samples/tvcg2017/torus.cpp: a squishy torus (Lame(1e6, 0.1), linear tets, WITH its
// self-collision proxy: binding::add_tetmesh registers a TetMeshCollision), lifted to y = 2 and tilted by 3 degrees about
// the x axis, dropped on a Floor at y = -1; global step = UzawaCG (-ls 2, 10 ADMM iterations, the reference's settings):
// floor contacts and self-contacts are the rows of C of the Schur-complement CG (src/UzawaCG.hpp).
// The reference loads samples/data/torus; here the torus comes from factory::make_torus (--mesh PREFIX loads a TetGen
// .node/.ele pair instead, e.g. the reference's own file).
//   usage: torus [Settings flags] [--frames N] [--cells M] [--mesh prefix] [--out prefix] [--out-every K] [--csv file]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include "AddMeshes.hpp"
#include "FrameLog.hpp"
#include "PassiveObject.hpp"

using namespace admm;

int main(int argc, char **argv) {
    Solver::Settings settings;
    settings.linsolver = 2;   // UzawaCG (torus.cpp:38)
    settings.admm_iters = 10; // torus.cpp:39
    FrameLog log;
    std::vector<char *> a1 = log.parse(argc, argv);
    int frames = 60, cells = 14;
    std::string mesh_prefix;
    std::vector<char *> rest = {a1[0]};
    for (size_t i = 1; i < a1.size(); ++i) {
        if (!strcmp(a1[i], "--frames") && i + 1 < a1.size()) frames = atoi(a1[++i]);
        else if (!strcmp(a1[i], "--cells") && i + 1 < a1.size()) cells = atoi(a1[++i]);
        else if (!strcmp(a1[i], "--mesh") && i + 1 < a1.size()) mesh_prefix = a1[++i];
        else rest.push_back(a1[i]);
    }
    if (settings.parse_args((int)rest.size(), rest.data())) return EXIT_SUCCESS;
    try {
        std::shared_ptr<TetMesh> mesh = mesh_prefix.empty() ? factory::make_torus(cells, 0.7, 0.3) : meshio::load_tetgen(mesh_prefix);
        mesh->renumber_for_locality();
        mesh->flags |= binding::LINEAR;
        const double ang = -3.0 * M_PI / 180.0, ca = std::cos(ang), sa = std::sin(ang);   // make_trans(0,2,0) * make_rot(-3 deg, x)
        for (Vec3 &p : mesh->vertices) { const double y = p[1], z = p[2]; p[1] = ca * y - sa * z + 2.0; p[2] = sa * y + ca * z; }
        Solver solver;
        binding::add_tetmesh(&solver, mesh, Lame(1000000, 0.1), settings.verbose > 0);
        const double floor_y = -1.0;
        solver.add_obstacle(std::make_shared<Floor>(floor_y));
        mesh->need_faces();
        if (!solver.initialize(settings)) return EXIT_FAILURE;
        double ymin = 0.0, ymax = 0.0;
        for (int f = 0; f < frames; ++f) {
            log.step(solver);
            log.frame(f, frames, solver, mesh->faces);
            ymin = 1e300; ymax = -1e300;
            for (int i = 0; i < solver.m_x.size() / 3; ++i) { ymin = std::min(ymin, solver.m_x[3 * i + 1]); ymax = std::max(ymax, solver.m_x[3 * i + 1]); }
            if (settings.verbose > 0) { printf("frame %d: y in [%.5f, %.5f]", f, ymin, ymax); Solver::RuntimeData rd = solver.runtime_data(); rd.print(solver.settings()); }
        }
        printf("torus: %d frames, %d tets, torus y in [%.5f, %.5f], floor %.1f\n", frames, (int)mesh->tets.size(), ymin, ymax, floor_y);
        if (!log.out_prefix.empty()) { meshio::save_positions(log.out_prefix + ".xyz", solver.m_x); meshio::save_obj(log.out_prefix + ".obj", solver.m_x, mesh->faces); }
    } catch (const std::exception &e) {
        std::cerr << "torus: " << e.what() << std::endl;
        return EXIT_FAILURE;
    }
    return EXIT_SUCCESS;
}
