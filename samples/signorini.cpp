// Headless version of the reference's samples/tvcg2017/signorini.cpp: a very soft ball (Lame(1e6, 0.299), linear tets,
// NOSELFCOLLISION) dropped on a Floor at y = -1, global step = nodal multi-colour Gauss-Seidel (-ls 1, the reference's
// setting) whose sweeps project the contacting nodes onto the plane (src/NodalMultiColorGS.hpp:218-262).
// The reference loads samples/data/sphere; here the ball comes from factory::make_ball (--mesh PREFIX loads a TetGen
// .node/.ele pair instead, e.g. the reference's own file).
//   usage: signorini [Settings flags] [--frames N] [--cells M] [--mesh prefix] [--out prefix] [--out-every K] [--csv file]
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
    settings.linsolver = 1; // NCMCGS (signorini.cpp:40)
    FrameLog log;
    std::vector<char *> a1 = log.parse(argc, argv);
    int frames = 60, cells = 10;
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
        std::shared_ptr<TetMesh> mesh = mesh_prefix.empty() ? factory::make_ball(cells, 0.5) : meshio::load_tetgen(mesh_prefix);
        mesh->renumber_for_locality();
        mesh->flags |= binding::NOSELFCOLLISION | binding::LINEAR;
        Solver solver;
        binding::add_tetmesh(&solver, mesh, Lame(1000000, 0.299), settings.verbose > 0);
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
        printf("signorini: %d frames, %d tets, ball y in [%.5f, %.5f], floor %.1f\n", frames, (int)mesh->tets.size(), ymin, ymax, floor_y);
        if (!log.out_prefix.empty()) { meshio::save_positions(log.out_prefix + ".xyz", solver.m_x); meshio::save_obj(log.out_prefix + ".obj", solver.m_x, mesh->faces); }
    } catch (const std::exception &e) {
        std::cerr << "signorini: " << e.what() << std::endl;
        return EXIT_FAILURE;
    }
    return EXIT_SUCCESS;
}
