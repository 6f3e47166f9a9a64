// Headless version of the reference's samples/tvcg2017/boxes.cpp: two unit boxes with self-collision proxies
// (binding::add_tetmesh without NOSELFCOLLISION registers a TetMeshCollision per mesh), the second one two units above
// the first, Lame::rubber(), a Floor at y = -1: the lower box lands on the floor, the upper one on the lower one.
// The reference's sample loads samples/data/box768 (768 tets); here the boxes come from factory::make_tet_blocks
// (5^3 cells = 750 tets).  Default solver as in the reference: the multi-colour GS (-ls 1) with the collision penalty
// C^T C added to the rows of the touched nodes; -ls 2 (UzawaCG, the set-up of tvcg2017/torus.cpp) works as well.
//   usage: boxes [Settings flags] [--frames N] [--cells M] [--gap G] [--out prefix]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include "AddMeshes.hpp"
#include "PassiveObject.hpp"

using namespace admm;

int main(int argc, char **argv) {
    Solver::Settings settings;
    settings.linsolver = 1; // NCMCGS (boxes.cpp:46)
    int frames = 48, cells = 5;
    double gap = 2.0;   // the reference's trans_up = i * 2
    std::string out;
    std::vector<char *> rest = {argv[0]};
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--frames") && i + 1 < argc) frames = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--cells") && i + 1 < argc) cells = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--gap") && i + 1 < argc) gap = atof(argv[++i]);
        else if (!strcmp(argv[i], "--out") && i + 1 < argc) out = argv[++i];
        else rest.push_back(argv[i]);
    }
    if (settings.parse_args((int)rest.size(), rest.data())) return EXIT_SUCCESS;

    std::vector<std::shared_ptr<TetMesh> > meshes = {factory::make_tet_blocks(cells, cells, cells), factory::make_tet_blocks(cells, cells, cells)};
    Solver solver;
    for (int i = 0; i < (int)meshes.size(); ++i) {
        meshes[i]->flags |= binding::LINEAR;
        meshes[i]->scale(1.0 / cells, 1.0 / cells, 1.0 / cells);
        // a slight sideways offset of the upper box: with perfectly aligned grids many query points are equidistant
        // from two surface triangles
        meshes[i]->translate(Vec3(-0.5 + 0.013 * i, -0.5 + i * gap, -0.5 + 0.007 * i));
        binding::add_tetmesh(&solver, meshes[i], Lame::rubber(), settings.verbose > 0);
    }
    const double floor_y = -1.0;
    solver.add_obstacle(std::make_shared<Floor>(floor_y));
    const int nv0 = (int)meshes[0]->vertices.size();
    double collision_ms = 0.0;
    try {
        if (!solver.initialize(settings)) return EXIT_FAILURE;
        for (int f = 0; f < frames; ++f) {
            solver.step();
            const Solver::RuntimeData &rd = solver.runtime_data();
            collision_ms += rd.collision_ms;
            double top0 = -1e300, bot1 = 1e300, bot0 = 1e300;
            for (int i = 0; i < nv0; ++i) { top0 = std::max(top0, solver.m_x[3 * i + 1]); bot0 = std::min(bot0, solver.m_x[3 * i + 1]); }
            for (int i = nv0; i < solver.m_x.size() / 3; ++i) bot1 = std::min(bot1, solver.m_x[3 * i + 1]);
            if (settings.verbose > 0)
                printf("frame %d: local %.3f ms, global %.3f ms, collision %.3f ms, inner iters %d; lower box y [%.4f, %.4f], upper box min y %.4f\n",
                       f, rd.local_ms, rd.global_ms, rd.collision_ms, rd.inner_iters, bot0, top0, bot1);
        }
    } catch (const std::exception &e) {
        std::cerr << "boxes: " << e.what() << std::endl;
        return EXIT_FAILURE;
    }
    double top0 = -1e300, bot1 = 1e300, bot0 = 1e300;
    for (int i = 0; i < nv0; ++i) { top0 = std::max(top0, solver.m_x[3 * i + 1]); bot0 = std::min(bot0, solver.m_x[3 * i + 1]); }
    for (int i = nv0; i < solver.m_x.size() / 3; ++i) bot1 = std::min(bot1, solver.m_x[3 * i + 1]);
    printf("boxes: %d frames, lower box y in [%.5f, %.5f], upper box min y %.5f, floor %.1f\n", frames, bot0, top0, bot1, floor_y);
    if (!out.empty()) {
        meshio::save_positions(out + ".xyz", solver.m_x);
        std::vector<Vec3i> all;
        for (Vec3i f : meshes[0]->faces) all.push_back(f);
        for (Vec3i f : meshes[1]->faces) { for (int c = 0; c < 3; ++c) f[c] += nv0; all.push_back(f); }
        meshio::save_obj(out + ".obj", solver.m_x, all);
    }
    return EXIT_SUCCESS;
}
