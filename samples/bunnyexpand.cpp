// Headless version of the reference's samples/sca2016/bunnyexpand.cpp: a Neo-Hookean body whose vertices are all thrown
// to random places in [-0.75, 0.75]^3 ("rand", the default) or to a single point ("point"), no gravity, exact global
// solve (-ls 0): the prox has to pull the mesh back to its rest shape through fully inverted states
// (bunnyexpand.cpp:41-59, set_vertices :112-135).  The reference loads samples/data/bunny_1124; here the body is a
// factory::make_tet_blocks cube unless --mesh <TetGen prefix> names a .node/.ele pair (e.g. the reference's data).
// With the nearly incompressible default material (nu = 0.499) the untangling takes hundreds of frames -- the CPU oracle
// shows the same pace (experiments/expand_check2.py); --lame soft (nu = 0.399) recovers within ten frames.
//   usage: bunnyexpand [point|rand] [Settings flags] [--frames N] [--cells M] [--size S] [--lame rubber|soft|verysoft] [--mesh prefix] [--seed S] [--out prefix]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <random>
#include "AddMeshes.hpp"

using namespace admm;

int main(int argc, char **argv) {
    Solver::Settings settings;
    settings.linsolver = 0;  // LDLT (bunnyexpand.cpp:56)
    settings.gravity = 0;    // :57
    settings.admm_iters = 20;
    int frames = 60, cells = 4;
    double size = 3.0;       // edge of the synthetic cube: twice the scramble box, so the start is a compressed state
    unsigned seed = 1;
    bool single_point_init = false;
    std::string out, mesh_prefix, lame_name = "rubber";   // the reference adds the mesh with the default Lame::rubber()
    std::vector<char *> rest = {argv[0]};
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "point")) single_point_init = true;
        else if (!strcmp(argv[i], "rand")) single_point_init = false;
        else if (!strcmp(argv[i], "--frames") && i + 1 < argc) frames = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--cells") && i + 1 < argc) cells = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--size") && i + 1 < argc) size = atof(argv[++i]);
        else if (!strcmp(argv[i], "--lame") && i + 1 < argc) lame_name = argv[++i];
        else if (!strcmp(argv[i], "--mesh") && i + 1 < argc) mesh_prefix = argv[++i];
        else if (!strcmp(argv[i], "--seed") && i + 1 < argc) seed = (unsigned)atoi(argv[++i]);
        else if (!strcmp(argv[i], "--out") && i + 1 < argc) out = argv[++i];
        else rest.push_back(argv[i]);
    }
    if (settings.parse_args((int)rest.size(), rest.data())) return EXIT_SUCCESS;

    int inverted = 0;
    double worst = 0.0;
    try {
        std::shared_ptr<TetMesh> mesh = mesh_prefix.empty() ? factory::make_tet_blocks(cells, cells, cells) : meshio::load_tetgen(mesh_prefix);
        if (mesh_prefix.empty()) mesh->scale(size / cells, size / cells, size / cells);
        else mesh->renumber_for_locality();   // mesh files number their vertices in insertion order: give the GPU a local numbering
        mesh->flags |= binding::NOSELFCOLLISION | binding::NEOHOOKEAN;
        Solver solver;
        const Lame lame = lame_name == "soft" ? Lame::soft_rubber() : lame_name == "verysoft" ? Lame::very_soft_rubber() : Lame::rubber();
        binding::add_tetmesh(&solver, mesh, lame, settings.verbose > 0);
        if (!solver.initialize(settings)) return EXIT_FAILURE;
        // set_vertices (:112-135)
        std::mt19937 gen(seed);
        std::uniform_real_distribution<double> dis(-0.75, 0.75);
        for (int i = 0; i < solver.m_x.size(); ++i) solver.m_x[i] = single_point_init ? 1e-9 * dis(gen) : dis(gen);
        for (int f = 0; f < frames; ++f) {
            solver.step();
            TetMesh now = *mesh;
            for (size_t i = 0; i < now.vertices.size(); ++i) now.vertices[i] = Vec3(solver.m_x[3 * i], solver.m_x[3 * i + 1], solver.m_x[3 * i + 2]);
            inverted = 0; worst = 0.0;
            for (size_t t = 0; t < now.tets.size(); ++t) {
                if (now.signed_volume((int)t) <= 0.0) ++inverted;
                for (int a = 0; a < 4; ++a) for (int b = a + 1; b < 4; ++b) {
                    const double l0 = (mesh->vertices[mesh->tets[t][a]] - mesh->vertices[mesh->tets[t][b]]).norm();
                    const double l1 = (now.vertices[now.tets[t][a]] - now.vertices[now.tets[t][b]]).norm();
                    worst = std::max(worst, std::abs(l1 / l0 - 1.0));
                }
            }
            if (settings.verbose > 0) printf("frame %d: %d of %d tets inverted, worst edge-length error %.3g\n", f, inverted, (int)now.tets.size(), worst);
        }
        if (!out.empty()) meshio::save_positions(out + ".xyz", solver.m_x);
    } catch (const std::exception &e) {
        std::cerr << "bunnyexpand: " << e.what() << std::endl;
        return EXIT_FAILURE;
    }
    printf("bunnyexpand: %d frames, %d tets inverted, worst edge-length error %.3g\n", frames, inverted, worst);
    return EXIT_SUCCESS;
}
