// Headless version of the reference's samples/sca2016/beams.cpp on the MI355X build: three 12x3x3-cell beams
// (linear / Neo-Hookean / StVK, soft rubber, 1 m tall, 1.75 m apart along y), the min-x / max-x faces pinned
// and pulled apart by dt * (1,0,0) per frame (beams.cpp:43-132).  No window: the scene is stepped for a fixed
// number of frames, RuntimeData is printed per frame and the final positions / surface can be written out.
//   usage: beams [Solver::Settings flags: -it -dt -g -ls -ck -v] [--frames N] [--cells C] [--out prefix]
// BASELINE configs[0] is this scene with -it 10.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include "AddMeshes.hpp"

using namespace admm;

static std::vector<int> left_pins, right_pins;   // -x, +x
static std::vector<Vec3> left_points, right_points;

static void find_pins(const std::vector<std::shared_ptr<TetMesh> > &meshes) {
    int nv_offset = 0;
    for (const auto &mesh : meshes) {
        Vec3 lo, hi;
        mesh->bounds(lo, hi);
        const double min_x = lo[0] + 1e-2, max_x = hi[0] - 1e-2;
        for (int j = 0; j < (int)mesh->vertices.size(); ++j) {
            const Vec3 &v = mesh->vertices[j];
            if (v[0] < min_x) { left_pins.push_back(j + nv_offset); left_points.push_back(v); }
            if (v[0] > max_x) { right_pins.push_back(j + nv_offset); right_points.push_back(v); }
        }
        nv_offset += (int)mesh->vertices.size();
    }
}

static void stretch_beams(Solver &solver) {
    const Vec3 move = Vec3(1.0, 0.0, 0.0) * solver.settings().timestep_s;
    std::vector<int> pins; std::vector<Vec3> points;
    for (size_t i = 0; i < left_pins.size(); ++i) { left_points[i] -= move; pins.push_back(left_pins[i]); points.push_back(left_points[i]); }
    for (size_t i = 0; i < right_pins.size(); ++i) { right_points[i] += move; pins.push_back(right_pins[i]); points.push_back(right_points[i]); }
    solver.set_pins(pins, points);
}

int main(int argc, char **argv) {
    Solver::Settings settings;
    settings.admm_iters = 20;   // the sample's own default (beams.cpp:40)
    int frames = 24, dim = 3;
    std::string out;
    std::vector<char *> rest = {argv[0]};
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--frames") && i + 1 < argc) frames = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--cells") && i + 1 < argc) dim = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--out") && i + 1 < argc) out = argv[++i];
        else rest.push_back(argv[i]);
    }
    if (settings.parse_args((int)rest.size(), rest.data())) return EXIT_SUCCESS;

    std::vector<std::shared_ptr<TetMesh> > meshes = {
        factory::make_tet_blocks(dim * 4, dim, dim), factory::make_tet_blocks(dim * 4, dim, dim), factory::make_tet_blocks(dim * 4, dim, dim)};
    const int flags[3] = {binding::NOSELFCOLLISION | binding::LINEAR, binding::NOSELFCOLLISION | binding::NEOHOOKEAN,
                          binding::NOSELFCOLLISION | binding::STVK};
    const double yoff[3] = {1.75, 0.0, -1.75};
    for (int i = 0; i < 3; ++i) {   // centre, make each beam 1 m tall, spread along y
        Vec3 lo, hi;
        meshes[i]->bounds(lo, hi);
        meshes[i]->translate((lo + hi) * -0.5);
        const double s = 1.0 / (hi[1] - lo[1]);
        meshes[i]->scale(s, s, s);
        meshes[i]->translate(Vec3(0.0, yoff[i], 0.0));
        meshes[i]->flags = flags[i];
    }
    Solver solver;
    const Lame softRubber(10000000, 0.399);
    for (auto &m : meshes) binding::add_tetmesh(&solver, m, softRubber, settings.verbose > 0);
    find_pins(meshes);
    stretch_beams(solver);   // initial pins (before initialize: they become SpringPin terms)
    try {
        if (!solver.initialize(settings)) return EXIT_FAILURE;
        for (int f = 0; f < frames; ++f) {
            if (f > 0) stretch_beams(solver);
            solver.step();
            const Solver::RuntimeData &rd = solver.runtime_data();
            if (settings.verbose > 0)
                printf("frame %d: local %.3f ms, global %.3f ms, collision %.3f ms, inner iters %d\n", f, rd.local_ms, rd.global_ms, rd.collision_ms, rd.inner_iters);
        }
    } catch (const std::exception &e) {
        std::cerr << "beams: " << e.what() << std::endl;
        return EXIT_FAILURE;
    }
    if (!out.empty()) {
        meshio::save_positions(out + ".xyz", solver.m_x);
        std::vector<Vec3i> faces; int off = 0;
        std::vector<Vec3i> all;
        for (auto &m : meshes) { m->surface_faces(faces); for (Vec3i f : faces) { for (int c = 0; c < 3; ++c) f[c] += off; all.push_back(f); } off += (int)m->vertices.size(); }
        meshio::save_obj(out + ".obj", solver.m_x, all);
    }
    return EXIT_SUCCESS;
}
