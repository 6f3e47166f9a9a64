// Headless version of the reference's samples/sca2016/trianglestrain.cpp: two cloth sheets side by side,
// Lame(100, 0.1), the left one with hard strain limits 0.95 / 1.05, the two top corners of each sheet pinned,
// falling under gravity (trianglestrain.cpp:34-58).  With --floor Y a passive floor is added and the scene needs
// a constraint-capable global solver (-ls 1 multi-colour GS or -ls 2 UzawaCG), as BASELINE configs[4].
//   usage: trianglestrain [Settings flags] [--frames N] [--cells M] [--floor Y] [--out prefix]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include "AddMeshes.hpp"
#include "PassiveObject.hpp"

using namespace admm;

// the two corners of the sheet's max-z edge (the "top" edge of the reference's upright plane)
static void get_pins(const TriangleMesh &mesh, std::vector<int> &pin_ids, int idx_offset) {
    double zmax = -1e300, xmin = 1e300, xmax = -1e300;
    for (const Vec3 &v : mesh.vertices) { zmax = std::max(zmax, v[2]); xmin = std::min(xmin, v[0]); xmax = std::max(xmax, v[0]); }
    int left = -1, right = -1;
    for (int i = 0; i < (int)mesh.vertices.size(); ++i) {
        const Vec3 &v = mesh.vertices[i];
        if (v[2] < zmax - 1e-9) continue;
        if (v[0] < xmin + 1e-9) left = i;
        if (v[0] > xmax - 1e-9) right = i;
    }
    pin_ids.push_back(left + idx_offset); pin_ids.push_back(right + idx_offset);
}

int main(int argc, char **argv) {
    Solver::Settings settings;
    int frames = 24, cells = 10;
    bool has_floor = false; double floor_y = 0.0;
    std::string out;
    std::vector<char *> rest = {argv[0]};
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--frames") && i + 1 < argc) frames = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--cells") && i + 1 < argc) cells = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--floor") && i + 1 < argc) { has_floor = true; floor_y = atof(argv[++i]); }
        else if (!strcmp(argv[i], "--out") && i + 1 < argc) out = argv[++i];
        else rest.push_back(argv[i]);
    }
    if (settings.parse_args((int)rest.size(), rest.data())) return EXIT_SUCCESS;

    std::vector<std::shared_ptr<TriangleMesh> > meshes = {factory::make_plane(cells, 2.0, 0.5), factory::make_plane(cells, 2.0, 0.5)};
    meshes[0]->flags = binding::NOSELFCOLLISION | binding::LINEAR;
    meshes[1]->flags = binding::NOSELFCOLLISION | binding::LINEAR;
    meshes[0]->translate(Vec3(-3.0, 0.0, -1.0));
    meshes[1]->translate(Vec3(1.0, 0.0, -1.0));

    Solver solver;
    Lame very_soft_rubber(100, 0.1);
    binding::add_trimesh(&solver, meshes[1], very_soft_rubber, settings.verbose > 0);   // right sheet first, as the sample does
    very_soft_rubber.limit_min = 0.95;
    very_soft_rubber.limit_max = 1.05;
    binding::add_trimesh(&solver, meshes[0], very_soft_rubber, settings.verbose > 0);
    std::vector<int> pins;
    get_pins(*meshes[1], pins, 0);
    get_pins(*meshes[0], pins, (int)meshes[1]->vertices.size());
    solver.set_pins(pins);
    if (has_floor) solver.add_obstacle(std::make_shared<Floor>(floor_y));
    try {
        if (!solver.initialize(settings)) return EXIT_FAILURE;
        for (int f = 0; f < frames; ++f) {
            solver.step();
            const Solver::RuntimeData &rd = solver.runtime_data();
            if (settings.verbose > 0)
                printf("frame %d: local %.3f ms, global %.3f ms, collision %.3f ms, inner iters %d\n", f, rd.local_ms, rd.global_ms, rd.collision_ms, rd.inner_iters);
        }
    } catch (const std::exception &e) {
        std::cerr << "trianglestrain: " << e.what() << std::endl;
        return EXIT_FAILURE;
    }
    if (!out.empty()) {
        meshio::save_positions(out + ".xyz", solver.m_x);
        std::vector<Vec3i> all;
        for (Vec3i f : meshes[1]->faces) all.push_back(f);
        for (Vec3i f : meshes[0]->faces) { for (int c = 0; c < 3; ++c) f[c] += (int)meshes[1]->vertices.size(); all.push_back(f); }
        meshio::save_obj(out + ".obj", solver.m_x, all);
    }
    return EXIT_SUCCESS;
}
