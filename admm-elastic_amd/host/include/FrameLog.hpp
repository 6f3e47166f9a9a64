// FrameLog.hpp -- per-frame output of the headless samples for offline comparison.  The reference prints
// Solver::RuntimeData per step (src/Solver.cpp:309-319) and can only save its matrix (Solver::save_matrix) and screenshots;
// a headless run needs the numbers and the geometry instead:
//   --csv FILE       one line per frame: frame, step_ms (wall clock of Solver::step), local_ms, global_ms, collision_ms,
//                    inner_iters, admm_iters  (the RuntimeData fields, src/Solver.hpp:54-61)
//   --out-every K    every K-th frame (and the last one) as <prefix>_%05d.xyz (positions, full precision) and, when the
//                    caller has surface triangles, <prefix>_%05d.obj;  <prefix> = the sample's --out
#ifndef ADMM_FRAMELOG_HPP
#define ADMM_FRAMELOG_HPP 1
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "Meshes.hpp"
#include "Solver.hpp"

namespace admm {

class FrameLog {
public:
    std::string csv_path, out_prefix;
    int out_every = 0;
    // takes its own flags out of argv (returns the remaining arguments, argv[0] kept)
    std::vector<char *> parse(int argc, char **argv) {
        std::vector<char *> rest = {argv[0]};
        for (int i = 1; i < argc; ++i) {
            if (!strcmp(argv[i], "--csv") && i + 1 < argc) csv_path = argv[++i];
            else if (!strcmp(argv[i], "--out-every") && i + 1 < argc) out_every = atoi(argv[++i]);
            else if (!strcmp(argv[i], "--out") && i + 1 < argc) out_prefix = argv[++i];
            else rest.push_back(argv[i]);
        }
        return rest;
    }
    ~FrameLog() { if (csv) fclose(csv); }
    // one Solver::step() with its wall clock
    void step(Solver &solver) {
        const auto t0 = std::chrono::steady_clock::now();
        solver.step();
        last_step_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    void frame(int f, int n_frames, Solver &solver, const std::vector<Vec3i> &faces = std::vector<Vec3i>()) {
        Solver::RuntimeData rd = solver.runtime_data();
        if (!csv_path.empty()) {
            if (!csv) {
                csv = fopen(csv_path.c_str(), "w");
                if (!csv) throw std::runtime_error("FrameLog: cannot write " + csv_path);
                fprintf(csv, "frame,step_ms,local_ms,global_ms,collision_ms,inner_iters,admm_iters\n");
            }
            fprintf(csv, "%d,%.6f,%.6f,%.6f,%.6f,%d,%d\n", f, last_step_ms, rd.local_ms, rd.global_ms, rd.collision_ms, rd.inner_iters,
                    solver.settings().admm_iters);
            fflush(csv);
        }
        if (!out_prefix.empty() && out_every > 0 && (f % out_every == 0 || f == n_frames - 1)) {
            char name[64];
            snprintf(name, sizeof(name), "_%05d", f);
            meshio::save_positions(out_prefix + name + ".xyz", solver.m_x);
            if (!faces.empty()) meshio::save_obj(out_prefix + name + ".obj", solver.m_x, faces);
        }
    }
private:
    FILE *csv = nullptr;
    double last_step_ms = 0.0;
};

} // namespace admm
#endif
