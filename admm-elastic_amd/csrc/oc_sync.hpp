// oc_sync.hpp -- the synchronisation primitives of the persistent on-chip PCG kernel (pcg_onchip2.hpp: k_pcg2, k_sync_probe).
//   * vectors and records that other blocks read are written with 16-byte WRITE-THROUGH (sc1) stores and read with sc1 loads:
//     the per-XCD L2s are not coherent with each other;
//   * grid barrier: eight monotonic counters (one per group of blocks = the XCD a block runs on, by observation; correctness
//     does not depend on it), ONE fire-and-forget arrival per block, relaxed agent-scope polling, BOUNDED spins -- a barrier
//     that cannot complete aborts the solve with an error instead of hanging the GPU; two counter sets alternate between
//     solves so nothing is cleared between launches; the barrier is split into arrive and wait so that work can be placed
//     inside its latency;
//   * neighbour hand-off: a block announces its published slice with a (solve, phase) flag and a consumer starts gathering
//     as soon as the <= 64 blocks its matrix rows reference have announced theirs -- no grid barrier in front of a gather.
// Measured alternatives (two-level barrier, tagged granules instead of barrier + records, an auxiliary reduction wave):
// profiles/HISTORY_rounds_1_3.md.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels.hpp"

namespace admm_k {

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
// LDS-qualified element types: pointers built by arithmetic on the dynamic LDS base otherwise decay to generic
// (flat) pointers and every matrix access becomes a flat_load instead of a ds_read
typedef __attribute__((address_space(3))) double LdsD;


constexpr int kOcSubK = 4;                // aggregates per block of the two-level preconditioner (= admm_host::kOcSub, pcg_onchip2.hpp)
constexpr unsigned kOcSpinLimit = 4000000u;
constexpr double kOcPipeFloor = 1e-18;    // squared relative residual below which the pipelined recurrences are not trusted
#ifndef ADMM_OC_TRIG
#define ADMM_OC_TRIG 0.9
#endif
constexpr double kOcTrig = ADMM_OC_TRIG;  // the recurrence must report gamma <= kOcTrig tol^2 b.M^-1 b before the true residual is checked
constexpr int kOcStagnation = 50;         // pipelined iterations without a new residual minimum before switching, once the
                                          // residual is within 100x of kOcPipeFloor (rounding-driven stagnation); above that
                                          // level plateaus of the residual norm are ordinary CG behaviour (measured: 13 of 40
                                          // solves of the 1M-tet bench plateau for > 50 iterations) and the window is 4x longer

__device__ __forceinline__ void oc_store_sc1(__amdgpu_buffer_rsrc_t rs, int byte_off, double a, double b) {
    union { double d[2]; v4u v; } t; t.d[0] = a; t.d[1] = b;
    __builtin_amdgcn_raw_buffer_store_b128(t.v, rs, byte_off, 0, 16 /* sc1: write-through */);
}
__device__ __forceinline__ void oc_store_sc1(__amdgpu_buffer_rsrc_t rs, int byte_off, double a) {
    union { double d; v2u v; } t; t.d = a;
    __builtin_amdgcn_raw_buffer_store_b64(t.v, rs, byte_off, 0, 16);
}
// published vectors are stored per axis (SoA): a wave's gather of one neighbour column is then three fully
// coalesced 512-byte requests (12 cache lines) instead of 32 lines of a padded 32-byte AoS record
__device__ __forceinline__ void oc_load_sc1(__amdgpu_buffer_rsrc_t rs, int byte_off, int axis_stride, double *g) {
    union { double d; v2u v; } t0, t1, t2;
    t0.v = __builtin_amdgcn_raw_buffer_load_b64(rs, byte_off, 0, 16);
    t1.v = __builtin_amdgcn_raw_buffer_load_b64(rs, byte_off + axis_stride, 0, 16);
    t2.v = __builtin_amdgcn_raw_buffer_load_b64(rs, byte_off + 2 * axis_stride, 0, 16);
    g[0] = t0.d; g[1] = t1.d; g[2] = t2.d;
}
__device__ __forceinline__ double oc_load_sc1_f64(__amdgpu_buffer_rsrc_t rs, int byte_off) {
    union { double d; v2u v; } t;
    t.v = __builtin_amdgcn_raw_buffer_load_b64(rs, byte_off, 0, 16);
    return t.d;
}

// Grid barrier: every payload store before it was a write-through (sc1) store, so no release fence is
// needed -- every wave drains its stores, one lane arrives.  Eight monotonic counters, one per group
// (blockIdx & 7 = the XCD the block runs on, by observation; correctness does not depend on it): a block
// arrives with ONE non-returning atomic on its group's counter (fire and forget: 32 arrivals per word, no
// second level to wait for) and lanes 0..7 of wave 0 poll the eight counters with relaxed agent-scope loads
// until each has reached (blocks in the group) x epoch.  Measured against the two-level form (per-group
// counter -> top counter -> generation word): see DESIGN.md.
__device__ __forceinline__ bool oc_barrier(unsigned *bar, unsigned epoch, int G, int *ok_lds, int *sig) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x < 64) {
        const int lane = (int)threadIdx.x;
        if (lane == 0) __hip_atomic_fetch_add(bar + 16 * ((int)blockIdx.x & 7), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int x = lane & 7;
        const unsigned need = (unsigned)((G + 7 - x) >> 3) * epoch;      // 0 for groups without blocks
        unsigned *word = bar + 16 * (lane < 8 ? x : 17);                  // lane 8 watches the abort word in the same load
        int ok = 1;
        unsigned spins = 0;
        while (true) {
            const unsigned v = (lane < 9) ? __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            if (__all(lane >= 8 || v >= need)) break;
            if (++spins > kOcSpinLimit || __any(lane == 8 && v != 0u)) {
                if (lane == 0) {
                    __hip_atomic_store(bar + 16 * 17, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(sig + 2, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (lane == 0) *ok_lds = ok;
    }
    __syncthreads();
    return *ok_lds != 0;
}

// Pipelined iteration, first half of the synchronisation: drain, announce (flag for the neighbours + arrival on the
// grid barrier, both fire-and-forget), then wait only for the blocks this block gathers from.
// (ARRIVE = false: flag and wait only -- pcg_onchip2.hpp crosses its grid barrier after the gather)
template <bool ARRIVE = true>
__device__ __forceinline__ bool oc_announce_and_wait_neighbours(unsigned *bar, unsigned long long *flags, const int *nbr, unsigned seq, unsigned epoch,
                                                                int *ok_lds, int *sig) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x < 64) {
        const int lane = (int)threadIdx.x;
        const unsigned long long tag = ((unsigned long long)seq << 32) | epoch;
        if (lane == 0) {
            __hip_atomic_store(flags + 8 * blockIdx.x, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ARRIVE) __hip_atomic_fetch_add(bar + 16 * ((int)blockIdx.x & 7), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const int nb = nbr[64 * blockIdx.x + lane];
        int ok = 1;
        unsigned spins = 0;
        while (true) {
            const unsigned long long v = (nb >= 0) ? __hip_atomic_load(flags + 8 * nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tag;
            if (__all(v >= tag)) break;
            if (++spins > kOcSpinLimit || ((spins & 255u) == 0u && __hip_atomic_load(bar + 16 * 17, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                if (lane == 0) {
                    __hip_atomic_store(bar + 16 * 17, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(sig + 2, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (lane == 0) *ok_lds = ok;
    }
    __syncthreads();
    return *ok_lds != 0;
}
// ... second half: the grid barrier this block has already arrived at (wave 0 polls the eight counters)
__device__ __forceinline__ bool oc_barrier_wait(unsigned *bar, unsigned epoch, int G, int *ok_lds, int *sig) {
    if (threadIdx.x < 64) {
        const int lane = (int)threadIdx.x;
        const int x = lane & 7;
        const unsigned need = (unsigned)((G + 7 - x) >> 3) * epoch;
        unsigned *word = bar + 16 * (lane < 8 ? x : 17);
        int ok = 1;
        unsigned spins = 0;
        while (true) {
            const unsigned v = (lane < 9) ? __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            if (__all(lane >= 8 || v >= need)) break;
            if (++spins > kOcSpinLimit || __any(lane == 8 && v != 0u)) {
                if (lane == 0) {
                    __hip_atomic_store(bar + 16 * 17, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(sig + 2, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (lane == 0) *ok_lds = ok;
    }
    __syncthreads();
    return *ok_lds != 0;
}

} // namespace admm_k
