// sync_latency.hip -- measured basis for the persistent multi-colour GS kernel: what does ONE neighbour hand-off cost
// on gfx950, by cache scope, by XCD placement, and by protocol (data + flag vs data that carries its own stamp)?
//   hipcc --offload-arch=gfx950 -O3 -o experiments/_build/sync_latency experiments/sync_latency.hip && experiments/_build/sync_latency
// Every spin is bounded; a run that cannot complete prints ABORT instead of hanging the GPU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
constexpr unsigned kSpin = 2000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7fffffff, 0x00020000); }

template <int AUX> __device__ __forceinline__ void st16(__amdgpu_buffer_rsrc_t rs, int off, double v, unsigned long long tag) {
    union { struct { double d; unsigned long long t; } s; v4u v; } u; u.s.d = v; u.s.t = tag;
    __builtin_amdgcn_raw_buffer_store_b128(u.v, rs, off, 0, AUX);
}
template <int AUX> __device__ __forceinline__ void ld16(__amdgpu_buffer_rsrc_t rs, int off, double &v, unsigned long long &tag) {
    union { struct { double d; unsigned long long t; } s; v4u v; } u;
    asm volatile("" ::: "memory");      // a polling load: must not be hoisted out of its loop
    u.v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, AUX);
    v = u.s.d; tag = u.s.t;
}

__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }

// ---- (1) ping-pong of ONE tagged 16-byte granule between block A and block B; AUX = cache policy of loads and stores ----
template <int AUX>
__global__ __launch_bounds__(64) void k_pingpong(int blkA, int blkB, int rounds, double *box /* 2 x 16 B, 256 B apart */, unsigned long long *out, unsigned *xcc) {
    const int b = blockIdx.x;
    if (threadIdx.x == 0) xcc[b] = xcc_id();
    if (b != blkA && b != blkB) return;
    if (threadIdx.x != 0) return;
    __amdgpu_buffer_rsrc_t rs = rsrc(box);
    const int mine = b == blkA ? 0 : 256, other = b == blkA ? 256 : 0;
    const unsigned long long t0 = wall_clock64();
    double v = 1.0; unsigned long long tag;
    bool ok = true;
    for (int i = 1; i <= rounds && ok; ++i) {
        if (b == blkA) {
            st16<AUX>(rs, mine, v, (unsigned long long)i);
            unsigned s = 0; double g;
            do { ld16<AUX>(rs, other, g, tag); } while (tag < (unsigned long long)i && ++s < kSpin);
            ok = s < kSpin; v = g + 1.0;
        } else {
            unsigned s = 0; double g;
            do { ld16<AUX>(rs, other, g, tag); } while (tag < (unsigned long long)i && ++s < kSpin);
            ok = s < kSpin; v = g + 1.0;
            st16<AUX>(rs, mine, v, (unsigned long long)i);
        }
    }
    const unsigned long long t1 = wall_clock64();
    if (b == blkA) { out[0] = t1 - t0; out[1] = ok ? 1 : 0; }
}

// ---- (2) phase exchange at scale: G blocks on a PX x PY grid of patches, every block has <= 8 neighbours; per phase every thread
// stores `ent` tagged granules into its block's outbox and polls `ent` granules of the neighbours' outboxes (thread t reads slot t of
// neighbour t % nnb).  PROTO 0: tagged data (one store, poll the data).  PROTO 1: data, drain, flag; poll the flag, then load. ----
template <int AUX, int PROTO>
__global__ __launch_bounds__(256) void k_phases(int PX, int PY, int xcd_tiles, int phases, int ent, double *box /* [G][ent*256] granules */,
                                                unsigned long long *flags, unsigned long long *out, int *abortw, int stride) {
    // stride = 8: only the blocks with blockIdx % 8 == 0 take part -- all of them on ONE XCD (round-robin dispatch)
    if ((int)blockIdx.x % stride) return;
    const int b = (int)blockIdx.x / stride, G = PX * PY;
    if (b >= G) return;
    // block -> patch: XCD-aware (xcd_tiles = 1): XCD x = b % 8 owns a compact tile of the patch grid; 0: row-major by block index
    int px, py;
    if (xcd_tiles) {
        const int x = b & 7, k = b >> 3;                 // XCD, index inside the XCD
        const int TX = PX / 4, TY = PY / 2;              // tile of one XCD (4 x 2 tiles)
        px = (x & 3) * TX + k % TX; py = (x >> 2) * TY + k / TX;
    } else { px = b % PX; py = b / PX; }
    auto blk_of = [&](int qx, int qy) -> int {
        if (!xcd_tiles) return qy * PX + qx;
        const int TX = PX / 4, TY = PY / 2;
        const int x = (qx / TX) + 4 * (qy / TY), k = (qy % TY) * TX + (qx % TX);
        return k * 8 + x;
    };
    __shared__ int nb[8]; __shared__ int nnb_s; __shared__ int ok_s;
    if (threadIdx.x == 0) {
        int n = 0;
        for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) {
            if (!dx && !dy) continue;
            const int qx = px + dx, qy = py + dy;
            if (qx < 0 || qy < 0 || qx >= PX || qy >= PY) continue;
            nb[n++] = blk_of(qx, qy);
        }
        nnb_s = n; ok_s = 1;
    }
    __syncthreads();
    const int nnb = nnb_s;
    __amdgpu_buffer_rsrc_t rs = rsrc(box);
    const int t = threadIdx.x;
    const int src_blk = nb[t % nnb];
    double acc = (double)b;
    const unsigned long long t0 = wall_clock64();
    bool ok = true;
    for (int p = 1; p <= phases && ok; ++p) {
        for (int e = 0; e < ent; ++e) st16<AUX>(rs, ((b * ent + e) * 256 + t) * 16, acc + e, (unsigned long long)p);
        if (PROTO == 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0) __hip_atomic_store(flags + 8 * b, (unsigned long long)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t < 64) {
                const int q = t < nnb ? nb[t] : -1;
                unsigned s = 0;
                while (true) {
                    const unsigned long long v = q >= 0 ? __hip_atomic_load(flags + 8 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (unsigned long long)p;
                    if (__all(v >= (unsigned long long)p)) break;
                    if (++s > kSpin) { ok_s = 0; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
            ok = ok_s != 0;
        }
        double sum = 0.0;
        for (int e = 0; e < ent && ok; ++e) {
            double g; unsigned long long tag; unsigned s = 0;
            do { ld16<AUX>(rs, ((src_blk * ent + e) * 256 + t) * 16, g, tag); } while (tag < (unsigned long long)p && ++s < kSpin);
            if (s >= kSpin) ok = false;
            sum += g;
        }
        acc = sum * 1e-3 + 1.0;
        if (PROTO == 0) {     // a block-level decision like the GS kernel's (everyone has its halo): one LDS barrier
            if (!ok) ok_s = 0;
            __syncthreads();
            ok = ok_s != 0;
        }
    }
    const unsigned long long t1 = wall_clock64();
    if (t == 0) { out[b] = t1 - t0; if (!ok) *abortw = 1; }
    if (acc == 12345.678) box[0] = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

int main() {
    double *box; unsigned long long *out, *flags; unsigned *xcc; int *abortw;
    const size_t box_bytes = (size_t)256 * 8 * 256 * 16;
    CK(hipMalloc(&box, box_bytes)); CK(hipMalloc(&out, 256 * 8)); CK(hipMalloc(&flags, 256 * 64)); CK(hipMalloc(&xcc, 256 * 4)); CK(hipMalloc(&abortw, 4));
    int khz = 0; CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0));
    const double us_per_tick = 1e3 / (double)khz;
    std::vector<unsigned> hx(256);
    const int rounds = 2000;
    struct { const char *name; int aux; } pol[] = {{"sc1 (agent, write-through)", 16}, {"sc0+sc1 (system)", 17}, {"sc0 (workgroup: L2 of the XCD)", 1}, {"none (wave)", 0}};
    for (auto &p : pol) {
        for (int pair = 0; pair < 2; ++pair) {
            const int A = 0, B = pair == 0 ? 8 : 1;      // same XCD (0, 8) / neighbouring XCDs (0, 1)
            CK(hipMemset(box, 0, 1024)); CK(hipMemset(out, 0, 16));
            switch (p.aux) {
            case 16: hipLaunchKernelGGL(k_pingpong<16>, dim3(256), dim3(64), 0, 0, A, B, rounds, box, out, xcc); break;
            case 17: hipLaunchKernelGGL(k_pingpong<17>, dim3(256), dim3(64), 0, 0, A, B, rounds, box, out, xcc); break;
            case 1: hipLaunchKernelGGL(k_pingpong<1>, dim3(256), dim3(64), 0, 0, A, B, rounds, box, out, xcc); break;
            default: hipLaunchKernelGGL(k_pingpong<0>, dim3(256), dim3(64), 0, 0, A, B, rounds, box, out, xcc); break;
            }
            CK(hipDeviceSynchronize());
            unsigned long long h[2]; CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hx.data(), xcc, 256 * 4, hipMemcpyDeviceToHost));
            printf("pingpong %-34s blocks %d,%d (xcc %u,%u): one way %.3f us %s\n", p.name, A, B, hx[A], hx[B], h[0] * us_per_tick / (2.0 * rounds), h[1] ? "" : "ABORT (stale data never seen)");
        }
    }
    printf("xcc of blocks 0..15:");
    for (int i = 0; i < 16; ++i) printf(" %u", hx[i]);
    printf("\n");
    bool rr = true;
    for (int i = 0; i < 256; ++i) rr = rr && hx[i] == hx[i & 7];
    printf("block b runs on the XCD of block b %% 8: %s\n", rr ? "yes" : "NO");
    const int phases = 2000;
    for (int G : {32, 64, 256}) {
        const int PX = G == 32 ? 8 : G == 64 ? 8 : 16, PY = G / PX;
        for (int ent : {1, 3, 6}) {
            for (int tiles = 0; tiles < 2; ++tiles) {
                if (tiles && (PX % 4 || PY % 2 || G < 64)) continue;
                for (int proto = 0; proto < 2; ++proto) {
                    for (int aux : {16, 1, 0}) {
                        // aux 1 / 0 (loads and stores that may be served by the XCD's L2): only coherent when every block runs on ONE XCD:
                        // G = 32 launched as 256 blocks of which those with blockIdx % 8 == 0 take part
                        if (aux != 16 && !(G == 32 && proto == 0)) continue;
                        const int stride = aux != 16 ? 8 : 1;
                        for (int one_xcd = 0; one_xcd < (aux == 16 && G == 32 ? 2 : 1); ++one_xcd) {
                        const int st = one_xcd ? 8 : stride;
                        CK(hipMemset(box, 0, box_bytes)); CK(hipMemset(flags, 0, 256 * 64)); CK(hipMemset(out, 0, 256 * 8)); CK(hipMemset(abortw, 0, 4));
                        if (proto == 0 && aux == 16) hipLaunchKernelGGL((k_phases<16, 0>), dim3(G * st), dim3(256), 0, 0, PX, PY, tiles, phases, ent, box, flags, out, abortw, st);
                        if (proto == 1 && aux == 16) hipLaunchKernelGGL((k_phases<16, 1>), dim3(G * st), dim3(256), 0, 0, PX, PY, tiles, phases, ent, box, flags, out, abortw, st);
                        if (proto == 0 && aux == 1) hipLaunchKernelGGL((k_phases<1, 0>), dim3(G * st), dim3(256), 0, 0, PX, PY, tiles, phases, ent, box, flags, out, abortw, st);
                        if (proto == 0 && aux == 0) hipLaunchKernelGGL((k_phases<0, 0>), dim3(G * st), dim3(256), 0, 0, PX, PY, tiles, phases, ent, box, flags, out, abortw, st);
                        CK(hipDeviceSynchronize());
                        std::vector<unsigned long long> h(256); int ab = 0;
                        CK(hipMemcpy(h.data(), out, 256 * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&ab, abortw, 4, hipMemcpyDeviceToHost));
                        unsigned long long mx = 0; for (int i = 0; i < G; ++i) mx = h[i] > mx ? h[i] : mx;
                        printf("phases G=%3d ent=%d/thread %-12s %-22s aux=%2d %s: %.3f us per phase%s\n", G, ent, tiles ? "xcd-tiles" : "row-major",
                               proto ? "data+drain+flag+poll" : "tagged data", aux, st == 8 ? "ONE XCD" : "all XCDs", mx * us_per_tick / phases, ab ? "  ABORT" : "");
                        }
                    }
                }
            }
        }
    }
    return 0;
}
