// Host stand-in for <hip/hip_runtime.h>, ONLY for tests/hostmath (tests/test_device_math_host.py): compiles csrc/device_math.hpp with g++ so that the
// numerics of a new SVD / Newton variant can be studied on the CPU before it goes to the GPU.  The hardware approximations are
// emulated with their measured precision (experiments/hw_prec.hip: v_rcp/rsq/sqrt_f64 2^-24, f32 versions 2^-23.4).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#define __device__
#define __host__
#define __forceinline__ inline
static inline double hm_noise() {
    static uint64_t s = 0x9E3779B97F4A7C15ull;
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    return (double)(int64_t)(s >> 11) / (double)(1ll << 52) - 1.0;   // [-1, 1)
}
static inline double hm_rcp(double x) { return (1.0 / x) * (1.0 + 4.6e-8 * hm_noise()); }
static inline double hm_rsq(double x) { return (1.0 / std::sqrt(x)) * (1.0 + 5.2e-8 * hm_noise()); }
static inline float hm_rcpf(float x) { return (float)((1.0 / (double)x) * (1.0 + 6e-8 * hm_noise())); }
static inline float hm_rsqf(float x) { return (float)((1.0 / std::sqrt((double)x)) * (1.0 + 6e-8 * hm_noise())); }
static inline float hm_sqrtf(float x) { return (float)(std::sqrt((double)x) * (1.0 + 6e-8 * hm_noise())); }
#define __builtin_amdgcn_rcp(x) hm_rcp(x)
#define __builtin_amdgcn_rsq(x) hm_rsq(x)
#define __builtin_amdgcn_rcpf(x) hm_rcpf(x)
#define __builtin_amdgcn_rsqf(x) hm_rsqf(x)
#define __builtin_amdgcn_sqrtf(x) hm_sqrtf(x)
static inline float hm_logf(float x) { return (float)(std::log((double)x) + 3e-6 * hm_noise()); }
#define __logf(x) hm_logf(x)
static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline int __double2hiint(double x) { int64_t b; std::memcpy(&b, &x, 8); return (int)(b >> 32); }
static inline int __double2loint(double x) { int64_t b; std::memcpy(&b, &x, 8); return (int)(b & 0xffffffff); }
static inline double __hiloint2double(int hi, int lo) { int64_t b = ((int64_t)hi << 32) | (uint32_t)lo; double x; std::memcpy(&x, &b, 8); return x; }
// wave votes: the harness runs one lane at a time
static inline int __any(int p) { return p; }
static inline int __all(int p) { return p; }
using std::fma; using std::fabs; using std::copysign; using std::fmax; using std::fmin; using std::log; using std::sqrt;
