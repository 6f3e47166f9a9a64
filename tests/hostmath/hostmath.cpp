// csrc/device_math.hpp compiled for the host (tests/test_device_math_host.py builds this with g++ -I tests/hostmath)
static long g_cnt[4];
#define ADMM_COUNT(slot) (++g_cnt[slot])
static double g_rec[32];
#define ADMM_RECORD(slot, value) (g_rec[slot] = (value))
#include "../../admm-elastic_amd/csrc/device_math.hpp"
using namespace admm_dev;
extern "C" {
void hm_counts(long *c) { for (int i = 0; i < 4; ++i) { c[i] = g_cnt[i]; g_cnt[i] = 0; } }
// per-matrix counts, for wave-level statistics (a wave runs as long as its slowest lane)
void hm_svd_counted(int n, const double *F, double *U, double *S, double *V, int *cnt, double *rec) {
    for (int i = 0; i < n; ++i) {
        for (int k = 0; k < 4; ++k) g_cnt[k] = 0;
        for (int k = 0; k < 32; ++k) g_rec[k] = -1.0;
        signed_svd3(F + 9 * i, U + 9 * i, S + 3 * i, V + 9 * i);
        for (int k = 0; k < 4; ++k) cnt[4 * i + k] = (int)g_cnt[k];
        for (int k = 0; k < 32; ++k) rec[32 * i + k] = g_rec[k];
    }
    for (int k = 0; k < 4; ++k) g_cnt[k] = 0;
}
// F, U, V: [n][9] column-major; S: [n][3]
void hm_svd(int n, const double *F, double *U, double *S, double *V) {
    for (int i = 0; i < n; ++i) signed_svd3(F + 9 * i, U + 9 * i, S + 3 * i, V + 9 * i);
}
void hm_prox(int kind, int n, double mu, double la, double k, double *S) {
    for (int i = 0; i < n; ++i) {
        if (kind == 1) prox_stretches<1>(mu, la, k, S + 3 * i);
        else if (kind == 2) prox_stretches<2>(mu, la, k, S + 3 * i);
        else if (kind == 3) prox_stretches<3>(mu, la, k, S + 3 * i);
        else if (kind == 0) prox_stretches<0>(mu, la, k, S + 3 * i);
    }
}
}
