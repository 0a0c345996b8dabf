#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define K2S_DEBUG 1
#include "../../deal-yolo-daya_amd/csrc/k2_filter.h"
using namespace dyd;
template <bool WANT_MAX>
__global__ __launch_bounds__(256) void rows_kernel(const double *box4, const int32_t *row_off, int64_t n_rows, double thr, uint8_t *out_high, double *out_max,
                                                   unsigned long long *dbg) {
    __shared__ WaveLdsF<8, 256> s_all[4];
    g_k2s_dbg = dbg;
    const int wave = threadIdx.x >> 6;
    const int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * 8;
    if (r0 >= n_rows) return;
    const int nr = (n_rows - r0 < 8) ? (int)(n_rows - r0) : 8;
    k2f_wave_rows<WANT_MAX, 8, 256>(box4, row_off, r0, nr, 2, thr, out_high, out_max, s_all[wave], nullptr);
}
static uint32_t ordf(float f) { uint32_t b; memcpy(&b, &f, 4); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
static float below(double v) { float f = (float)v; if ((double)f > v) f = nextafterf(f, -INFINITY); return f; }
static float above(double v) { float f = (float)v; if ((double)f < v) f = nextafterf(f, INFINITY); return f; }
int main(int argc, char **argv) {
    const int n = 256, rows = argc > 1 ? atoi(argv[1]) : 64;
    srand(7);
    std::vector<double> b(4 * (size_t)n * rows);
    std::vector<int32_t> off(rows + 1);
    for (int r = 0; r <= rows; ++r) off[r] = r * n;
    for (size_t i = 0; i < (size_t)n * rows; ++i) {
        double cx = rand() / (double)RAND_MAX * 1920, cy = rand() / (double)RAND_MAX * 1080;
        double w = 40 + rand() / (double)RAND_MAX * 60, h = 40 + rand() / (double)RAND_MAX * 60;
        b[4 * i] = round((cx - w / 2) * 100) / 100; b[4 * i + 1] = round((cy - h / 2) * 100) / 100;
        b[4 * i + 2] = round((cx + w / 2) * 100) / 100; b[4 * i + 3] = round((cy + h / 2) * 100) / 100;
    }
    double *d, *dmax; int32_t *doff; unsigned long long *dbg; uint8_t *high;
    hipMalloc(&d, 8 * b.size()); hipMalloc(&doff, 4 * off.size()); hipMalloc(&dbg, 128); hipMalloc(&high, rows); hipMalloc(&dmax, 8 * rows);
    hipMemcpy(d, b.data(), 8 * b.size(), hipMemcpyHostToDevice); hipMemcpy(doff, off.data(), 4 * off.size(), hipMemcpyHostToDevice);
    for (int mode = 0; mode < 4; ++mode) {
        const double thr = mode == 1 ? 0.9 : (mode == 2 ? 1.5 : 0.98);
        const bool want_max = mode == 3;
        hipMemset(dbg, 0, 128);
        const int blocks = (rows + 31) / 32;
        if (want_max) hipLaunchKernelGGL(rows_kernel<true>, dim3(blocks), dim3(256), 0, 0, d, doff, (int64_t)rows, thr, high, dmax, dbg);
        else hipLaunchKernelGGL(rows_kernel<false>, dim3(blocks), dim3(256), 0, 0, d, doff, (int64_t)rows, thr, high, (double *)nullptr, dbg);
        unsigned long long c[16];
        hipMemcpy(c, dbg, 128, hipMemcpyDeviceToHost);
        const double tl = want_max ? 0.0 : thr * 0.999;
        long its = 0, cand = 0;
        for (int r = 0; r < rows; ++r) {
            const double *bb = b.data() + 4 * (size_t)n * r;
            std::vector<uint32_t> key(n), lim(n); std::vector<float> y1(n), y2(n);
            for (int k = 0; k < n; ++k) {
                double x1 = bb[4 * k], x2 = bb[4 * k + 2];
                key[k] = (ordf(below(x1)) & ~0xffu) | k;
                lim[k] = (ordf(above(x2 - tl * (x2 - x1))) + 256u) | 0xffu;
                y1[k] = below(bb[4 * k + 1]); y2[k] = above(bb[4 * k + 3]);
            }
            std::vector<uint32_t> s = key; std::sort(s.begin(), s.end());
            for (int p0 = 0; p0 < n - 1; p0 += 64) {
                int maxd = 0;
                for (int p = p0; p < std::min(p0 + 64, n - 1); ++p) {
                    int ia = s[p] & 0xff, dd = 1;
                    while (p + dd < n && s[p + dd] <= lim[ia]) { int ib = s[p + dd] & 0xff; cand += y2[ia] > y1[ib] && y2[ib] > y1[ia]; ++dd; }
                    maxd = std::max(maxd, dd);
                }
                its += maxd;
            }
        }
        printf("rows=%d thr=%.3f max=%d  gpu: swept=%llu bad=%llu iterations=%llu candidates=%llu drains=%llu | host: iterations=%ld candidates=%ld\n", rows, thr,
               (int)want_max, c[0], c[1], c[2], c[3], c[4], its, cand);
    }
    return 0;
}
