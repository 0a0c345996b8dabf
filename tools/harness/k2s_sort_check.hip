#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../deal-yolo-daya_amd/csrc/k2_filter.h"
using namespace dyd;
template <int E>
__global__ void sort_kernel(const uint32_t *in, uint32_t *out) {
    const int lane = threadIdx.x & 63;
    uint32_t v[E];
    for (int r = 0; r < E; ++r) v[r] = in[blockIdx.x * 64 * E + lane + 64 * r];
    k2s_sort_regs<E>(v, lane);
    for (int r = 0; r < E; ++r) out[blockIdx.x * 64 * E + lane * E + r] = v[r];
}
template <int E>
int run() {
    const int nb = 1000, P = 64 * E;
    std::vector<uint32_t> h(nb * P), o(nb * P);
    for (auto &x : h) x = (uint32_t)rand() * 2654435761u;
    uint32_t *di, *dout;
    hipMalloc(&di, 4 * h.size()); hipMalloc(&dout, 4 * h.size());
    hipMemcpy(di, h.data(), 4 * h.size(), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(sort_kernel<E>, dim3(nb), dim3(64), 0, 0, di, dout);
    hipMemcpy(o.data(), dout, 4 * h.size(), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < nb; ++b) {
        std::sort(h.begin() + b * P, h.begin() + (b + 1) * P);
        if (!std::equal(h.begin() + b * P, h.begin() + (b + 1) * P, o.begin() + b * P)) ++bad;
    }
    printf("E=%d blocks=%d not sorted=%d\n", E, nb, bad);
    if (bad) { for (int i = 0; i < 16; ++i) printf("%08x %08x\n", h[i], o[i]); }
    return bad;
}
int main() { return run<1>() + run<2>() + run<4>(); }
