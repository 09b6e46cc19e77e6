// Dependent-load latency of ONE wavefront over a buffer of a given size (pointer chase with a random stride pattern), and the same
// with 16 independent chains per lane group -- a yardstick for the per-hop cost inside k_fused (tools/probe is not part of the library).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
__global__ void chase(const uint32_t *next, uint32_t start, uint32_t hops, uint32_t *out, unsigned long long *ticks)
{
    uint32_t p = start + threadIdx.x * 0u;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (uint32_t i = 0; i < hops; i++) p = __builtin_nontemporal_load(next + (size_t)p * 768u);   // one dword per 3 KB row
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { *out = p; *ticks = t1 - t0; }
}
int main(int argc, char **argv)
{
    const size_t rows_list[] = {20000, 200000, 1000000, 10000000};
    for (size_t rows : rows_list) {
        uint32_t *d; const size_t bytes = rows * 3072;
        if (hipMalloc((void **)&d, bytes) != hipSuccess) { printf("alloc %zu failed\n", bytes); continue; }
        std::vector<uint32_t> perm(rows); for (size_t i = 0; i < rows; i++) perm[i] = (uint32_t)i;
        std::mt19937 g(1); for (size_t i = rows - 1; i > 0; i--) std::swap(perm[i], perm[g() % (i + 1)]);
        std::vector<uint32_t> nx(rows); for (size_t i = 0; i < rows; i++) nx[perm[i]] = perm[(i + 1) % rows];
        // scatter the `next` pointers: element r*768 of the buffer
        // only the first dword of each 3 KB row is used: a strided 2-D copy places them
        hipMemcpy2D(d, 3072, nx.data(), 4, 4, rows, hipMemcpyHostToDevice);
        uint32_t *out; unsigned long long *tk; hipMalloc((void **)&out, 4); hipMalloc((void **)&tk, 8);
        const uint32_t hops = 2000;
        hipLaunchKernelGGL(chase, dim3(1), dim3(64), 0, 0, d, 0u, hops, out, tk);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(chase, dim3(1), dim3(64), 0, 0, d, 5u, hops, out, tk);
        hipDeviceSynchronize();
        unsigned long long t; hipMemcpy(&t, tk, 8, hipMemcpyDeviceToHost);
        printf("rows %zu (%.1f GB): %.0f ns per dependent hop (100 MHz realtime clock)\n", rows, bytes / 1e9, (double)t * 10.0 / hops);
        hipFree(d); hipFree(out); hipFree(tk);
    }
    return 0;
}
