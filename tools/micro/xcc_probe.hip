// Which XCD does workgroup b of a launch land on?  The tile mappings (tile_index, sph_kernels.h) assume b mod 8.
// Prints, for several launches of different grid sizes, the histogram of (XCC_ID - b) mod 8 over the workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_probe(unsigned* out) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) out[blockIdx.x] = x & 0xF;
}
__global__ void k_spin(unsigned* out, int n) {  // some unrelated work between the probes
    unsigned a = threadIdx.x;
    for (int i = 0; i < n; i++) a = a * 1664525u + 1013904223u;
    if (a == 12345u) out[0] = a;
}
int main() {
    unsigned* d;
    hipMalloc(&d, 1 << 20);
    const int grids[] = {4824, 4824, 4821, 4824, 19294, 4824, 1207, 4824, 4824};
    for (int g : grids) {
        hipLaunchKernelGGL(k_probe, dim3(g), dim3(256), 0, 0, d);
        std::vector<unsigned> h(g);
        hipMemcpy(h.data(), d, g * 4, hipMemcpyDeviceToHost);
        int hist[8] = {0};
        for (int b = 0; b < g; b++) hist[(h[b] + 8 - (b & 7)) & 7]++;
        printf("grid %5d: (xcc - b) mod 8 histogram:", g);
        for (int k = 0; k < 8; k++) printf(" %d", hist[k]);
        printf("   first blocks:");
        for (int b = 0; b < 10; b++) printf(" %u", h[b]);
        printf("\n");
        hipLaunchKernelGGL(k_spin, dim3(g % 977 + 13), dim3(256), 0, 0, d + 100000, 1000);
    }
    return 0;
}
