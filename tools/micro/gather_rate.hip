// Microbenchmark: cost of one divergent 16-byte gather instruction (global_load_dwordx4, 64 lanes) on gfx950 as a function of how the
// lanes' addresses group.  g = log2(lanes per group): the 2^g adjacent lanes of a group load consecutive 16-byte pieces of ONE record of
// 16 << g bytes; different groups load pseudo-random records of a table of `bytes` bytes.  g = 0: 64 unrelated records (the force
// kernel today); g = 1: lane pairs share a 32-byte record; g = 2: a hardware quad shares 64 bytes; g = 6: fully coalesced 1 KiB.
// Reports nanoseconds and core cycles (at 2.4 GHz) per gather instruction and CU with 8 waves per SIMD resident.
//   hipcc -O3 --offload-arch=gfx950 -o gather_rate gather_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int G>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ tab, uint32_t recMask, float* out, int iters) {
    const uint32_t lane = threadIdx.x & 63u, grp = lane >> G, sub = lane & ((1u << G) - 1u);
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    float acc = 0;
    uint32_t s = mix(wave * 64u + grp + 1u);
    for (int i = 0; i < iters; i++) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            s = s * 1664525u + 1013904223u;
            const uint32_t rec = (s >> 8) & recMask;
            v[u] = tab[((size_t)rec << G) + sub];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) acc += v[u].x + v[u].w;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int G>
void run(size_t bytes, int blocks_per_cu) {
    float4* tab; float* out;
    const int nb = 256 * blocks_per_cu;
    hipMalloc(&tab, bytes); hipMemset(tab, 0, bytes); hipMalloc(&out, nb * 256 * sizeof(float));
    const uint32_t nrec = (uint32_t)(bytes / (16u << G));  // power of two
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<G>, dim3(nb), dim3(256), 0, 0, tab, nrec - 1u, out, 50);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<G>, dim3(nb), dim3(256), 0, 0, tab, nrec - 1u, out, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double gathers_per_cu = (double)iters * 4 * 4 * blocks_per_cu;  // wave-instructions per CU
    const double ns = ms * 1e6 / gathers_per_cu;
    printf("table %8zu KiB  lanes/record %2d  waves/SIMD %d: %7.2f ns = %6.1f cycles per gather instruction and CU  (%.2f TB/s useful)\n", bytes >> 10, 1 << G,
           blocks_per_cu, ns, ns * 2.4, 256.0 * 1024.0 / ns * 1e-3);
    hipFree(tab); hipFree(out);
}
// coalesced loads of W dwords per lane (one wave instruction = 64 x W x 4 contiguous bytes), streaming through a table of `bytes`
template <int W> struct VecT;
template <> struct VecT<1> { typedef float T; };
template <> struct VecT<2> { typedef float2 T; };
template <> struct VecT<4> { typedef float4 T; };
template <int W>
__global__ __launch_bounds__(256) void kc(const typename VecT<W>::T* __restrict__ tab, uint32_t mask, float* out, int iters) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    float acc = 0;
    uint32_t pos = mix(wave + 1u);
    for (int i = 0; i < iters; i++) {
        typename VecT<W>::T v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { pos += 977u; v[u] = tab[(((size_t)(pos & mask)) << 6) + lane]; }
#pragma unroll
        for (int u = 0; u < 4; u++) acc += *reinterpret_cast<const float*>(&v[u]);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int W>
void runc(size_t bytes, int blocks_per_cu) {
    void* tab; float* out;
    const int nb = 256 * blocks_per_cu;
    hipMalloc(&tab, bytes); hipMemset(tab, 0, bytes); hipMalloc(&out, nb * 256 * sizeof(float));
    const uint32_t nchunk = (uint32_t)(bytes / (256u * W));
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kc<W>, dim3(nb), dim3(256), 0, 0, (const typename VecT<W>::T*)tab, nchunk - 1u, out, 50);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kc<W>, dim3(nb), dim3(256), 0, 0, (const typename VecT<W>::T*)tab, nchunk - 1u, out, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / ((double)iters * 4 * 4 * blocks_per_cu);
    printf("coalesced, table %8zu KiB, %d dword(s) per lane, waves/SIMD %d: %7.2f ns per load instruction and CU  (%.2f TB/s)\n", bytes >> 10, W, blocks_per_cu, ns,
           256.0 * 256.0 * W / ns * 1e-3);
    hipFree(tab); hipFree(out);
}
int main() {
    for (size_t bytes : {(size_t)16 << 10, (size_t)2 << 20, (size_t)256 << 20}) { runc<1>(bytes, 8); runc<2>(bytes, 8); runc<4>(bytes, 8); }
    for (size_t bytes : {(size_t)16 << 10, (size_t)2 << 20, (size_t)64 << 20, (size_t)1 << 30}) {
        for (int b : {4, 8}) {
            run<0>(bytes, b); run<1>(bytes, b); run<2>(bytes, b); run<3>(bytes, b); run<6>(bytes, b);
        }
    }
    return 0;
}
