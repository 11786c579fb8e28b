// Microbenchmark: sustained VALU issue rate on gfx950 for the instruction kinds the SPH pair loops use.
// Round 2: modes 0 / 1 (C++ mul / add / fma on eight independent floats) are auto-packed by the compiler into v_pk_* instructions, which is
// why round 1 read "2.5 cycles per operation"; modes 6 / 7 pin the instruction with inline asm.  Measured (cycles per wave64 instruction and
// SIMD at 1 / 4 / 8 waves per SIMD, loop overhead included): v_fma_f32 7.8 / 3.4 / 2.9, v_pk_fma_f32 11.8 / 6.0 / 5.6 -- the SIMD-32 retires a
// scalar FP32 instruction in 2 cycles and a packed one in 4 once several waves are ready; packing saves issues, not arithmetic time.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    double d0 = x0, d1 = x1;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {  // 16 independent-ish mul/add (no fma)
            x0 = x0 * a; x1 = x1 * a; x2 = x2 * a; x3 = x3 * a; x4 = x4 * a; x5 = x5 * a; x6 = x6 * a; x7 = x7 * a;
            x0 = x0 + b; x1 = x1 + b; x2 = x2 + b; x3 = x3 + b; x4 = x4 + b; x5 = x5 + b; x6 = x6 + b; x7 = x7 + b;
        } else if (MODE == 1) {  // 16 fma
            x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
            x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
            x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
            x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
        } else if (MODE == 2) {  // 2 float divisions
            x0 = x0 / (x1 + 2.0f); x2 = x2 / (x3 + 2.0f);
        } else if (MODE == 3) {  // 2 double divisions
            d0 = d0 / (d1 + 2.0); d1 = d1 / (d0 + 2.0);
        } else if (MODE == 4) {  // 4 sqrt
            x0 = sqrtf(x0 + 1.0f); x1 = sqrtf(x1 + 1.0f); x2 = sqrtf(x2 + 1.0f); x3 = sqrtf(x3 + 1.0f);
        } else if (MODE == 6) {  // 8 scalar v_fma_f32 (inline asm: not packable)
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
        } else if (MODE == 7) {  // 4 packed v_pk_fma_f32 = 8 fma
            typedef float v2 __attribute__((ext_vector_type(2)));
            v2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, pa = {a, a}, pb = {b, b};
            asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pa), "v"(pb));
            x0 = p0.x; x1 = p0.y; x2 = p1.x; x3 = p1.y; x4 = p2.x; x5 = p2.y; x6 = p3.x; x7 = p3.y;
        } else if (MODE == 5) {  // 8 double fma
            d0 = __builtin_fma(d0, (double)a, (double)b); d1 = __builtin_fma(d1, (double)a, (double)b);
            d0 = __builtin_fma(d0, (double)a, (double)b); d1 = __builtin_fma(d1, (double)a, (double)b);
            d0 = __builtin_fma(d0, (double)a, (double)b); d1 = __builtin_fma(d1, (double)a, (double)b);
            d0 = __builtin_fma(d0, (double)a, (double)b); d1 = __builtin_fma(d1, (double)a, (double)b);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (float)(d0 + d1);
}
template <int MODE>
void run(const char* name, int ops_per_iter, int blocks_per_cu) {
    float* out; int nb = 256 * blocks_per_cu;
    hipMalloc(&out, nb * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int iters = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(256), 0, 0, out, 100, 1.0001f, 0.5f);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double waves_per_simd = blocks_per_cu;  // 4 waves per block, 4 SIMDs per CU
    double inst = (double)iters * ops_per_iter * waves_per_simd;  // source-level ops per SIMD
    double cycles = ms * 1e-3 * 2.4e9;
    printf("%-30s blocks/CU=%d  %.3f ms  -> %.2f cycles per instruction (or source op) per SIMD (at 2.4 GHz)\n", name, blocks_per_cu, ms, cycles / inst);
    hipFree(out);
}
int main() {
    for (int b : {1, 2, 4, 8}) {
        run<0>("fp32 mul+add x16", 16, b);
        run<1>("fp32 fma x16", 16, b);
        run<2>("fp32 div x2", 2, b);
        run<3>("fp64 div x2", 2, b);
        run<4>("fp32 sqrt x4", 4, b);
        run<5>("fp64 fma x8", 8, b);
        run<6>("fp32 v_fma_f32 x8 (asm)", 8, b);
        run<7>("fp32 v_pk_fma_f32 x4 (asm)", 4, b);
    }
    return 0;
}
