// micro-benchmark: cycles per v_mfma_f32_32x32x16_bf16 for different operand register classes (one wave per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define MM "v_mfma_f32_32x32x16_bf16 "
template <int MODE>
__global__ void __launch_bounds__(256) k(unsigned long long *out, const bf16x8 *in, float *sink) {
    bf16x8 a0 = in[threadIdx.x], a1 = in[threadIdx.x + 256], b = in[threadIdx.x + 512];
    f32x16 c0, c1;
    for (int i = 0; i < 16; i++) { c0[i] = 0.f; c1[i] = 0.f; }
    asm volatile("" : "+a"(a0), "+a"(a1));
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 64; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (MODE == 0) asm volatile(MM "%0, %2, %4, %0\n\t" MM "%1, %3, %4, %1" : "+v"(c0), "+v"(c1) : "a"(a0), "a"(a1), "v"(b));      // A agpr, B vgpr, C vgpr
            if (MODE == 1) asm volatile(MM "%0, %2, %4, %0\n\t" MM "%1, %3, %4, %1" : "+a"(c0), "+a"(c1) : "a"(a0), "a"(a1), "v"(b));      // C agpr
            if (MODE == 2) asm volatile(MM "%0, %2, %4, %0\n\t" MM "%1, %3, %4, %1" : "+a"(c0), "+a"(c1) : "v"(a0), "v"(a1), "v"(b));      // A vgpr, C agpr
            if (MODE == 3) asm volatile(MM "%0, %2, %4, %0\n\t" MM "%1, %3, %4, %1" : "+v"(c0), "+v"(c1) : "v"(a0), "v"(a1), "v"(b));      // all vgpr
            if (MODE == 4) asm volatile(MM "%0, %2, %4, %0\n\t" MM "%0, %3, %4, %0" : "+v"(c0), "+v"(c1) : "a"(a0), "a"(a1), "v"(b));      // one accumulator chain, C vgpr
            if (MODE == 5) asm volatile(MM "%0, %4, %2, %0\n\t" MM "%1, %4, %3, %1" : "+v"(c0), "+v"(c1) : "a"(a0), "a"(a1), "v"(b));      // A vgpr (shared), B agpr
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 16; i++) s += c0[i] + c1[i];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
int main() {
    unsigned long long *out; bf16x8 *in; float *sink;
    hipMalloc(&out, 8 * 1024); hipMalloc(&in, 16 * 1024); hipMalloc(&sink, 4 * 256 * 1024); hipMemset(in, 0x3c, 16 * 1024);
    const char *names[] = {"A agpr, B vgpr, C vgpr (2 chains)", "A agpr, B vgpr, C agpr", "A vgpr, B vgpr, C agpr", "all vgpr", "A agpr, C vgpr, ONE chain", "A vgpr shared, B agpr, C vgpr"};
    for (int grid : {1, 256}) for (int m = 0; m < 6; m++) {
        for (int rep = 0; rep < 2; rep++) {
            if (m == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out, in, sink);
            if (m == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, in, sink);
            if (m == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, out, in, sink);
            if (m == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, out, in, sink);
            if (m == 4) hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, out, in, sink);
            if (m == 5) hipLaunchKernelGGL(k<5>, dim3(grid), dim3(256), 0, 0, out, in, sink);
            hipDeviceSynchronize();
        }
        unsigned long long h[256]; hipMemcpy(h, out, 8 * grid, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < grid; i++) s += h[i];
        printf("grid %3d  %-36s %.1f cycles per MFMA\n", grid, names[m], s / grid / (64.0 * 32));
    }
    return 0;
}
