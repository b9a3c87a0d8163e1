// micro-benchmark: the W-loop iteration of fused_fstat_kernel (wait, ring read, 2 MFMAs, squares) piece by piece
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define MM "v_mfma_f32_32x32x16_bf16 "
template <int MODE>
__global__ void __launch_bounds__(256) k(unsigned long long *out, const bf16x8 *in, float *sink) {
    extern __shared__ unsigned char smem[];
    for (int i = threadIdx.x; i < 65536 / 4; i += 256) ((unsigned *)smem)[i] = 0x3c003c00u;
    __syncthreads();
    bf16x8 a0 = in[threadIdx.x], a1 = in[threadIdx.x + 256];
    bf16x8 b0 = in[threadIdx.x + 512], b1 = b0, b2 = b0, b3 = b0, b4 = b0;
    f32x16 c0, c1;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int i = 0; i < 16; i++) { c0[i] = 0.f; c1[i] = 0.f; }
    asm volatile("" : "+a"(a0), "+a"(a1));
    const unsigned lane = threadIdx.x & 63;
    const unsigned g = 2 * (lane & 31) + (lane >> 5);
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem + ((g ^ ((g >> 4) & 1)) * 16);
    const char *gsrc = (const char *)in + (threadIdx.x & 63) * 16;
    const unsigned ldsdst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem + 32768 + (threadIdx.x >> 6) * 4096);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 64; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (MODE == 0) asm volatile(MM "%0, %2, %4, %0\n\t" MM "%1, %3, %4, %1" : "+v"(c0), "+v"(c1) : "a"(a0), "a"(a1), "v"(b0));
            if (MODE == 1 || MODE == 2 || MODE == 3) // + counted wait + ring read
                asm volatile("s_waitcnt lgkmcnt(3)\n\tds_read_b128 %5, %6 offset:%7\n\t" MM "%0, %2, %4, %0\n\t" MM "%1, %3, %4, %1"
                             : "+v"(c0), "+v"(c1) : "a"(a0), "a"(a1), "v"(u % 5 == 0 ? b0 : u % 5 == 1 ? b1 : u % 5 == 2 ? b2 : u % 5 == 3 ? b3 : b4),
                               "v"(u % 5 == 4 ? b0 : u % 5 == 0 ? b1 : u % 5 == 1 ? b2 : u % 5 == 2 ? b3 : b4), "v"(addr), "i"(u * 4096) : "memory");
            if (MODE == 2) { s0 = fmaf(c0[u], c0[u], s0); s1 = fmaf(c1[u], c1[u], s1); }                     // + 2 scalar fma
            if (MODE == 3) { asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(*(double *)&s0) : "v"(*(double *)&s2)); } // + 1 packed fma
            if (MODE == 4) { asm volatile("s_waitcnt lgkmcnt(3)\n\t" MM "%0, %2, %4, %0\n\t" MM "%1, %3, %4, %1" : "+v"(c0), "+v"(c1) : "a"(a0), "a"(a1), "v"(b0)); }
            if (MODE == 6) // squares as fillers INSIDE the statement, one v_fma_f32 behind each MFMA (independent registers)
                asm volatile("s_waitcnt lgkmcnt(3)\n\tds_read_b128 %5, %6 offset:%7\n\t" MM "%0, %2, %4, %0\n\tv_fma_f32 %8, %10, %10, %8\n\t" MM "%1, %3, %4, %1\n\tv_fma_f32 %9, %11, %11, %9"
                             : "+v"(c0), "+v"(c1) : "a"(a0), "a"(a1), "v"(b0), "v"(b1), "v"(addr), "i"(u * 4096), "v"(s0), "v"(s1), "v"(s2), "v"(s3) : "memory");
            if (MODE == 7) // two fillers behind each MFMA
                asm volatile("s_waitcnt lgkmcnt(3)\n\tds_read_b128 %5, %6 offset:%7\n\t" MM "%0, %2, %4, %0\n\tv_fma_f32 %8, %10, %10, %8\n\tv_fma_f32 %9, %11, %11, %9\n\t" MM "%1, %3, %4, %1\n\tv_fma_f32 %8, %10, %10, %8\n\tv_fma_f32 %9, %11, %11, %9"
                             : "+v"(c0), "+v"(c1) : "a"(a0), "a"(a1), "v"(b0), "v"(b1), "v"(addr), "i"(u * 4096), "v"(s0), "v"(s1), "v"(s2), "v"(s3) : "memory");
            if (MODE == 8) { // compiler-placed independent scalar fma between statements
                asm volatile("s_waitcnt lgkmcnt(3)\n\tds_read_b128 %5, %6 offset:%7\n\t" MM "%0, %2, %4, %0\n\t" MM "%1, %3, %4, %1"
                             : "+v"(c0), "+v"(c1) : "a"(a0), "a"(a1), "v"(u % 5 == 0 ? b0 : u % 5 == 1 ? b1 : u % 5 == 2 ? b2 : u % 5 == 3 ? b3 : b4),
                               "v"(u % 5 == 4 ? b0 : u % 5 == 0 ? b1 : u % 5 == 1 ? b2 : u % 5 == 2 ? b3 : b4), "v"(addr), "i"(u * 4096) : "memory");
                s0 = fmaf(s2, s2, s0);
            }
            if (MODE == 9) { // every 4th iteration carries one LDS-DMA piece (M0 written in the same statement)
                if ((u & 3) == 1)
                    asm volatile("s_waitcnt lgkmcnt(3)\n\tds_read_b128 %5, %6 offset:%7\n\t" MM "%0, %2, %4, %0\n\ts_mov_b32 m0, %9\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %8, off\n\t" MM "%1, %3, %4, %1"
                                 : "+v"(c0), "+v"(c1) : "a"(a0), "a"(a1), "v"(b0), "v"(b1), "v"(addr), "i"(u * 4096), "v"(gsrc), "s"(ldsdst + (u >> 2) * 1024) : "memory");
                else
                    asm volatile("s_waitcnt lgkmcnt(3)\n\tds_read_b128 %5, %6 offset:%7\n\t" MM "%0, %2, %4, %0\n\t" MM "%1, %3, %4, %1"
                                 : "+v"(c0), "+v"(c1) : "a"(a0), "a"(a1), "v"(b0), "v"(b1), "v"(addr), "i"(u * 4096) : "memory");
            }
            if (MODE == 10) { // same pieces, but all four of an unrolled block back to back at its start
                if (u == 0)
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024\n\tglobal_load_lds_dwordx4 %0, off offset:2048\n\tglobal_load_lds_dwordx4 %0, off offset:3072"
                                 :: "v"(gsrc), "s"(ldsdst) : "memory");
                asm volatile("s_waitcnt lgkmcnt(3)\n\tds_read_b128 %5, %6 offset:%7\n\t" MM "%0, %2, %4, %0\n\t" MM "%1, %3, %4, %1"
                             : "+v"(c0), "+v"(c1) : "a"(a0), "a"(a1), "v"(b0), "v"(b1), "v"(addr), "i"(u * 4096) : "memory");
            }
            if (MODE == 11) // the score-tile form: A = ring fragment (VGPR), B = resident fragment (AGPR)
                asm volatile("s_waitcnt lgkmcnt(3)\n\tds_read_b128 %5, %6 offset:%7\n\t" MM "%0, %4, %2, %0\n\t" MM "%1, %4, %3, %1"
                             : "+v"(c0), "+v"(c1) : "a"(a0), "a"(a1), "v"(b0), "v"(b1), "v"(addr), "i"(u * 4096) : "memory");
            if (MODE == 5) { asm volatile("ds_read_b128 %5, %6 offset:%7\n\t" MM "%0, %2, %4, %0\n\t" MM "%1, %3, %4, %1\n\ts_waitcnt lgkmcnt(0)"
                             : "+v"(c0), "+v"(c1) : "a"(a0), "a"(a1), "v"(b0), "v"(b1), "v"(addr), "i"(u * 4096) : "memory"); }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = s0 + s1 + s2 + s3;
    for (int i = 0; i < 16; i++) s += c0[i] + c1[i];
    sink[blockIdx.x * 256 + threadIdx.x] = s + (float)b1[0] + (float)b2[0] + (float)b3[0] + (float)b4[0];
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
int main() {
    unsigned long long *out; bf16x8 *in; float *sink;
    hipMalloc(&out, 8 * 1024); hipMalloc(&in, 16 * 1024); hipMalloc(&sink, 4 * 256 * 1024); hipMemset(in, 0x3c, 16 * 1024);
    const char *names[] = {"2 MFMA", "wait(3) + ds_read_b128 + 2 MFMA", "... + 2 v_fma_f32 on the accumulators", "... + 1 v_pk_fma_f32 (independent)", "wait(3) + 2 MFMA (no read)", "ds_read + 2 MFMA + wait(0)", "in-asm: 1 v_fma behind each MFMA", "in-asm: 2 v_fma behind each MFMA", "compiler-placed 1 scalar v_fma per iteration", "1 LDS-DMA piece per 4 iterations, in-stream", "4 LDS-DMA pieces back to back per 16 iterations", "wait + read + 2 MFMA, A = VGPR fragment, B = AGPR"};
    for (int grid : {256}) for (int m = 0; m < 12; m++) {
        for (int rep = 0; rep < 2; rep++) {
#define L(M) if (m == M) { hipFuncSetAttribute((const void *)k<M>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536); hipLaunchKernelGGL(k<M>, dim3(grid), dim3(256), 65536, 0, out, in, sink); }
            L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11)
            hipDeviceSynchronize();
        }
        unsigned long long h[256]; hipMemcpy(h, out, 8 * grid, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < grid; i++) s += h[i];
        printf("grid %3d  %-44s %.1f cycles per MFMA\n", grid, names[m], s / grid / (64.0 * 32));
    }
    return 0;
}
