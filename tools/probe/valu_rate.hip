// Issue-rate probe: cycles per instruction of plain and packed single-precision VALU streams at 1, 2
// and 4 waves per SIMD (one workgroup of 256 / 512 / 1024 threads on one CU).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ void k(float *out, unsigned long long *cyc, int iters)
{
    v2f a[8];
    for (int i = 0; i < 8; ++i) a[i] = v2f{(float)threadIdx.x + i, 1.0f + i};
    v2f m = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) { // 2 plain fma
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(m.x), "v"(c.x));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].y) : "v"(m.y), "v"(c.y));
                } else if (KIND == 1) { // 1 packed fma
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
                } else if (KIND == 2) { // packed add with operand selects
                    asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(a[i]) : "v"(c));
                } else if (KIND == 3) { // dpp mov
                    asm volatile("v_mov_b32_dpp %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(a[i].x));
                } else if (KIND == 4) { // 1 plain fma
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(m.x), "v"(c.x));
                } else if (KIND == 5) { // s_nop
                    asm volatile("s_nop 0");
                } else if (KIND == 6) { // salu
                    asm volatile("s_add_u32 s20, s20, 1" ::: "s20");
                }
            }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int KIND>
void run(const char *name, int perIter)
{
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 1024 * 4 * 4); hipMalloc(&cyc, 64 * 8);
    const int iters = 2000;
    for (int threads : {64, 256, 512, 1024}) {
        hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
        hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        unsigned long long h[16]; hipMemcpy(h, cyc, sizeof(unsigned long long) * (threads / 64), hipMemcpyDeviceToHost);
        double mx = 0; for (int i = 0; i < threads / 64; ++i) mx = h[i] > mx ? h[i] : mx;
        printf("%-28s waves/SIMD %.2f: %.2f cycles per instruction per wave (%.2f per SIMD-instruction)\n", name, threads / 256.0,
               mx / (iters * 32.0 * perIter), mx / (iters * 32.0 * perIter) / (threads >= 256 ? threads / 256.0 : 1.0));
    }
}
int main()
{
    run<4>("v_fma_f32", 1);
    run<0>("2 x v_fma_f32", 2);
    run<1>("v_pk_fma_f32", 1);
    run<2>("v_pk_add_f32 op_sel", 1);
    run<3>("v_mov_b32_dpp", 1);
    run<5>("s_nop 0", 1);
    run<6>("s_add_u32", 1);
    return 0;
}
