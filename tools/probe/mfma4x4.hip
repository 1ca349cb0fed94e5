// Probe: operand / result layout of v_mfma_f32_4x4x1 (16 blocks) on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *a, const float *b, float *d)
{
    const int l = threadIdx.x;
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}
int main()
{
    float ha[64], hb[64], hd[256], *da, *db, *dd;
    for (int l = 0; l < 64; ++l) {
        ha[l] = 1.0f + l;           // A value of lane l
        hb[l] = 1000.0f * (1 + l);  // B value of lane l
    }
    hipMalloc(&da, sizeof ha), hipMalloc(&db, sizeof hb), hipMalloc(&dd, sizeof hd);
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice);
    hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    k<<<1, 64>>>(da, db, dd);
    hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost);
    // hypothesis: lane (blk, j) element r = A[lane 4 blk + r] * B[lane 4 blk + j]
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int blk = l / 4, j = l % 4;
            const float want = ha[4 * blk + r] * hb[4 * blk + j];
            if (hd[l * 4 + r] != want) ++bad;
        }
    printf("hypothesis D[lane(blk,j)][r] = A[blk,r] * B[blk,j]: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
    for (int l = 0; l < 8; ++l) printf("lane %d: %g %g %g %g\n", l, hd[4 * l], hd[4 * l + 1], hd[4 * l + 2], hd[4 * l + 3]);
    return 0;
}
