// Probe: operand / result layout of v_mfma_f64_4x4x4 (4 blocks of 16 lanes) on gfx950, and its issue cost.
#include <hip/hip_runtime.h>
#include <stdio.h>
// one-hot: A = 1 in lane x only, B = 1 in lane y only; records the lanes whose result is 1
__global__ void k(unsigned long long *hit)
{
    const int l = threadIdx.x;
    for (int x = 0; x < 64; ++x)
        for (int y = 0; y < 64; ++y) {
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(l == x ? 1.0 : 0.0, l == y ? 1.0 : 0.0, 0.0, 0, 0, 0);
            const unsigned long long m = __builtin_amdgcn_ballot_w64(d != 0.0);
            if (l == 0) hit[x * 64 + y] = m;
        }
}
__global__ void rate(double *out, int n)
{
    double a = threadIdx.x * 0.5, b = 1.0 / (1 + threadIdx.x), c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    const long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
    }
    const long long t1 = clock64();
    out[threadIdx.x] = c0 + c1 + c2 + c3;
    if (threadIdx.x == 0) out[64] = (double)(t1 - t0) / (4.0 * n);
}
int main()
{
    unsigned long long *dh, hh[4096];
    hipMalloc(&dh, sizeof hh);
    k<<<1, 64>>>(dh);
    hipMemcpy(hh, dh, sizeof hh, hipMemcpyDeviceToHost);
    // for lane x as A source and lane y as B source inside block 0: which lane receives the product
    printf("rows: A lane x (0..15), columns: B lane y (0..15); entry: result lane or -- if none\n");
    for (int x = 0; x < 16; ++x) {
        for (int y = 0; y < 16; ++y) {
            const unsigned long long m = hh[x * 64 + y];
            if (!m) printf(" --");
            else printf(" %2d", __builtin_ctzll(m));
        }
        printf("\n");
    }
    int cross = 0;
    for (int x = 0; x < 64; ++x)
        for (int y = 0; y < 64; ++y)
            if (x / 16 != y / 16 && hh[x * 64 + y]) ++cross;
    printf("products across blocks of 16 lanes: %d\n", cross);
    // same pattern in every block?
    int same = 1;
    for (int b = 1; b < 4; ++b)
        for (int x = 0; x < 16; ++x)
            for (int y = 0; y < 16; ++y)
                if ((hh[(16 * b + x) * 64 + 16 * b + y] >> (16 * b)) != hh[x * 64 + y]) same = 0;
    printf("blocks 1..3 repeat block 0's pattern: %s\n", same ? "yes" : "NO");
    double *dout, hout[65];
    hipMalloc(&dout, sizeof hout);
    rate<<<1, 64>>>(dout, 1000);
    hipMemcpy(hout, dout, sizeof hout, hipMemcpyDeviceToHost);
    printf("clock64 ticks per back-to-back independent v_mfma_f64_4x4x4 (one wave): %.2f\n", hout[64]);
    return 0;
}
