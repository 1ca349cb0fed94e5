// Calibration of the FETCH_SIZE counter for the access widths the kernels use: every kernel below
// reads a 1 GiB buffer (four times the Infinity Cache) exactly once, so the bytes that reach the
// memory side are known; rocprofv3 --pmc FETCH_SIZE reports what the counter makes of each pattern.
//   read16: 16 bytes per lane, a wave reads 1024 consecutive bytes (global_load_dwordx4)
//   read12: 12 bytes per lane, a wave reads 768 consecutive bytes (global_load_dwordx3, unaligned type)
//   read4:   4 bytes per lane, a wave reads 256 consecutive bytes (global_load_dword)
// build: hipcc --offload-arch=gfx950 -O3 tools/calib/fetch_calib.hip -o tools/calib/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u3u __attribute__((ext_vector_type(3), aligned(1)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
__global__ void read16(const u4 *p, size_t n, unsigned *out)
{
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const u4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void read12(const unsigned char *p, size_t n, unsigned *out)
{
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const u3u v = *reinterpret_cast<const u3u *>(p + 12 * i);
        acc ^= v.x ^ v.y ^ v.z;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void read4(const unsigned *p, size_t n, unsigned *out)
{
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
    if (acc == 0x12345678u) out[0] = acc;
}
int main()
{
    const size_t bytes = (size_t)1 << 30;
    unsigned char *buf;
    unsigned *out;
    if (hipMalloc(&buf, bytes + 64) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
    hipMemset(buf, 1, bytes + 64);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(read16, dim3(2048), dim3(256), 0, 0, reinterpret_cast<const u4 *>(buf), bytes / 16, out);
        hipLaunchKernelGGL(read12, dim3(2048), dim3(256), 0, 0, buf, bytes / 12, out);
        hipLaunchKernelGGL(read4, dim3(2048), dim3(256), 0, 0, reinterpret_cast<const unsigned *>(buf), bytes / 4, out);
    }
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    printf("each kernel read %zu bytes once\n", bytes);
    return 0;
}
