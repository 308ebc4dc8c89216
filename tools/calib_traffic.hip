// Calibration (not product): kernels that move a KNOWN number of bytes with 2-, 4-, 8- and 16-byte-per-lane accesses, for
// reading rocprofv3's FETCH_SIZE / WRITE_SIZE / TCC_EA0_RDREQ_* / TCC_EA0_WRREQ_* against (MI355X_MICROARCH.md, HBM: only
// the 16-byte streaming case is calibrated there).  Buffers are 1 GiB (beyond the 256 MiB Infinity Cache); every kernel
// touches 512 MiB exactly once.    hipcc --offload-arch=gfx950 -O3 -o tools/calib_traffic.bin tools/calib_traffic.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
template <typename T> __global__ void cal_read(const T* a, unsigned* sink, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned acc = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) { T v = a[i]; acc += ((const unsigned char*)&v)[0]; }
    if (acc == 0x7fffffffu) sink[0] = acc;
}
template <typename T> __global__ void cal_write(T* a, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    T v; for (unsigned k = 0; k < sizeof(T); k++) ((unsigned char*)&v)[k] = (unsigned char)(i + k);
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = v;
}
// the access shape of the codec's block kernels: 8-byte rows of 8x8 blocks inside 352-byte lines (every lane its own row)
__global__ void cal_read_rows8(const uint8_t* a, unsigned* sink, int frames)
{
    const int W = 352, H = 288;
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // one lane per block row
    const long long nrows = (long long)frames * (W / 8) * H;
    if (id >= nrows) return;
    const int f = (int)(id / ((W / 8) * H)), r = (int)(id % ((W / 8) * H));
    const int blk = r / 8, i = r % 8, by = blk / (W / 8), bx = blk % (W / 8);
    const uint2 v = *(const uint2*)(a + (size_t)f * W * H + (by * 8 + i) * W + bx * 8);
    if (v.x == 0x12345678u && v.y == 0x9abcdef0u) sink[0] = 1;
}
int main()
{
    const size_t bytes = (size_t)1 << 30, half = bytes / 2;
    uint8_t* a; unsigned* sink;
    (void)hipMalloc(&a, bytes); (void)hipMalloc(&sink, 64); (void)hipMemset(a, 1, bytes);
    const dim3 grid(256 * 16), blk(256);
    hipLaunchKernelGGL(cal_read<uint16_t>, grid, blk, 0, 0, (const uint16_t*)a, sink, half / 2);
    hipLaunchKernelGGL(cal_read<uint32_t>, grid, blk, 0, 0, (const uint32_t*)(a + half), sink, half / 4);
    hipLaunchKernelGGL(cal_read<uint2>, grid, blk, 0, 0, (const uint2*)a, sink, half / 8);
    hipLaunchKernelGGL(cal_read<uint4>, grid, blk, 0, 0, (const uint4*)(a + half), sink, half / 16);
    hipLaunchKernelGGL(cal_write<uint16_t>, grid, blk, 0, 0, (uint16_t*)a, half / 2);
    hipLaunchKernelGGL(cal_write<uint32_t>, grid, blk, 0, 0, (uint32_t*)(a + half), half / 4);
    hipLaunchKernelGGL(cal_write<uint2>, grid, blk, 0, 0, (uint2*)a, half / 8);
    hipLaunchKernelGGL(cal_write<uint4>, grid, blk, 0, 0, (uint4*)(a + half), half / 16);
    const int frames = 5000;                                                    // 5000 CIF luma planes = 507 MB
    hipLaunchKernelGGL(cal_read_rows8, dim3((unsigned)(((long long)frames * 44 * 288 + 255) / 256)), blk, 0, 0, a, sink, frames);
    (void)hipDeviceSynchronize();
    printf("known bytes: 2/4/8/16-byte reads and writes %zu each; rows8 read %zu\n", half, (size_t)frames * 352 * 288);
    return 0;
}
