// Does an 8-byte LDS store to an arbitrary byte address work on this stack (SH_MEM alignment mode)?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out)
{
    __shared__ unsigned char buf[64 * 40];
    for (int i = threadIdx.x; i < 64 * 40; i += 64) buf[i] = 0xee;
    __syncthreads();
    const unsigned off = threadIdx.x * 40 + (threadIdx.x % 8);      // every alignment 0..7
    const unsigned long long v = 0x0807060504030201ull + threadIdx.x;
    asm volatile("ds_write_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" :: "v"((unsigned) (size_t) 0 + off), "v"(v) : "memory");
    __syncthreads();
    unsigned ok = 1;
    for (int j = 0; j < 8; ++j) ok &= buf[off + j] == (unsigned char) ((v >> (8 * j)) & 0xff);
    ok &= buf[off + 8] == 0xee;
    if (off) ok &= buf[off - 1] == 0xee;
    out[threadIdx.x] = ok;
}
int main()
{
    unsigned *d, h[64];
    hipMalloc(&d, 256);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) bad += !h[i];
    printf("unaligned ds_write_b64: %s (%d of 64 lanes wrong)\n", bad ? "BROKEN" : "ok", bad);
    return 0;
}
