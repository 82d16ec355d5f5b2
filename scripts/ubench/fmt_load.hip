// Does a typed buffer load (16-bit SSCALED descriptor) hand back (float)(int16) exactly, for all 65536
// values, on gfx950?  -> yes/no.  (K1s: takes the int->float convert out of the VALU stream.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
extern "C" __device__ float fmt_load_f32(v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.format.f32");
__global__ void k(const short *x, float *o, int n)
{
    const unsigned long long a = (unsigned long long) x;
    const v4i r = {(int) (a & 0xffffffffu), (int) ((a >> 32) & 0xffff), n * 2, 0x13004};
    for (int i = threadIdx.x + blockIdx.x * blockDim.x; i < n; i += gridDim.x * blockDim.x)
        o[i] = fmt_load_f32(r, (i & 1023) * 2, (i >> 10) * 2048, 0);
}
int main()
{
    const int n = 65536;
    short *h = (short *) malloc(n * 2), *d;
    float *o, *ho = (float *) malloc(n * 4);
    for (int i = 0; i < n; ++i) h[i] = (short) (i - 32768);
    hipMalloc(&d, n * 2); hipMalloc(&o, n * 4);
    hipMemcpy(d, h, n * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(64), dim3(256), 0, 0, d, o, n);
    hipMemcpy(ho, o, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) if (ho[i] != (float) h[i]) { if (bad < 5) printf("x=%d got %g\n", h[i], ho[i]); ++bad; }
    printf("typed buffer load int16->f32: %d mismatches of %d\n", bad, n);
    return bad != 0;
}
