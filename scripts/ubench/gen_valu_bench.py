"""Generates valu_rate.hip: exact instruction streams (inline asm, fixed VGPRs)
to measure per-instruction issue cost of scalar / packed fp32 VALU and MFMA
mixes on gfx950."""
def clob(lo, hi): return ",".join(f'"v{i}"' for i in range(lo, hi))
modes = {}
# A: 32 independent v_add_f32
modes["add32"] = [f"v_add_f32 v{100+i}, v{100+i}, %0" for i in range(32)]
# B: K1-like: 32 x (v_mul tmp ; v_add acc) with 4 rotating temps
b = []
for i in range(32):
    b.append(f"v_mul_f32 v{140+i%4}, %0, v{100+(i*7)%32}")
    b.append(f"v_add_f32 v{100+i}, v{100+i}, v{140+i%4}")
modes["mul_add32"] = b
# C: 32 independent v_pk_add_f32 on register pairs
modes["pk_add32"] = [f"v_pk_add_f32 v[{100+2*i}:{101+2*i}], v[{100+2*i}:{101+2*i}], %1" for i in range(32)]
# D: 32 x (v_pk_mul tmp ; v_pk_add acc)
d = []
for i in range(32):
    t = 170 + 2*(i % 4)
    d.append(f"v_pk_mul_f32 v[{t}:{t+1}], %1, v[{100+2*((i*7)%32)}:{101+2*((i*7)%32)}]")
    d.append(f"v_pk_add_f32 v[{100+2*i}:{101+2*i}], v[{100+2*i}:{101+2*i}], v[{t}:{t+1}]")
modes["pk_mul_add32"] = d
# E: 32 x v_mul only
modes["mul32"] = [f"v_mul_f32 v{100+i}, %0, v{100+i}" for i in range(32)]
# F: integer ops: 32 x v_add_u32
modes["iadd32"] = [f"v_add_u32 v{100+i}, v{100+i}, %0" for i in range(32)]
# G: 32 x v_fma_f32
modes["fma32"] = [f"v_fma_f32 v{100+i}, v{100+i}, %0, %0" for i in range(32)]
# H: 32 x v_pk_fma
modes["pk_fma32"] = [f"v_pk_fma_f32 v[{100+2*i}:{101+2*i}], v[{100+2*i}:{101+2*i}], %1, %1" for i in range(32)]
src = ['#include <hip/hip_runtime.h>', '#include <cstdio>', 'typedef float f32x2 __attribute__((ext_vector_type(2)));']
names = list(modes)
for k, (name, ins) in enumerate(modes.items()):
    body = "\\n\\t".join(ins)
    src.append(f'''__global__ __launch_bounds__(256) void k{k}(float *out, int iters, float b) {{
  f32x2 bb = {{b, b}};
  for (int it = 0; it < iters; ++it)
    asm volatile("{body}" :: "v"(b), "v"(bb) : {clob(100, 180)});
  out[blockIdx.x * blockDim.x + threadIdx.x] = b;
}}''')
src.append('typedef void (*kern_t)(float*, int, float);')
src.append('int main() { float *d; (void) hipMalloc(&d, 8192 * 256 * 4); const int iters = 20000;')
src.append('  kern_t ks[] = {' + ",".join(f"k{k}" for k in range(len(names))) + '};')
src.append('  const char *nm[] = {' + ",".join(f'"{n}"' for n in names) + '};')
src.append('  const int ninstr[] = {' + ",".join(str(len(modes[n])) for n in names) + '};')
src.append('''  for (int wps = 1; wps <= 8; wps *= 2) for (int m = 0; m < (int)(sizeof(ks)/sizeof(ks[0])); ++m) {
    hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    hipLaunchKernelGGL(ks[m], dim3(256 * wps), dim3(256), 0, 0, d, 10, 1.0f);
    (void) hipEventRecord(e0);
    hipLaunchKernelGGL(ks[m], dim3(256 * wps), dim3(256), 0, 0, d, iters, 1.0f);
    (void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
    float ms; (void) hipEventElapsedTime(&ms, e0, e1);
    printf("waves/SIMD=%d %-14s %8.3f ms  %.3f ns per wave-instruction per SIMD\\n", wps, nm[m], ms, ms * 1e6 / iters / wps / ninstr[m]);
  }
  return 0; }''')
open("valu_rate.hip", "w").write("\n".join(src))
