"""Generates int_rate.hip: issue cost per wave-instruction and SIMD of the integer / dot / mixed VALU operations a cheaper
certified FIR sum could be built from, beside the fp32 ones it uses now (same harness as gen_valu_bench.py)."""
def clob(lo, hi): return ",".join(f'"v{i}"' for i in range(lo, hi))
m = {}
m["add_f32"] = [f"v_add_f32 v{100+i}, v{100+i}, %0" for i in range(32)]
m["mul_f32"] = [f"v_mul_f32 v{100+i}, %0, v{100+i}" for i in range(32)]
m["fma_f32"] = [f"v_fma_f32 v{100+i}, v{100+i}, %0, %0" for i in range(32)]
m["fmac_f32_vvv"] = [f"v_fmac_f32 v{100+i}, v{140+i%8}, v{150+i%8}" for i in range(32)]
m["pk_fma_f32"] = [f"v_pk_fma_f32 v[{100+2*(i%16)}:{101+2*(i%16)}], v[{100+2*(i%16)}:{101+2*(i%16)}], %1, %1" for i in range(32)]
m["dot2_i32_i16"] = [f"v_dot2_i32_i16 v{100+i}, v{140+i%8}, v{150+i%8}, v{100+i}" for i in range(32)]
m["dot2_u32_u16"] = [f"v_dot2_u32_u16 v{100+i}, v{140+i%8}, v{150+i%8}, v{100+i}" for i in range(32)]
m["dot4_i32_i8"] = [f"v_dot4_i32_i8 v{100+i}, v{140+i%8}, v{150+i%8}, v{100+i}" for i in range(32)]
m["dot2c_i32_i16"] = [f"v_dot2c_i32_i16 v{100+i}, v{140+i%8}, v{150+i%8}" for i in range(32)]
m["mad_i32_i24"] = [f"v_mad_i32_i24 v{100+i}, v{140+i%8}, v{150+i%8}, v{100+i}" for i in range(32)]
m["mad_u32_u24"] = [f"v_mad_u32_u24 v{100+i}, v{140+i%8}, v{150+i%8}, v{100+i}" for i in range(32)]
m["mad_u32_u16"] = [f"v_mad_u32_u16 v{100+i}, v{140+i%8}, v{150+i%8}, v{100+i}" for i in range(32)]
m["mad_i32_i16"] = [f"v_mad_i32_i16 v{100+i}, v{140+i%8}, v{150+i%8}, v{100+i}" for i in range(32)]
m["pk_mad_i16"] = [f"v_pk_mad_i16 v{100+i}, v{140+i%8}, v{150+i%8}, v{100+i}" for i in range(32)]
m["pk_mul_lo_u16"] = [f"v_pk_mul_lo_u16 v{100+i}, v{140+i%8}, v{150+i%8}" for i in range(32)]
m["pk_add_i16"] = [f"v_pk_add_i16 v{100+i}, v{140+i%8}, v{100+i}" for i in range(32)]
m["add3_u32"] = [f"v_add3_u32 v{100+i}, v{140+i%8}, v{150+i%8}, v{100+i}" for i in range(32)]
m["add_u32"] = [f"v_add_u32 v{100+i}, v{100+i}, %0" for i in range(32)]
m["mul_lo_u32"] = [f"v_mul_lo_u32 v{100+i}, v{140+i%8}, v{150+i%8}" for i in range(32)]
m["mul_u32_u24"] = [f"v_mul_u32_u24 v{100+i}, v{140+i%8}, v{150+i%8}" for i in range(32)]
m["alignbit"] = [f"v_alignbit_b32 v{100+i}, v{100+i}, v{140+i%8}, 30" for i in range(32)]
m["max3_i32"] = [f"v_max3_i32 v{100+i}, v{100+i}, v{140+i%8}, v{150+i%8}" for i in range(32)]
m["perm_b32"] = [f"v_perm_b32 v{100+i}, v{140+i%8}, v{150+i%8}, v{100+i}" for i in range(32)]
m["cvt_f32_i32"] = [f"v_cvt_f32_i32 v{100+i}, v{100+i}" for i in range(32)]
m["sad_u32"] = [f"v_sad_u32 v{100+i}, v{140+i%8}, v{150+i%8}, v{100+i}" for i in range(32)]
m["lshl_add_u32"] = [f"v_lshl_add_u32 v{100+i}, v{140+i%8}, 3, v{100+i}" for i in range(32)]
m["mul_sdwa"] = [f"v_mul_u32_u24_sdwa v{100+i}, v{140+i%8}, v{150+i%8} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" for i in range(32)]
m["cmp_gt_u32"] = [f"v_cmp_gt_u32 vcc, v{100+i}, v{140+i%8}" for i in range(32)]
# the direct-form stream per output as it is: 6 adds, 1 mul, 5 fmac (SGPR taps), 1 alignbit, max every output
sc = []
for smp in range(8):
    for q in range(6): sc.append(f"v_add_f32 v{140+q}, v{100+(smp+q)%32}, v{100+(smp+11-q)%32}")
    sc.append("v_mul_f32 v146, s20, v140")
    for q in range(1, 6): sc.append(f"v_fmac_f32 v146, s{20+q}, v{140+q}")
    sc.append("v_alignbit_b32 v151, v151, v146, 30")
    sc.append(f"v_max_i32 v148, v148, v{100+smp}")
m["direct12_now_8out"] = ["s_mov_b32 s%d, 0x3f000000" % r for r in range(20, 26)] + sc
# the same output from six dot2 on packed int16 pairs + alignbit + compare (taps in VGPR pairs)
dc = []
for smp in range(8):
    dc.append(f"v_dot2_i32_i16 v146, v{100+smp%16}, v160, v159")
    for q in range(1, 6): dc.append(f"v_dot2_i32_i16 v146, v{100+(smp+q)%16}, v{160+q}, v146")
    if smp % 2: dc.append(f"v_dot2_i32_i16 v146, v{100+(smp+6)%16}, v166, v146")
    dc.append("v_alignbit_b32 v151, v151, v146, 31")
    dc.append("v_cmp_gt_u32 vcc, v158, v146")
    if smp % 2: dc.append(f"v_pk_max_i16 v148, v148, v{100+smp%16}")
m["dot2_12_8out"] = dc
src = ['#include <hip/hip_runtime.h>', '#include <cstdio>', 'typedef float f32x2 __attribute__((ext_vector_type(2)));']
names = list(m)
for k, (name, ins) in enumerate(m.items()):
    body = "\\n\\t".join(ins)
    src.append(f'''__global__ __launch_bounds__(256) void k{k}(float *out, int iters, float b) {{
  f32x2 bb = {{b, b}};
  for (int it = 0; it < iters; ++it)
    asm volatile("{body}" :: "v"(b), "v"(bb) : {clob(100, 170)}, "vcc", "s20","s21","s22","s23","s24","s25");
  out[blockIdx.x * blockDim.x + threadIdx.x] = b;
}}''')
src.append('typedef void (*kern_t)(float*, int, float);')
src.append('int main() { float *d; (void) hipMalloc(&d, 8192 * 256 * 4); const int iters = 20000;')
src.append('  kern_t ks[] = {' + ",".join(f"k{k}" for k in range(len(names))) + '};')
src.append('  const char *nm[] = {' + ",".join(f'"{n}"' for n in names) + '};')
src.append('  const int ninstr[] = {' + ",".join(str(len([i for i in m[n] if not i.startswith("s_")])) for n in names) + '};')
src.append('''  for (int wps : {1, 4, 5}) for (int q = 0; q < (int)(sizeof(ks)/sizeof(ks[0])); ++q) {
    hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    hipLaunchKernelGGL(ks[q], dim3(256 * wps), dim3(256), 0, 0, d, 10, 1.0f);
    (void) hipEventRecord(e0);
    hipLaunchKernelGGL(ks[q], dim3(256 * wps), dim3(256), 0, 0, d, iters, 1.0f);
    (void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
    float ms; (void) hipEventElapsedTime(&ms, e0, e1);
    printf("waves/SIMD=%d %-20s %8.3f ms  %.3f ns per wave-instruction per SIMD\\n", wps, nm[q], ms, ms * 1e6 / iters / wps / ninstr[q]);
  }
  return 0; }''')
open("int_rate.hip", "w").write("\n".join(src))
