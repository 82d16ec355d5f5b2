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
# G2: 32 x v_fmac_f32 (VOP2 encoding, accumulate in place), SGPR and VGPR multiplicand
modes["fmac32"] = [f"v_fmac_f32 v{100+i}, %0, v{140+i%8}" for i in range(32)]
modes["fmac32_sgpr"] = ["s_mov_b32 s20, 0x3f000000"] + [f"v_fmac_f32 v{100+i}, s20, v{140+i%8}" for i in range(32)]
# K1s-like with FMA: 1 mul + 11 fmac per "sample" on 12 rotating accumulators
kf = []
for smp in range(4):
    for q in range(6):
        a0 = 100 + (smp + 11 - q) % 12
        a1 = 100 + (smp + q) % 12
        if q == 0:
            kf.append(f"v_mul_f32 v{a0}, s20, v{140+smp}")
        else:
            kf.append(f"v_fmac_f32 v{a0}, s20, v{140+smp}")
        kf.append(f"v_fmac_f32 v{a1}, s20, v{140+smp}")
modes["k1s_fma_4samples"] = ["s_mov_b32 s20, 0x3f000000"] + kf
# H: 32 x v_pk_fma
modes["pk_fma32"] = [f"v_pk_fma_f32 v[{100+2*i}:{101+2*i}], v[{100+2*i}:{101+2*i}], %1, %1" for i in range(32)]

# the transposed FIR on register PAIRS: acc pair += tap pair (VGPR or SGPR pair, halves swapped by op_sel) x one sample
# broadcast from either half of a sample pair (op_sel / op_sel_hi on the sample operand)
modes["pk_fma_acc_vtap"] = [f"v_pk_fma_f32 v[{100+2*i}:{101+2*i}], v[{170+2*(i%4)}:{171+2*(i%4)}], v[164:165], v[{100+2*i}:{101+2*i}] op_sel_hi:[1,0,1]" for i in range(32)]
modes["pk_fma_acc_vtap_swz"] = [f"v_pk_fma_f32 v[{100+2*i}:{101+2*i}], v[{170+2*(i%4)}:{171+2*(i%4)}], v[164:165], v[{100+2*i}:{101+2*i}] op_sel:[{i%2},1,0] op_sel_hi:[{1-i%2},1,1]" for i in range(32)]
modes["pk_fma_acc_stap"] = ["s_mov_b32 s20, 0x3f000000", "s_mov_b32 s21, 0x3f100000"] + [f"v_pk_fma_f32 v[{100+2*i}:{101+2*i}], s[20:21], v[164:165], v[{100+2*i}:{101+2*i}] op_sel_hi:[1,0,1]" for i in range(32)]
modes["pk_fma_acc_stap_swz"] = ["s_mov_b32 s20, 0x3f000000", "s_mov_b32 s21, 0x3f100000"] + [f"v_pk_fma_f32 v[{100+2*i}:{101+2*i}], s[20:21], v[164:165], v[{100+2*i}:{101+2*i}] op_sel:[1,1,0] op_sel_hi:[0,1,1]" for i in range(32)]
modes["fmac32_vtap"] = [f"v_fmac_f32 v{100+i}, v{170+i%8}, v{164+i%2}" for i in range(32)]

# the packed transposed 12-tap stream of fir_sign_pk.hip, 12 samples (six pair steps), registers renamed
modes["pk12_stream_12samples"] = ["s_mov_b32 s%d, 0x3f000000" % r for r in range(20, 34)] + ['v_pk_fma_f32 v[74:75], s[28:29], v[152:153], v[74:75] op_sel:[1,0,0] op_sel_hi:[0,0,1]', 'v_pk_fma_f32 v[76:77], s[30:31], v[152:153], v[76:77] op_sel:[1,0,0] op_sel_hi:[0,0,1]', 'v_pk_fma_f32 v[72:73], s[26:27], v[152:153], v[72:73] op_sel:[0,0,0] op_sel_hi:[1,0,1]', 'v_pk_fma_f32 v[78:79], s[32:33], v[152:153], v[78:79] op_sel:[0,0,0] op_sel_hi:[1,0,1]', 'v_pk_fma_f32 v[80:81], s[30:31], v[152:153], v[80:81] op_sel:[0,0,0] op_sel_hi:[1,0,1]', 'v_pk_fma_f32 v[82:83], s[28:29], v[152:153], v[82:83] op_sel:[0,0,0] op_sel_hi:[1,0,1]', 'v_sub_f32 v84, |v73|, v154', 'v_alignbit_b32 v151, v151, v84, 31', 'v_alignbit_b32 v150, v150, v73, 31', 'v_mov_b32 v73, 0', 'v_pk_fma_f32 v[74:75], s[20:21], v[152:153], v[74:75] op_sel:[0,1,0] op_sel_hi:[1,1,1]', 'v_pk_fma_f32 v[76:77], s[22:23], v[152:153], v[76:77] op_sel:[0,1,0] op_sel_hi:[1,1,1]', 'v_pk_fma_f32 v[78:79], s[24:25], v[152:153], v[78:79] op_sel:[0,1,0] op_sel_hi:[1,1,1]', 'v_pk_fma_f32 v[80:81], s[24:25], v[152:153], v[80:81] op_sel:[1,1,0] op_sel_hi:[0,1,1]', 'v_pk_fma_f32 v[82:83], s[22:23], v[152:153], v[82:83] op_sel:[1,1,0] op_sel_hi:[0,1,1]', 'v_pk_fma_f32 v[72:73], s[20:21], v[152:153], v[72:73] op_sel:[1,1,0] op_sel_hi:[0,1,1]', 'v_sub_f32 v84, |v74|, v154', 'v_alignbit_b32 v151, v151, v84, 31', 'v_alignbit_b32 v150, v150, v74, 31', 'v_mov_b32 v74, 0', 'v_pk_fma_f32 v[76:77], s[28:29], v[152:153], v[76:77] op_sel:[1,0,0] op_sel_hi:[0,0,1]', 'v_pk_fma_f32 v[78:79], s[30:31], v[152:153], v[78:79] op_sel:[1,0,0] op_sel_hi:[0,0,1]', 'v_pk_fma_f32 v[74:75], s[26:27], v[152:153], v[74:75] op_sel:[0,0,0] op_sel_hi:[1,0,1]', 'v_pk_fma_f32 v[80:81], s[32:33], v[152:153], v[80:81] op_sel:[0,0,0] op_sel_hi:[1,0,1]', 'v_pk_fma_f32 v[82:83], s[30:31], v[152:153], v[82:83] op_sel:[0,0,0] op_sel_hi:[1,0,1]', 'v_pk_fma_f32 v[72:73], s[28:29], v[152:153], v[72:73] op_sel:[0,0,0] op_sel_hi:[1,0,1]', 'v_sub_f32 v84, |v75|, v154', 'v_alignbit_b32 v151, v151, v84, 31', 'v_alignbit_b32 v150, v150, v75, 31', 'v_mov_b32 v75, 0', 'v_pk_fma_f32 v[76:77], s[20:21], v[152:153], v[76:77] op_sel:[0,1,0] op_sel_hi:[1,1,1]', 'v_pk_fma_f32 v[78:79], s[22:23], v[152:153], v[78:79] op_sel:[0,1,0] op_sel_hi:[1,1,1]', 'v_pk_fma_f32 v[80:81], s[24:25], v[152:153], v[80:81] op_sel:[0,1,0] op_sel_hi:[1,1,1]', 'v_pk_fma_f32 v[82:83], s[24:25], v[152:153], v[82:83] op_sel:[1,1,0] op_sel_hi:[0,1,1]', 'v_pk_fma_f32 v[72:73], s[22:23], v[152:153], v[72:73] op_sel:[1,1,0] op_sel_hi:[0,1,1]', 'v_pk_fma_f32 v[74:75], s[20:21], v[152:153], v[74:75] op_sel:[1,1,0] op_sel_hi:[0,1,1]', 'v_sub_f32 v84, |v76|, v154', 'v_alignbit_b32 v151, v151, v84, 31', 'v_alignbit_b32 v150, v150, v76, 31', 'v_mov_b32 v76, 0', 'v_pk_fma_f32 v[78:79], s[28:29], v[152:153], v[78:79] op_sel:[1,0,0] op_sel_hi:[0,0,1]', 'v_pk_fma_f32 v[80:81], s[30:31], v[152:153], v[80:81] op_sel:[1,0,0] op_sel_hi:[0,0,1]', 'v_pk_fma_f32 v[76:77], s[26:27], v[152:153], v[76:77] op_sel:[0,0,0] op_sel_hi:[1,0,1]', 'v_pk_fma_f32 v[82:83], s[32:33], v[152:153], v[82:83] op_sel:[0,0,0] op_sel_hi:[1,0,1]', 'v_pk_fma_f32 v[72:73], s[30:31], v[152:153], v[72:73] op_sel:[0,0,0] op_sel_hi:[1,0,1]', 'v_pk_fma_f32 v[74:75], s[28:29], v[152:153], v[74:75] op_sel:[0,0,0] op_sel_hi:[1,0,1]', 'v_sub_f32 v84, |v77|, v154', 'v_alignbit_b32 v151, v151, v84, 31', 'v_alignbit_b32 v150, v150, v77, 31', 'v_mov_b32 v77, 0', 'v_pk_fma_f32 v[78:79], s[20:21], v[152:153], v[78:79] op_sel:[0,1,0] op_sel_hi:[1,1,1]', 'v_pk_fma_f32 v[80:81], s[22:23], v[152:153], v[80:81] op_sel:[0,1,0] op_sel_hi:[1,1,1]', 'v_pk_fma_f32 v[82:83], s[24:25], v[152:153], v[82:83] op_sel:[0,1,0] op_sel_hi:[1,1,1]', 'v_pk_fma_f32 v[72:73], s[24:25], v[152:153], v[72:73] op_sel:[1,1,0] op_sel_hi:[0,1,1]', 'v_pk_fma_f32 v[74:75], s[22:23], v[152:153], v[74:75] op_sel:[1,1,0] op_sel_hi:[0,1,1]', 'v_pk_fma_f32 v[76:77], s[20:21], v[152:153], v[76:77] op_sel:[1,1,0] op_sel_hi:[0,1,1]', 'v_sub_f32 v84, |v78|, v154', 'v_alignbit_b32 v151, v151, v84, 31', 'v_alignbit_b32 v150, v150, v78, 31', 'v_mov_b32 v78, 0', 'v_pk_fma_f32 v[80:81], s[28:29], v[152:153], v[80:81] op_sel:[1,0,0] op_sel_hi:[0,0,1]', 'v_pk_fma_f32 v[82:83], s[30:31], v[152:153], v[82:83] op_sel:[1,0,0] op_sel_hi:[0,0,1]', 'v_pk_fma_f32 v[78:79], s[26:27], v[152:153], v[78:79] op_sel:[0,0,0] op_sel_hi:[1,0,1]', 'v_pk_fma_f32 v[72:73], s[32:33], v[152:153], v[72:73] op_sel:[0,0,0] op_sel_hi:[1,0,1]', 'v_pk_fma_f32 v[74:75], s[30:31], v[152:153], v[74:75] op_sel:[0,0,0] op_sel_hi:[1,0,1]', 'v_pk_fma_f32 v[76:77], s[28:29], v[152:153], v[76:77] op_sel:[0,0,0] op_sel_hi:[1,0,1]', 'v_sub_f32 v84, |v79|, v154', 'v_alignbit_b32 v151, v151, v84, 31', 'v_alignbit_b32 v150, v150, v79, 31', 'v_mov_b32 v79, 0', 'v_pk_fma_f32 v[80:81], s[20:21], v[152:153], v[80:81] op_sel:[0,1,0] op_sel_hi:[1,1,1]', 'v_pk_fma_f32 v[82:83], s[22:23], v[152:153], v[82:83] op_sel:[0,1,0] op_sel_hi:[1,1,1]', 'v_pk_fma_f32 v[72:73], s[24:25], v[152:153], v[72:73] op_sel:[0,1,0] op_sel_hi:[1,1,1]', 'v_pk_fma_f32 v[74:75], s[24:25], v[152:153], v[74:75] op_sel:[1,1,0] op_sel_hi:[0,1,1]', 'v_pk_fma_f32 v[76:77], s[22:23], v[152:153], v[76:77] op_sel:[1,1,0] op_sel_hi:[0,1,1]', 'v_pk_fma_f32 v[78:79], s[20:21], v[152:153], v[78:79] op_sel:[1,1,0] op_sel_hi:[0,1,1]', 'v_sub_f32 v84, |v80|, v154', 'v_alignbit_b32 v151, v151, v84, 31', 'v_alignbit_b32 v150, v150, v80, 31', 'v_mov_b32 v80, 0', 'v_pk_fma_f32 v[82:83], s[28:29], v[152:153], v[82:83] op_sel:[1,0,0] op_sel_hi:[0,0,1]', 'v_pk_fma_f32 v[72:73], s[30:31], v[152:153], v[72:73] op_sel:[1,0,0] op_sel_hi:[0,0,1]', 'v_pk_fma_f32 v[80:81], s[26:27], v[152:153], v[80:81] op_sel:[0,0,0] op_sel_hi:[1,0,1]', 'v_pk_fma_f32 v[74:75], s[32:33], v[152:153], v[74:75] op_sel:[0,0,0] op_sel_hi:[1,0,1]', 'v_pk_fma_f32 v[76:77], s[30:31], v[152:153], v[76:77] op_sel:[0,0,0] op_sel_hi:[1,0,1]', 'v_pk_fma_f32 v[78:79], s[28:29], v[152:153], v[78:79] op_sel:[0,0,0] op_sel_hi:[1,0,1]', 'v_sub_f32 v84, |v81|, v154', 'v_alignbit_b32 v151, v151, v84, 31', 'v_alignbit_b32 v150, v150, v81, 31', 'v_mov_b32 v81, 0', 'v_pk_fma_f32 v[82:83], s[20:21], v[152:153], v[82:83] op_sel:[0,1,0] op_sel_hi:[1,1,1]', 'v_pk_fma_f32 v[72:73], s[22:23], v[152:153], v[72:73] op_sel:[0,1,0] op_sel_hi:[1,1,1]', 'v_pk_fma_f32 v[74:75], s[24:25], v[152:153], v[74:75] op_sel:[0,1,0] op_sel_hi:[1,1,1]', 'v_pk_fma_f32 v[76:77], s[24:25], v[152:153], v[76:77] op_sel:[1,1,0] op_sel_hi:[0,1,1]', 'v_pk_fma_f32 v[78:79], s[22:23], v[152:153], v[78:79] op_sel:[1,1,0] op_sel_hi:[0,1,1]', 'v_pk_fma_f32 v[80:81], s[20:21], v[152:153], v[80:81] op_sel:[1,1,0] op_sel_hi:[0,1,1]', 'v_sub_f32 v84, |v82|, v154', 'v_alignbit_b32 v151, v151, v84, 31', 'v_alignbit_b32 v150, v150, v82, 31', 'v_mov_b32 v82, 0', 'v_pk_fma_f32 v[72:73], s[28:29], v[152:153], v[72:73] op_sel:[1,0,0] op_sel_hi:[0,0,1]', 'v_pk_fma_f32 v[74:75], s[30:31], v[152:153], v[74:75] op_sel:[1,0,0] op_sel_hi:[0,0,1]', 'v_pk_fma_f32 v[82:83], s[26:27], v[152:153], v[82:83] op_sel:[0,0,0] op_sel_hi:[1,0,1]', 'v_pk_fma_f32 v[76:77], s[32:33], v[152:153], v[76:77] op_sel:[0,0,0] op_sel_hi:[1,0,1]', 'v_pk_fma_f32 v[78:79], s[30:31], v[152:153], v[78:79] op_sel:[0,0,0] op_sel_hi:[1,0,1]', 'v_pk_fma_f32 v[80:81], s[28:29], v[152:153], v[80:81] op_sel:[0,0,0] op_sel_hi:[1,0,1]', 'v_sub_f32 v84, |v83|, v154', 'v_alignbit_b32 v151, v151, v84, 31', 'v_alignbit_b32 v150, v150, v83, 31', 'v_mov_b32 v83, 0', 'v_pk_fma_f32 v[72:73], s[20:21], v[152:153], v[72:73] op_sel:[0,1,0] op_sel_hi:[1,1,1]', 'v_pk_fma_f32 v[74:75], s[22:23], v[152:153], v[74:75] op_sel:[0,1,0] op_sel_hi:[1,1,1]', 'v_pk_fma_f32 v[76:77], s[24:25], v[152:153], v[76:77] op_sel:[0,1,0] op_sel_hi:[1,1,1]', 'v_pk_fma_f32 v[78:79], s[24:25], v[152:153], v[78:79] op_sel:[1,1,0] op_sel_hi:[0,1,1]', 'v_pk_fma_f32 v[80:81], s[22:23], v[152:153], v[80:81] op_sel:[1,1,0] op_sel_hi:[0,1,1]', 'v_pk_fma_f32 v[82:83], s[20:21], v[152:153], v[82:83] op_sel:[1,1,0] op_sel_hi:[0,1,1]', 'v_sub_f32 v84, |v72|, v154', 'v_alignbit_b32 v151, v151, v84, 31', 'v_alignbit_b32 v150, v150, v72, 31', 'v_mov_b32 v72, 0']
# the scalar direct-form stream of fir_sign_kernel per sample: 6 adds, 1 mul + 5 fmac with SGPR taps, |y|-eps, two alignbits, max3 every other sample
sc = []
for smp in range(12):
    for q in range(6):
        sc.append(f"v_add_f32 v{140+q}, v{100+(smp+q)%32}, v{100+(smp+11-q)%32}")
    sc.append("v_mul_f32 v146, s20, v140")
    for q in range(1, 6):
        sc.append(f"v_fmac_f32 v146, s{20+q}, v{140+q}")
    sc.append("v_sub_f32 v147, |v146|, v154")
    sc.append("v_alignbit_b32 v151, v151, v147, 31")
    sc.append("v_alignbit_b32 v150, v150, v146, 31")
    if smp % 2: sc.append(f"v_max3_i32 v148, v148, v{100+smp}, v{101+smp}")
modes["scalar12_stream_12samples"] = ["s_mov_b32 s%d, 0x3f000000" % r for r in range(20, 26)] + sc
# integer / bit ops used by the PLL core
modes["bfi32"] = [f"v_bfi_b32 v{100+i}, v{100+i}, %0, %0" for i in range(32)]
modes["bfe32"] = [f"v_bfe_i32 v{100+i}, v{100+i}, 3, 1" for i in range(32)]
modes["ashr32"] = [f"v_ashrrev_i32 v{100+i}, 31, v{100+i}" for i in range(32)]
modes["addco32"] = [f"v_add_co_u32 v{100+i}, vcc, v{100+i}, %0" for i in range(32)]
modes["xor32"] = [f"v_xor_b32 v{100+i}, v{100+i}, %0" for i in range(32)]
modes["cndmask32"] = [f"v_cndmask_b32 v{100+i}, v{100+i}, %0, vcc" for i in range(32)]
modes["cmp32"] = [f"v_cmp_gt_f32 vcc, v{100+i}, %0" for i in range(32)]
# the PLL recurrence itself: one dependent chain, 6 instr / sample, 8 samples
pl = []
for i in range(8):
    pl += [f"v_bfe_i32 v101, v100, {i}, 1", "v_ashrrev_i32 v102, 31, v103",
           "v_addc_co_u32 v104, vcc, v104, v104, vcc",
           "v_bfi_b32 v105, v102, %0, %0", "v_bfi_b32 v105, v101, v105, %0",
           "v_add_co_u32 v103, vcc, v103, v105"]
modes["pll_chain"] = pl

modes["cndmask_e64_s"] = ["s_mov_b64 s[20:21], -1"] + [f"v_cndmask_b32_e64 v{100+i}, v{100+i}, %0, s[20:21]" for i in range(32)]
modes["cndmask_01_vcc"] = [f"v_cndmask_b32_e64 v{100+i}, 0, 1, vcc" for i in range(32)]
modes["lshl_or32"] = [f"v_lshl_or_b32 v{100+i}, v{100+i}, 1, %0" for i in range(32)]
modes["lshl_add_u64"] = [f"v_lshl_add_u64 v[{100+2*i}:{101+2*i}], v[{100+2*i}:{101+2*i}], 1, v[{100+2*i}:{101+2*i}]" for i in range(16)]
modes["cvt32"] = [f"v_cvt_f32_i32 v{100+i}, v{100+i}" for i in range(32)]
modes["max3_32"] = [f"v_max3_i32 v{100+i}, v{100+i}, %0, %0" for i in range(32)]
f1 = []; f2 = []; f3 = []
for i in range(16):
    f1 += [f"v_cmp_lt_f32 vcc, 0, v{100+i}", "v_cndmask_b32_e64 v140, 0, 1, vcc", "v_lshl_or_b32 v141, v141, 1, v140",
           f"v_cmp_le_f32 s[20:21], |v{100+i}|, %0", "v_cndmask_b32_e64 v142, 0, 1, s[20:21]", "v_lshl_or_b32 v143, v143, 1, v142"]
    f2 += [f"v_cmp_lt_f32 vcc, 0, v{100+i}", "s_nop 0", "v_addc_co_u32 v141, vcc, v141, v141, vcc",
           f"v_cmp_le_f32 vcc, |v{100+i}|, %0", "s_nop 0", "v_addc_co_u32 v143, vcc, v143, v143, vcc"]
    # interleaved: cmp A ; cmp B(sgpr) ; addc A ; addc B
    f3 += [f"v_cmp_lt_f32 vcc, 0, v{100+i}", f"v_cmp_le_f32 s[20:21], |v{100+i}|, %0",
           "v_addc_co_u32 v141, vcc, v141, v141, vcc", "v_addc_co_u32 v143, s[22:23], v143, v143, s[20:21]"]
modes["flags_cur"] = f1
modes["flags_addc"] = f2
modes["flags_addc_il"] = f3
src = ['#include <hip/hip_runtime.h>', '#include <cstdio>', 'typedef float f32x2 __attribute__((ext_vector_type(2)));']
names = list(modes)
for k, (name, ins) in enumerate(modes.items()):
    body = "\\n\\t".join(ins)
    src.append(f'''__global__ __launch_bounds__(256) void k{k}(float *out, int iters, float b, int lim) {{
  f32x2 bb = {{b, b}};
  if ((int) (threadIdx.x & 63) >= lim) return;
  for (int it = 0; it < iters; ++it)
    asm volatile("{body}" :: "v"(b), "v"(bb) : {clob(72, 180)}, "vcc", "s20","s21","s22","s23","s24","s25","s26","s27","s28","s29","s30","s31","s32","s33");
  out[blockIdx.x * blockDim.x + threadIdx.x] = b;
}}''')
src.append('typedef void (*kern_t)(float*, int, float, int);')
src.append('int main() { float *d; (void) hipMalloc(&d, 8192 * 256 * 4); const int iters = 20000;')
src.append('  kern_t ks[] = {' + ",".join(f"k{k}" for k in range(len(names))) + '};')
src.append('  const char *nm[] = {' + ",".join(f'"{n}"' for n in names) + '};')
src.append('  const int ninstr[] = {' + ",".join(str(len(modes[n])) for n in names) + '};')
src.append('''  for (int lim = 64; lim >= 64; lim /= 2) for (int wps = 1; wps <= 5; ++wps) for (int m = 0; m < (int)(sizeof(ks)/sizeof(ks[0])); ++m) {
    hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    hipLaunchKernelGGL(ks[m], dim3(256 * wps), dim3(256), 0, 0, d, 10, 1.0f, lim);
    (void) hipEventRecord(e0);
    hipLaunchKernelGGL(ks[m], dim3(256 * wps), dim3(256), 0, 0, d, iters, 1.0f, lim);
    (void) hipEventRecord(e1); (void) hipEventSynchronize(e1);
    float ms; (void) hipEventElapsedTime(&ms, e0, e1);
    printf("lanes=%d waves/SIMD=%d %-14s %8.3f ms  %.3f ns per wave-instruction per SIMD\\n", lim, wps, nm[m], ms, ms * 1e6 / iters / wps / ninstr[m]);
  }
  return 0; }''')
open("valu_rate.hip", "w").write("\n".join(src))
