"""Generates gnuais_amd/csrc/fir_sign_pk_asm.inc: the instruction streams of fir_sign_pk.hip's pair steps.

The accumulator ring lives in FIXED registers above the compiler's own budget (amdgpu_num_vgpr(B)): slot s of the
ring is v[B + s].  inline-asm operands cannot name one half of a register pair, and what the compiler makes of
`pair[1] = 0` on a vector value is two moves and a copy per output; with fixed registers an output is read where it
lies and cleared with one v_mov_b32.  One macro per ring phase P (even): 2 x NC/2 v_pk_fma_f32, and after each half
the flags of the output that has just completed (|y| - eps -> amb, sign -> neg, both through v_alignbit_b32).

Operands of a full step:   %0 neg (+v)  %1 amb (+v)  %2 sample pair (v)  %3 eps (v)  %4.. tap pairs
  NC = 12: %4..%6 = E[0..2], %7..%10 = O[0..3]   (SGPR pairs)
  NC = 40, 48: no tap operands: O[j] (NC/4 + 1 pairs) sits in v[B+NC+2+2j : +1], E[j] (NC/4 pairs) right above them
  (PK40_LOAD_O / PK48_LOAD_O put them there)
A warm-up step (no outputs): %0 sample pair, %1.. tap pairs.
Order inside a half: the pair whose slot is read next goes first and the pair that was just cleared goes last, so that
no packed instruction sits next to an instruction that depends on it (the assembler adds no wait states in inline asm).
"""
import os
NCS = (40, 48)
BASE = {12: 72, 40: int(os.environ.get("PK40_BASE", "44")), 48: int(os.environ.get("PK48_BASE", "44"))}      # first register of the ring per instantiation
GAP = 8     # what the compiler's own code may use ends at least GAP registers below the ring (PK*_VGPR_BUDGET, for the
            # record: nothing enforces it in the language -- scripts/check_pk_registers.py, run by the Makefile on every
            # build, scans the generated ISA for compiler code that touches the ring and checks the wave's allocation)

def e_base(nc):                 # first register of E[0] (NC >= 40: the tap pairs live in registers)
    return BASE[nc] + nc + 2 + 2 * (nc // 4 + 1)

def gen(nc, warm):
    B = BASE[nc]
    npairs = nc // 2
    e_first = 1 if warm else 4
    o_first = e_first + nc // 4                      # NC = 12 only
    x = "%0" if warm else "%2"
    out = {}
    for P in range(0, nc, 2):
        for half_only in ((False, True) if (warm and P == 0) else (False,)):
            lines = ["s_nop 1"] if warm else []       # a warm-up step may follow the v_cvt that made its sample pair
            def acc(a): return f"v[{B + a}:{B + a + 1}]"
            def flags(slot):
                if warm:
                    return [f"v_mov_b32 v{B + slot}, 0"]
                return [f"v_sub_f32 v{B + nc}, |v{B + slot}|, %3",
                        f"v_alignbit_b32 %1, %1, v{B + nc}, 31",
                        f"v_alignbit_b32 %0, %0, v{B + slot}, 31",
                        f"v_mov_b32 v{B + slot}, 0"]
            # first half: x_P, tap pairs O(k)
            # pair P/2: its low half was cleared by the last instruction of the step before and its high half is read
            # right after this group -- third, away from both (a v_pk_fma_f32 next to an instruction it depends on, or
            # that depends on it, is not interlocked reliably: seen as wrong sums when a v_cvt sat right before one)
            order1 = [(P // 2 + 1 + i) % npairs for i in range(npairs)]
            order1.remove(P // 2)
            order1.insert(2, P // 2)
            if not half_only:
                for kk in order1:
                    a = 2 * kk
                    k = (P - a) % nc
                    j = k // 2
                    if j <= nc // 4:
                        src, swap = j, 0
                    else:
                        src, swap = nc // 2 - j, 1
                    tap = f"%{o_first + src}" if nc == 12 else f"v[{B + nc + 2 + 2 * src}:{B + nc + 3 + 2 * src}]"
                    lines.append(f"v_pk_fma_f32 {acc(a)}, {tap}, {x}, {acc(a)} op_sel:[{swap},0,0] op_sel_hi:[{1 - swap},0,1]")
            lines += flags((P + 1) % nc)
            # second half: x_P+1, reversed E(k); pair (P+2)/2 first, pair P/2 (just cleared) last
            order2 = [((P + 2) // 2 + i) % npairs for i in range(npairs)]
            for kk in order2:
                a = 2 * kk
                k = (P - a) % nc
                j = k // 2
                if j < nc // 4:
                    src, swap = j, 1
                else:
                    src, swap = nc // 2 - 1 - j, 0
                tap = f"%{e_first + src}" if nc == 12 else f"v[{e_base(nc) + 2 * src}:{e_base(nc) + 1 + 2 * src}]"
                lines.append(f"v_pk_fma_f32 {acc(a)}, {tap}, {x}, {acc(a)} op_sel:[{swap},1,0] op_sel_hi:[{1 - swap},1,1]")
            lines += flags((P + 2) % nc)
            name = f"PK{nc}_{'WARM' if warm else 'STEP'}_{P}" + ("_HALF" if half_only else "")
            out[name] = lines
    return out

def clobbers(nc):
    B = BASE[nc]
    regs = list(range(B, B + nc + 2))
    if nc != 12:
        regs += list(range(B + nc + 2, B + nc + 2 + 2 * (nc // 4 + 1) + 2 * (nc // 4)))
    return ", ".join(f'"v{r}"' for r in regs)

path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gnuais_amd", "csrc", "fir_sign_pk_asm.inc")
with open(path, "w") as f:
    f.write("// GENERATED by scripts/gen_fir_pk_asm.py -- do not edit.  See that script for the layout.\n")
    for nc in NCS:
        B = BASE[nc]
        f.write(f"#define PK{nc}_VGPR_BASE {B}\n")
        f.write(f"#define PK{nc}_VGPR_BUDGET {B - GAP}\n")
        f.write(f"#define PK{nc}_CLOBBERS {clobbers(nc)}\n")
        zero = "\\n\\t".join(f"v_mov_b32 v{B + s}, 0" for s in range(nc + 1))
        f.write(f'#define PK{nc}_ZERO "{zero}"\n')
        for warm in (True, False):
            for name, lines in gen(nc, warm).items():
                f.write(f'#define {name} "' + "\\n\\t".join(lines) + '"\n')
    # NC = 40, 48: the tap pairs into their registers, operands %0.. = O[0..NC/4], then E[0..NC/4-1] (SGPR pairs)
    for nc in NCS:
        if nc == 12:
            continue
        B, no, ne = BASE[nc], nc // 4 + 1, nc // 4
        ld = [f"v_mov_b64 v[{B + nc + 2 + 2 * j}:{B + nc + 3 + 2 * j}], %{j}" for j in range(no)]
        ld += [f"v_mov_b64 v[{e_base(nc) + 2 * j}:{e_base(nc) + 1 + 2 * j}], %{no + j}" for j in range(ne)]
        f.write(f'#define PK{nc}_LOAD_O "' + "\\n\\t".join(ld) + '"\n')
    # C++ dispatch: one function template per kind, the ring phase as template argument
    for nc in NCS:
        ne, no = nc // 4, nc // 4 + 1
        taps_e = ", ".join(f'"s"(tp.E[{j}])' for j in range(ne))
        taps_o = ", ".join(f'"s"(tp.O[{j}])' for j in range(no))
        taps = (taps_e + ", " + taps_o) if nc == 12 else ""
        tapsc = (", " + taps) if taps else ""
        f.write(f"template <int P> __device__ __forceinline__ void pk{nc}_step(uint32_t &neg, uint32_t &amb, pk_f2 x, float eps, const PkTaps<{nc}> &tp)\n{{\n")
        for i, P in enumerate(range(0, nc, 2)):
            f.write(f"    {'if' if i == 0 else 'else if'} constexpr (P == {P}) asm volatile(PK{nc}_STEP_{P} : \"+v\"(neg), \"+v\"(amb) : \"v\"(x), \"v\"(eps){tapsc} : PK{nc}_CLOBBERS);\n")
        f.write("}\n")
        f.write(f"template <int P, bool HALF> __device__ __forceinline__ void pk{nc}_warm(pk_f2 x, const PkTaps<{nc}> &tp)\n{{\n")
        f.write(f"    if constexpr (HALF) asm volatile(PK{nc}_WARM_0_HALF :: \"v\"(x){tapsc} : PK{nc}_CLOBBERS);\n")
        for P in range(0, nc, 2):
            f.write(f"    else if constexpr (P == {P}) asm volatile(PK{nc}_WARM_{P} :: \"v\"(x){tapsc} : PK{nc}_CLOBBERS);\n")
        f.write("}\n")
    for nc in NCS:
        if nc == 12:
            continue
        no, ne = nc // 4 + 1, nc // 4
        f.write(f"__device__ __forceinline__ void pk{nc}_load_o(const PkTaps<{nc}> &tp)\n{{\n    asm volatile(PK{nc}_LOAD_O :: " +
                ", ".join([f'\"s\"(tp.O[{j}])' for j in range(no)] + [f'\"s\"(tp.E[{j}])' for j in range(ne)]) + f" : PK{nc}_CLOBBERS);\n}}\n")
    # one name for all instantiations
    f.write("template <int NC, int P> __device__ __forceinline__ void pk_step(uint32_t &neg, uint32_t &amb, pk_f2 x, float eps, const PkTaps<NC> &tp)\n{\n")
    for i, nc in enumerate(NCS):
        f.write(f"    {'if' if i == 0 else 'else if'} constexpr (NC == {nc}) pk{nc}_step<P>(neg, amb, x, eps, tp);\n")
    f.write("}\n")
    f.write("template <int NC, int P, bool HALF> __device__ __forceinline__ void pk_warm(pk_f2 x, const PkTaps<NC> &tp)\n{\n")
    for i, nc in enumerate(NCS):
        f.write(f"    {'if' if i == 0 else 'else if'} constexpr (NC == {nc}) pk{nc}_warm<P, HALF>(x, tp);\n")
    f.write("}\n")
    f.write("template <int NC> __device__ __forceinline__ void pk_zero_ring(const PkTaps<NC> &tp)\n{\n")
    for i, nc in enumerate(NCS):
        f.write(f"    {'if' if i == 0 else 'else if'} constexpr (NC == {nc}) {{ asm volatile(PK{nc}_ZERO ::: PK{nc}_CLOBBERS);" + (f" pk{nc}_load_o(tp);" if nc != 12 else "") + " }\n")
    f.write("}\n")
print("wrote", path)
