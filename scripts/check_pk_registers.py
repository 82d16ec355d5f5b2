"""fir_sign_pk.hip keeps its accumulator ring in fixed VGPRs above the compiler's budget.  The budget attribute is
not a hard limit, so this compiles the file to ISA and checks that no compiler-generated instruction (anything outside
the ASMSTART/ASMEND blocks) names a register of the ring or above.  Exit status 0 = clean."""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gnuais_amd", "csrc", "fir_sign_pk.hip")
inc = open(os.path.join(root, "gnuais_amd", "csrc", "fir_sign_pk_asm.inc")).read()
base = {nc: int(re.search(rf"#define PK{nc}_VGPR_BASE (\d+)", inc).group(1)) for nc in (12, 48)}
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "pk.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
                           "-mllvm", "-pragma-unroll-threshold=200000", "-fno-slp-vectorize", "-S", "--cuda-device-only",
                           "-w", src, "-o", out])
    text = open(out).read().splitlines()
bad = 0
for nc in (12, 48):
    inside = in_asm = False
    top = -1
    for line in text:
        if re.match(rf"^_ZN.*fir_sign_pk{nc}_kernel.*:", line): inside = True
        if not inside: continue
        if "ASMSTART" in line: in_asm = True
        elif "ASMEND" in line: in_asm = False
        elif not in_asm and not line.lstrip().startswith((";", ".")):
            for m in re.finditer(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]", line):
                top = max(top, int(m.group(1) or m.group(3)))
        if "s_endpgm" in line: break
    print(f"fir_sign_pk{nc}_kernel: compiler code uses v0..v{top}, the ring starts at v{base[nc]}")
    bad += top >= base[nc]
sys.exit(1 if bad else 0)
