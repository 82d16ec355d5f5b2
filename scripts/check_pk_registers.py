"""fir_sign_pk.hip keeps its accumulator ring (and, for 40 and 48 taps, its tap pairs) in fixed VGPRs above what the compiler
uses.  The asm statements list them as clobbers (so the compiler keeps nothing of its own there ACROSS a statement and
counts them into the wave's allocation), but a clobber list cannot protect state BETWEEN two statements, so the ISA is
scanned: (1) no compiler-generated instruction (anything outside the ASMSTART/ASMEND blocks) of the two kernels may name
a register the streams own (base .. the highest clobbered one; a register ABOVE them is the compiler's to use -- it
parks spilled SGPRs there when v0..base-1 are taken); (2) the kernel descriptor must allocate the wave every register the streams name.  A
violation of (1) shows as rare wrong sign bits, of (2) as corrupted neighbours, neither as a crash -- so the Makefile
runs this on the ISA of the very object it builds (same flags) and fails the build on a finding.

usage: check_pk_registers.py [file.s]     without an argument the file is compiled here with the Makefile's flags
Exit status 0 = clean."""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gnuais_amd", "csrc", "fir_sign_pk.hip")
inc = open(os.path.join(root, "gnuais_amd", "csrc", "fir_sign_pk_asm.inc")).read()
base = {nc: int(re.search(rf"#define PK{nc}_VGPR_BASE (\d+)", inc).group(1)) for nc in (40, 48)}
# the highest register the generated streams name: the wave must have been allocated at least that many
ring_top = {nc: max(int(r) for r in re.findall(r'"v(\d+)"', re.search(rf"#define PK{nc}_CLOBBERS (.*)", inc).group(1)))
            for nc in (40, 48)}
if len(sys.argv) > 1:
    text = open(sys.argv[1]).read().splitlines()
else:
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "pk.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
                               "-mllvm", "-pragma-unroll-threshold=200000", "-fno-slp-vectorize", "-S", "--cuda-device-only",
                               "-w", src, "-o", out])
        text = open(out).read().splitlines()
bad = 0
for nc in (40, 48):
    inside = in_asm = found = False
    top = -1                     # highest register of the compiler's own code below the streams' block
    above = -1                   # ... and above it
    where = None
    for no, line in enumerate(text, 1):
        if re.match(rf"^_ZN.*fir_sign_pk{nc}_kernel.*:", line): inside = found = True
        if not inside: continue
        if "ASMSTART" in line: in_asm = True
        elif "ASMEND" in line: in_asm = False
        elif not in_asm and not line.lstrip().startswith((";", ".")):
            for m in re.finditer(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]", line):
                lo = int(m.group(1) or m.group(2))
                r = int(m.group(1) or m.group(3))
                if lo > ring_top[nc]: above = max(above, r)
                elif r > top: top, where = r, (no, line.strip())
        if "s_endpgm" in line: break
    if not found:
        print(f"fir_sign_pk{nc}_kernel: not found in the ISA")
        bad += 1
        continue
    ok = top < base[nc]
    print(f"fir_sign_pk{nc}_kernel: compiler code uses v0..v{top}" + (f" and v{ring_top[nc] + 1}..v{above}" if above >= 0 else "") +
          f", the streams own v{base[nc]}..v{ring_top[nc]}" +
          ("" if ok else f"  <-- VIOLATION at line {where[0]}: {where[1]}"))
    bad += not ok
    # the kernel descriptor: registers the hardware gives a wave.  The streams' registers count only because the asm
    # statements list them as clobbers; a wave allocated fewer would write into its neighbours' registers
    alloc = None
    in_desc = False
    for line in text:
        if re.match(rf"^\s*\.amdhsa_kernel\s+_ZN.*fir_sign_pk{nc}_kernel", line): in_desc = True
        elif in_desc and ".amdhsa_next_free_vgpr" in line:
            alloc = int(line.split()[-1]); break
        elif in_desc and ".end_amdhsa_kernel" in line: break
    ok2 = alloc is not None and alloc > ring_top[nc]
    print(f"fir_sign_pk{nc}_kernel: the wave is allocated v0..v{(alloc or 0) - 1}, the streams reach v{ring_top[nc]}" +
          ("" if ok2 else "  <-- VIOLATION: the streams' registers are not part of the wave's allocation"))
    bad += not ok2
sys.exit(1 if bad else 0)
