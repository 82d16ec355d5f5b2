#!/usr/bin/env python3
"""Build-time check of what a kernel's descriptor asks the hardware for (run by gnuais_amd/csrc/Makefile on the ISA of the
object it is about to build).  Two things went unnoticed for rounds because nothing fails when they happen:
  * scratch: the deframer kept two state words in private memory (scratch loads / stores inside its event loop, beside a
    FIR that saturates the memory pipeline);
  * a padded register request: K3 used 59 VGPRs and asked for 104 -- the compiler derives an occupancy from a workgroup's
    STATIC shared memory and raises NumVGPRsForWavesPerEU to match it, and a wave that asks for 104 registers waits for a
    FIR wave to retire where one that asks for 72 fits.
usage: check_kernel_resources.py file.s name_substring [name_substring ...]   (every kernel whose symbol contains one)"""
import re
import sys


def kernels(text):
    name = None
    info = {}
    for line in text.splitlines():
        m = re.match(r"\s+\.amdhsa_kernel\s+(\S+)", line)
        if m:
            name = m.group(1)
            info[name] = {}
            continue
        m = re.match(r";\s+(NumVgprs|NumAgprs|ScratchSize|NumVGPRsForWavesPerEU):\s+(\d+)", line)
        if m and name:
            info[name][m.group(1)] = int(m.group(2))
    return info


def main():
    path, wanted = sys.argv[1], sys.argv[2:]
    bad = 0
    seen = 0
    for name, k in kernels(open(path).read()).items():
        if not any(w in name for w in wanted) or "NumVgprs" not in k:
            continue
        seen += 1
        used = k["NumVgprs"] + k.get("NumAgprs", 0)
        asked = k.get("NumVGPRsForWavesPerEU", used)
        line = f"{name[:70]}: {used} registers in use, {asked} requested, {k.get('ScratchSize', 0)} bytes of scratch"
        if k.get("ScratchSize", 0) > 0:
            print(line + "  <-- scratch memory in a kernel of the chain")
            bad += 1
        elif asked > used + 8:
            print(line + "  <-- padded request (static shared memory? make it dynamic)")
            bad += 1
        else:
            print(line)
    if not seen:
        print(f"{path}: no kernel matching {wanted}")
        return 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
