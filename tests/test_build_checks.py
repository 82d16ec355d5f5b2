"""scripts/check_kernel_resources.py -- the build-time check of what the chain's kernels ask the hardware for -- on
hand-made ISA listings: a clean kernel passes, scratch fails, a padded register request fails, an unknown name fails."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "scripts", "check_kernel_resources.py")


def listing(name, used, asked, scratch, agpr=0):
    return (f"\t.amdhsa_kernel {name}\n\t\t.amdhsa_next_free_vgpr {asked}\n\t.end_amdhsa_kernel\n"
            f"; Kernel info:\n; NumVgprs: {used}\n; NumAgprs: {agpr}\n; TotalNumVgprs: {used + agpr}\n"
            f"; ScratchSize: {scratch}\n; NumVGPRsForWavesPerEU: {asked}\n; Occupancy: 4\n")


def run(tmp_path, text, *names):
    p = tmp_path / "k.s"
    p.write_text(text)
    r = subprocess.run([sys.executable, SCRIPT, str(p), *names], capture_output=True, text=True)
    return r.returncode, r.stdout


def test_clean_kernel_passes(tmp_path):
    rc, out = run(tmp_path, listing("_ZN6gnuais15hdlc_crc_kernelE", 68, 68, 0), "hdlc_crc_kernel")
    assert rc == 0 and "68 registers in use, 68 requested" in out


def test_scratch_fails(tmp_path):
    rc, out = run(tmp_path, listing("_ZN6gnuais18hdlc_events_kernelILi16EE", 119, 119, 12), "hdlc_events_kernelILi16E")
    assert rc == 1 and "scratch" in out


def test_padded_request_fails(tmp_path):
    rc, out = run(tmp_path, listing("_ZN6gnuais15hdlc_crc_kernelE", 59, 97, 0), "hdlc_crc_kernel")
    assert rc == 1 and "padded request" in out


def test_only_the_named_kernels_are_judged(tmp_path):
    text = listing("_ZN6gnuais15hdlc_crc_kernelE", 68, 68, 0) + listing("_ZN6gnuais19hdlc_deframe_kernelE", 72, 72, 96)
    assert run(tmp_path, text, "hdlc_crc_kernel")[0] == 0
    assert run(tmp_path, text, "hdlc_deframe_kernel")[0] == 1
    assert run(tmp_path, text, "no_such_kernel")[0] == 1


def test_the_makefile_runs_the_check_on_the_chain_kernels():
    mk = open(os.path.join(ROOT, "gnuais_amd", "csrc", "Makefile")).read()
    for obj in ("hdlc_events.s", "hdlc_crc.s", "pll_h3.s"):
        assert f"$(CHECK_RES) $(BUILD)/{obj}" in mk
