"""CPU-side checks of the drop-in boundary: the HIP library builds for gfx950,
loads, and exports every symbol include/gnuais_hip.h declares.  No compute."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    so = os.path.join(ROOT, "gnuais_amd", "libgnuais_hip.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "gnuais_amd", "csrc")])
    from gnuais_amd import lib as L
    return L


def header_symbols():
    text = open(os.path.join(ROOT, "include", "gnuais_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gnuais_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    handle = lib.load()
    declared = header_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in gnuais_hip.h but not exported"
    assert sorted(lib.SYMBOLS) == declared, "python binding table out of sync with the header"


def test_frame_record_layout(lib):
    assert lib.FRAME_DTYPE.itemsize == 64
    assert lib.FRAME_DTYPE.fields["payload"][1] == 8
    assert lib.FRAME_DTYPE.fields["nbits"][1] == 62


def test_default_taps_are_the_reference_table(lib):
    from gnuais_amd import params
    t = lib.default_taps()
    assert np.array_equal(t.view(np.uint32), params.taps_48k().view(np.uint32))
    g = np.load(os.path.join(ROOT, "tests", "golden", "fir_kat.npz"))
    assert np.array_equal(t.view(np.uint32), g["taps"])


def test_no_silent_cpu_fallback(lib):
    """Without a HIP device the product path must fail loudly."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from gnuais_amd import ReceiverBatch
    with pytest.raises(lib.GnuaisError) as e:
        ReceiverBatch(2)
    assert e.value.code == lib.E_HIP


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under gnuais_amd/ or include/ may name it."""
    bad = []
    for base in ("gnuais_amd", "include"):
        for d, _, files in os.walk(os.path.join(ROOT, base)):
            if "build" in d.split(os.sep):
                continue
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".c", ".cpp", "Makefile")):
                    txt = open(os.path.join(d, f), errors="ignore").read()
                    if re.search(r"ais_oracle|oracle_lib|libgnuais_ref|oracle/", txt):
                        bad.append(os.path.join(d, f))
    assert not bad, bad


@pytest.mark.parametrize("name", ["receiver_hip.c", "sinks_batch.c", "protodec_hip.c"])
def test_dropin_c_file_compiles_standalone_and_in_tree(tmp_path, name):
    """The host-side C files that a gnuais tree adds (the receiver drop-in, the batched sink
    adapter, the reference-named filter_* / protodec_* shims) are plain C against the public headers, and (where the reference tree is present)
    against the reference's own headers."""
    src = os.path.join(ROOT, "gnuais_amd", "csrc", name)
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=gnu11", "-Wall", "-Werror", "-I", inc, "-c", src, "-o",
                           str(tmp_path / "a.o")])
    ref = "/root/reference"
    if os.path.isdir(os.path.join(ref, "src")):
        subprocess.check_call(["cmake", f"-DREF={ref}", f"-DOUT={tmp_path}", "-P",
                               os.path.join(ROOT, "oracle", "gen_config.cmake")],
                              stdout=subprocess.DEVNULL)
        subprocess.check_call(["gcc", "-std=gnu11", "-w", "-fcommon", "-DGNUAIS_TREE", "-I", inc,
                               "-I", str(tmp_path), "-I", os.path.join(ref, "src"), "-c", src,
                               "-o", str(tmp_path / "b.o")])


def test_public_struct_layout_matches_reference_sizes():
    """SURVEY section 4 probe: sizeof(struct receiver) = 56, demod_state_t = 144."""
    import tempfile
    code = ('#include <stdio.h>\n#include "gnuais_receiver_abi.h"\n'
            'int main(void){printf("%zu %zu\\n", sizeof(struct receiver), '
            'sizeof(struct demod_state_t));return 0;}\n')
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(code)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"),
                               "-o", os.path.join(d, "s")])
        out = subprocess.check_output([os.path.join(d, "s")]).decode().split()
    assert out == ["56", "144"]


def test_packed_fir_keeps_its_ring_out_of_the_compilers_registers():
    """fir_sign_pk.hip holds its accumulator ring in fixed VGPRs above what the compiler uses for the rest of the kernel;
    amdgpu_num_vgpr is only a hint, so the generated ISA is scanned: no compiler-generated instruction may name a
    register of the ring (scripts/check_pk_registers.py; a violation shows up as rare wrong signs, not as a crash)."""
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_pk_registers.py")], capture_output=True, timeout=600)
    assert r.returncode == 0, r.stdout.decode() + r.stderr.decode()


def test_dropin_fatal_handler_runs_before_abort(tmp_path):
    """receiver_hip.c has no CPU path to fall back to: without a usable device init_receiver() ends the program
    (receiver.c:104-105 does the same on its own fatal errors).  A host's handler (gnuais_receiver_on_fatal) is
    called with the message first.  Runs with every device hidden, so it needs no GPU."""
    import signal
    code = r'''
#include <stdio.h>
#include "gnuais_receiver_abi.h"
void protodec_initialize(struct demod_state_t *d, struct serial_state_t *s, struct ipc_state_t *i, char c) { (void) d; (void) s; (void) i; (void) c; }
void protodec_getdata(int n, struct demod_state_t *d) { (void) n; (void) d; }
static void handler(const char *m) { printf("handler: %s\n", m); fflush(stdout); }
int main(void) { gnuais_receiver_on_fatal(handler); init_receiver('A', 2, 0, NULL, NULL); printf("not reached\n"); return 0; }
'''
    src = tmp_path / "f.c"
    src.write_text(code)
    exe = tmp_path / "f"
    subprocess.check_call(["gcc", "-std=gnu11", "-I", os.path.join(ROOT, "include"), str(src),
                           os.path.join(ROOT, "gnuais_amd", "csrc", "receiver_hip.c"),
                           "-L" + os.path.join(ROOT, "gnuais_amd"), "-lgnuais_hip", "-lpthread",
                           "-Wl,-rpath," + os.path.join(ROOT, "gnuais_amd"), "-o", str(exe)])
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
    p = subprocess.run([str(exe)], capture_output=True, env=env, timeout=120)
    assert p.returncode == -signal.SIGABRT, (p.returncode, p.stderr.decode())
    out = p.stdout.decode()
    assert out.startswith("handler: gnuais-hip: ") and "not reached" not in out
