"""The host-side C / C++ of the drop-in under AddressSanitizer + UndefinedBehaviorSanitizer.

`make -C gnuais_amd/csrc asan` builds receiver_hip.c, protodec_hip.c, sinks_batch.c, wavio.c and nmea.cpp with
-fsanitize=address,undefined -fno-sanitize-recover=all into tests/c/asan_host.bin.  The two drop-in files run over a
test double of the C ABI on the CPU oracle (tests/c/fake_gnuais_hip.c) -- what they do with tables, locks, rounds and
queues does not depend on who computes the frames -- the others as they are.  A sanitizer finding ends the program
with a non-zero status; what the program wrote is compared with the golden vectors as well, so this is also a
functional test of the drop-in's host logic on the CPU."""
import os
import subprocess

import numpy as np
import pytest

from gnuais_amd import io
from oracle_lib import FRAME_DTYPE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
EXE = os.path.join(ROOT, "tests", "c", "asan_host.bin")


def poly_hash(values):
    """sum = sum * 1000003 + v (mod 2^64), what asan_host_main.c prints"""
    h = 0
    for v in values:
        h = (h * 1000003 + int(v)) & 0xFFFFFFFFFFFFFFFF
    return h


def frame_line(name, f):
    n = int(f["nbits"])
    return f"ch {name} bits {n} payload " + bytes(f["payload"][: n // 8]).hex()


@pytest.fixture(scope="module")
def run(tmp_path_factory):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "gnuais_amd", "csrc"), "asan"])
    d = tmp_path_factory.mktemp("asan")
    g = np.load(os.path.join(G, "chain_48k.npz"))
    g["x"].astype("<i2").tofile(d / "stereo.raw")
    io.write_wav(str(d / "stereo.wav"), 48000, g["x"])
    (d / "truncated.wav").write_bytes(open(d / "stereo.wav", "rb").read()[:20])
    n = np.load(os.path.join(G, "nmea.npz"))
    n["synthetic_frames"].tofile(d / "frames.bin")
    g["bits0"].astype(np.uint8).tofile(d / "bits_a.bin")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([EXE, str(d)], capture_output=True, timeout=600, env=env)
    if p.returncode != 0 and b"LeakSanitizer has encountered a fatal error" in p.stderr:
        env["ASAN_OPTIONS"] = "detect_leaks=0"         # no ptrace in this container: everything but the leak check
        p = subprocess.run([EXE, str(d)], capture_output=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr.decode()[-4000:]
    assert b"asan_host: done" in p.stdout
    return d, g, n


def test_file_readers_under_the_sanitizers(run):
    d, g, _ = run
    lines = open(d / "out_wav.txt").read().splitlines()
    want = poly_hash(g["x"].astype("<u2").reshape(-1))
    assert lines[0] == f"stereo.raw channels 2 rate 48000 frames 6400 sum {want}"
    assert lines[1] == f"stereo.wav channels 2 rate 48000 frames 6400 sum {want}"
    assert lines[2:] == ["missing 1", "truncated 1"]


def test_message_layer_and_sinks_under_the_sanitizers(run):
    d, _, n = run
    assert open(d / "out_nmea.bin", "rb").read() == bytes(n["synthetic_text"])
    assert open(d / "out_text.bin", "rb").read() == bytes(n["synthetic_stdout"])
    misc = open(d / "out_misc.txt").read().splitlines()
    n_sent = bytes(n["synthetic_text"]).count(b"\r\n")
    n_lines = bytes(n["synthetic_stdout"]).count(b"\n")
    assert misc[0].startswith(f"sentences {n_sent} lines {n_lines} vessels ")
    full, reduced = int(misc[1].split()[-1]), int(misc[2].split()[-1])
    assert misc[1].startswith("sql keepsmall 0") and misc[2].startswith("sql keepsmall 1") and 0 < reduced < full
    log = open(d / "out_sinks.txt", "rb").read().decode("latin-1").splitlines()
    summary = dict(zip(log[-1].split()[1::2], log[-1].split()[2::2]))
    assert int(summary["frames"]) == len(n["synthetic_frames"]) and int(summary["sentences"]) == n_sent
    assert 1 <= int(summary["serial_calls"]) <= 4 and summary["ipc_calls"] == summary["serial_calls"]   # <= one per batch and sink
    assert int(summary["sql"]) == full + reduced
    assert sum(l.startswith("nmea ") for l in log) == 2 * n_sent                        # both keepsmall settings log every sentence
    assert sum(l.startswith("position ") for l in log) > 0 and sum(l.startswith("cpos ") for l in log) > 0


def test_dropin_over_the_test_double_under_the_sanitizers(run):
    d, g, _ = run
    fr = np.frombuffer(np.ascontiguousarray(g["frames"]).tobytes(), dtype=FRAME_DTYPE)
    per = {name: [frame_line(name, f) for f in fr if int(f["channel"]) == ch] for ch, name in enumerate("AB")}
    tails = [f"{'AB'[c]}: received {g['counters'][c][0]} lost {g['counters'][c][1]} lost2 {g['counters'][c][2]} pll {g['pll'][c][0]}"
             for c in range(2)]
    got = open(d / "out_dropin.txt").read().splitlines()
    assert got[-2:] == tails and len(got) == len(fr) + 2
    for name in "AB":
        assert [l for l in got[:-2] if l.startswith(f"ch {name} ")] == per[name]
    # two groups driven from two threads at once: the stereo group again and a mono group that hears channel B
    mt = open(d / "out_dropin_mt.txt").read().splitlines()
    for name in "AB":
        assert [l for l in mt if l.startswith(f"ch {name} ")] == per[name]
    assert [l for l in mt if l.startswith("ch M ")] == [l.replace("ch B ", "ch M ") for l in per["B"]]
    assert tails[0] in mt and tails[1] in mt and tails[1].replace("B:", "M:") in mt


def test_reference_named_shims_over_the_test_double_under_the_sanitizers(run):
    d, g, _ = run
    fr = np.frombuffer(np.ascontiguousarray(g["frames"]).tobytes(), dtype=FRAME_DTYPE)
    got = open(d / "out_shims.txt").read().splitlines()
    assert got[0] == f"filter sum {poly_hash(g['filtered_u32'][:, 0])} peak {int(g['maxval'][0])}"
    assert got[1].startswith("filter_run ")
    frames = [l for l in got if l.startswith("ch A ")]
    assert frames == [frame_line("A", f) for f in fr if int(f["channel"]) == 0] and len(frames) > 0
    c = g["counters"][0]
    assert any(l.startswith(f"A: received {c[0]} lost {c[1]} lost2 {c[2]} state ") for l in got)
    assert "crc 906e empty 0000" in got
    good = [l for l in got if l.startswith("calculate_crc good ")][0].split()
    assert good[2] == "1"
    body = bytes([0x04, 0x43, 0x12, 0x34, 0x56, 0x78])
    assert good[3:] == [format(b, "08b") for b in body]                # the payload, most significant bit first
    assert "calculate_crc bad 0 nonpositive 0 0 huge 0" in got
    # protodec_reset() every 997 bits / d->buffer every 500 / protodec_deinit(): what the oracle says for the same bit stream
    from oracle_lib import Oracle
    bits = np.asarray(g["bits0"] if "bits0" in g else [], dtype=np.uint8)
    line = [l for l in got if l.startswith("R: ")]
    assert len(line) == 1
    if len(bits):
        o = Oracle(1)
        h, inside = 1469598103934665603, 0
        for i in range(len(bits)):
            if i % 997 == 500:
                o.protodec_reset()
            o.decode_bits(0, bits[i:i + 1])
            if i % 500 == 499 and o.hdlc(0)["state"] in (4, 5):
                inside += 1
                for v in o.frame_cells(0):
                    h = ((h ^ int(v)) * 1099511628211) & 0xffffffffffffffff
        st = o.hdlc(0)
        assert line[0] == (f"R: received {st['receivedframes']} lost {st['lostframes']} lost2 {st['lostframes2']} "
                           f"state {st['state']} inside {inside} cells {h:016x}")
        assert inside > 0


def _check_shims_from_two_threads(d, g):
    """out_shims_mt.txt: both threads decoded the same golden bit stream through decoders of their own while each
    thread's filter_run_buf() flushed the other's and scratch decoders came and went (gnuais_protodec_release)."""
    fr = np.frombuffer(np.ascontiguousarray(g["frames"]).tobytes(), dtype=FRAME_DTYPE)
    got = open(d / "out_shims_mt.txt").read().splitlines()
    c = g["counters"][0]
    for name in "AP":
        assert [l for l in got if l.startswith(f"ch {name} ")] == [frame_line(name, f) for f in fr if int(f["channel"]) == 0]
        assert f"{name}: received {c[0]} lost {c[1]} lost2 {c[2]}" in got


def test_reference_named_shims_from_two_threads_under_the_sanitizers(run):
    d, g, _ = run
    _check_shims_from_two_threads(d, g)


def test_buffer_mismatch_inside_a_round_is_refused_and_the_handler_may_free(run):
    """receiver_hip.c: a receiver that has not been served from the round in progress and brings a different buffer would
    make the shared batch advance every channel with the wrong samples -- it is refused loudly; the fatal handler runs with
    the drop-in's (recursive) lock held and may call free_receiver() without deadlocking."""
    d, _, _ = run
    p = subprocess.run([EXE, str(d), "mismatch"], capture_output=True, timeout=120,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    err = p.stderr.decode()
    assert p.returncode == 7, err
    assert "another buffer" in err and "buffer mismatch inside a round" in err and "handler: receivers freed" in err


def test_host_code_under_thread_sanitizer(run):
    """The same program under -fsanitize=thread (`make -C tests/c tsan`): its two-thread section drives two receiver
    groups at once through receiver_hip.c (process-wide recursive lock around the device work, frames delivered
    unlocked) -- no data race reported, and the two threads' outputs are what the single-threaded run printed."""
    d, g, _ = run
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "c"), "tsan"])
    exe = os.path.join(ROOT, "tests", "c", "tsan_host.bin")
    p = subprocess.run([exe, str(d)], capture_output=True, timeout=900,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1"))
    err = p.stderr.decode()
    if p.returncode != 0 and "FATAL: ThreadSanitizer" in err and "WARNING" not in err:
        pytest.skip("ThreadSanitizer cannot run in this container: " + err.splitlines()[0])
    assert p.returncode == 0 and "WARNING: ThreadSanitizer" not in err, err[-4000:]
    fr = np.frombuffer(np.ascontiguousarray(g["frames"]).tobytes(), dtype=FRAME_DTYPE)
    mt = open(d / "out_dropin_mt.txt").read().splitlines()
    for ch, name in enumerate("AB"):
        assert [l for l in mt if l.startswith(f"ch {name} ")] == [frame_line(name, f) for f in fr if int(f["channel"]) == ch]
    # and the reference-named shims driven from two threads (protodec_decode / filter_run_buf / gnuais_protodec_release
    # on distinct objects: flush_all() holds a reference on every entry it serves)
    _check_shims_from_two_threads(d, g)
