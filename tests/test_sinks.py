"""Row f3: the batched sink adapter (gnuais_amd/csrc/sinks_batch.c) in front of the reference's
OWN, unchanged sink functions.  The adapter is linked against oracle/_ref (the reference's
serial.c, ipc.c, cache.c as they lie in /root/reference), handed frame records in batches, and
must leave
  * on the serial fd the bytes of the reference's per-sentence serial_write() calls,
  * on the ipc client socket the bytes of its per-sentence ipc_write() calls,
  * on stdout's stand-in the lines its protodec_getdata() prints,
  * in the reference's position cache the entries its per-message cache_*() calls leave,
with a handful of sink calls per batch instead of one or more per message.  CPU only."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import cases
from oracle_lib import REF_SO, have_reference, reference

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not have_reference(), reason="oracle/_ref not built")


class Serial(C.Structure):                       # src/serial.h:24-26
    _fields_ = [("fd", C.c_int)]


class Ipc(C.Structure):                          # src/ipc.h:29-32
    _fields_ = [("numclientsockets", C.c_int), ("clientsocket", C.c_int * 20)]


class Sinks(C.Structure):                        # include/gnuais_sinks.h
    _fields_ = [("serial", C.POINTER(Serial)), ("ipc", C.POINTER(Ipc)), ("text_out", C.c_void_p),
                ("use_cache", C.c_int), ("seqnr", C.c_void_p), ("chanid", C.c_char_p), ("n_channels", C.c_int),
                ("frames", C.c_long), ("sentences", C.c_long), ("vessels", C.c_long),
                ("serial_calls", C.c_long), ("ipc_calls", C.c_long), ("cache_calls", C.c_long),
                ("flushes", C.c_long),
                ("nmea", C.c_void_p), ("text", C.c_void_p), ("ipcbuf", C.c_void_p),
                ("nmea_cap", C.c_size_t), ("text_cap", C.c_size_t), ("ipc_cap", C.c_size_t),
                ("table", C.c_void_p), ("table_cap", C.c_int)]


@pytest.fixture(scope="module")
def adapter(tmp_path_factory):
    """sinks_batch.c built as the gnuais tree would build it, resolved against the reference's
    own sink objects inside oracle/_ref and against libgnuais_hip.so."""
    from gnuais_amd import lib
    ref = reference()                            # loads libgnuais_ref.so
    so = str(tmp_path_factory.mktemp("sinks") / "libsinks.so")
    subprocess.check_call(["gcc", "-std=gnu11", "-Wall", "-Werror", "-shared", "-fPIC", "-I",
                           os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "gnuais_amd", "csrc", "sinks_batch.c"), "-o", so])
    C.CDLL(REF_SO, mode=C.RTLD_GLOBAL)
    C.CDLL(lib.LIB_PATH, mode=C.RTLD_GLOBAL)
    L = C.CDLL(so)
    L.gnuais_sinks_deliver.argtypes = [C.POINTER(Sinks), C.c_void_p, C.c_int]
    return L, ref


def mixed_traffic():
    a, n_ch = cases.vessel_frames(seed=91, n=700, n_mmsi=120)
    b, _ = cases.nmea_frames(seed=92, n_channels=n_ch, n_random=300)
    fr = np.concatenate([a, b])
    rng = np.random.default_rng(93)
    fr = fr[rng.permutation(len(fr))]
    return np.ascontiguousarray(fr), n_ch


def test_adapter_leaves_what_the_per_message_path_leaves(adapter, tmp_path):
    L, ref = adapter
    fr, n_ch = mixed_traffic()
    want_nmea, want_seq, want_text = ref.nmea_of_frames(fr, n_ch, stdout=True)
    want_cache = ref.cache_of_frames(fr, n_ch)

    libc = C.CDLL(None)
    libc.fdopen.restype = C.c_void_p
    libc.fdopen.argtypes = [C.c_int, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    paths = [str(tmp_path / n) for n in ("serial", "ipc", "text")]
    fds = [os.open(p, os.O_RDWR | os.O_CREAT, 0o600) for p in paths]
    ser, ipc = Serial(fds[0]), Ipc(1)
    ipc.clientsocket[0] = fds[1]
    seq = np.zeros(n_ch, dtype=np.uint8)
    chanid = bytes(ord("A") + c for c in range(n_ch))
    s = Sinks()
    s.serial, s.ipc = C.pointer(ser), C.pointer(ipc)
    s.text_out = libc.fdopen(fds[2], b"w")
    s.use_cache, s.seqnr, s.chanid, s.n_channels = 1, seq.ctypes.data, chanid, n_ch

    from gnuais_amd import VESSEL_DTYPE
    ref.lib.ref_cache_enable()
    scratch = np.zeros(len(fr), dtype=VESSEL_DTYPE)
    ref.lib.ref_cache_take(scratch.ctypes.data_as(C.c_void_p), 0)           # empty the reference's cache
    cuts = [0, 1, 2, 150, 151, 600, len(fr)]
    for lo, hi in zip(cuts[:-1], cuts[1:]):                                  # batches of very different sizes
        part = np.ascontiguousarray(fr[lo:hi])
        assert L.gnuais_sinks_deliver(C.byref(s), part.ctypes.data, len(part)) == 0
    n = ref.lib.ref_cache_take(scratch.ctypes.data_as(C.c_void_p), len(scratch))
    got_cache = scratch[:n]
    libc.fclose(s.text_out)
    got = [open(p, "rb").read() for p in paths]
    os.close(fds[0]); os.close(fds[1])

    assert got[0] == want_nmea
    assert got[1] == want_nmea.replace(b"\r\n", b"")
    assert got[2] == want_text
    assert np.array_equal(seq, want_seq)
    assert got_cache.tobytes() == want_cache.tobytes()
    # and it got there in a few calls per batch, not per message
    n_batches = len(cuts) - 1
    assert s.frames == len(fr) and s.serial_calls <= n_batches and s.ipc_calls <= n_batches
    assert s.flushes <= n_batches
    n_cache_msgs = int(sum((want_cache["set"] != 0)))
    assert 0 < s.cache_calls <= 4 * s.vessels and s.vessels >= n_cache_msgs
    L.gnuais_sinks_free(C.byref(s))


def test_adapter_without_listeners_still_advances_the_sequence_digits(adapter):
    L, ref = adapter
    fr, n_ch = mixed_traffic()
    _, want_seq = ref.nmea_of_frames(fr, n_ch)
    seq = np.zeros(n_ch, dtype=np.uint8)
    s = Sinks()
    s.seqnr, s.n_channels = seq.ctypes.data, n_ch
    assert L.gnuais_sinks_deliver(C.byref(s), fr.ctypes.data, len(fr)) == 0
    assert np.array_equal(seq, want_seq)
    assert s.serial_calls == s.ipc_calls == s.cache_calls == s.flushes == 0
    assert L.gnuais_sinks_deliver(None, fr.ctypes.data, len(fr)) == -1
    L.gnuais_sinks_free(C.byref(s))


def _open_sinks(tmp_path, n_ch, tag):
    libc = C.CDLL(None)
    libc.fdopen.restype = C.c_void_p
    libc.fdopen.argtypes = [C.c_int, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    paths = [str(tmp_path / (tag + n)) for n in ("serial", "ipc", "text")]
    fds = [os.open(p, os.O_RDWR | os.O_CREAT, 0o600) for p in paths]
    ser, ipc = Serial(fds[0]), Ipc(1)
    ipc.clientsocket[0] = fds[1]
    seq = np.zeros(n_ch, dtype=np.uint8)
    chanid = bytes(ord("A") + c for c in range(n_ch))
    s = Sinks()
    s.serial, s.ipc = C.pointer(ser), C.pointer(ipc)
    s.text_out = libc.fdopen(fds[2], b"w")
    s.use_cache, s.seqnr, s.chanid, s.n_channels = 1, seq.ctypes.data, chanid, n_ch
    keep = (ser, ipc, seq, chanid)

    def close():
        libc.fclose(s.text_out)
        out = [open(p, "rb").read() for p in paths]
        os.close(fds[0]); os.close(fds[1])
        return out
    return s, seq, chanid, close, keep


def test_formatted_delivery_equals_frame_delivery(adapter, tmp_path):
    """gnuais_sinks_deliver_formatted(): sentences, stdout lines and vessel entries produced elsewhere (here
    by the host formatter and fold; on a GPU box by the device ones) go through the same sink calls and
    leave the same bytes and the same cache."""
    from gnuais_amd import VESSEL_DTYPE, messages_from_frames, vessels_from_frames
    L, ref = adapter
    L.gnuais_sinks_deliver_formatted.argtypes = [C.POINTER(Sinks), C.c_int, C.c_int, C.c_char_p, C.c_size_t,
                                                 C.c_char_p, C.c_size_t, C.c_void_p, C.c_int]
    fr, n_ch = mixed_traffic()
    want_nmea, want_seq, want_text = ref.nmea_of_frames(fr, n_ch, stdout=True)
    want_cache = ref.cache_of_frames(fr, n_ch)
    s, seq, chanid, close, keep = _open_sinks(tmp_path, n_ch, "f_")
    ref.lib.ref_cache_enable()
    scratch = np.zeros(len(fr), dtype=VESSEL_DTYPE)
    ref.lib.ref_cache_take(scratch.ctypes.data_as(C.c_void_p), 0)
    cuts = [0, 3, 200, 201, len(fr)]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        part = np.ascontiguousarray(fr[lo:hi])
        nm, tx = messages_from_frames(part, seq, chanid)
        tab = vessels_from_frames(part)
        assert L.gnuais_sinks_deliver_formatted(C.byref(s), len(part), nm.count(b"\r\n"), nm, len(nm), tx, len(tx),
                                                tab.ctypes.data, len(tab)) == 0
    n = ref.lib.ref_cache_take(scratch.ctypes.data_as(C.c_void_p), len(scratch))
    got = close()
    assert got[0] == want_nmea and got[1] == want_nmea.replace(b"\r\n", b"") and got[2] == want_text
    assert np.array_equal(seq, want_seq)
    assert scratch[:n].tobytes() == want_cache.tobytes()
    assert s.frames == len(fr) and s.serial_calls <= len(cuts) - 1
    L.gnuais_sinks_free(C.byref(s))


@pytest.mark.gpu
def test_device_message_layer_feeds_the_reference_sinks(adapter, tmp_path):
    """Row f3 end to end from the device: frames in the HBM ring -> gnuais_batch_fold_vessels() +
    gnuais_batch_drain_messages() -> gnuais_sinks_deliver_formatted() -> the reference's unchanged
    serial.c / ipc.c / cache.c / stdout: the same bytes and cache contents as the reference's own
    per-message path over the same frames."""
    from gnuais_amd import ReceiverBatch, VESSEL_DTYPE, synth
    L, ref = adapter
    L.gnuais_sinks_deliver_formatted.argtypes = [C.POINTER(Sinks), C.c_int, C.c_int, C.c_char_p, C.c_size_t,
                                                 C.c_char_p, C.c_size_t, C.c_void_p, C.c_int]
    src, n_ch = mixed_traffic()
    streams = [[np.zeros(8, dtype=np.uint8)] for _ in range(n_ch)]
    for f in src:
        body = bytes(f["payload"][: int(f["nbits"]) // 8])
        if len(body) < 1:
            continue
        streams[int(f["channel"])].append(synth.hdlc_frame_bits(body, training_bits=24))
        streams[int(f["channel"])].append(np.zeros(5, dtype=np.uint8))
    streams = [np.concatenate(st).astype(np.uint8) for st in streams]
    a, b = ReceiverBatch(n_ch, max_len=48000), ReceiverBatch(n_ch, max_len=48000)
    s, seq, chanid, close, keep = _open_sinks(tmp_path, n_ch, "d_")
    ref.lib.ref_cache_enable()
    scratch = np.zeros(len(src) + 8, dtype=VESSEL_DTYPE)
    ref.lib.ref_cache_take(scratch.ctypes.data_as(C.c_void_p), 0)           # empty the reference's cache
    all_frames = []
    for piece in (0, 1):
        half = [st[: len(st) // 2] if piece == 0 else st[len(st) // 2:] for st in streams]
        a.decode_bits(half)
        b.decode_bits(half)
        all_frames.append(a.drain_frames())
        tab = b.fold_vessels()
        nm, tx, n_sent, n_lines, n_frames = b.drain_messages(seq, chanid)
        assert n_frames == len(all_frames[-1]) > 100
        assert L.gnuais_sinks_deliver_formatted(C.byref(s), n_frames, n_sent, nm, len(nm), tx, len(tx),
                                                tab.ctypes.data, len(tab)) == 0
    got = close()
    fr = np.concatenate(all_frames)
    want_nmea, want_seq, want_text = ref.nmea_of_frames(fr, n_ch, stdout=True)
    assert got[0] == want_nmea and got[1] == want_nmea.replace(b"\r\n", b"") and got[2] == want_text
    assert np.array_equal(seq, want_seq)
    # the cache after both batches == the reference's after the same frames one message at a time
    n = ref.lib.ref_cache_take(scratch.ctypes.data_as(C.c_void_p), len(scratch))
    want_cache = ref.cache_of_frames(fr, n_ch)
    assert scratch[:n].tobytes() == want_cache.tobytes()
    L.gnuais_sinks_free(C.byref(s))
