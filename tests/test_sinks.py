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
                ("table", C.c_void_p), ("table_cap", C.c_int), ("sql", C.c_void_p), ("sql_cap", C.c_int)]


Sinks2 = Sinks


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


MYSQL_STUB = r'''
/* test stub of the MySQL sink's API (src/out_mysql.h:37-45): writes every call down, in the format of the
 * reference-side log (oracle/ref_shim.c) */
#include <stdio.h>
#include <stdlib.h>
#include <stdarg.h>
#include <time.h>
struct mysql_state_t;
static char *g; static size_t gn, gc;
static void lg(const char *fmt, ...)
{
	va_list ap; int n;
	if (gn + 512 > gc) { gc = gc * 2 + 65536; g = realloc(g, gc); }
	va_start(ap, fmt); n = vsnprintf(g + gn, 512, fmt, ap); va_end(ap);
	if (n > 0) gn += (size_t) (n < 512 ? n : 511);
}
int myout_ais_position(struct mysql_state_t *m, time_t t, int mmsi, float lat, float lon, float hdg, float course, float sog)
{ lg("position %d %.9g %.9g %.9g %.9g %.9g\n", mmsi, lat, lon, hdg, course, sog); return 0; }
int myout_ais_basestation(struct mysql_state_t *m, time_t t, int mmsi, float lat, float lon)
{ lg("basestation %d %.9g %.9g\n", mmsi, lat, lon); return 0; }
int myout_ais_vesseldata(struct mysql_state_t *m, time_t t, int mmsi, char *name, char *destination, float draught, int A, int B, int C, int D)
{ lg("vesseldata %d %.9g %d %d %d %d |%s|%s|\n", mmsi, draught, A, B, C, D, name, destination); return 0; }
int myout_ais_vesseldatab(struct mysql_state_t *m, time_t t, int mmsi, int A, int B, int C, int D)
{ lg("vesseldatab %d %d %d %d %d\n", mmsi, A, B, C, D); return 0; }
int myout_ais_vesselname(struct mysql_state_t *m, time_t t, int mmsi, const char *name, const char *destination)
{ lg("vesselname %d |%s|%s|\n", mmsi, name, destination); return 0; }
int myout_nmea(struct mysql_state_t *m, time_t t, char *nmea)
{ lg("nmea %s\n", nmea); return 0; }
const char *stub_log(void) { return g; }
size_t stub_log_bytes(void) { return gn; }
'''


def sql_state(log: bytes, keepsmall: bool = True):
    """What the statements of a call log leave in the database (src/out_mysql.c:133-166, 174-283).  keepsmall: every
    myout_ais_*() call is UPDATE of its own columns WHERE mmsi, INSERT only when no row was touched -- one row per vessel
    and table.  Not keepsmall (the reference's default, cfg.c:74): every call INSERTs a row of its own, in call order.
    myout_nmea() always appends a row (:286-297)."""
    tables = {"ais_position": {}, "ais_basestation": {}, "ais_vesseldata": {}} if keepsmall else \
             {"ais_position": [], "ais_basestation": [], "ais_vesseldata": []}
    nmea, calls = [], 0

    def put(table, mmsi, **cols):
        if keepsmall:
            tables[table].setdefault(mmsi, {}).update(cols)
        else:
            tables[table].append(dict(cols, mmsi=mmsi))

    for line in log.decode("latin-1").splitlines():
        kind, _, rest = line.partition(" ")
        if kind == "nmea":
            nmea.append(rest)
            continue
        calls += 1
        head, *strs = rest.split("|")
        v = head.split()
        mmsi = int(v[0])
        if kind == "position":
            put("ais_position", mmsi, **dict(zip(("lat", "lon", "hdg", "course", "sog"), v[1:6])))
        elif kind == "basestation":
            put("ais_basestation", mmsi, **dict(zip(("lat", "lon"), v[1:3])))
        elif kind == "vesseldata":
            put("ais_vesseldata", mmsi, **dict(zip(("draught", "A", "B", "C", "D"), v[1:6])), name=strs[0], destination=strs[1])
        elif kind == "vesseldatab":
            put("ais_vesseldata", mmsi, **dict(zip(("A", "B", "C", "D"), v[1:5])))
        elif kind == "vesselname":
            put("ais_vesseldata", mmsi, name=strs[0], destination=strs[1])
        else:
            raise AssertionError(line)
    return tables, nmea, calls


@pytest.mark.parametrize("keepsmall", [1, 0])
def test_mysql_front_leaves_the_rows_of_the_per_message_path(tmp_path, keepsmall):
    """gnuais_sinks_deliver_mysql() under both settings of the reference's mysql_keepsmall (out_mysql.c:140; the adapter
    reads the reference's own global).  On: per batch only the last myout_ais_*() call of each kind per vessel (the
    statements are UPDATE ... WHERE mmsi, else INSERT).  Off, the reference's default: every call INSERTs a row, so every
    call must be issued, in order.  Either way one myout_nmea() per sentence.  The reference's own protodec_getdata() is
    run over the same frames with its MySQL sink switched on and its myout_*() calls written down (--wrap in oracle/_ref);
    both call logs are applied to a model of the three tables for that setting: identical rows (argument values as printed
    with nine significant digits, i.e. bit for bit), identical sentence log, and -- with keepsmall on -- far fewer
    statements."""
    from gnuais_amd import lib, nmea_from_frames
    ref = reference()
    fr, n_ch = mixed_traffic()
    ref.lib.ref_sql_bytes.restype = C.c_size_t
    ref.lib.ref_sql_ptr.restype = C.c_void_p
    ref.lib.ref_mysql_enable(1)
    try:
        ref.lib.ref_sql_clear()
        ref.nmea_of_frames(fr, n_ch, stdout=True)
        want_log = C.string_at(ref.lib.ref_sql_ptr(), ref.lib.ref_sql_bytes())
    finally:
        ref.lib.ref_mysql_enable(0)
    (tmp_path / "stub.c").write_text(MYSQL_STUB)
    so = str(tmp_path / "libsinks_my.so")
    subprocess.check_call(["gcc", "-std=gnu11", "-w", "-shared", "-fPIC", "-Wl,-Bsymbolic-functions", "-I",
                           os.path.join(ROOT, "include"), os.path.join(ROOT, "gnuais_amd", "csrc", "sinks_batch.c"),
                           str(tmp_path / "stub.c"), "-o", so])
    refso = C.CDLL(REF_SO, mode=C.RTLD_GLOBAL)
    C.CDLL(lib.LIB_PATH, mode=C.RTLD_GLOBAL)
    L = C.CDLL(so)
    flag = C.c_int.in_dll(refso, "mysql_keepsmall")    # cfg.h:80: the switch both sides read
    old_flag = flag.value
    flag.value = keepsmall
    L.gnuais_sinks_deliver_mysql.argtypes = [C.POINTER(Sinks2), C.c_void_p, C.c_long, C.c_void_p, C.c_int, C.c_char_p,
                                             C.c_size_t, C.POINTER(C.c_long)]
    L.stub_log.restype = C.c_void_p
    L.stub_log_bytes.restype = C.c_size_t
    s = Sinks2()
    counts = (C.c_long * 2)(0, 0)
    seq = np.zeros(n_ch, dtype=np.uint8)
    cuts = [0, 1, 2, 150, 151, 600, len(fr)]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        part = np.ascontiguousarray(fr[lo:hi])
        text = nmea_from_frames(part, seq)
        assert L.gnuais_sinks_deliver_mysql(C.byref(s), C.c_void_p(16), 1234, part.ctypes.data, len(part), text, len(text), counts) == 0
    flag.value = old_flag
    got_log = C.string_at(L.stub_log(), L.stub_log_bytes())
    want_tables, want_nmea, want_calls = sql_state(want_log, bool(keepsmall))
    got_tables, got_nmea, got_calls = sql_state(got_log, bool(keepsmall))
    assert got_tables == want_tables and sum(len(t) for t in want_tables.values()) > 100
    assert got_nmea == want_nmea and len(want_nmea) == counts[1] > 900
    if keepsmall:
        assert counts[0] == got_calls < want_calls          # fewer statements already with six small batches of 120 vessels
    else:
        assert counts[0] == got_calls == want_calls         # a row per call: none may go missing
        assert sum(len(t) for t in want_tables.values()) == want_calls
    # one batch: exactly one call per (vessel, kind)
    from gnuais_amd.lib import load
    plan = np.zeros(2 * len(fr) + 1, dtype=np.dtype([("kind", "<i4"), ("mmsi", "<i4"), ("f", "<f4", (6,)), ("abcd", "<i4", (4,)), ("name", "S24"), ("destination", "S24")]))
    n = C.c_int(0)
    assert load().gnuais_sql_plan_from_frames(fr.ctypes.data, len(fr), plan.ctypes.data, len(plan), C.byref(n)) == 0
    keys = list(zip(plan["mmsi"][: n.value].tolist(), plan["kind"][: n.value].tolist()))
    assert len(keys) == len(set(keys)) and n.value < 0.75 * want_calls
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
