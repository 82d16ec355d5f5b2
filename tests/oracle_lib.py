"""ctypes bindings of the CPU checkers (TEST INFRASTRUCTURE).

  Oracle    -- oracle/libais_oracle.so, the restatement (always available)
  Reference -- oracle/_ref/libgnuais_ref.so, the REAL gnuais objects compiled in
               place from /root/reference (present in the build container and,
               as a prebuilt .so, on the GPU box)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libais_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libgnuais_ref.so")

FRAME_DTYPE = np.dtype([("channel", "<u4"), ("end_bit", "<u4"), ("payload", "u1", (53,)),
                        ("flags", "u1"), ("nbits", "<u2")])
assert FRAME_DTYPE.itemsize == 64


class HdlcState(C.Structure):
    _fields_ = [("state", C.c_int32), ("nstartsign", C.c_int32), ("antallpreamble", C.c_int32),
                ("antallenner", C.c_int32), ("bitstuff", C.c_int32), ("last", C.c_int32),
                ("bufferpos", C.c_int32), ("receivedframes", C.c_int32), ("lostframes", C.c_int32),
                ("lostframes2", C.c_int32), ("bits_seen", C.c_uint32), ("buffer", C.c_uint8 * 450)]


class RunOut(C.Structure):
    _fields_ = [("filtered", C.c_void_p), ("bits", C.c_void_p), ("nbits", C.c_void_p),
                ("bits_cap", C.c_uint32), ("maxval", C.c_void_p)]


def build_oracle(force: bool = False) -> None:
    src = os.path.join(ORACLE_DIR, "ais_oracle.c")
    if (force or not os.path.exists(ORACLE_SO)
            or os.path.getmtime(ORACLE_SO) < max(os.path.getmtime(src),
                                                 os.path.getmtime(src[:-1] + "h"))):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "oracle"])


def build_ref() -> bool:
    """(Re)build the real reference when its tree is present; True if usable."""
    if os.path.isdir("/root/reference/src"):
        shim = os.path.join(ORACLE_DIR, "ref_shim.c")
        if not os.path.exists(REF_SO) or os.path.getmtime(REF_SO) < os.path.getmtime(shim):
            subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "ref"])
    return os.path.exists(REF_SO)


def default_taps() -> np.ndarray:
    lib = _oracle_lib()
    t = np.zeros(36, dtype=np.float32)
    lib.ais_oracle_default_taps(t.ctypes.data_as(C.c_void_p))
    return t


_ORACLE = None


def _oracle_lib():
    global _ORACLE
    if _ORACLE is None:
        build_oracle()
        lib = C.CDLL(ORACLE_SO)
        lib.ais_oracle_create.restype = C.c_void_p
        lib.ais_oracle_create.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_uint]
        lib.ais_oracle_destroy.argtypes = [C.c_void_p]
        lib.ais_oracle_reset.argtypes = [C.c_void_p]
        lib.ais_oracle_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        lib.ais_oracle_run_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        lib.ais_oracle_run_planar_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        lib.ais_oracle_filter_channel.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                                  C.c_int, C.c_void_p, C.c_void_p]
        lib.ais_oracle_decode_bits.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        lib.ais_oracle_frame_count.restype = C.c_size_t
        lib.ais_oracle_frame_count.argtypes = [C.c_void_p]
        lib.ais_oracle_frames.restype = C.c_void_p
        lib.ais_oracle_frames.argtypes = [C.c_void_p]
        lib.ais_oracle_sort_frames.argtypes = [C.c_void_p]
        lib.ais_oracle_clear_frames.argtypes = [C.c_void_p]
        lib.ais_oracle_hdlc.restype = C.POINTER(HdlcState)
        lib.ais_oracle_hdlc.argtypes = [C.c_void_p, C.c_int]
        lib.ais_oracle_get_pll.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.ais_oracle_get_history.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.ais_crc16_x25.restype = C.c_uint16
        lib.ais_crc16_x25.argtypes = [C.c_void_p, C.c_uint]
        lib.ais_oracle_default_taps.argtypes = [C.c_void_p]
        _ORACLE = lib
    return _ORACLE


def crc16_x25(data: bytes) -> int:
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    return _oracle_lib().ais_crc16_x25(buf, len(data))


class Oracle:
    """The CPU restatement over a batch of channels."""

    def __init__(self, n_ch: int, taps=None, pllinc: int = 0):
        self.lib = _oracle_lib()
        self.taps = np.ascontiguousarray(default_taps() if taps is None else taps, dtype=np.float32)
        self.n_ch = n_ch
        self.n_taps = int(self.taps.size)
        self.h = self.lib.ais_oracle_create(n_ch, self.taps.ctypes.data_as(C.c_void_p),
                                            self.n_taps, pllinc)
        assert self.h

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ais_oracle_destroy(self.h)
            self.h = None

    def reset(self):
        self.lib.ais_oracle_reset(self.h)

    def run(self, x: np.ndarray, want_filtered=False, want_bits=False, threads: int = 0):
        """x: int16 [len][n_ch].  Returns dict with optional 'filtered' [len][n_ch],
        'bits' (list of uint8 arrays per channel), 'maxval' [n_ch]."""
        x = np.ascontiguousarray(x, dtype=np.int16)
        assert x.ndim == 2 and x.shape[1] == self.n_ch
        n = x.shape[0]
        if threads > 1:
            self.lib.ais_oracle_run_mt(self.h, x.ctypes.data_as(C.c_void_p), n, threads)
            return {}
        out = RunOut()
        res = {}
        maxval = np.zeros(self.n_ch, dtype=np.int16)
        out.maxval = maxval.ctypes.data
        if want_filtered:
            filt = np.zeros((n, self.n_ch), dtype=np.float32)
            out.filtered = filt.ctypes.data
        if want_bits:
            cap = n // 2 + 64
            bits = np.zeros((self.n_ch, cap), dtype=np.uint8)
            nbits = np.zeros(self.n_ch, dtype=np.uint32)
            out.bits, out.nbits, out.bits_cap = bits.ctypes.data, nbits.ctypes.data, cap
        self.lib.ais_oracle_run(self.h, x.ctypes.data_as(C.c_void_p), n, C.byref(out))
        res["maxval"] = maxval
        if want_filtered:
            res["filtered"] = filt
        if want_bits:
            res["bits"] = [bits[c, : nbits[c]].copy() for c in range(self.n_ch)]
        return res

    def run_planar(self, xp: np.ndarray, threads: int):
        """xp: int16 [n_ch][len], each channel contiguous (benchmark context: the CPU's best case)."""
        xp = np.ascontiguousarray(xp, dtype=np.int16)
        assert xp.ndim == 2 and xp.shape[0] == self.n_ch
        self.lib.ais_oracle_run_planar_mt(self.h, xp.ctypes.data_as(C.c_void_p), int(xp.shape[1]), threads)

    def decode_bits(self, ch: int, bits: np.ndarray):
        b = np.ascontiguousarray(bits, dtype=np.uint8)
        self.lib.ais_oracle_decode_bits(self.h, ch, b.ctypes.data_as(C.c_void_p), int(b.size))

    def frames(self, sort: bool = True) -> np.ndarray:
        if sort:
            self.lib.ais_oracle_sort_frames(self.h)
        n = self.lib.ais_oracle_frame_count(self.h)
        if n == 0:
            return np.zeros(0, dtype=FRAME_DTYPE)
        ptr = self.lib.ais_oracle_frames(self.h)
        buf = (C.c_uint8 * (n * 64)).from_address(ptr)
        return np.frombuffer(bytes(buf), dtype=FRAME_DTYPE).copy()

    def clear_frames(self):
        self.lib.ais_oracle_clear_frames(self.h)

    def hdlc(self, ch: int) -> dict:
        s = self.lib.ais_oracle_hdlc(self.h, ch).contents
        return {k: getattr(s, k) for k in ("state", "nstartsign", "antallpreamble", "antallenner",
                                           "bitstuff", "last", "bufferpos", "receivedframes",
                                           "lostframes", "lostframes2", "bits_seen")}

    def frame_cells(self, ch: int) -> np.ndarray:
        """d->buffer[0 .. bufferpos) (protodec.h:52): the stored bits of the frame the decoder is in"""
        s = self.lib.ais_oracle_hdlc(self.h, ch).contents
        return np.frombuffer(bytes(s.buffer), dtype=np.uint8)[: s.bufferpos].copy()

    def protodec_reset(self):
        """protodec_reset() (protodec.c:87-100) on every channel's decoder"""
        self.lib.ais_oracle_protodec_reset.argtypes = [C.c_void_p, C.c_int]
        for c in range(self.n_ch):
            self.lib.ais_oracle_protodec_reset(self.h, c)

    def counters(self) -> np.ndarray:
        out = np.zeros((self.n_ch, 3), dtype=np.int32)
        for c in range(self.n_ch):
            s = self.lib.ais_oracle_hdlc(self.h, c).contents
            out[c] = (s.receivedframes, s.lostframes, s.lostframes2)
        return out

    def pll(self, ch: int):
        p, a, b = C.c_uint32(), C.c_int(), C.c_int()
        self.lib.ais_oracle_get_pll(self.h, ch, C.byref(p), C.byref(a), C.byref(b))
        return p.value, a.value, b.value

    def history(self, ch: int) -> np.ndarray:
        h = np.zeros(self.n_taps, dtype=np.int16)
        self.lib.ais_oracle_get_history(self.h, ch, h.ctypes.data_as(C.c_void_p))
        return h


class Reference:
    """The real gnuais objects behind oracle/ref_shim.c.  One global instance."""

    def __init__(self):
        if not build_ref():
            raise FileNotFoundError(REF_SO)
        lib = C.CDLL(REF_SO)
        lib.ref_receiver_new.argtypes = [C.c_char, C.c_int, C.c_int]
        lib.ref_receiver_set_params.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_uint]
        lib.ref_receiver_run.argtypes = [C.c_int, C.c_void_p, C.c_int]
        lib.ref_run_stream.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.ref_get_taps.argtypes = [C.c_int, C.c_void_p, C.c_int]
        lib.ref_get_pll.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.ref_get_counters.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.ref_get_fsm.argtypes = [C.c_int, C.c_void_p]
        lib.ref_capture_bits_of.argtypes = [C.c_int]
        lib.ref_bits_count.restype = C.c_size_t
        lib.ref_bits_ptr.restype = C.c_void_p
        lib.ref_frames_count.restype = C.c_size_t
        lib.ref_frames_ptr.restype = C.c_void_p
        lib.ref_filter_stream.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                          C.c_int, C.c_void_p, C.c_void_p]
        lib.ref_decode_bits.argtypes = [C.c_int, C.c_void_p, C.c_int]
        lib.ref_sdlc_crc.restype = C.c_uint
        lib.ref_sdlc_crc.argtypes = [C.c_void_p, C.c_uint]
        lib.ref_getdata.argtypes = [C.c_int, C.c_void_p, C.c_int]
        lib.ref_getdata_text.argtypes = [C.c_int, C.c_void_p, C.c_int]
        lib.ref_set_location.argtypes = [C.c_float, C.c_float]
        lib.ref_best_range.argtypes = [C.c_int]
        lib.ref_best_range.restype = C.c_float
        lib.ref_cache_take.argtypes = [C.c_void_p, C.c_int]
        lib.ref_text_bytes.restype = C.c_size_t
        lib.ref_text_ptr.restype = C.c_void_p
        lib.ref_nmea_bytes.restype = C.c_size_t
        lib.ref_nmea_ptr.restype = C.c_void_p
        lib.ref_bench_run.restype = C.c_long
        lib.ref_bench_run.argtypes = [C.c_void_p, C.c_int, C.c_int]
        assert lib.ref_frame_size() == 64
        self.lib = lib
        self.lib.ref_set_text(0)
        self.n = 0

    def reset(self):
        self.lib.ref_reset_all()
        self.n = 0

    def add_receivers(self, num_ch: int, taps=None, pllinc: int = 0):
        """One reference receiver per interleaved channel (ais.c:139-147 style)."""
        for c in range(num_ch):
            idx = self.lib.ref_receiver_new(bytes([65 + c % 26]), num_ch, c)
            assert idx == self.n
            self.n += 1
            if taps is not None or pllinc:
                t = None if taps is None else np.ascontiguousarray(taps, dtype=np.float32)
                self.lib.ref_receiver_set_params(idx, None if t is None else t.ctypes.data,
                                                 0 if t is None else int(t.size), pllinc)

    def nmea_of_frames(self, frames: np.ndarray, n_channels: int, stdout: bool = False):
        """Hand frame records to the reference's own protodec_getdata(), one fresh receiver per
        channel; returns (every sentence serial_write() received, concatenated; final seqnr[])
        and, with stdout=True, also what it printed on stdout."""
        self.reset()
        self.add_receivers(n_channels)
        self.lib.ref_nmea_clear()
        self.lib.ref_text_clear()
        call = self.lib.ref_getdata_text if stdout else self.lib.ref_getdata
        for f in frames:
            pay = np.ascontiguousarray(f["payload"])
            call(int(f["channel"]), pay.ctypes.data, int(f["nbits"]))
        n = self.lib.ref_nmea_bytes()
        text = C.string_at(self.lib.ref_nmea_ptr(), n) if n else b""
        seq = np.array([self.lib.ref_get_seqnr(c) for c in range(n_channels)], dtype=np.uint8)
        if not stdout:
            return text, seq
        m = self.lib.ref_text_bytes()
        return text, seq, (C.string_at(self.lib.ref_text_ptr(), m) if m else b"")

    def range_of_frames(self, frames: np.ndarray, n_channels: int, lat_deg: float, lon_deg: float) -> np.ndarray:
        """best_range of every channel's decoder after the reference's protodec_getdata() saw these
        frames with the station at (lat, lon) degrees (update_range, range.c:32-45)."""
        self.lib.ref_set_location(lat_deg, lon_deg)
        try:
            self.nmea_of_frames(frames, n_channels, stdout=True)     # nothing is skipped on this path
            return np.array([self.lib.ref_best_range(c) for c in range(n_channels)], dtype=np.float32)
        finally:
            self.lib.ref_set_location(-200.0, -200.0)

    def cache_of_frames(self, frames: np.ndarray, n_channels: int, pieces=None) -> np.ndarray:
        """The reference's own position cache (cache.c) after its protodec_getdata() saw these
        frames, flattened in key order into gnuais_vessel records.  `pieces` (split points)
        only proves that the cache carries across calls -- it is taken out once, at the end."""
        from gnuais_amd.receiver import VESSEL_DTYPE
        self.lib.ref_cache_enable()
        out = np.zeros(max(1, len(frames)), dtype=VESSEL_DTYPE)
        self.lib.ref_cache_take(out.ctypes.data_as(C.c_void_p), 0)          # start from an empty cache
        self.reset()
        self.add_receivers(n_channels)
        for lo, hi in zip([0] + list(pieces or []), list(pieces or []) + [len(frames)]):
            for f in frames[lo:hi]:
                pay = np.ascontiguousarray(f["payload"])
                self.lib.ref_getdata_text(int(f["channel"]), pay.ctypes.data, int(f["nbits"]))
        n = self.lib.ref_cache_take(out.ctypes.data_as(C.c_void_p), len(out))
        assert n <= len(out)
        return out[:n].copy()

    def taps(self, idx: int = 0) -> np.ndarray:
        t = np.zeros(1024, dtype=np.float32)
        n = self.lib.ref_get_taps(idx, t.ctypes.data_as(C.c_void_p), 1024)
        return t[:n].copy()

    def run_stream(self, x: np.ndarray, chunk: int = 1020, capture_bits_of: int = -1):
        x = np.ascontiguousarray(x, dtype=np.int16)
        self.lib.ref_capture_bits_of(capture_bits_of)
        self.lib.ref_run_stream(x.ctypes.data_as(C.c_void_p), int(x.shape[0]), chunk)

    def bits(self) -> np.ndarray:
        n = self.lib.ref_bits_count()
        if n == 0:
            return np.zeros(0, dtype=np.uint8)
        buf = (C.c_uint8 * n).from_address(self.lib.ref_bits_ptr())
        return np.frombuffer(bytes(buf), dtype=np.uint8).copy()

    def frames(self, sort: bool = True) -> np.ndarray:
        n = self.lib.ref_frames_count()
        if n == 0:
            return np.zeros(0, dtype=FRAME_DTYPE)
        buf = (C.c_uint8 * (n * 64)).from_address(self.lib.ref_frames_ptr())
        f = np.frombuffer(bytes(buf), dtype=FRAME_DTYPE).copy()
        if sort:
            f = f[np.lexsort((f["end_bit"], f["channel"]))]
        return f

    def counters(self) -> np.ndarray:
        out = np.zeros((self.n, 3), dtype=np.int32)
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        for i in range(self.n):
            self.lib.ref_get_counters(i, C.byref(a), C.byref(b), C.byref(c))
            out[i] = (a.value, b.value, c.value)
        return out

    def pll(self, idx: int):
        p, a, b = C.c_uint(), C.c_int(), C.c_int()
        self.lib.ref_get_pll(idx, C.byref(p), C.byref(a), C.byref(b))
        return p.value, a.value, b.value

    def fsm(self, idx: int) -> dict:
        v = (C.c_int * 7)()
        self.lib.ref_get_fsm(idx, v)
        return dict(zip(("state", "nstartsign", "antallpreamble", "antallenner", "bitstuff",
                         "last", "bufferpos"), list(v)))

    def filter_stream(self, taps, x: np.ndarray, step: int, total_len: int, chunk: int):
        t = np.ascontiguousarray(taps, dtype=np.float32)
        x = np.ascontiguousarray(x, dtype=np.int16)
        out = np.zeros(total_len, dtype=np.float32)
        nchunks = (total_len + chunk - 1) // chunk
        mv = np.zeros(nchunks, dtype=np.int16)
        self.lib.ref_filter_stream(t.ctypes.data, int(t.size), x.ctypes.data, step, total_len,
                                   chunk, out.ctypes.data, mv.ctypes.data)
        return out, mv

    def decode_bits(self, idx: int, bits: np.ndarray):
        b = np.ascontiguousarray(bits, dtype=np.uint8)
        self.lib.ref_decode_bits(idx, b.ctypes.data, int(b.size))

    def sdlc_crc(self, data: bytes) -> int:
        buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
        return self.lib.ref_sdlc_crc(buf, len(data))


_REF = None


def reference() -> Reference:
    global _REF
    if _REF is None:
        _REF = Reference()
    _REF.reset()
    return _REF


def have_reference() -> bool:
    try:
        return build_ref()
    except Exception:
        return False
