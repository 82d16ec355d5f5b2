"""Host-side mirror of the reference receiver interface over the HIP batch.

The reference creates one `struct receiver` per audio channel with
init_receiver(name, num_ch, ch_ofs, ...) (src/receiver.c:52-74) and calls
receiver_run(rx, buf, len) once per channel on the same interleaved buffer
(src/ais.c:237-247).  `ReceiverBatch` is those N receivers at once: the same
buffer layout, the same carried state, the same counters, one launch.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import lib as _lib
from .lib import COUNTERS_DTYPE, E_OVERFLOW, FRAME_DTYPE, FSM_DTYPE, OK, PLL_DTYPE, GnuaisError, check


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


class ReceiverBatch:
    """N independent AIS receivers on one MI355X.

    n_channels  -- num_ch of the interleaved input (receiver.c:102,107)
    taps/pllinc -- filter_init() table and rx->pllinc; None/0 = reference values
    max_len     -- largest len (frames per channel) of one run() call
    """

    def __init__(self, n_channels: int, taps=None, pllinc: int = 0, max_len: int = 48000,
                 device: int = 0, frame_capacity: int = 0):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        t = None if taps is None else np.ascontiguousarray(taps, dtype=np.float32)
        check(self._lib.gnuais_batch_create(C.byref(self._h), device, n_channels,
                                            None if t is None else t.ctypes.data,
                                            0 if t is None else int(t.size), pllinc, max_len,
                                            frame_capacity))
        self.n_channels = n_channels
        self.n_taps = self._lib.gnuais_batch_n_taps(self._h)
        self.on_overflow = "raise"        # stream_nmea(): "keep" returns an overflowed slot's text and counts it
        self.stream_overflows = 0
        self.max_len = max_len
        self.device = device

    # -- lifetime -----------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.gnuais_batch_destroy(self._h)
            self._h = None

    __del__ = close

    def reset(self):
        check(self._lib.gnuais_batch_reset(self._h))

    def set_option(self, name: str, value: int):
        check(self._lib.gnuais_batch_set_option(self._h, name.encode(), int(value)))

    # -- receiver_run() -------------------------------------------------------
    def run(self, samples, stream: Optional[int] = None, sync: bool = True):
        """samples: interleaved int16 [len][n_channels]; a CUDA/HIP torch tensor
        (used in place, asynchronous on `stream` or torch's current stream) or a
        numpy array (copied host->device, synchronous)."""
        if _is_torch(samples):
            import torch
            assert samples.is_cuda and samples.dtype == torch.int16 and samples.is_contiguous()
            assert samples.dim() == 2 and samples.shape[1] == self.n_channels
            if stream is None:
                stream = torch.cuda.current_stream(samples.device).cuda_stream
            check(self._lib.gnuais_batch_run(self._h, samples.data_ptr(), int(samples.shape[0]),
                                             C.c_void_p(stream)))
            if sync:
                self.sync()
        else:
            x = np.ascontiguousarray(samples, dtype=np.int16)
            assert x.ndim == 2 and x.shape[1] == self.n_channels
            check(self._lib.gnuais_batch_run_host(self._h, x.ctypes.data, int(x.shape[0])))

    def run_host_async(self, samples: np.ndarray):
        """Host input without waiting for the device: pinned double-buffered staging inside the
        library (gnuais_batch_run_host_async); results after sync()."""
        x = np.ascontiguousarray(samples, dtype=np.int16)
        assert x.ndim == 2 and x.shape[1] == self.n_channels
        check(self._lib.gnuais_batch_run_host_async(self._h, x.ctypes.data, int(x.shape[0])))

    def autotune(self, samples, stream: Optional[int] = None) -> float:
        """Measure the stage -> stream assignments on `samples` (CUDA/HIP int16 tensor) and keep the
        fastest; resets the batch.  Returns the best ms per call seen."""
        import torch
        assert samples.is_cuda and samples.dtype == torch.int16 and samples.is_contiguous()
        if stream is None:
            stream = torch.cuda.current_stream(samples.device).cuda_stream
        ms = C.c_float(0)
        check(self._lib.gnuais_batch_autotune(self._h, samples.data_ptr(), int(samples.shape[0]),
                                              C.c_void_p(stream), C.byref(ms)))
        return ms.value

    def autotune_delivery(self, samples, stream: Optional[int] = None) -> float:
        """gnuais_batch_autotune_delivery(): place the copy stream of stream_nmea() by measurement; resets the
        batch and leaves it streaming.  Returns the best ms per run + stream_nmea seen."""
        import torch
        assert samples.is_cuda and samples.dtype == torch.int16 and samples.is_contiguous()
        if stream is None:
            stream = torch.cuda.current_stream(samples.device).cuda_stream
        ms = C.c_float(0)
        check(self._lib.gnuais_batch_autotune_delivery(self._h, samples.data_ptr(), int(samples.shape[0]),
                                                       C.c_void_p(stream), C.byref(ms)))
        return ms.value

    def sync(self):
        check(self._lib.gnuais_batch_sync(self._h))

    # -- stage taps -----------------------------------------------------------
    def filter(self, samples):
        """filter_run_buf() for every channel -> float32 [len][n_channels] (torch, device)."""
        import torch
        if not _is_torch(samples):
            samples = torch.from_numpy(np.ascontiguousarray(samples, dtype=np.int16)).to(
                f"cuda:{self.device}")
        out = torch.empty(samples.shape, dtype=torch.float32, device=samples.device)
        stream = torch.cuda.current_stream(samples.device).cuda_stream
        check(self._lib.gnuais_batch_filter(self._h, samples.data_ptr(), int(samples.shape[0]),
                                            out.data_ptr(), C.c_void_p(stream)))
        self.sync()
        return out

    def decode_bits(self, bits_per_channel):
        """protodec_decode() for every channel; bits_per_channel: list of uint8 arrays."""
        assert len(bits_per_channel) == self.n_channels
        stride = max(1, max(len(b) for b in bits_per_channel))
        buf = np.zeros((self.n_channels, stride), dtype=np.uint8)
        cnt = np.zeros(self.n_channels, dtype=np.int32)
        for c, b in enumerate(bits_per_channel):
            buf[c, : len(b)] = b
            cnt[c] = len(b)
        check(self._lib.gnuais_batch_decode_bits(self._h, buf.ctypes.data, stride, cnt.ctypes.data))

    def last_bits(self):
        stride = self.max_len // 2 + 64
        buf = np.zeros((self.n_channels, stride), dtype=np.uint8)
        cnt = np.zeros(self.n_channels, dtype=np.int32)
        check(self._lib.gnuais_batch_last_bits(self._h, buf.ctypes.data, stride, cnt.ctypes.data))
        return [buf[c, : cnt[c]].copy() for c in range(self.n_channels)]

    def last_signs(self, length: int) -> np.ndarray:
        """The slicer's decisions of the last run call: uint8 [n_channels][length]."""
        out = np.zeros((self.n_channels, length), dtype=np.uint8)
        check(self._lib.gnuais_batch_last_signs(self._h, out.ctypes.data, length))
        return out

    def info(self, name: str) -> float:
        v = C.c_double()
        check(self._lib.gnuais_batch_info(self._h, name.encode(), C.byref(v)))
        return v.value

    # -- results --------------------------------------------------------------
    def pending_frames(self) -> int:
        n = C.c_int()
        check(self._lib.gnuais_batch_pending_frames(self._h, C.byref(n)))
        return n.value

    def discard_frames(self, stream: Optional[int] = None):
        if stream is None:
            import torch
            stream = torch.cuda.current_stream(self.device).cuda_stream
        check(self._lib.gnuais_batch_discard_frames(self._h, C.c_void_p(stream)))

    def drain_frames(self) -> np.ndarray:
        n = self.pending_frames()
        out = np.zeros(max(n, 1), dtype=FRAME_DTYPE)
        got = C.c_int()
        check(self._lib.gnuais_batch_drain_frames(self._h, out.ctypes.data, int(out.size),
                                                  C.byref(got)))
        return out[: got.value].copy()

    def drain_nmea(self, seqnr: np.ndarray):
        """The queued frames as !AIVDM sentences, formatted on the device (row f1); consumes them.
        seqnr: uint8[n_channels], updated in place.  Returns (text bytes, sentences, frames)."""
        assert seqnr.dtype == np.uint8 and seqnr.flags.c_contiguous and len(seqnr) == self.n_channels
        n = self.pending_frames()
        out = np.empty(164 * max(n, 1), dtype=np.uint8)
        ln, ns, nf = C.c_size_t(0), C.c_int(0), C.c_int(0)
        check(self._lib.gnuais_batch_drain_nmea(self._h, seqnr.ctypes.data, out.ctypes.data, out.size,
                                                C.byref(ln), C.byref(ns), C.byref(nf)))
        return out[: ln.value].tobytes(), ns.value, nf.value

    @property
    def stream_depth(self) -> int:
        return int(self.info("stream_depth"))

    def stream_nmea(self, copy: bool = True):
        """gnuais_batch_stream_nmea(): call after every run(); returns (text, sentences, frames) of the
        call self.stream_depth calls ago -- frames == -1 while the pipeline fills.  copy=False returns a
        uint8 view of the library's pinned buffer (valid until the next call) instead of bytes."""
        ptr, ln, ns, nf = C.c_void_p(), C.c_size_t(0), C.c_int(0), C.c_int(0)
        rc = self._lib.gnuais_batch_stream_nmea(self._h, C.cast(C.byref(ptr), C.POINTER(C.c_char_p)), C.byref(ln),
                                                C.byref(ns), C.byref(nf))
        # A slot's LATE error (its frame ring overflowed, a PLL-stage watchdog) comes with that slot's text: the text
        # that did fit is handed out and the pipeline goes on.  Build the result first, then raise with it attached
        # (GnuaisError.partial) -- or, for an overflow with on_overflow="keep", return it and count the event.
        if ln.value:
            view = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(ln.value,))
            res = ((view.tobytes() if copy else view), ns.value, nf.value)
        else:
            res = ((b"" if copy else np.zeros(0, dtype=np.uint8)), ns.value, nf.value)
        if rc != OK:
            if rc == E_OVERFLOW and self.on_overflow == "keep":
                self.stream_overflows += 1
                return res
            err = GnuaisError(rc, self._lib.gnuais_last_error().decode())
            err.partial = res
            raise err
        return res

    def fold_vessels(self) -> np.ndarray:
        """gnuais_batch_fold_vessels(): the vessel table of the queued frames (gnuais_vessel per MMSI, sorted),
        folded on the device; the frames stay queued."""
        out = np.zeros(max(self.pending_frames(), 1), dtype=VESSEL_DTYPE)
        n = C.c_int(0)
        check(self._lib.gnuais_batch_fold_vessels(self._h, out.ctypes.data, len(out), C.byref(n)))
        return out[: n.value].copy()

    def vessel_table_enable(self, capacity: int) -> None:
        """gnuais_batch_vessel_table_enable(): an empty position cache for `capacity` vessels, carried on the device
        from batch to batch; stream_nmea() folds every span it takes off into it from now on."""
        check(self._lib.gnuais_batch_vessel_table_enable(self._h, capacity))
        self._vt_capacity = capacity

    def vessel_table_update(self) -> None:
        """gnuais_batch_vessel_table_update(): the queued frames into the carried table (drain-type use: call it
        before the drain that consumes them)."""
        check(self._lib.gnuais_batch_vessel_table_update(self._h))

    def vessel_table(self) -> np.ndarray:
        """gnuais_batch_vessel_table(): the carried table, sorted by MMSI."""
        out = np.zeros(max(getattr(self, "_vt_capacity", 1), 1), dtype=VESSEL_DTYPE)
        n = C.c_int(0)
        check(self._lib.gnuais_batch_vessel_table(self._h, out.ctypes.data, len(out), C.byref(n)))
        return out[: n.value].copy()

    def vessel_table_clear(self) -> None:
        check(self._lib.gnuais_batch_vessel_table_clear(self._h))

    def drain_messages(self, seqnr: np.ndarray, chanid: Optional[bytes] = None):
        """gnuais_batch_drain_messages(): sentences and stdout lines of everything queued, both formatted
        on the device -> (nmea bytes, text bytes, sentences, lines, frames)."""
        assert seqnr.dtype == np.uint8 and seqnr.flags.c_contiguous and len(seqnr) == self.n_channels
        assert chanid is None or len(chanid) == self.n_channels
        n = max(self.pending_frames(), 1)
        nm = np.empty(164 * n, dtype=np.uint8)
        tx = np.empty(512 * n, dtype=np.uint8)
        nl, tl, ns, nlines, nf = C.c_size_t(0), C.c_size_t(0), C.c_int(0), C.c_int(0), C.c_int(0)
        check(self._lib.gnuais_batch_drain_messages(self._h, seqnr.ctypes.data, chanid, nm.ctypes.data, nm.size,
                                                    C.byref(nl), C.byref(ns), tx.ctypes.data, tx.size, C.byref(tl),
                                                    C.byref(nlines), C.byref(nf)))
        return nm[: nl.value].tobytes(), tx[: tl.value].tobytes(), ns.value, nlines.value, nf.value

    def drain_frames_nmea(self, seqnr: np.ndarray):
        """Records and device-formatted sentences of the same drained span: (frames, text, sentences)."""
        assert seqnr.dtype == np.uint8 and seqnr.flags.c_contiguous and len(seqnr) == self.n_channels
        n = self.pending_frames()
        fr = np.zeros(max(n, 1), dtype=FRAME_DTYPE)
        out = np.empty(164 * max(n, 1), dtype=np.uint8)
        ln, ns, nf = C.c_size_t(0), C.c_int(0), C.c_int(0)
        check(self._lib.gnuais_batch_drain_frames_nmea(self._h, fr.ctypes.data, int(fr.size), C.byref(nf),
                                                       seqnr.ctypes.data, out.ctypes.data, out.size,
                                                       C.byref(ln), C.byref(ns)))
        return fr[: nf.value].copy(), out[: ln.value].tobytes(), ns.value

    def _struct_array(self, fn, dtype):
        out = np.zeros(self.n_channels, dtype=dtype)
        check(fn(self._h, out.ctypes.data))
        return out

    def counters(self) -> np.ndarray:
        return self._struct_array(self._lib.gnuais_batch_counters, COUNTERS_DTYPE)

    def total_received(self) -> int:
        t = C.c_longlong()
        check(self._lib.gnuais_batch_total_received(self._h, C.byref(t)))
        return t.value

    def pll_state(self) -> np.ndarray:
        return self._struct_array(self._lib.gnuais_batch_pll_state, PLL_DTYPE)

    def fsm_state(self) -> np.ndarray:
        return self._struct_array(self._lib.gnuais_batch_fsm_state, FSM_DTYPE)

    def protodec_reset(self):
        """protodec_reset() (protodec.c:87-100) for every decoder: back to ST_SKURR, counters stay."""
        check(self._lib.gnuais_batch_protodec_reset(self._h))

    def frame_bits(self, channel: int):
        """d->buffer of one channel (protodec.h:52): the stored bits of the frame in progress or of the last frame that
        reached its stop bit, one per byte; None when neither is on record."""
        out = np.zeros(450, dtype=np.uint8)
        n = C.c_int(0)
        check(self._lib.gnuais_batch_frame_bits(self._h, int(channel), out.ctypes.data, out.size, C.byref(n)))
        return None if n.value < 0 else out[:min(n.value, out.size)].copy()

    def maxval(self) -> np.ndarray:
        out = np.zeros(self.n_channels, dtype=np.int16)
        check(self._lib.gnuais_batch_maxval(self._h, out.ctypes.data))
        return out

    def history(self) -> np.ndarray:
        out = np.zeros((self.n_channels, self.n_taps), dtype=np.int16)
        check(self._lib.gnuais_batch_history(self._h, out.ctypes.data))
        return out

    # -- timing ---------------------------------------------------------------
    KERNELS = ("fir_slice", "pll", "hdlc_deframe", "hdlc_crc")

    def set_timing(self, on: bool):
        check(self._lib.gnuais_batch_set_timing(self._h, int(on)))

    def mean_timing(self):
        ms = (C.c_float * 5)()
        n = C.c_int()
        check(self._lib.gnuais_batch_mean_timing(self._h, ms, C.byref(n)))
        d = dict(zip(self.KERNELS + ("total",), ms))
        d["calls"] = n.value
        return d

    def last_timing(self):
        ms = (C.c_float * 5)()
        check(self._lib.gnuais_batch_last_timing(self._h, ms))
        return dict(zip(self.KERNELS + ("total",), ms))


def crc16_batch(messages, device: int = 0) -> np.ndarray:
    """protodec_sdlc_crc() of each byte string, on the device."""
    lib = _lib.load()
    stride = max(1, max(len(m) for m in messages))
    data = np.zeros((len(messages), stride), dtype=np.uint8)
    lens = np.zeros(len(messages), dtype=np.int32)
    for i, m in enumerate(messages):
        data[i, : len(m)] = np.frombuffer(m, dtype=np.uint8)
        lens[i] = len(m)
    out = np.zeros(len(messages), dtype=np.uint16)
    check(lib.gnuais_crc16_batch(device, data.ctypes.data, stride, lens.ctypes.data,
                                 len(messages), out.ctypes.data))
    return out


def nmea_from_frames(frames: np.ndarray, seqnr: np.ndarray) -> bytes:
    """The "!AIVDM,...*hh\\r\\n" sentences the reference emits for these frame records
    (protodec_getdata / protodec_generate_nmea), concatenated.  `seqnr` (uint8 per channel)
    is the rolling sequence digit state, updated in place."""
    lib = _lib.load()
    frames = np.ascontiguousarray(frames, dtype=FRAME_DTYPE)
    assert seqnr.dtype == np.uint8 and seqnr.flags.c_contiguous
    cap = 164 * max(1, len(frames))
    out = np.zeros(cap, dtype=np.uint8)
    need = C.c_size_t(0)
    n_sent = C.c_int(0)
    check(lib.gnuais_nmea_from_frames(frames.ctypes.data, len(frames), seqnr.ctypes.data, len(seqnr),
                                      out.ctypes.data, cap, C.byref(need), C.byref(n_sent)))
    return out[: need.value].tobytes()


def messages_from_frames(frames: np.ndarray, seqnr: np.ndarray, chanid: Optional[bytes] = None):
    """(NMEA sentences, stdout text) exactly as the reference's protodec_getdata() produces them
    for these frame records.  `seqnr` as in nmea_from_frames; `chanid` one byte per channel."""
    lib = _lib.load()
    frames = np.ascontiguousarray(frames, dtype=FRAME_DTYPE)
    assert seqnr.dtype == np.uint8 and seqnr.flags.c_contiguous
    assert chanid is None or len(chanid) == len(seqnr)
    nm = np.zeros(164 * max(1, len(frames)), dtype=np.uint8)
    tx = np.zeros(1024 * max(1, len(frames)), dtype=np.uint8)
    nm_len, tx_len = C.c_size_t(0), C.c_size_t(0)
    check(lib.gnuais_messages_from_frames(frames.ctypes.data, len(frames), seqnr.ctypes.data, chanid,
                                          len(seqnr), nm.ctypes.data, nm.size, C.byref(nm_len), None,
                                          tx.ctypes.data, tx.size, C.byref(tx_len), None))
    return nm[: nm_len.value].tobytes(), tx[: tx_len.value].tobytes()


def range_from_frames(frames: np.ndarray, best_range_km: np.ndarray, my_lat_deg: float, my_lon_deg: float):
    """update_range() (range.c:32-45) over these frame records: best_range_km[channel] (float32,
    updated in place) keeps the farthest plausible position of a type 1-3 / 4 / 18 report."""
    frames = np.ascontiguousarray(frames, dtype=FRAME_DTYPE)
    assert best_range_km.dtype == np.float32 and best_range_km.flags.c_contiguous
    check(_lib.load().gnuais_range_from_frames(frames.ctypes.data, len(frames), len(best_range_km),
                                               C.c_float(my_lat_deg), C.c_float(my_lon_deg),
                                               best_range_km.ctypes.data))
    return best_range_km


# struct gnuais_vessel (include/gnuais_hip.h) = the reference's struct cache_ent, strings inline
VESSEL_DTYPE = np.dtype([("mmsi", "<i4"), ("set", "<u4"), ("lat", "<f4"), ("lon", "<f4"), ("hdg", "<i4"),
                         ("course", "<f4"), ("sog", "<f4"), ("navstat", "<i4"), ("imo", "<i4"),
                         ("shiptype", "<i4"), ("A", "<i4"), ("B", "<i4"), ("C", "<i4"), ("D", "<i4"),
                         ("draught", "<f4"), ("persons_on_board", "<i4"), ("callsign", "S8"),
                         ("name", "S24"), ("destination", "S24")])
assert VESSEL_DTYPE.itemsize == 120


def vessels_from_frames(frames: np.ndarray, table: Optional[np.ndarray] = None) -> np.ndarray:
    """Fold frame records into the vessel table the reference's position cache would hold
    (cache.c:204-384), continuing from `table` (sorted by MMSI) or from an empty cache."""
    frames = np.ascontiguousarray(frames, dtype=FRAME_DTYPE)
    n_old = 0 if table is None else len(table)
    out = np.zeros(n_old + len(frames), dtype=VESSEL_DTYPE)
    if n_old:
        out[:n_old] = table
    n = C.c_int(n_old)
    check(_lib.load().gnuais_vessels_from_frames(frames.ctypes.data, len(frames), out.ctypes.data, len(out),
                                                 C.byref(n)))
    return out[: n.value].copy()


def tile_channels(base, n_channels: int):
    """Device-side benchmark input builder: base torch int16 [K][L] ->
    interleaved [L][n_channels]."""
    import torch
    assert base.is_cuda and base.dtype == torch.int16 and base.is_contiguous()
    k, n = base.shape
    out = torch.empty((n, n_channels), dtype=torch.int16, device=base.device)
    stream = torch.cuda.current_stream(base.device).cuda_stream
    check(_lib.load().gnuais_tile_channels(base.data_ptr(), k, n, out.data_ptr(), n_channels,
                                           C.c_void_p(stream)))
    return out
