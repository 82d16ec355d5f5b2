"""Input side of the receive chain: sample files -> the interleaved int16
[frames][channels] layout receiver_run() reads (gnuais src/receiver.c:102,107).

read_raw()  what the reference itself does with a sound file (src/ais.c:173-182, 214-217): the
            file is a bare stream of little-endian int16 frames, read 1020 frames at a time; a WAV
            header, if there is one, is demodulated as if it were 11 stereo frames of audio.
read_wav()  a proper RIFF/WAVE reader: 16-bit PCM (plain or WAVE_FORMAT_EXTENSIBLE), any number of
            channels, unknown chunks skipped, odd chunk sizes padded, truncated data tolerated.
planar()    [channels][frames] (one file or array per channel) -> interleaved.
chunks()    the reference's read loop: 1020 frames per receiver_run() call.
SampleFile  the same two readers on the C boundary (gnuais_wav_*, gnuais_amd/csrc/wavio.c), streaming.
"""
from __future__ import annotations

import struct
from typing import Iterator, Sequence, Tuple

import numpy as np

REFERENCE_CHUNK = 1020          # ais.c:178-180: 1024 - 1024 % 5


def read_raw(path: str, n_channels: int = 2) -> np.ndarray:
    data = np.fromfile(path, dtype="<i2")
    frames = data.size // n_channels            # fread() drops a partial trailing frame
    return np.ascontiguousarray(data[: frames * n_channels].reshape(frames, n_channels))


def read_wav(path: str) -> Tuple[int, np.ndarray]:
    """-> (sample rate, int16 [frames][channels])"""
    with open(path, "rb") as f:
        blob = f.read()
    if len(blob) < 12 or blob[:4] != b"RIFF" or blob[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, data = 12, None, None
    while pos + 8 <= len(blob):
        tag, size = blob[pos:pos + 4], struct.unpack_from("<I", blob, pos + 4)[0]
        body = blob[pos + 8: pos + 8 + size]     # a truncated last chunk just ends early
        if tag == b"fmt ":
            if len(body) < 16:
                raise ValueError(f"{path}: short fmt chunk")
            code, ch, rate, _, align, bits = struct.unpack_from("<HHIIHH", body)
            if code == 0xFFFE and len(body) >= 40:          # WAVE_FORMAT_EXTENSIBLE: sub-format GUID
                code = struct.unpack_from("<H", body, 24)[0]
            fmt = (code, ch, rate, align, bits)
        elif tag == b"data" and data is None:
            data = body
        pos += 8 + size + (size & 1)
    if fmt is None or data is None:
        raise ValueError(f"{path}: missing fmt or data chunk")
    code, ch, rate, align, bits = fmt
    if code != 1 or bits != 16 or ch < 1 or align != 2 * ch:
        raise ValueError(f"{path}: only 16-bit PCM is supported (format {code}, {bits} bits, {ch} channels)")
    frames = len(data) // align
    x = np.frombuffer(data, dtype="<i2", count=frames * ch).reshape(frames, ch)
    return rate, np.array(x, dtype=np.int16)             # a writable copy


def write_wav(path: str, rate: int, x: np.ndarray) -> None:
    x = np.ascontiguousarray(x, dtype="<i2")
    assert x.ndim == 2
    ch, n = x.shape[1], x.size * 2
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + n) + b"WAVE")
        f.write(b"fmt " + struct.pack("<IHHIIHH", 16, 1, ch, rate, rate * 2 * ch, 2 * ch, 16))
        f.write(b"data" + struct.pack("<I", n))
        f.write(x.tobytes())


def planar(channels: Sequence[np.ndarray]) -> np.ndarray:
    """[channel][frame] -> interleaved [frame][channel]; channels are cut to the shortest."""
    n = min(len(c) for c in channels)
    return np.ascontiguousarray(np.stack([np.asarray(c[:n], dtype=np.int16) for c in channels], axis=1))


def chunks(x: np.ndarray, frames: int = REFERENCE_CHUNK) -> Iterator[np.ndarray]:
    for i in range(0, x.shape[0], frames):
        yield x[i:i + frames]


class SampleFile:
    """Streaming reader over the C ABI (include/gnuais_hip.h gnuais_wav_*): `raw_channels` > 0 reads the
    file as the reference does (src/ais.c:216), 0 parses RIFF/WAVE."""

    def __init__(self, path: str, raw_channels: int = 0):
        import ctypes as C
        from . import lib as _lib
        self._lib = _lib.load()
        self._h = C.c_void_p()
        rc = self._lib.gnuais_wav_open(C.byref(self._h), path.encode(), raw_channels)
        if rc != _lib.OK:
            raise ValueError(f"{path}: not a readable {'raw' if raw_channels else '16-bit PCM RIFF/WAVE'} file")
        self.channels = self._lib.gnuais_wav_channels(self._h)
        self.rate = self._lib.gnuais_wav_rate(self._h)

    def read(self, frames: int) -> np.ndarray:
        out = np.empty((frames, self.channels), dtype=np.int16)
        got = self._lib.gnuais_wav_read(self._h, out.ctypes.data, frames)
        if got < 0:
            raise ValueError("gnuais_wav_read failed")
        return out[:got]

    def __iter__(self) -> Iterator[np.ndarray]:
        while True:
            x = self.read(REFERENCE_CHUNK)
            if not len(x):
                return
            yield x

    def close(self):
        if getattr(self, "_h", None):
            self._lib.gnuais_wav_close(self._h)
            self._h = None

    __del__ = close
