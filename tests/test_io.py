"""Input side (row f2): raw files as the reference reads them, RIFF/WAVE files read properly."""
import os
import struct

import numpy as np
import pytest

from gnuais_amd import io


def test_raw_is_what_the_reference_reads(tmp_path):
    x = np.arange(-3000, 3007, dtype=np.int16)             # 6007 samples: one dangling sample
    p = tmp_path / "a.raw"
    x.tofile(p)
    got = io.read_raw(str(p), 2)
    assert got.shape == (3003, 2) and got[0, 0] == -3000 and got[-1, 1] == 3005
    sizes = [c.shape[0] for c in io.chunks(got)]
    assert sizes == [1020, 1020, 963] and io.REFERENCE_CHUNK == 1020


def test_wav_round_trip_and_header_is_not_audio(tmp_path):
    rng = np.random.default_rng(5)
    x = rng.integers(-32768, 32768, (5000, 2)).astype(np.int16)
    p = tmp_path / "a.wav"
    io.write_wav(str(p), 48000, x)
    rate, got = io.read_wav(str(p))
    assert rate == 48000 and np.array_equal(got, x)
    raw = io.read_raw(str(p), 2)                            # the reference's view: 44 header bytes = 11 frames
    assert raw.shape[0] == 5011 and np.array_equal(raw[11:], x)


def test_wav_extra_chunks_extensible_multichannel_truncated(tmp_path):
    x = np.arange(6 * 700, dtype=np.int16).reshape(700, 6)
    body = x.tobytes()
    fmt = struct.pack("<HHIIHHHHIH", 0xFFFE, 6, 48000, 48000 * 12, 12, 16, 22, 16, 0x3F, 1) + b"\x00" * 14
    junk = b"LIST" + struct.pack("<I", 3) + b"abc" + b"\x00"             # odd size: padded
    blob = b"RIFF" + struct.pack("<I", 0) + b"WAVE" + junk + b"fmt " + struct.pack("<I", len(fmt)) + fmt
    blob += b"data" + struct.pack("<I", len(body) + 1000) + body            # claims more than there is
    p = tmp_path / "b.wav"
    p.write_bytes(blob)
    rate, got = io.read_wav(str(p))
    assert rate == 48000 and np.array_equal(got, x)


@pytest.mark.parametrize("blob", [b"", b"RIFFxxxxWAVE", b"RIFF\0\0\0\0AVI ",
                                  b"RIFF\0\0\0\0WAVEfmt " + struct.pack("<IHHIIHH", 16, 3, 2, 48000, 0, 8, 32)
                                  + b"data" + struct.pack("<I", 0)])
def test_wav_errors_are_loud(tmp_path, blob):
    p = tmp_path / "c.wav"
    p.write_bytes(blob)
    with pytest.raises(ValueError):
        io.read_wav(str(p))


def test_planar_to_interleaved():
    a, b, c = np.arange(10), np.arange(100, 108), np.arange(200, 212)
    x = io.planar([a, b, c])
    assert x.shape == (8, 3) and x.dtype == np.int16 and list(x[3]) == [3, 103, 203]


# ---------------------------------------------------------------- the same readers on the C boundary

def test_c_reader_matches_python_reader(tmp_path):
    rng = np.random.default_rng(6)
    x = rng.integers(-32768, 32768, (5003, 2)).astype(np.int16)
    p = tmp_path / "a.wav"
    io.write_wav(str(p), 44100, x)
    f = io.SampleFile(str(p))
    assert (f.channels, f.rate) == (2, 44100)
    got = np.concatenate(list(f))
    assert np.array_equal(got, x) and len(f.read(10)) == 0
    raw = io.SampleFile(str(p), raw_channels=2)               # the reference's view: header and all
    parts = list(raw)
    assert [len(c) for c in parts[:2]] == [1020, 1020]
    assert np.array_equal(np.concatenate(parts), io.read_raw(str(p), 2))


def test_c_reader_extensible_extra_chunks_truncated_and_errors(tmp_path):
    x = np.arange(6 * 700, dtype=np.int16).reshape(700, 6)
    body = x.tobytes()
    fmt = struct.pack("<HHIIHHHHIH", 0xFFFE, 6, 48000, 48000 * 12, 12, 16, 22, 16, 0x3F, 1) + b"\x00" * 14
    junk = b"LIST" + struct.pack("<I", 3) + b"abc" + b"\x00"
    blob = b"RIFF" + struct.pack("<I", 0) + b"WAVE" + junk + b"fmt " + struct.pack("<I", len(fmt)) + fmt
    blob += b"data" + struct.pack("<I", len(body) + 1000) + body + b"\x01"     # claims more; a dangling byte
    p = tmp_path / "b.wav"
    p.write_bytes(blob)
    f = io.SampleFile(str(p))
    assert f.channels == 6 and np.array_equal(f.read(10000), x)
    for bad in (b"", b"RIFFxxxxWAVE", b"RIFF\0\0\0\0AVI ",
                b"RIFF\0\0\0\0WAVEfmt " + struct.pack("<IHHIIHH", 16, 3, 2, 48000, 0, 8, 32) + b"data" + struct.pack("<I", 0)):
        q = tmp_path / "c.wav"
        q.write_bytes(bad)
        with pytest.raises(ValueError):
            io.SampleFile(str(q))
    with pytest.raises(ValueError):
        io.SampleFile(str(tmp_path / "missing.wav"))
