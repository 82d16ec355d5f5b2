"""The CPU oracle against the golden vectors generated from the real reference
(tests/golden/make_golden.py).  Runs everywhere (no GPU, no /root/reference)."""
import os

import numpy as np
import pytest

from gnuais_amd import params, synth
from oracle_lib import FRAME_DTYPE, Oracle, crc16_x25, default_taps

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name + ".npz"))


def frames_of(raw):
    return np.frombuffer(np.ascontiguousarray(raw).tobytes(), dtype=FRAME_DTYPE)


def test_default_taps_match_reference_bits():
    g = load("fir_kat")
    assert np.array_equal(default_taps().view(np.uint32), g["taps"])
    assert np.array_equal(params.taps_48k().view(np.uint32), g["taps"])
    assert np.array_equal(params.taps_192k().view(np.uint32), g["taps192"])
    t = default_taps().view(np.uint32)
    assert t[0] == 0 and t[1] == 0 and t[2] == 0x69 and t[17] == 0x3F50B242


@pytest.mark.parametrize("name", ["chain_48k", "chain_192k", "chain_long"])
def test_full_chain_golden(name):
    g = load(name)
    x = g["x"]
    taps = g["taps"].view(np.float32)
    o = Oracle(x.shape[1], taps=taps, pllinc=int(g["pllinc"]))
    r = o.run(x, want_filtered=True, want_bits=True)
    assert np.array_equal(r["filtered"].view(np.uint32), g["filtered_u32"])
    assert np.array_equal(r["maxval"], g["maxval"])
    for c in range(x.shape[1]):
        assert np.array_equal(r["bits"][c], g[f"bits{c}"])
        assert o.pll(c) == tuple(int(v) for v in g["pll"][c])
        h = o.hdlc(c)
        assert [h[k] for k in ("state", "nstartsign", "antallpreamble", "antallenner",
                               "bitstuff", "last", "bufferpos")] == g["fsm"][c].tolist()
    assert np.array_equal(o.counters(), g["counters"])
    assert o.frames().tobytes() == frames_of(g["frames"]).tobytes()


def test_chain_is_chunk_size_independent():
    g = load("chain_48k")
    x = g["x"]
    for chunk in (1, 7, 333, 1020, 4096):
        o = Oracle(2)
        for p in range(0, x.shape[0], chunk):
            o.run(x[p:p + chunk])
        assert o.frames().tobytes() == frames_of(g["frames"]).tobytes()
        assert o.pll(0) == tuple(int(v) for v in g["pll"][0])
        if chunk > 1:
            continue
        assert np.array_equal(o.counters(), g["counters"])


def test_fir_known_answers():
    g = load("fir_kat")
    taps = g["taps"].view(np.float32)
    for k in g.files:
        if not k.startswith("x_"):
            continue
        x = g[k]
        o = Oracle(1, taps=taps)
        r = o.run(x[:, None], want_filtered=True)
        assert np.array_equal(r["filtered"][:, 0].view(np.uint32), g["y_" + k[2:]]), k
    o = Oracle(1, taps=g["taps192"].view(np.float32))
    r = o.run(g["x_noise_full"][:, None], want_filtered=True)
    assert np.array_equal(r["filtered"][:, 0].view(np.uint32), g["y192_noise_full"])


def test_fir_denormal_taps_are_live():
    """+-1 impulse: the first non-zero outputs are products with the subnormal tap."""
    g = load("fir_kat")
    y = g["y_imp_p1"]
    nz = np.nonzero(y)[0]
    assert nz[0] == 40 + 36 - 33 and y[nz[0]] == 0x69          # tap 33 hits first
    ym = g["y_imp_m1"]
    assert ym[nz[0]] == 0x80000069


def test_crc16_known_answers():
    g = load("crc16")
    data, pos = g["data"].tobytes(), 0
    for n, want in zip(g["lens"], g["crc"]):
        blob = data[pos:pos + n]
        pos += n
        assert crc16_x25(blob) == int(want)
        assert synth.crc16_x25(blob) == int(want)
    assert crc16_x25(b"123456789") == 0x906E
    # a frame followed by its FCS (low byte first) leaves the magic residue
    body = b"hello AIS"
    fcs = crc16_x25(body)
    assert crc16_x25(body + bytes([fcs & 0xFF, fcs >> 8])) == 0x0F47   # protodec.c:166


@pytest.mark.parametrize("name", ["random_p50", "random_p70", "random_p30", "alternating",
                                  "crafted", "mixed"])
def test_deframer_golden(name):
    g = load("deframer_bits")
    bits = np.unpackbits(g["bits_" + name])[: int(g["n_" + name])]
    o = Oracle(1)
    o.decode_bits(0, bits)
    assert o.counters()[0].tolist() == g["counters_" + name].tolist()
    assert o.frames().tobytes() == frames_of(g["frames_" + name]).tobytes()
    h = o.hdlc(0)
    assert [h[k] for k in ("state", "nstartsign", "antallpreamble", "antallenner", "bitstuff",
                           "last", "bufferpos")] == g["fsm_" + name].tolist()


def test_generated_frames_decode_to_their_payloads():
    x, placed = synth.make_stream(10 * 1280, seed=9, channel=3, occupancy=1.0, sigma=500.0)
    o = Oracle(1)
    o.run(x[:, None])
    got = [bytes(f["payload"][:21]) for f in o.frames()]
    want = [p for _, p in placed]
    assert len(got) >= len(want) - 1 and all(p in want for p in got)
