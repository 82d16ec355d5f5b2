"""Pins the CPU restatement (oracle/ais_oracle.c) against the REAL reference
objects (oracle/_ref/libgnuais_ref.so) on randomised inputs.  Needs the prebuilt
reference library (built in the build container from /root/reference; shipped to
the GPU box as a binary); skipped where it is absent."""
import numpy as np
import pytest

from gnuais_amd import params, synth
from oracle_lib import Oracle, have_reference, reference

pytestmark = pytest.mark.skipif(not have_reference(), reason="oracle/_ref not built")


def compare(x, taps=None, pllinc=0, chunk=1020, bits_of=0):
    ref = reference()
    n_ch = x.shape[1]
    ref.add_receivers(n_ch, taps=taps, pllinc=pllinc)
    ref.run_stream(x, chunk, capture_bits_of=bits_of)
    o = Oracle(n_ch, taps=ref.taps(0), pllinc=pllinc)
    r = o.run(x, want_bits=True, want_filtered=True)
    assert np.array_equal(ref.bits(), r["bits"][bits_of])
    assert ref.frames().tobytes() == o.frames().tobytes()
    assert np.array_equal(ref.counters(), o.counters())
    for c in range(n_ch):
        assert ref.pll(c) == o.pll(c)
        h = o.hdlc(c)
        f = ref.fsm(c)
        assert all(h[k] == f[k] for k in f)
    return ref, o, r


def test_taps_identical():
    ref = reference()
    ref.add_receivers(1)
    assert np.array_equal(ref.taps(0).view(np.uint32), params.taps_48k().view(np.uint32))


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_signal_streams(seed):
    n = 20 * 1280
    x = np.stack([synth.make_stream(n, seed=seed, channel=c, sigma=s)[0]
                  for c, s in enumerate((1000.0, 3000.0, 6000.0, 12000.0))], axis=1)
    ref, o, r = compare(x, bits_of=seed % 4)
    assert o.counters()[:, 0].sum() > 10


def test_pure_noise_and_full_scale():
    rng = np.random.default_rng(4)
    x = np.stack([rng.normal(0, 3000, 60000), rng.integers(-32768, 32768, 60000),
                  rng.normal(0, 30, 60000)], axis=1)
    x = np.clip(np.rint(x), -32768, 32767).astype(np.int16)
    compare(x, chunk=4096, bits_of=1)


def test_filter_floats_bit_equal_any_chunking():
    rng = np.random.default_rng(6)
    x = rng.integers(-32768, 32768, 9000).astype(np.int16)
    ref = reference()
    for taps in (params.taps_48k(), params.taps_192k()):
        for chunk in (1, 5, 1020, 4096):
            f, _ = ref.filter_stream(taps, x, 1, x.size, chunk)
            o = Oracle(1, taps=taps)
            r = o.run(x[:, None], want_filtered=True)
            assert np.array_equal(f.view(np.uint32), r["filtered"][:, 0].view(np.uint32))


def test_192k_parameters():
    x, _ = synth.make_stream(8 * 5120, seed=8, channel=0, sps=20, sigma=2000.0, occupancy=0.7)
    compare(x[:, None], taps=params.taps_192k(), pllinc=params.PLLINC_192K, chunk=4096)


def test_deframer_random_bitstreams():
    rng = np.random.default_rng(12)
    for trial in range(6):
        parts = []
        for i in range(80):
            parts.append((rng.random(int(rng.integers(0, 120))) < rng.random()).astype(np.uint8))
            if rng.random() < 0.7:
                n = int(rng.choice([0, 1, 11, 21, 21, 40, 53, 54]))
                bits = synth.hdlc_frame_bits(bytes(rng.integers(0, 256, n, dtype=np.uint8)),
                                             training_bits=int(rng.integers(0, 40)))
                if rng.random() < 0.2:
                    bits[int(rng.integers(0, bits.size))] ^= 1
                parts.append(bits)
        bits = np.concatenate(parts).astype(np.uint8)
        ref = reference()
        ref.add_receivers(1)
        ref.decode_bits(0, bits)
        o = Oracle(1)
        o.decode_bits(0, bits)
        assert ref.frames().tobytes() == o.frames().tobytes()
        assert np.array_equal(ref.counters(), o.counters())
        f, h = ref.fsm(0), o.hdlc(0)
        assert all(h[k] == f[k] for k in f)
