"""Row f3 (cache side): gnuais_vessels_from_frames() against the reference's own position cache
(cache.c:163-384) as its per-type decoders fill it one cache_*() call per message
(protodec.c:282,390,435,516,619,676-678,740,772) -- field for field, floats bit for bit.

CPU tests: committed golden (tests/golden/vessels.npz, made by make_golden.py from oracle/_ref)
and, where oracle/_ref is present, fresh random traffic.  Host code inside libgnuais_hip.so."""
import os

import numpy as np
import pytest

import cases
from oracle_lib import FRAME_DTYPE, have_reference, reference

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def as_frames(a):
    return np.frombuffer(np.ascontiguousarray(a).tobytes(), dtype=FRAME_DTYPE)


def test_vessels_golden():
    from gnuais_amd import VESSEL_DTYPE, vessels_from_frames
    g = np.load(os.path.join(G, "vessels.npz"))
    fr = as_frames(g["frames"])
    want = np.frombuffer(g["vessels"].tobytes(), dtype=VESSEL_DTYPE)
    got = vessels_from_frames(fr)
    assert len(got) == len(want)
    for name in VESSEL_DTYPE.names:               # a readable failure before the byte compare
        assert np.array_equal(got[name], want[name]), name
    assert got.tobytes() == want.tobytes()
    # the table is the cache in key order, and the traffic leaves entries in every state
    assert np.all(np.diff(got["mmsi"].astype(np.int64)) > 0)
    assert len(np.unique(got["set"])) >= 8
    only_pos = got[got["set"] == 1]
    assert len(only_pos) and np.all(only_pos["shiptype"] == -1) and np.all(only_pos["persons_on_board"] == -1)
    class_b = got[(got["destination"] == b"CLASS B")]
    assert len(class_b)


def test_vessels_fold_continues_across_batches():
    """Draining in pieces and carrying the table gives the table of the whole: the fold is the
    cache's own, entry by entry."""
    from gnuais_amd import vessels_from_frames
    g = np.load(os.path.join(G, "vessels.npz"))
    fr = as_frames(g["frames"])
    whole = vessels_from_frames(fr)
    tab = None
    for i in range(0, len(fr), 131):
        tab = vessels_from_frames(fr[i:i + 131], tab)
    assert tab.tobytes() == whole.tobytes()
    assert vessels_from_frames(fr[:0], whole).tobytes() == whole.tobytes()


def test_vessels_argument_errors():
    import ctypes as C
    from gnuais_amd import VESSEL_DTYPE, lib
    g = np.load(os.path.join(G, "vessels.npz"))
    fr = as_frames(g["frames"])
    L = lib.load()
    small = np.zeros(3, dtype=VESSEL_DTYPE)
    n = C.c_int(0)
    assert L.gnuais_vessels_from_frames(fr.ctypes.data, len(fr), small.ctypes.data, 3, C.byref(n)) == -3
    assert n.value == 0 and not small["mmsi"].any()               # left as on entry
    small["mmsi"] = [5, 5, 9]                                      # not strictly ascending
    n = C.c_int(3)
    assert L.gnuais_vessels_from_frames(fr.ctypes.data, 0, small.ctypes.data, 3, C.byref(n)) == -1


@pytest.mark.skipif(not have_reference(), reason="oracle/_ref not built")
@pytest.mark.parametrize("seed,n_mmsi", [(7, 40), (8, 900)])
def test_vessels_random_against_reference(seed, n_mmsi):
    from gnuais_amd import vessels_from_frames
    fr, n_ch = cases.vessel_frames(seed=200 + seed, n=900, n_mmsi=n_mmsi)
    want = reference().cache_of_frames(fr, n_ch, pieces=[300, 301])
    assert vessels_from_frames(fr).tobytes() == want.tobytes()


@pytest.mark.gpu
def test_device_fold_equals_host_fold():
    """gnuais_batch_fold_vessels(): the same table folded on the device from the queued frames (sort on
    (MMSI, arrival order), one thread per vessel, last writer per field group) == the host fold over the
    drained records -- which the tests above pin to the reference's cache.  Traffic: the golden frames and
    random ones over a small MMSI pool, through the device deframer, several channels, two batches."""
    from gnuais_amd import ReceiverBatch, VESSEL_DTYPE, synth, vessels_from_frames
    g = np.load(os.path.join(G, "vessels.npz"))
    gold = as_frames(g["frames"])
    rnd, n_ch = cases.vessel_frames(seed=73, n_channels=5, n=2500, n_mmsi=300)
    for span, src in enumerate((list(gold), list(rnd))):
        streams = [[np.zeros(8, dtype=np.uint8)] for _ in range(n_ch)]
        for f in src:
            body = bytes(f["payload"][: int(f["nbits"]) // 8])
            if len(body) < 1:
                continue
            c = int(f["channel"]) % n_ch
            streams[c].append(synth.hdlc_frame_bits(body, training_bits=24))
            streams[c].append(np.zeros(5, dtype=np.uint8))
        streams = [np.concatenate(s).astype(np.uint8) for s in streams]
        b = ReceiverBatch(n_ch, max_len=48000)
        b.decode_bits(streams)
        got = b.fold_vessels()
        assert b.pending_frames() > 500                       # not consumed
        frames = b.drain_frames()
        want = vessels_from_frames(frames)
        assert len(got) == len(want) > 100
        for name in VESSEL_DTYPE.names:
            assert np.array_equal(got[name], want[name]), (span, name)
        assert got.tobytes() == want.tobytes()
        assert len(np.unique(want["set"])) >= 6


def _vessel_traffic(seed, n_ch, n, n_mmsi):
    """bit streams per channel carrying `n` random cache-touching frames over a small MMSI pool"""
    from gnuais_amd import synth
    rnd, _ = cases.vessel_frames(seed=seed, n_channels=n_ch, n=n, n_mmsi=n_mmsi)
    streams = [[np.zeros(8, dtype=np.uint8)] for _ in range(n_ch)]
    for f in rnd:
        body = bytes(f["payload"][: int(f["nbits"]) // 8])
        if len(body) < 1:
            continue
        c = int(f["channel"]) % n_ch
        streams[c].append(synth.hdlc_frame_bits(body, training_bits=24))
        streams[c].append(np.zeros(5, dtype=np.uint8))
    return [np.concatenate(s).astype(np.uint8) for s in streams]


@pytest.mark.gpu
def test_carried_table_equals_the_host_fold_span_by_span():
    """gnuais_batch_vessel_table_*(): the position cache kept on the device across batches (hash table by MMSI, two
    passes per span: stamps by atomicMax, the standing stamp writes its group) == gnuais_vessels_from_frames() fed
    the same spans one after another -- which the CPU tests above pin to the reference's cache.  Drain-type use:
    update() before every drain; the table is read after every span, cleared, and filled again."""
    from gnuais_amd import ReceiverBatch, VESSEL_DTYPE, vessels_from_frames
    n_ch = 7
    b = ReceiverBatch(n_ch, max_len=48000)
    b.vessel_table_enable(2000)
    want = np.zeros(0, dtype=VESSEL_DTYPE)
    for span, (seed, n_mmsi) in enumerate(((5, 40), (6, 300), (7, 300), (8, 700))):
        b.decode_bits(_vessel_traffic(seed, n_ch, 1800, n_mmsi))
        b.vessel_table_update()
        frames = b.drain_frames()
        assert len(frames) > 1000
        want = vessels_from_frames(frames, want)
        got = b.vessel_table()
        assert len(got) == len(want)
        for name in VESSEL_DTYPE.names:
            assert np.array_equal(got[name], want[name]), (span, name)
        assert got.tobytes() == want.tobytes()
    assert len(want) > 600 and len(np.unique(want["set"])) >= 6
    b.vessel_table_clear()
    assert len(b.vessel_table()) == 0
    b.decode_bits(_vessel_traffic(9, n_ch, 500, 50))
    b.vessel_table_update()
    assert b.vessel_table().tobytes() == vessels_from_frames(b.drain_frames()).tobytes()


@pytest.mark.gpu
def test_carried_table_overflow_is_reported():
    from gnuais_amd import ReceiverBatch
    from gnuais_amd.lib import GnuaisError
    b = ReceiverBatch(4, max_len=48000)
    b.vessel_table_enable(100)
    b.decode_bits(_vessel_traffic(11, 4, 3000, 2500))
    b.vessel_table_update()
    with pytest.raises(GnuaisError) as e:
        b.vessel_table()
    assert e.value.code == -3 and "more vessels" in str(e.value)


@pytest.mark.gpu
def test_streamed_delivery_carries_the_table_on_the_device():
    """A streaming batch (run + gnuais_batch_stream_nmea per call) with the table enabled folds every span into it
    behind the span's formatter, no host round trip: after call i the table == the host fold over the frames a
    second, drain-type batch drained for calls 0..i.  Sample-domain input, mixed message types over 200 MMSIs."""
    import torch
    from gnuais_amd import ReceiverBatch, VESSEL_DTYPE, synth, vessels_from_frames
    n_ch, call, n_calls = 96, 2 * 1280, 8
    pool, _ = cases.vessel_frames(seed=31, n_channels=1, n=4000, n_mmsi=200)
    bodies = [bytes(f["payload"][: int(f["nbits"]) // 8]) for f in pool if int(f["nbits"]) >= 8]

    def payloads(rng, slot):
        return bodies[int(rng.integers(0, len(bodies)))] if rng.random() < 0.8 else None

    x = np.stack([synth.make_stream(call * n_calls, seed=33, channel=c, payloads=payloads)[0] for c in range(n_ch)], axis=1)
    xd = torch.from_numpy(x).cuda()
    a, b = ReceiverBatch(n_ch, max_len=call), ReceiverBatch(n_ch, max_len=call)
    b.vessel_table_enable(500)
    want = np.zeros(0, dtype=VESSEL_DTYPE)
    total = 0
    for i in range(n_calls):
        a.run(xd[i * call:(i + 1) * call])
        fr = a.drain_frames()
        total += len(fr)
        want = vessels_from_frames(fr, want)
        b.run(xd[i * call:(i + 1) * call], sync=False)
        b.stream_nmea()
        if i in (0, 3, n_calls - 1):
            got = b.vessel_table()
            assert got.tobytes() == want.tobytes(), i
    for _ in range(b.stream_depth):
        b.stream_nmea()
    assert b.vessel_table().tobytes() == want.tobytes()
    assert total > 300 and len(want) > 100 and len(np.unique(want["set"])) >= 5


@pytest.mark.gpu
def test_carried_table_at_c3_size():
    """BASELINE C3's shape through the streamed delivery with the table enabled: 16 384 channels x 48 000 samples,
    three calls (the third on a shifted window, so its frames differ), ~3 x 10^5 frames per call over a few thousand
    vessels, against the host fold over the frames a drain-type batch drained call by call."""
    import torch
    from gnuais_amd import ReceiverBatch, VESSEL_DTYPE, synth, tile_channels, vessels_from_frames
    n_ch, total = 16384, 48000
    base, _ = synth.make_base_streams(256, total)
    x = tile_channels(torch.from_numpy(base).cuda(), n_ch)
    a, b = ReceiverBatch(n_ch, max_len=total), ReceiverBatch(n_ch, max_len=total)
    b.vessel_table_enable(1 << 16)
    want = np.zeros(0, dtype=VESSEL_DTYPE)
    n_frames = 0
    for lo, hi in ((0, total), (0, total), (2000, 30000)):
        a.run(x[lo:hi])
        fr = a.drain_frames()
        n_frames += len(fr)
        want = vessels_from_frames(fr, want)
        b.run(x[lo:hi], sync=False)
        b.stream_nmea(copy=False)
    got = b.vessel_table()
    assert n_frames > 500000 and len(want) > 2000
    assert len(got) == len(want)
    for name in VESSEL_DTYPE.names:
        assert np.array_equal(got[name], want[name]), name
    assert got.tobytes() == want.tobytes()
