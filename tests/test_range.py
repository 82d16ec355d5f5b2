"""Row f4: gnuais_range_from_frames() against the reference's update_range() (range.c:32-45) as
its position decoders call it (protodec.c:399,441,628) -- float for float.

CPU tests: committed golden (tests/golden/range.npz, made by make_golden.py from oracle/_ref) and,
where oracle/_ref is present, fresh random fixes.  Host code inside libgnuais_hip.so."""
import os

import numpy as np
import pytest

import cases
from oracle_lib import FRAME_DTYPE, have_reference, reference

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def ranges(frames, n_ch, lat, lon, start=None):
    from gnuais_amd import range_from_frames
    best = np.zeros(n_ch, dtype=np.float32) if start is None else start
    return range_from_frames(frames, best, lat, lon)


def as_frames(a):
    return np.frombuffer(np.ascontiguousarray(a).tobytes(), dtype=FRAME_DTYPE)


def test_range_golden():
    g = np.load(os.path.join(G, "range.npz"))
    fr, n_ch = as_frames(g["frames"]), int(g["nch"][0])
    for (lat, lon), want in zip(g["stations"], g["best_range"]):
        assert ranges(fr, n_ch, float(lat), float(lon)).tobytes() == want.tobytes()
    fr1 = as_frames(g["single_frames"])
    for (lat, lon), want in zip(g["stations"], g["single_range"]):
        got = ranges(fr1, len(fr1), float(lat), float(lon))
        assert got.tobytes() == want.tobytes()
    # what the numbers mean: a station without a position tracks nothing; the 0/0 fix, fixes beyond
    # 89 / 180.01 degrees and frames of other types leave their channel at 0
    assert not g["single_range"][4].any() and not g["single_range"][5].any()
    near = g["single_range"][0]
    assert (near > 0).sum() > 200 and (near == 0).sum() > 50
    assert near[(near > 0)].min() < 10.0 and near.max() > 15000.0


def test_range_is_a_running_maximum():
    g = np.load(os.path.join(G, "range.npz"))
    fr, n_ch = as_frames(g["frames"]), int(g["nch"][0])
    lat, lon = (float(v) for v in g["stations"][1])
    whole = ranges(fr, n_ch, lat, lon)
    best = np.zeros(n_ch, dtype=np.float32)
    for i in range(0, len(fr), 97):                       # drained in pieces, carried like d->best_range
        ranges(fr[i:i + 97], n_ch, lat, lon, start=best)
    assert best.tobytes() == whole.tobytes()
    high = np.full(n_ch, 30000.0, dtype=np.float32)       # nothing on earth is farther
    assert ranges(fr, n_ch, lat, lon, start=high.copy()).tobytes() == high.tobytes()


def test_range_argument_errors():
    from gnuais_amd.lib import GnuaisError
    g = np.load(os.path.join(G, "range.npz"))
    fr = as_frames(g["frames"])
    with pytest.raises(GnuaisError):
        ranges(fr, 2, 60.0, 10.0)                         # a frame of channel >= n_channels


@pytest.mark.skipif(not have_reference(), reason="oracle/_ref not built")
@pytest.mark.parametrize("seed", [5, 6])
def test_range_random_against_reference(seed):
    rng = np.random.default_rng(seed)
    fr, n_ch = cases.range_frames(seed=100 + seed, n_random=300, own_channel=True)
    ref = reference()
    for _ in range(4):
        lat, lon = float(rng.uniform(-89.9, 89.9)), float(rng.uniform(-179.9, 179.9))
        want = ref.range_of_frames(fr, n_ch, lat, lon)
        assert ranges(fr, n_ch, lat, lon).tobytes() == want.tobytes()
