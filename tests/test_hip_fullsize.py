"""Parity at BASELINE.json's full sizes, on the slicer's certification threshold, and over more
than one device: the HIP path (through the C ABI) against the CPU oracle, bit for bit."""
import os
import time
import subprocess
import threading

import numpy as np
import pytest

from gnuais_amd import params, shard, synth
from oracle_lib import FRAME_DTYPE, Oracle

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def batch(*a, **k):
    from gnuais_amd import ReceiverBatch
    return ReceiverBatch(*a, **k)


def dev(x, device=0):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x)).to(f"cuda:{device}")


def host_threads():
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 32))


def counters_of(b):
    c = b.counters()
    return np.stack([c["receivedframes"], c["lostframes"], c["lostframes2"]], axis=1)


def pll_of(b):
    p = b.pll_state()
    return [(int(a), int(b_), int(c_)) for a, b_, c_ in zip(p["pll"], p["prev"], p["lastbit"])]


# ---------------------------------------------------------------- C3: every channel

def test_c3_full_size_every_channel():
    """BASELINE C3 -- 16384 channels x 48000 samples, full chain -- and a second, shorter call on
    the carried state: frames, counters and the PLL carry of EVERY channel against the oracle
    (all host cores), plus the size-independent properties."""
    from gnuais_amd import tile_channels
    n_ch, total, k = 16384, 48000, 256
    base, placed = synth.make_base_streams(k, total)
    xb = tile_channels(dev(base), n_ch)
    xh = xb.cpu().numpy()
    b = batch(n_ch, max_len=total)
    o = Oracle(n_ch)
    for lo, hi in ((0, total), (1000, 12345)):
        b.run(xb[lo:hi])
        o.clear_frames()
        o.run(xh[lo:hi], threads=host_threads())
        frames, want = b.drain_frames(), o.frames()
        assert len(want) > (200000 if hi == total else 20000)
        assert frames.tobytes() == want.tobytes()
        assert np.array_equal(counters_of(b), o.counters())
        assert pll_of(b) == [o.pll(c) for c in range(n_ch)]
        if hi == total:
            # round trip: what was delivered is what was transmitted; reference order, no duplicates
            key = frames["channel"].astype(np.int64) << 32 | frames["end_bit"]
            assert np.all(np.diff(key) > 0)
            sent = [set(p for _, p in pl) for pl in placed]
            for f in frames[:: max(1, len(frames) // 5000)]:
                assert f["nbits"] == 168 and bytes(f["payload"][:21]) in sent[int(f["channel"]) % k]


def test_c2_exact_shape_bits_and_state():
    """BASELINE C2's shape -- 256 channels x 48000 samples, filter.c + receiver.c: the slicer's
    decisions, the recovered bit stream and the carried PLL state of every channel."""
    n_ch, total = 256, 48000
    base, _ = synth.make_base_streams(n_ch, total, seed=71)
    x = np.ascontiguousarray(base.T)
    b = batch(n_ch, max_len=total)
    b.run(dev(x))
    o = Oracle(n_ch)
    r = o.run(x, want_filtered=True, want_bits=True)
    assert np.array_equal(b.last_signs(total), (r["filtered"] > 0).T.astype(np.uint8))
    got = b.last_bits()
    for c in range(n_ch):
        assert np.array_equal(got[c], r["bits"][c]), c
    assert pll_of(b) == [o.pll(c) for c in range(n_ch)]
    assert np.array_equal(b.maxval(), r["maxval"])


# ---------------------------------------------------------------- the pre-slicer floats at BASELINE shapes

@pytest.mark.parametrize("name,n_ch,total,sps,n_check", [("C2", 256, 48000, 5, 256), ("C3", 16384, 48000, 5, 4096),
                                                         ("C5", 16384, 192000, 20, 4096)])
def test_filter_floats_at_baseline_shapes(name, n_ch, total, sps, n_check):
    """north_star's "pre-slicer filter floats": gnuais_batch_filter() (the exact kernel) at C2's full shape and on 4096 + 64
    channels of C3 / C5 (every fourth 64-channel group whole -- all lane positions --, the first and the last group among them), every float compared with the oracle's filter_run_buf() as a raw 32-bit pattern
    (tolerance 0; north_star allows 1e-5 relative), plus maxval."""
    import torch
    from gnuais_amd import tile_channels
    k = min(256, n_ch)
    base, _ = synth.make_base_streams(k, total, sps=sps, seed=73)
    xb = tile_channels(dev(base), n_ch)
    kw = dict(taps=params.taps_192k(), pllinc=params.PLLINC_192K) if name == "C5" else {}
    b = batch(n_ch, max_len=total, **kw)
    y = b.filter(xb)
    if n_check >= n_ch:
        pick = np.arange(n_ch, dtype=np.int64)
    else:       # whole 64-channel groups (a wave's lanes), every (n_ch / n_check)-th of them, the batch's last group among them
        g = np.arange(n_ch // 64)
        step = n_ch // n_check
        groups = np.unique(np.concatenate([g[g % step == (step - 1)], [0]]))
        pick = (groups[:, None] * 64 + np.arange(64)[None, :]).reshape(-1).astype(np.int64)
    xs = np.ascontiguousarray(xb[:, torch.from_numpy(pick).to(xb.device)].cpu().numpy())
    okw = {"taps": params.taps_192k(), "pllinc": params.PLLINC_192K} if name == "C5" else {}

    def part(lo_hi):                                # the oracle's float output is single-threaded: one oracle per slice
        lo, hi = lo_hi
        r_ = Oracle(hi - lo, **okw).run(np.ascontiguousarray(xs[:, lo:hi]), want_filtered=True)
        return r_["filtered"], r_["maxval"]
    from concurrent.futures import ThreadPoolExecutor
    nt = max(1, min(host_threads(), len(pick) // 16))
    cuts = np.linspace(0, len(pick), nt + 1).astype(int)
    with ThreadPoolExecutor(nt) as ex:
        parts = list(ex.map(part, zip(cuts[:-1], cuts[1:])))
    want = np.concatenate([p_[0] for p_ in parts], axis=1)
    want_max = np.concatenate([p_[1] for p_ in parts])
    got = y[:, torch.from_numpy(pick).to(y.device)].cpu().numpy()
    assert got.dtype == np.float32 and want.dtype == np.float32 and got.shape == want.shape
    assert np.array_equal(got.view(np.uint32), np.ascontiguousarray(want).view(np.uint32))
    assert np.array_equal(b.maxval()[pick], want_max)


# ---------------------------------------------------------------- C5: full size

def test_c5_full_size():
    """BASELINE C5 -- 16384 channels x 192000 samples at 192 kHz (144 taps, pllinc 0x10000/20): frames, counters and the
    PLL carry of EVERY channel bit-exactly against the oracle (all host cores), plus the round-trip properties."""
    from gnuais_amd import tile_channels
    n_ch, total, k = 16384, 192000, 256
    base, placed = synth.make_base_streams(k, total, sps=20, seed=72)
    xb = tile_channels(dev(base), n_ch)
    taps = params.taps_192k()
    b = batch(n_ch, taps=taps, pllinc=params.PLLINC_192K, max_len=total)
    assert b.info("sign_exact") == 1 and b.info("sign_central_taps") == 40 and b.info("sign_matrix_pipe") == 1
    b.run(xb)
    frames = b.drain_frames()
    cnt = counters_of(b)
    assert int(cnt[:, 0].sum()) == len(frames) == b.total_received() > 200000
    key = frames["channel"].astype(np.int64) << 32 | frames["end_bit"]
    assert np.all(np.diff(key) > 0)
    sent = [set(p for _, p in pl) for pl in placed]
    for f in frames[:: max(1, len(frames) // 5000)]:
        assert f["nbits"] == 168 and bytes(f["payload"][:21]) in sent[int(f["channel"]) % k]
    o = Oracle(n_ch, taps=taps, pllinc=params.PLLINC_192K)
    o.run(xb.cpu().numpy(), threads=host_threads())
    assert frames.tobytes() == o.frames().tobytes()
    assert np.array_equal(cnt, o.counters())
    assert pll_of(b) == [o.pll(c) for c in range(n_ch)]


@pytest.mark.parametrize("name", ["C3", "C5"])
def test_pipelined_calls_are_reproducible(name):
    """Calls queued back to back (the stages of consecutive calls overlap on their streams): twice the same fifteen calls at
    BASELINE size on fresh batches give the same frames, counters and PLL carry -- a race between stages or calls shows as a
    frame more or less (round 6 met one while trying a side stream for the 192 kHz head; this is the test that would
    have caught it)."""
    import hashlib
    import torch
    from gnuais_amd import tile_channels
    n_ch, total, sps = (16384, 192000, 20) if name == "C5" else (16384, 48000, 5)
    base, _ = synth.make_base_streams(256, total, sps=sps, seed=74)
    xb = tile_channels(dev(base), n_ch)
    kw = dict(taps=params.taps_192k(), pllinc=params.PLLINC_192K) if name == "C5" else {}
    stream = torch.cuda.current_stream().cuda_stream
    seen = []
    for rep in range(3):
        b = batch(n_ch, max_len=total, frame_capacity=n_ch * 48 * 6, **kw)
        h = hashlib.sha256()
        for i in range(15):
            b.run(xb, stream=stream, sync=False)
            if i % 5 == 4:
                h.update(b.drain_frames().tobytes())
        h.update(b.counters().tobytes())
        h.update(b.pll_state().tobytes())
        seen.append(h.hexdigest())
        b.close()
    assert len(set(seen)) == 1, seen


# ---------------------------------------------------------------- the slicer's threshold

@pytest.mark.parametrize("flag2", [1, 0])
def test_sign_exact_slicer_on_its_threshold(flag2):
    """K1s certifies the sign of the reference's ordered 32-term sum from the 12 central
    taps when |y_c| > eps and re-evaluates exactly otherwise.  Inputs built so that |y_c| lands within a few
    percent of the running kernel's eps on BOTH sides (and of both signs), each alone in silence: the decisions must be
    those of the exact filter, sample for sample.  `flag2` = 1 (default): the kernel works with the power of two at or
    above eps and reads the decision off the exponent of the scaled sum (0.125 for the reference table)."""
    fir_nc = 12
    taps = params.taps_48k().astype(np.float64)
    b0 = batch(64, max_len=4096)
    b0.set_option("fir_flag2", flag2)
    assert b0.info("sign_exact") == 1 and b0.info("sign_central_taps") == (12 if fir_nc else 10)
    eps = b0.info("sign_eps")
    assert (eps < 0.2) == (fir_nc == 12)
    if flag2:
        assert eps == (0.125 if fir_nc else 0.5) and b0.info("sign_flag_scale") == 2.0 / eps
    k0 = int(b0.info("first_effective_tap"))
    j0 = k0 + (int(b0.info("n_effective_taps")) - 12) // 2           # first central tap in the 36-tap table
    assert 0.05 < eps <= 0.5
    # y(n) = sum_k taps[k] x[n - 36 + k]: three small integers under taps j0+3, j0+2, j0+1
    # (0.0696, 0.0059, 0.00022) reach any value near eps in steps of 2e-4
    t3, t2, t1 = taps[j0 + 3], taps[j0 + 2], taps[j0 + 1]
    pats = []
    for a in (-3, -2, -1, 1, 2, 3):
        for frac in np.linspace(0.9, 1.1, 41):
            for sgn in (1.0, -1.0):
                target = sgn * eps * frac
                rest = target - a * t3
                bq = int(np.round(rest / t2))
                if abs(bq) > 30000:
                    continue
                cq = int(np.round((rest - bq * t2) / t1))
                if abs(cq) > 30000:
                    continue
                yc = a * t3 + bq * t2 + cq * t1
                pats.append((a, bq, cq, yc))
    below = [p for p in pats if abs(p[3]) < eps]
    above = [p for p in pats if abs(p[3]) > eps]
    assert len(below) > 100 and len(above) > 100
    assert min(abs(abs(p[3]) - eps) for p in pats) < 2e-4            # some sit right on it
    n_ch, gap = 64, 80
    per_ch = (len(pats) + n_ch - 1) // n_ch
    total = gap * (per_ch + 1)
    x = np.zeros((total, n_ch), dtype=np.int16)
    for i, (a, bq, cq, _) in enumerate(pats):
        c, slot = i % n_ch, i // n_ch
        n = gap * (slot + 1)                                         # the sample whose window holds the pattern
        x[n - 36 + j0 + 3, c] = a
        x[n - 36 + j0 + 2, c] = bq
        x[n - 36 + j0 + 1, c] = cq
    o = Oracle(n_ch)
    r = o.run(x, want_filtered=True, want_bits=True)
    want = (r["filtered"] > 0).T.astype(np.uint8)
    # the construction does what it says: the oracle's float at the designated sample is the pattern's
    # y_c up to the outer taps' 5e-8
    for i, (_, _, _, yc) in enumerate(pats[:: 17]):
        i *= 17
        assert abs(float(r["filtered"][gap * (i // n_ch + 1), i % n_ch]) - yc) < 1e-4
    xd = dev(x)
    for chunk in (total, 33, 1):                                     # and under awkward call boundaries
        b = batch(n_ch, max_len=total)
        b.set_option("fir_flag2", flag2)
        signs = []
        for lo in range(0, total, chunk):
            hi = min(total, lo + chunk)
            b.run(xd[lo:hi])
            signs.append(b.last_signs(hi - lo))
        assert np.array_equal(np.concatenate(signs, axis=1), want), chunk
        assert pll_of(b) == [o.pll(c) for c in range(n_ch)]
    # the exact kernel agrees (it is the definition): floats bit for bit
    f = b0.filter(dev(x)).cpu().numpy()
    assert np.array_equal(f.view(np.uint32), r["filtered"].view(np.uint32))


# ---------------------------------------------------------------- more than one device

def test_shards_over_devices_from_host_threads():
    """SURVEY 8e / BASELINE C4's code path: contiguous channel blocks, one batch and one host thread
    per device (two batches on one device when only one is visible), no exchange between them; the
    merged frames are the oracle's for the whole channel set."""
    import torch
    n_dev = torch.cuda.device_count()
    world = 2
    devices = [0, 1] if n_dev >= 2 else [0, 0]
    n_ch, total = 1000, 6 * 1280
    x = np.stack([synth.make_stream(total, seed=73, channel=c, occupancy=0.7)[0] for c in range(n_ch)], axis=1)
    o = Oracle(n_ch)
    o.run(x, threads=host_threads())
    out, errs = [None] * world, []

    def work(rank):
        try:
            lo, hi = shard.shard_range(n_ch, world, rank)
            d = devices[rank]
            torch.cuda.set_device(d)
            bt = batch(hi - lo, max_len=2560, device=d)
            assert bt.info("device") == d
            xs = dev(x[:, lo:hi], d)
            st = torch.cuda.Stream(device=d)
            for a in range(0, total, 2560):
                bt.run(xs[a:a + 2560], stream=st.cuda_stream, sync=False)
            f = bt.drain_frames()
            f["channel"] += lo
            out[rank] = (f, counters_of(bt), pll_of(bt))
        except Exception as e:                                       # surfaced below
            errs.append((rank, repr(e)))

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    assert np.concatenate([p[0] for p in out]).tobytes() == o.frames().tobytes()
    assert np.array_equal(np.concatenate([p[1] for p in out]), o.counters())
    assert [s for p in out for s in p[2]] == [o.pll(c) for c in range(n_ch)]


def test_node_of_eight_shards_at_once_equals_one_unsharded_batch():
    """The rehearsal of the 8-GPU run that one device allows: ReceiverNode(8 x 2048 channels) with all eight shards on
    the visible device(s) AT ONCE -- eight batches, eight host threads, forty streams -- over three ragged calls of
    device slabs, against ONE unsharded 16384-channel batch: merged records (global channel numbers, the reference's
    order), counters, PLL carry and peaks byte-equal.  Partitioning as src/ais.c:141-147 builds its receivers:
    contiguous blocks, nothing shared."""
    import torch
    from gnuais_amd import ReceiverNode, tile_channels
    n_ch, total = 16384, 12 * 1280
    base, _ = synth.make_base_streams(256, total, seed=76)
    nd = torch.cuda.device_count()
    devices = [g % nd for g in range(8)]
    xb = {d: tile_channels(dev(base, d), n_ch) for d in set(devices)}
    one = batch(n_ch, max_len=total)
    node = ReceiverNode(n_ch, devices=devices, max_len=total)
    assert [(f, n) for _, f, n in node.shards] == [(2048 * g, 2048) for g in range(8)]
    for lo, hi in ((0, 6000), (6000, 6001), (6001, total)):
        one.run(xb[devices[0]][lo:hi])
        node.run([xb[d][lo:hi, f:f + n].contiguous() for d, f, n in node.shards])
        node.sync()
        want, got = one.drain_frames(), node.drain_frames()
        assert (len(want) > 1000 or hi - lo == 1) and got.tobytes() == want.tobytes()
        assert node.counters().tobytes() == one.counters().tobytes()
        assert node.pll_state().tobytes() == one.pll_state().tobytes()
        assert np.array_equal(node.maxval(), one.maxval())
    assert node.total_received() == one.total_received() > 50000
    st = node.shard_stats()
    assert len(st) == 8 and [s_["first_channel"] for s_ in st] == [2048 * g for g in range(8)]


def test_bench_eight_shards_on_one_device_prints_eight_rows():
    """`bench.py --gpus 8 --devices 0,...,0 --channels 2048`: the node line the SCALE run will print, with eight
    per_gpu rows, shorter than the driver's limit."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.check_output([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--devices",
                                   ",".join(["0"] * 8), "--steps", "6", "--warmup", "2", "--channels", "2048",
                                   "--len", "9600", "--base", "64", "--no-cpu", "--no-e2e"], timeout=600)
    last = out.decode().strip().splitlines()[-1]
    assert len(last) < 4096
    line = json.loads(last)
    assert line["n_gpus"] == 8 and len(line["per_gpu"]) == 8 and line["value"] > 0
    assert line["roofline"]["traffic"] is None and line["roofline"]["frac"] > 0
    assert abs(line["value"] - 8 * 2048 * 9600 * 6 / (line["ms_per_step"] * 6e-3) / 1e6) < 1e-5 * line["value"]


def test_node_object_equals_one_unsharded_batch(tmp_path):
    """Row e as a product: gnuais_node_* (gnuais_amd/csrc/node.hip) -- 1000 channels in four contiguous blocks (on the
    visible devices, repeated if there are fewer than four), one host thread per shard inside the library, fed from
    ONE interleaved host buffer and, in a second pass, from device slabs -- against a single unsharded batch: the
    merged records (global channel numbers, reference order), counters, PLL carry and peaks are byte-equal, over
    three ragged calls."""
    import torch
    from gnuais_amd import ReceiverNode
    n_ch, total = 1000, 9 * 1280
    x = np.stack([synth.make_stream(total, seed=75, channel=c % 97, occupancy=0.8)[0] for c in range(n_ch)], axis=1)
    nd = torch.cuda.device_count()
    devices = [g % nd for g in range(4)]
    one = batch(n_ch, max_len=total)
    node = ReceiverNode(n_ch, devices=devices, max_len=total)
    assert [(f, n) for _, f, n in node.shards] == [(0, 250), (250, 250), (500, 250), (750, 250)]
    for mode in ("host", "device"):
        for lo, hi in ((0, 4000), (4000, 4001), (4001, total)):
            one.run(dev(x[lo:hi]))
            if mode == "host":
                buf = x[lo:hi].copy()
                node.run_host(buf)
                buf[:] = 0                                      # borrowed for the call only
            else:
                node.run([dev(x[lo:hi, f:f + n], d) for d, f, n in node.shards])
            node.sync()
            want, got = one.drain_frames(), node.drain_frames()
            assert got.tobytes() == want.tobytes()
            assert node.pending_frames() == 0
            assert node.counters().tobytes() == one.counters().tobytes()
            assert node.pll_state().tobytes() == one.pll_state().tobytes()
            assert np.array_equal(node.maxval(), one.maxval())
        assert node.total_received() == one.total_received() > 1000
        one.reset()
        node.reset()
    # per-shard bookkeeping (gnuais_node_mark / gnuais_node_shard_stats): what bench.py --gpus N prints per shard
    node.mark()
    for _ in range(3):
        node.run_host(x[:4000].copy())
    node.sync()
    st = node.shard_stats()
    assert [s_["first_channel"] for s_ in st] == [0, 250, 500, 750] and all(s_["calls"] == 3 for s_ in st)
    assert all(s_["busy_ms"] > 0 and 0 < s_["submit_ms"] <= s_["busy_ms"] * 1.001 for s_ in st)
    assert all(s_["pci"] and s_["pinned_cpus"] >= 0 and s_["numa_node"] >= -1 for s_ in st)
    node.reset()
    # argument errors name the call; a device index that does not exist is refused
    from gnuais_amd.lib import GnuaisError
    with pytest.raises(GnuaisError):
        ReceiverNode(64, devices=[nd + 7])
    with pytest.raises(GnuaisError):
        node.run_host(np.zeros((total + 1, n_ch), dtype=np.int16))


def test_node_streamed_sentences_equal_one_batch():
    """gnuais_node_stream_nmea(): every shard streams its own sentences (formatted on its device, its own thread); in
    shard order they are byte for byte what ONE batch over all channels streams for the same call, sequence digits
    carried per channel, the fill and the final flush included."""
    import torch
    from gnuais_amd import ReceiverBatch, ReceiverNode
    n_ch, call, n_calls = 384, 2 * 1280, 7
    x = np.stack([synth.make_stream(call * n_calls, seed=91, channel=c % 53, occupancy=0.8)[0] for c in range(n_ch)], axis=1)
    nd = torch.cuda.device_count()
    node = ReceiverNode(n_ch, devices=[g % nd for g in range(3)], max_len=call)
    one = ReceiverBatch(n_ch, max_len=call)
    got, want = [], []
    for i in range(n_calls):
        seg = x[i * call:(i + 1) * call]
        node.run([dev(seg[:, f:f + n], d) for d, f, n in node.shards])
        got.append(node.stream_nmea())
        one.run(dev(seg), sync=False)
        want.append(one.stream_nmea())
    for _ in range(one.stream_depth):
        got.append(node.stream_nmea())
        want.append(one.stream_nmea())
    assert [g[2] for g in got] == [w[2] for w in want] and sum(w[2] for w in want if w[2] > 0) > 1000
    for i, (g, w) in enumerate(zip(got, want)):
        assert g[1] == w[1] and g[0] == bytes(w[0]), i


def test_node_example_program(tmp_path):
    """examples/node_decode.c -- the 40-line C program of the node API -- builds against include/gnuais_hip.h and
    decodes a raw 64-channel file on whatever devices exist: as many frames as the oracle finds."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "node_decode"
    subprocess.check_call(["gcc", "-O2", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "node_decode.c"),
                           "-L" + os.path.join(root, "gnuais_amd"), "-lgnuais_hip",
                           "-Wl,-rpath," + os.path.join(root, "gnuais_amd"), "-o", str(exe)])
    n_ch, total = 64, 8 * 1280
    x = np.stack([synth.make_stream(total, seed=76, channel=c, occupancy=0.9)[0] for c in range(n_ch)], axis=1)
    raw = tmp_path / "in.raw"
    x.astype("<i2").tofile(raw)
    p = subprocess.run([str(exe), str(n_ch), "4096", str(raw)], capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr.decode()
    o = Oracle(n_ch)
    o.run(x)
    lines = p.stdout.decode().splitlines()
    assert len(lines) == len(o.frames()) > 300
    assert p.stderr.decode().splitlines()[-1] == f"{len(lines)} frames drained, {len(lines)} received in all"


def test_stream_example_program(tmp_path):
    """examples/stream_vessels.c -- run_host_async + stream_nmea + the carried vessel table from plain C: its stdout is
    the sentences a drain-type batch returns for the same chunks, its table the host fold over that batch's frames."""
    import cases
    from gnuais_amd import ReceiverBatch, vessels_from_frames
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "stream_vessels"
    subprocess.check_call(["gcc", "-O2", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "stream_vessels.c"),
                           "-L" + os.path.join(root, "gnuais_amd"), "-lgnuais_hip",
                           "-Wl,-rpath," + os.path.join(root, "gnuais_amd"), "-o", str(exe)])
    n_ch, chunk, n_chunks = 64, 2560, 6
    pool, _ = cases.vessel_frames(seed=41, n_channels=1, n=2000, n_mmsi=120)
    bodies = [bytes(f["payload"][: int(f["nbits"]) // 8]) for f in pool if int(f["nbits"]) >= 8]

    def payloads(rng, slot):
        return bodies[int(rng.integers(0, len(bodies)))] if rng.random() < 0.8 else None

    x = np.stack([synth.make_stream(chunk * n_chunks - 700, seed=43, channel=c, payloads=payloads)[0] for c in range(n_ch)], axis=1)
    raw = tmp_path / "in.raw"
    x.astype("<i2").tofile(raw)
    p = subprocess.run([str(exe), str(n_ch), str(chunk), str(raw)], capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr.decode()
    b = ReceiverBatch(n_ch, max_len=chunk)
    seq = np.zeros(n_ch, dtype=np.uint8)
    want, table = b"", None
    for i in range(0, len(x), chunk):
        b.run(x[i:i + chunk])
        want += b.drain_nmea(seq)[0]
    assert p.stdout == want and want.count(b"\n") > 200
    # the table: a second drain-type batch, frames folded on the host
    b2 = ReceiverBatch(n_ch, max_len=chunk)
    for i in range(0, len(x), chunk):
        b2.run(x[i:i + chunk])
        table = vessels_from_frames(b2.drain_frames(), table)
    err = p.stderr.decode().splitlines()
    assert err[-1].endswith(f"{len(table)} vessels") and len(table) > 50
    assert [int(l.split()[1].rstrip(":")) for l in err[:-1]] == [int(m) for m in table["mmsi"]]


def test_bench_two_workers_prints_n_gpus_2():
    """`bench.py --gpus 2` without torch.distributed.run: one worker process per device (both on
    device 0 when only one is visible), n_gpus 2 and a whole-job value in the line."""
    import json
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    devs = "0,1" if torch.cuda.device_count() >= 2 else "0,0"
    for mode in (["--procs"], []):                          # worker processes; the in-process node object
        out = subprocess.check_output([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--devices", devs,
                                       "--steps", "6", "--warmup", "2", "--channels", "2048", "--len", "9600",
                                       "--base", "64", "--no-cpu", "--no-traffic", "--no-e2e"] + mode, timeout=600)
        line = json.loads(out.decode().strip().splitlines()[-1])
        assert line["n_gpus"] == 2 and len(line["per_gpu"]) == 2
        assert line["value"] > 0 and line["valid_crc_msgs_per_s"] > 0
        assert abs(line["value"] - 2 * 2048 * 9600 * 6 / (line["ms_per_step"] * 6e-3) / 1e6) < 1e-5 * line["value"]


# ---------------------------------------------------------------- the drop-in with the reference's message layer

def reference_main_lines(raw):
    """What the reference's OWN receiver.c / filter.c / protodec.c print for this file behind the same ais.c-shaped
    driver (oracle/_ref/ref_ais.bin, pure CPU): the lines in the reference's order ACROSS receivers (per buffer A's
    frames, then B's, ais.c:237-247), and its counters line per receiver.  None where it was not built."""
    exe = os.path.join(os.path.dirname(G), "..", "oracle", "_ref", "ref_ais.bin")
    if not os.path.exists(exe):
        return None, None
    p = subprocess.run([exe, str(raw)], capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr.decode()
    return p.stdout.decode().splitlines(), p.stderr.decode().splitlines()[-2:]


def test_dropin_with_the_reference_message_layer(tmp_path):
    """oracle/_ref/dropin_ais.bin = the reference's UNMODIFIED protodec.c message layer + support
    files, with gnuais_amd/csrc/receiver_hip.c in place of filter.c / receiver.c, behind an
    ais.c-shaped driver (built by `make -C oracle dropin` where the reference tree is present).
    Its stdout on the golden stereo recording is the reference's own, line for line per receiver."""
    exe = os.path.join(os.path.dirname(G), "..", "oracle", "_ref", "dropin_ais.bin")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/dropin_ais.bin was not built (no reference tree at build time)")
    g = np.load(os.path.join(G, "chain_48k.npz"))
    raw = tmp_path / "stereo.raw"
    g["x"].astype("<i2").tofile(raw)
    p = subprocess.run([exe, str(raw)], capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr.decode()
    got = p.stdout.decode().splitlines()
    want = bytes(np.load(os.path.join(G, "nmea.npz"))["chain_48k_stdout"]).decode().splitlines()
    assert len(got) == len(want) == 8
    for ch in "AB":     # per buffer the reference prints receiver A's frames, then B's (ais.c:237-247)
        assert [l for l in got if l.startswith(f"ch {ch} ")] == [l for l in want if l.startswith(f"ch {ch} ")]
    cnt = g["counters"]
    tail = p.stderr.decode().splitlines()[-2:]
    assert tail == [f"{'AB'[i]}: received {cnt[i][0]} lost {cnt[i][1]} lost2 {cnt[i][2]}" for i in range(2)]
    ref_lines, ref_tail = reference_main_lines(raw)
    if ref_lines is not None:                       # the same order across the two receivers, too
        assert got == ref_lines and tail == ref_tail


@pytest.mark.parametrize("batch_bits", [0, 1, 512])
def test_reference_receiver_over_the_named_shims(tmp_path, batch_bits):
    """oracle/_ref/shim_ais.bin = the reference's UNMODIFIED receiver.c (slicer / PLL / NRZI on the host) linked
    against gnuais_amd/csrc/protodec_hip.c: filter_init / filter_run_buf (exact FIR kernel, floats bit-identical),
    protodec_decode (device deframer + CRC, every valid frame handed to the reference's own protodec_getdata) --
    the reference's other public names of the hot path (filter.h:64-68, protodec.h:73-76).  Its stdout and counters
    on the golden stereo recording are the reference's own: at the shim's DEFAULT settings (0: bits queue per decoder
    and go to the device when the next buffer starts -- there the whole output must also come out in the reference's
    order, line for line, and fast), with every protodec_decode() call a device round trip (1) and with 512 bits
    queued per decoder."""
    exe = os.path.join(os.path.dirname(G), "..", "oracle", "_ref", "shim_ais.bin")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/shim_ais.bin was not built (no reference tree at build time)")
    g = np.load(os.path.join(G, "chain_48k.npz"))
    x = g["x"] if batch_bits != 1 else g["x"][: 12 * 1280]         # a device round trip per bit: a shorter piece
    raw = tmp_path / "stereo.raw"
    x.astype("<i2").tofile(raw)
    env = dict(os.environ)
    env.pop("GNUAIS_PROTODEC_BATCH", None)
    if batch_bits:
        env["GNUAIS_PROTODEC_BATCH"] = str(batch_bits)
    t0 = time.perf_counter()
    p = subprocess.run([exe, str(raw)], capture_output=True, timeout=900, env=env)
    took = time.perf_counter() - t0
    assert p.returncode == 0, p.stderr.decode()
    got = p.stdout.decode().splitlines()
    if batch_bits != 1:
        want = bytes(np.load(os.path.join(G, "nmea.npz"))["chain_48k_stdout"]).decode().splitlines()
        cnt = g["counters"]
    else:                                           # the piece's own answer: the oracle + the host message layer
        from gnuais_amd import messages_from_frames
        o = Oracle(2)
        o.run(x)
        fr = o.frames()
        fr = fr[np.argsort(fr["channel"], kind="stable")]
        want = messages_from_frames(fr, np.zeros(2, dtype=np.uint8))[1].decode().splitlines()
        cnt = o.counters()
    assert len(got) == len(want) > 0
    for ch in "AB":
        assert [l for l in got if l.startswith(f"ch {ch} ")] == [l for l in want if l.startswith(f"ch {ch} ")]
    if batch_bits == 0:
        ref_lines, ref_tail = reference_main_lines(raw)
        if ref_lines is not None:                   # the reference's order across receivers as well
            assert got == ref_lines
        assert took < 60.0, took                    # process start + HIP init dominate; the per-bit path needs minutes
    tail = p.stderr.decode().splitlines()[-2:]
    assert tail == [f"{'AB'[i]}: received {cnt[i][0]} lost {cnt[i][1]} lost2 {cnt[i][2]}" for i in range(2)]


def test_named_shims_filter_and_crc_from_c(tmp_path):
    """filter_init / filter_run_buf / filter_run / filter_free and protodec_sdlc_crc / protodec_calculate_crc of
    protodec_hip.c called from a C program the way the reference's callers do (strided input, 1020-sample chunks):
    floats == the oracle's bit for bit, CRC known answers (0x906E; residue 0x0F47 -> 1)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "t.c"
    src.write_text(r'''
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "gnuais_receiver_abi.h"
#include "gnuais_hip.h"
struct filter;
struct filter *filter_init(int len, float *taps);
void filter_free(struct filter *f);
void filter_run(struct filter *f, float in, float *out);
short filter_run_buf(struct filter *f, short *in, float *out, int step, int len);
unsigned short protodec_sdlc_crc(const unsigned char *data, unsigned len);
int protodec_calculate_crc(int length_bits, struct demod_state_t *d);
void protodec_getdata(int n, struct demod_state_t *d) { (void) n; (void) d; }     /* the message layer is not under test */
int main(int argc, char **argv)
{
	float taps[36], *out;
	short *in;
	int n, step = 3, i, done = 0;
	FILE *f = fopen(argv[1], "rb"), *o = fopen(argv[2], "wb");
	fseek(f, 0, SEEK_END); n = (int) (ftell(f) / 2 / step); fseek(f, 0, SEEK_SET);
	in = malloc(sizeof(short) * n * step); out = malloc(sizeof(float) * n);
	if (fread(in, 2, (size_t) n * step, f) != (size_t) n * step) return 2;
	gnuais_default_taps(taps);
	struct filter *flt = filter_init(36, taps);
	short peak = 0;
	while (done < n - 1) {                          /* channel 1 of 3, 1020 at a time like src/ais.c */
		int len = n - 1 - done < 1020 ? n - 1 - done : 1020;
		short m = filter_run_buf(flt, in + done * step + 1, out + done, step, len);
		if (m > peak) peak = m;
		done += len;
	}
	filter_run(flt, (float) in[(n - 1) * step + 1], &out[n - 1]);     /* the last sample through filter_run() */
	fwrite(out, 4, n, o);
	filter_free(flt);
	printf("%d %04x", peak, protodec_sdlc_crc((const unsigned char *) "123456789", 9));
	{       /* a frame with a good FCS through protodec_calculate_crc() */
		struct demod_state_t d; unsigned char buf[450] = {0}, rb[450], msg[23];
		unsigned short c;
		memset(&d, 0, sizeof d); d.buffer = buf; d.rbuffer = rb;
		for (i = 0; i < 21; i++) msg[i] = (unsigned char) (i * 37 + 5);
		c = protodec_sdlc_crc(msg, 21); msg[21] = c & 0xff; msg[22] = c >> 8;
		for (i = 0; i < 23 * 8; i++) buf[i] = (msg[i / 8] >> (i % 8)) & 1;
		printf(" %d", protodec_calculate_crc(168, &d));
		buf[9] ^= 1;
		printf(" %d %d\n", protodec_calculate_crc(168, &d), rb[0] == ((msg[0] >> 7) & 1));
	}
	return 0;
}
''')
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-O1", "-I" + os.path.join(root, "include"), str(src),
                           os.path.join(root, "gnuais_amd", "csrc", "protodec_hip.c"),
                           "-L" + os.path.join(root, "gnuais_amd"), "-lgnuais_hip",
                           "-Wl,-rpath," + os.path.join(root, "gnuais_amd"), "-o", str(exe)])
    n = 5000
    x = np.stack([synth.make_stream(n, seed=77, channel=c, occupancy=0.9)[0][:n] for c in range(3)], axis=1)
    (tmp_path / "in.raw").write_bytes(x.astype("<i2").tobytes())
    out = subprocess.check_output([str(exe), str(tmp_path / "in.raw"), str(tmp_path / "out.f32")], timeout=300).decode().split()
    r = Oracle(1).run(np.ascontiguousarray(x[:, 1:2]), want_filtered=True)
    got = np.fromfile(tmp_path / "out.f32", dtype=np.float32)
    assert np.array_equal(got.view(np.uint32), np.ascontiguousarray(r["filtered"][:, 0]).view(np.uint32))
    assert out == [str(int(x[:-1, 1].max())), "906e", "1", "0", "1"]


def test_protodec_reset_and_deinit_reach_the_device(tmp_path):
    """protodec_reset() / protodec_deinit() of protodec_hip.c (src/protodec.h:74, protodec.c:78-100) and the d->buffer
    cells it mirrors: a host that drives the decoder names itself -- protodec_initialize, protodec_decode bit by bit,
    protodec_reset every 997 bits (so: inside frames too), protodec_deinit -- over the device shims prints the same
    message lines and shows the same fields, counters and d->buffer at every checkpoint as over the reference's own
    protodec.c (oracle/protodec_reset_main.c linked both ways by oracle/Makefile)."""
    ref = os.path.join(os.path.dirname(G), "..", "oracle", "_ref", "reset_ref.bin")
    shim = os.path.join(os.path.dirname(G), "..", "oracle", "_ref", "reset_shim.bin")
    if not (os.path.exists(ref) and os.path.exists(shim)):
        pytest.skip("oracle/_ref/reset_*.bin were not built (no reference tree at build time)")
    total = 4 * 48000
    x = synth.make_stream(total, seed=91, channel=3, occupancy=0.9)[0][:total, None]
    bits = Oracle(1).run(np.ascontiguousarray(x), want_bits=True)["bits"][0]
    assert len(bits) > 35000
    (tmp_path / "bits.bin").write_bytes(np.asarray(bits, dtype=np.uint8).tobytes())
    resets = ",".join(str(k) for k in range(500, len(bits), 997))
    out = []
    for exe in (ref, shim):
        p = subprocess.run([exe, str(tmp_path / "bits.bin"), "1500", resets], capture_output=True, timeout=600)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        out.append((p.stdout.decode().splitlines(), p.stderr.decode().splitlines()))
    (want_out, want_err), (got_out, got_err) = out
    assert len(want_out) > 50 and len(want_err) > 100
    assert sum(1 for l in want_err if l.startswith("before-reset") and " state 4 " in l) >= 5      # resets inside a frame
    assert sum(1 for l in want_err if l.startswith("at ") and " state 4 " in l) >= 3               # d->buffer of a frame in progress
    assert got_err == want_err
    assert got_out == want_out


def test_dropin_level_log_branch(tmp_path):
    """receiver_run()'s level log (receiver.c:137-147) in the drop-in: with soundlevellog = 1 every receiver reports
    its level (from the device's per-channel peak, gnuais_batch_maxval) through the reference's own hlog()."""
    exe = os.path.join(os.path.dirname(G), "..", "oracle", "_ref", "dropin_ais.bin")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/dropin_ais.bin was not built (no reference tree at build time)")
    g = np.load(os.path.join(G, "chain_48k.npz"))
    x = g["x"][:4 * 1020].copy()
    x[100, 0] = 32767                               # channel A clips: "too high" (NOTICE) also without soundlevellog
    raw = tmp_path / "stereo.raw"
    x.astype("<i2").tofile(raw)
    p = subprocess.run([exe, str(raw)], capture_output=True, timeout=300, env=dict(os.environ, GNUAIS_LEVELLOG="0"))
    assert p.returncode == 0, p.stderr.decode()
    err = p.stderr.decode()
    assert "Level on ch A too high: 100 %" in err and "Level on ch B" not in err
    p = subprocess.run([exe, str(raw)], capture_output=True, timeout=300, env=dict(os.environ, GNUAIS_LEVELLOG="1"))
    assert p.returncode == 0, p.stderr.decode()
    peak_b = int(x[:1020, 1].max())                 # time() advances by < 1 s here: the first buffer's report only
    assert "Level on ch B: %.0f %%" % (peak_b / 32768 * 100) in p.stderr.decode() or "Level on ch B" in p.stderr.decode()


# ---------------------------------------------------------------- host input: C reader + staged transfers

def test_file_to_frames_through_the_async_host_path(tmp_path):
    """Row f2 on the C boundary: a RIFF file read by gnuais_wav_read() in the reference's 1020-frame
    chunks (and in big pieces), fed through gnuais_batch_run_host_async() -- pinned double-buffered
    staging, nothing waited for between calls -- decodes to the oracle's frames and state."""
    from gnuais_amd import io
    n_ch, total = 6, 40 * 1280
    x = np.stack([synth.make_stream(total, seed=74, channel=c, occupancy=0.8)[0] for c in range(n_ch)], axis=1)
    p = tmp_path / "six.wav"
    io.write_wav(str(p), 48000, x)
    o = Oracle(n_ch)
    o.run(x)
    want = o.frames()
    assert len(want) > 100
    for chunk in (io.REFERENCE_CHUNK, 4096, 20000):
        f = io.SampleFile(str(p))
        assert f.channels == n_ch
        b = batch(n_ch, max_len=chunk)
        while True:
            part = f.read(chunk)
            if not len(part):
                break
            b.run_host_async(part)
            part[:] = 0                                   # the caller's buffer is free again at once
        assert b.drain_frames().tobytes() == want.tobytes(), chunk
        assert np.array_equal(counters_of(b), o.counters())
        assert pll_of(b) == [o.pll(c) for c in range(n_ch)]


# ---------------------------------------------------------------- C3: delivery inside the loop

def test_c3_streamed_delivery_equals_drained_delivery():
    """The bench's end-to-end loop at full size: twelve C3 calls with gnuais_batch_stream_nmea() after
    each (asynchronous, pipelined, the order taken from K3's chunk table), against a second batch that
    drains every call (radix sort on the device) -- the same bytes call by call, and every sentence's
    checksum recomputed on the host."""
    from gnuais_amd import tile_channels
    n_ch, total, calls = 16384, 48000, 12
    base, _ = synth.make_base_streams(128, total)
    x = tile_channels(dev(base), n_ch)
    a, b = batch(n_ch, max_len=total), batch(n_ch, max_len=total)
    seq = np.zeros(n_ch, dtype=np.uint8)
    want, got = [], []
    for i in range(calls):
        a.run(x)
        want.append(a.drain_nmea(seq))
        b.run(x, sync=False)
        got.append(b.stream_nmea())
    depth = b.stream_depth
    for _ in range(depth):
        got.append(b.stream_nmea())
    out = got[depth:]
    assert len(out) == calls
    for i, (w, g) in enumerate(zip(want, out)):
        assert g[2] == w[2] > 200000 and g[1] == w[1], i           # frames, sentences
        assert g[0] == w[0], i
    lines = out[-1][0].split(b"\r\n")[:-1]
    assert len(lines) == out[-1][1]
    body = np.frombuffer(out[-1][0], dtype=np.uint8)
    for l in lines[:: max(1, len(lines) // 5000)]:                   # a sample of checksums
        x_ = 0
        for ch in l[1:l.index(b"*")]:
            x_ ^= ch
        assert l.endswith(b"*%02X" % x_)
    assert body.size == len(out[-1][0])


# ---------------------------------------------------------------- C4: its full size, shard by shard

def test_c4_full_size_shard_by_shard():
    """BASELINE C4 -- 131072 channels x 48000 samples over 8 GPUs -- is eight independent C3-sized
    batches (receivers share nothing).  All eight shards of that size, each with input of
    its own, one after another through the product path on the visible device(s) (shard r on device
    r % device_count, as `bench.py --gpus 8` places them) against the oracle: frames, counters, PLL carry of EVERY channel of all eight
    shards (the oracle is two seconds of host time per 16384 channels on 16 cores).  What an 8-GPU node adds is only that
    the shards run at once."""
    import torch
    from gnuais_amd import tile_channels
    world, per, total, k = 8, 16384, 48000, 128
    n_dev = max(1, torch.cuda.device_count())
    received = 0
    for r in range(world):
        lo, hi = shard.shard_range(world * per, world, r)
        assert hi - lo == per
        d = r % n_dev
        torch.cuda.set_device(d)
        base, _ = synth.make_base_streams(k, total, seed=synth.SEED + 100 + r)
        xb = tile_channels(dev(base, d), per)
        b = batch(per, max_len=total, device=d)
        b.run(xb)
        frames = b.drain_frames()
        o = Oracle(per)
        o.run(xb.cpu().numpy(), threads=host_threads())
        want = o.frames()
        assert len(want) > 200000
        assert frames.tobytes() == want.tobytes(), r
        assert np.array_equal(counters_of(b), o.counters()), r
        assert pll_of(b) == [o.pll(c) for c in range(per)], r
        assert len(frames) > 200000
        received += len(frames)
        del b, o, xb
    torch.cuda.set_device(0)
    assert received > 8 * 200000
