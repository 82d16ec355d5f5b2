"""Row f1 (first part): gnuais_nmea_from_frames() against the reference's own
protodec_getdata() / protodec_generate_nmea() -- byte for byte.

CPU tests: committed golden text (tests/golden/nmea.npz, made by make_golden.py from
oracle/_ref) and, where oracle/_ref is present, fresh random frames against the reference.
The function is host code inside libgnuais_hip.so; no device call is made here."""
import os

import numpy as np
import pytest

import cases
from oracle_lib import FRAME_DTYPE, have_reference, reference

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def nmea(frames, n_ch, seq=None):
    from gnuais_amd import nmea_from_frames
    seq = np.zeros(n_ch, dtype=np.uint8) if seq is None else seq
    return nmea_from_frames(frames, seq), seq


def test_nmea_golden_synthetic():
    g = np.load(os.path.join(G, "nmea.npz"))
    fr = np.frombuffer(np.ascontiguousarray(g["synthetic_frames"]).tobytes(), dtype=FRAME_DTYPE)
    want = g["synthetic_text"].tobytes()
    got, seq = nmea(fr, int(g["synthetic_nch"][0]))
    assert got == want
    assert np.array_equal(seq, g["synthetic_seqnr"])
    # known shapes: a 168-bit type-1 message is one 28-character sentence on "channel A";
    # anything above 366 bits splits, carries the sequence digit and an empty channel field
    lines = want.split(b"\r\n")[:-1]
    assert all(l.startswith(b"!AIVDM,") for l in lines)
    assert any(l.startswith(b"!AIVDM,2,2,") and not l.endswith(b",0*" + l[-2:]) for l in lines)
    for l in lines:                               # XOR checksum of the bytes between '!' and '*'
        body, chk = l[1:].split(b"*")
        x = 0
        for ch in body:
            x ^= ch
        assert chk == b"%02X" % x


@pytest.mark.parametrize("name", ["chain_48k", "chain_long"])
def test_nmea_golden_decoded_frames(name):
    g = np.load(os.path.join(G, "nmea.npz"))
    c = np.load(os.path.join(G, name + ".npz"))
    fr = np.frombuffer(np.ascontiguousarray(c["frames"]).tobytes(), dtype=FRAME_DTYPE)
    got, seq = nmea(fr, int(fr["channel"].max()) + 1)
    assert got == g[name + "_text"].tobytes()
    assert np.array_equal(seq, g[name + "_seqnr"])


def test_stdout_text_golden():
    """The line protodec_getdata() prints for every accepted frame, all per-type decoders."""
    from gnuais_amd import messages_from_frames
    g = np.load(os.path.join(G, "nmea.npz"))
    fr = np.frombuffer(np.ascontiguousarray(g["synthetic_frames"]).tobytes(), dtype=FRAME_DTYPE)
    n_ch = int(g["synthetic_nch"][0])
    nm, tx = messages_from_frames(fr, np.zeros(n_ch, dtype=np.uint8))
    assert nm == g["synthetic_text"].tobytes()
    want = g["synthetic_stdout"].tobytes()
    if tx != want:                                # show the first differing line
        for a, b in zip(tx.split(b"\n"), want.split(b"\n")):
            assert a == b
    assert tx == want
    types = {l.split(b" type ")[1].split(b" ")[0] for l in want.split(b"\n") if b" type " in l}
    assert types == {str(t).encode() for t in range(1, 25)}      # every accepted AIS type occurs
    assert b"(tide-weather) lat" in want and b"(persons-on-board) persons-on-board" in want
    for name in ("chain_48k", "chain_long"):
        c = np.load(os.path.join(G, name + ".npz"))
        fr = np.frombuffer(np.ascontiguousarray(c["frames"]).tobytes(), dtype=FRAME_DTYPE)
        n_ch = int(fr["channel"].max()) + 1
        nm, tx = messages_from_frames(fr, np.zeros(n_ch, dtype=np.uint8))
        assert tx == g[name + "_stdout"].tobytes()


@pytest.mark.skipif(not have_reference(), reason="oracle/_ref not built (reference tree absent)")
@pytest.mark.parametrize("seed", [71, 72])
def test_stdout_text_random_frames_vs_reference(seed):
    from gnuais_amd import messages_from_frames
    fr, n_ch = cases.nmea_frames(seed=seed, n_channels=4, n_random=1500)
    want_nm, seq_want, want_tx = reference().nmea_of_frames(fr, n_ch, stdout=True)
    seq = np.zeros(n_ch, dtype=np.uint8)
    nm, tx = messages_from_frames(fr, seq)
    assert nm == want_nm and np.array_equal(seq, seq_want)
    if tx != want_tx:
        for a, b in zip(tx.split(b"\n"), want_tx.split(b"\n")):
            assert a == b
    assert tx == want_tx


def test_threaded_formatting_equals_serial():
    """>= 16384 frames grouped by channel are formatted by several host threads (ranges cut at
    channel boundaries); the bytes and the sequence state must equal the serial path's."""
    from gnuais_amd import messages_from_frames
    fr, n_ch = cases.nmea_frames(seed=53, n_channels=97, n_random=3000)
    big = np.tile(fr, 12)
    grouped = big[np.argsort(big["channel"], kind="stable")]
    s1 = np.zeros(n_ch, dtype=np.uint8)
    nm1, tx1 = messages_from_frames(grouped, s1)                       # threaded
    s2 = np.zeros(n_ch, dtype=np.uint8)
    parts = [messages_from_frames(grouped[i:i + 5000], s2) for i in range(0, len(grouped), 5000)]   # serial
    assert nm1 == b"".join(p[0] for p in parts)
    assert tx1 == b"".join(p[1] for p in parts)
    assert np.array_equal(s1, s2)
    s3 = np.zeros(n_ch, dtype=np.uint8)
    nm3, tx3 = messages_from_frames(big, s3)                            # not grouped: serial, other order
    assert len(nm3) == len(nm1) and len(tx3) == len(tx1)


def test_nmea_state_carries_and_sizes():
    fr, n_ch = cases.nmea_frames(seed=52, n_random=200)
    whole, seq_w = nmea(fr, n_ch)
    seq = np.zeros(n_ch, dtype=np.uint8)
    parts = b"".join(nmea(fr[i:i + 37], n_ch, seq)[0] for i in range(0, len(fr), 37))
    assert parts == whole and np.array_equal(seq, seq_w)
    # sizing call and loud failures
    import ctypes as C
    from gnuais_amd import lib as L
    lib = L.load()
    need, ns = C.c_size_t(0), C.c_int(0)
    s0 = np.zeros(n_ch, dtype=np.uint8)
    assert lib.gnuais_nmea_from_frames(fr.ctypes.data, len(fr), s0.ctypes.data, n_ch, None, 0,
                                       C.byref(need), C.byref(ns)) == 0
    assert need.value == len(whole) and ns.value == whole.count(b"\r\n")
    small = np.zeros(10, dtype=np.uint8)
    s0[:] = 0
    assert lib.gnuais_nmea_from_frames(fr.ctypes.data, len(fr), s0.ctypes.data, n_ch, small.ctypes.data,
                                       10, C.byref(need), C.byref(ns)) == -3   # GNUAIS_E_OVERFLOW
    bad = fr.copy()
    bad["channel"][3] = n_ch
    assert lib.gnuais_nmea_from_frames(bad.ctypes.data, len(bad), s0.ctypes.data, n_ch, None, 0,
                                       C.byref(need), C.byref(ns)) == -1   # GNUAIS_E_ARG


@pytest.mark.skipif(not have_reference(), reason="oracle/_ref not built (reference tree absent)")
@pytest.mark.parametrize("seed", [61, 62, 63])
def test_nmea_random_frames_vs_reference(seed):
    fr, n_ch = cases.nmea_frames(seed=seed, n_channels=7, n_random=900)
    want, seq_want = reference().nmea_of_frames(fr, n_ch)
    got, seq = nmea(fr, n_ch)
    assert got == want
    assert np.array_equal(seq, seq_want)


@pytest.mark.gpu
def test_nmea_of_device_chain_frames():
    """Full chain on the GPU -> frame records -> sentences == the golden text the reference
    produced from ITS frames for the same input."""
    import torch
    from gnuais_amd import ReceiverBatch
    g = np.load(os.path.join(G, "nmea.npz"))
    for name in ("chain_48k", "chain_long"):
        c = np.load(os.path.join(G, name + ".npz"))
        x = c["x"]
        b = ReceiverBatch(x.shape[1], max_len=x.shape[0])
        b.run(torch.from_numpy(np.ascontiguousarray(x)).cuda())
        fr = b.drain_frames()
        got, _ = nmea(fr, x.shape[1])
        assert got == g[name + "_text"].tobytes()


@pytest.mark.gpu
def test_device_formatters_directly_against_the_golden_text():
    """The DEVICE formatters pinned to the reference without the host formatter in between: full chain on the GPU,
    then gnuais_batch_drain_messages() / gnuais_batch_drain_nmea() / the streamed path -- sentences and stdout lines
    formatted on the device -- against the text the reference itself printed for the same input (nmea.npz), and the
    golden frames of all 24 message types (sorted into print order, re-sent through the device deframer) against the
    reference's text for them."""
    import torch
    from gnuais_amd import ReceiverBatch, synth
    g = np.load(os.path.join(G, "nmea.npz"))
    for name in ("chain_48k", "chain_long"):
        x = np.load(os.path.join(G, name + ".npz"))["x"]
        xd = torch.from_numpy(np.ascontiguousarray(x)).cuda()
        n_ch = x.shape[1]
        b = ReceiverBatch(n_ch, max_len=x.shape[0])
        b.run(xd)
        seq = np.zeros(n_ch, dtype=np.uint8)
        nm, tx, n_sent, n_lines, n_frames = b.drain_messages(seq)
        assert nm == g[name + "_text"].tobytes() and tx == g[name + "_stdout"].tobytes()
        assert np.array_equal(seq, g[name + "_seqnr"]) and n_frames > 0
        b2 = ReceiverBatch(n_ch, max_len=x.shape[0])
        b2.run(xd)
        seq2 = np.zeros(n_ch, dtype=np.uint8)
        assert b2.drain_nmea(seq2)[0] == g[name + "_text"].tobytes() and np.array_equal(seq2, seq)
        b3 = ReceiverBatch(n_ch, max_len=x.shape[0])            # streamed: handed out stream_depth calls later
        b3.run(xd)
        texts = [b3.stream_nmea()[0] for _ in range(b3.stream_depth + 1)]
        assert b"".join(texts) == g[name + "_text"].tobytes()
    # every message type: the golden frames in the reference's print order (stable by channel)
    gold = np.frombuffer(np.ascontiguousarray(g["synthetic_frames"]).tobytes(), dtype=FRAME_DTYPE)
    n_ch = int(g["synthetic_nch"][0])
    keep = gold[(gold["nbits"] >= 8) & (gold["nbits"] % 8 == 0)]
    keep = keep[np.argsort(keep["channel"], kind="stable")]
    from gnuais_amd import messages_from_frames
    want_nm, want_tx = messages_from_frames(keep, np.zeros(n_ch, dtype=np.uint8))     # == reference: test_stdout_text_golden
    streams = [[np.zeros(8, dtype=np.uint8)] for _ in range(n_ch)]
    for f in keep:
        streams[int(f["channel"])].append(synth.hdlc_frame_bits(bytes(f["payload"][: int(f["nbits"]) // 8]), training_bits=24))
        streams[int(f["channel"])].append(np.zeros(5, dtype=np.uint8))
    b = ReceiverBatch(n_ch, max_len=48000)
    b.decode_bits([np.concatenate(s_).astype(np.uint8) for s_ in streams])
    nm, tx, _, _, n_frames = b.drain_messages(np.zeros(n_ch, dtype=np.uint8))
    assert n_frames == len(keep) > 500 and nm == want_nm and tx == want_tx
    # and those lines are the reference's own: the golden stdout holds each of them verbatim
    ref_lines = set(g["synthetic_stdout"].tobytes().split(b"\n"))
    body = lambda l: l.split(b" (!AIVDM")[0]                     # the sequence digit differs with the order
    ref_bodies = {body(l) for l in ref_lines}
    assert all(body(l) in ref_bodies for l in tx.split(b"\n") if l)


@pytest.mark.gpu
def test_wav_file_to_sentences_end_to_end(tmp_path):
    """f2 + hot path + f1: a stereo WAV of the golden input, read properly, decodes to the golden
    sentences whatever the call size."""
    import torch
    from gnuais_amd import ReceiverBatch, io, messages_from_frames
    g = np.load(os.path.join(G, "nmea.npz"))
    c = np.load(os.path.join(G, "chain_48k.npz"))
    p = tmp_path / "in.wav"
    io.write_wav(str(p), 48000, c["x"])
    rate, x = io.read_wav(str(p))
    assert rate == 48000 and np.array_equal(x, c["x"])
    for call in (1020, 7777, x.shape[0]):
        b = ReceiverBatch(x.shape[1], max_len=call)
        seq = np.zeros(x.shape[1], dtype=np.uint8)
        frames = []
        for part in io.chunks(x, call):
            b.run(torch.from_numpy(np.array(part, dtype=np.int16)).cuda())
            frames.append(b.drain_frames())
        fr = np.concatenate(frames)
        fr = fr[np.argsort(fr["channel"], kind="stable")]          # the golden text is per receiver, in time order
        nm, tx = messages_from_frames(fr, seq)
        assert nm == g["chain_48k_text"].tobytes()
        assert tx == g["chain_48k_stdout"].tobytes()


@pytest.mark.gpu
def test_device_formatter_matches_host_formatter_on_arbitrary_frames():
    """gnuais_batch_drain_nmea(): sentences formatted on the device from the HBM frame ring.  Frames of
    every length (1..53 bytes: one- and two-part sentences, every fill-bit count) and every type
    field (accepted and ignored ones) are put into the ring through the device deframer; the text,
    the sentence count and the carried sequence digits must equal the host formatter's over the
    drained records -- which the tests above pin to the reference."""
    from gnuais_amd import ReceiverBatch, synth
    rng = np.random.default_rng(77)
    n_ch = 7
    streams = []
    for c in range(n_ch):
        parts = [np.zeros(8, dtype=np.uint8)]
        for k in range(int(rng.integers(0, 60)) if c != 3 else 0):       # channel 3 stays empty
            nbytes = int(rng.integers(1, 54)) if k % 3 else int(rng.choice([1, 21, 45, 46, 47, 53]))
            body = bytearray(rng.integers(0, 256, nbytes, dtype=np.uint8).tobytes())
            t = int(rng.integers(0, 64)) if k % 4 == 0 else int(rng.integers(1, 25))
            body[0] = (t << 2) | (body[0] & 3)
            parts.append(synth.hdlc_frame_bits(bytes(body), training_bits=24))
            parts.append(np.zeros(int(rng.integers(1, 30)), dtype=np.uint8))
        streams.append(np.concatenate(parts).astype(np.uint8))
    seq0 = rng.integers(0, 10, n_ch).astype(np.uint8)
    a, b = ReceiverBatch(n_ch, max_len=48000), ReceiverBatch(n_ch, max_len=48000)
    for piece in (0, 1):                                  # two spans: the digits carry over
        half = [st[: len(st) // 2] if piece == 0 else st[len(st) // 2:] for st in streams]
        a.decode_bits(half)
        b.decode_bits(half)
        if piece == 0:
            seq_a, seq_b = seq0.copy(), seq0.copy()
        frames = a.drain_frames()
        want = nmea(frames, n_ch, seq_a)[0]
        got, n_sent, n_frames = b.drain_nmea(seq_b)
        assert n_frames == len(frames) > 20
        assert got == want
        assert n_sent == want.count(b"\r\n")
        assert np.array_equal(seq_a, seq_b)
        assert b.pending_frames() == 0
    assert b"!AIVDM,2,2," in want or b"!AIVDM,2,2," in got


@pytest.mark.gpu
def test_device_formatter_on_the_chain_over_several_calls():
    """Full chain, three calls queued before one drain (frames of different calls interleave per
    channel in the ring), more channels than a block: device text == host text == golden order."""
    import torch
    from gnuais_amd import ReceiverBatch, synth
    n_ch, per = 300, 6 * 1280
    x = np.stack([synth.make_stream(3 * per, seed=81, channel=c, occupancy=0.9)[0] for c in range(n_ch)], axis=1)
    a, b = ReceiverBatch(n_ch, max_len=per), ReceiverBatch(n_ch, max_len=per)
    xd = torch.from_numpy(x).cuda()
    for i in range(3):
        a.run(xd[i * per:(i + 1) * per])
        b.run(xd[i * per:(i + 1) * per])
    frames = a.drain_frames()
    seq_a, seq_b = np.zeros(n_ch, dtype=np.uint8), np.zeros(n_ch, dtype=np.uint8)
    want = nmea(frames, n_ch, seq_a)[0]
    got, n_sent, n_frames = b.drain_nmea(seq_b)
    assert n_frames == len(frames) > 3000 and n_sent == n_frames
    assert got == want and np.array_equal(seq_a, seq_b)
    got2, n2, f2 = b.drain_nmea(seq_b)                     # nothing left
    assert got2 == b"" and n2 == 0 and f2 == 0
    # records and sentences of one span in one drain
    for i in range(2):
        a.run(xd[i * per:(i + 1) * per])
        b.run(xd[i * per:(i + 1) * per])
    frames = a.drain_frames()
    want = nmea(frames, n_ch, seq_a)[0]
    fr_b, text_b, n_sent = b.drain_frames_nmea(seq_b)
    assert fr_b.tobytes() == frames.tobytes() and text_b == want and n_sent == len(frames)
    assert np.array_equal(seq_a, seq_b) and b.pending_frames() == 0


@pytest.mark.gpu
def test_streamed_sentences_equal_drained_sentences():
    """gnuais_batch_stream_nmea(): one call per run, the text arriving from pinned
    memory `stream_depth` calls later -- byte for byte what gnuais_batch_drain_nmea() returns for the same runs, the sequence
    digits carried across runs, empty runs and the final flush included."""
    import torch
    from gnuais_amd import ReceiverBatch, synth
    n_ch, call, n_calls = 300, 2 * 1280, 9
    x = np.stack([synth.make_stream(call * n_calls, seed=81, channel=c, occupancy=0.8)[0] for c in range(n_ch)], axis=1)
    x[3 * call:4 * call] = 0                                    # a run without a single frame
    xd = torch.from_numpy(x).cuda()
    a, b = ReceiverBatch(n_ch, max_len=call), ReceiverBatch(n_ch, max_len=call)
    seq = np.zeros(n_ch, dtype=np.uint8)
    want, got = [], []
    for i in range(n_calls):
        a.run(xd[i * call:(i + 1) * call])
        want.append(a.drain_nmea(seq))
        b.run(xd[i * call:(i + 1) * call], sync=False)
        got.append(b.stream_nmea())
    depth = b.stream_depth
    for _ in range(depth):                                      # flush
        got.append(b.stream_nmea())
    assert all(g[2] == -1 for g in got[:depth]) and all(g[2] >= 0 for g in got[depth:])
    out = got[depth:]
    assert len(out) == n_calls
    assert sum(w[2] for w in want) > 1000 and want[3][2] < want[2][2]
    for i, (w, g) in enumerate(zip(want, out)):
        assert g[2] == w[2] and g[1] == w[1], i
        assert g[0] == w[0], i


@pytest.mark.gpu
def test_streamed_sentences_with_several_runs_per_call():
    """A ring that received more than one run has no single chunk table: the formatter falls back to
    the radix sort; rings with exactly one run take their order from K3's chunk table.  Both mixed in
    one sequence, against drain_nmea() over the same spans."""
    import torch
    from gnuais_amd import ReceiverBatch, synth
    n_ch, call = 200, 2 * 1280
    plan = [1, 2, 1, 1, 3, 1, 2, 1, 1]                          # runs per stream call
    total = sum(plan)
    x = np.stack([synth.make_stream(call * total, seed=83, channel=c, occupancy=0.9)[0] for c in range(n_ch)], axis=1)
    xd = torch.from_numpy(x).cuda()
    a, b = ReceiverBatch(n_ch, max_len=call), ReceiverBatch(n_ch, max_len=call)
    b.autotune(xd[:call].contiguous())
    assert b.autotune_delivery(xd[:call].contiguous()) > 0      # places the copy stream; resets, stays streaming
    seq = np.zeros(n_ch, dtype=np.uint8)
    want, got = [], []
    i = 0
    for runs in plan:
        for _ in range(runs):
            a.run(xd[i * call:(i + 1) * call])
            b.run(xd[i * call:(i + 1) * call], sync=False)
            i += 1
        want.append(a.drain_nmea(seq))
        got.append(b.stream_nmea())
    for _ in range(b.stream_depth):
        got.append(b.stream_nmea())
    out = got[b.stream_depth:]
    assert len(out) == len(plan) and sum(w[2] for w in want) > 1000
    for k, (w, g) in enumerate(zip(want, out)):
        assert g[2] == w[2] and g[1] == w[1], k
        assert g[0] == w[0], k


@pytest.mark.gpu
def test_device_message_lines_match_host_formatter():
    """gnuais_batch_drain_messages(): the stdout line of every accepted frame formatted on the device
    (every per-type decoder, %.6f / %.1f / %.0f of the reference's float expressions done in integer
    arithmetic) == the host formatter's text over the drained records -- which test_stdout_text_* pin to
    the reference's own printf.  Frames: the golden set (all 24 types, extreme coordinates) and random
    payloads of every length, through the device deframer; two spans so the digits carry over."""
    from gnuais_amd import ReceiverBatch, synth, messages_from_frames
    g = np.load(os.path.join(G, "nmea.npz"))
    gold = np.frombuffer(np.ascontiguousarray(g["synthetic_frames"]).tobytes(), dtype=FRAME_DTYPE)
    rnd, n_ch = cases.nmea_frames(seed=91, n_channels=6, n_random=1500)
    # coordinates whose sixth decimal sits on a rounding boundary: lat / 600000 for many values
    rng = np.random.default_rng(92)
    extra = []
    for k in range(600):
        bits = rng.integers(0, 2, 53 * 8).astype(np.uint8)
        t = (1, 2, 3, 18, 4)[k % 5]
        for i in range(6):
            bits[i] = (t >> (5 - i)) & 1
        lat = int(rng.integers(-(1 << 26), 1 << 26)) if k % 4 else int(rng.choice([0, 1, -1, 3, -3, 300000, -300000, 54000000]))
        pos = {1: 89, 2: 89, 3: 89, 18: 85, 4: 107}[t]
        for i in range(27):
            bits[pos + i] = (lat >> (26 - i)) & 1
        f = np.zeros(1, dtype=FRAME_DTYPE)[0]
        f["channel"] = k % n_ch
        f["payload"] = np.packbits(bits)
        f["nbits"] = 168
        extra.append(f)
    allf = list(gold) + list(rnd) + extra
    streams = [[np.zeros(8, dtype=np.uint8)] for _ in range(n_ch)]
    for f in allf:
        body = bytes(f["payload"][: int(f["nbits"]) // 8])
        if len(body) < 1:
            continue
        c = int(f["channel"]) % n_ch
        streams[c].append(synth.hdlc_frame_bits(body, training_bits=24))
        streams[c].append(np.zeros(5, dtype=np.uint8))
    streams = [np.concatenate(s).astype(np.uint8) for s in streams]
    chanid = b"ABXYZQ"
    a, b = ReceiverBatch(n_ch, max_len=48000), ReceiverBatch(n_ch, max_len=48000)
    seq_a, seq_b = np.zeros(n_ch, dtype=np.uint8), np.zeros(n_ch, dtype=np.uint8)
    types = set()
    for piece in (0, 1):
        half = [st[: len(st) // 2] if piece == 0 else st[len(st) // 2:] for st in streams]
        a.decode_bits(half)
        b.decode_bits(half)
        frames = a.drain_frames()
        want_nm, want_tx = messages_from_frames(frames, seq_a, chanid if piece else None)
        nm, tx, n_sent, n_lines, n_frames = b.drain_messages(seq_b, chanid if piece else None)
        assert n_frames == len(frames) > 500
        assert nm == want_nm and n_sent == want_nm.count(b"\r\n")
        if tx != want_tx:                             # show the first differing line
            for x, y in zip(tx.split(b"\n"), want_tx.split(b"\n")):
                assert x == y
        assert tx == want_tx and n_lines == want_tx.count(b"\n")
        assert np.array_equal(seq_a, seq_b) and b.pending_frames() == 0
        types |= {l.split(b" type ")[1].split(b" ")[0] for l in want_tx.split(b"\n") if b" type " in l}
    assert types == {str(t).encode() for t in range(1, 25)}
    assert b"(tide-weather) lat" in want_tx or b"(tide-weather) lat" in tx


@pytest.mark.gpu
def test_streamed_delivery_reports_a_ring_overflow_with_its_text():
    """A call with more CRC-valid frames than `frame_capacity`: the streamed path says so
    (GNUAIS_E_OVERFLOW) on the call that hands that call's text out -- the text itself holds whole
    sentences of real frames (K3's chunk counts are clipped to the ring) -- and the calls before and
    after it are delivered whole."""
    import ctypes as C
    import torch
    from gnuais_amd import ReceiverBatch, lib, synth
    n_ch, call = 40, 10 * 1280
    dense = np.stack([synth.make_stream(call, seed=97, channel=c, occupancy=1.0)[0] for c in range(n_ch)], axis=1)
    quiet = np.stack([synth.make_stream(call, seed=98, channel=c, occupancy=0.02)[0] for c in range(n_ch)], axis=1)
    xd, xq = torch.from_numpy(dense).cuda(), torch.from_numpy(quiet).cuda()
    a = ReceiverBatch(n_ch, max_len=call)
    b = ReceiverBatch(n_ch, max_len=call, frame_capacity=64)
    plan = [xq, xd, xq, xq]
    seq = np.zeros(n_ch, dtype=np.uint8)
    want = []
    for x in plan:
        a.run(x)
        want.append(a.drain_nmea(seq))
    assert want[1][2] > 64 >= max(want[0][2], want[2][2], want[3][2])
    depth = b.stream_depth
    got = []
    for i in range(len(plan) + depth):
        if i < len(plan):
            b.run(plan[i], sync=False)
        text, ln, ns, nf = C.c_char_p(), C.c_size_t(0), C.c_int(0), C.c_int(0)
        rc = b._lib.gnuais_batch_stream_nmea(b._h, C.byref(text), C.byref(ln), C.byref(ns), C.byref(nf))
        got.append((rc, C.string_at(text, ln.value) if ln.value else b"", ns.value, nf.value))
    out = got[depth:]
    assert [g[0] for g in out] == [0, lib.E_OVERFLOW, 0, 0]
    assert out[0][1] == want[0][0] and out[0][3] == want[0][2]
    assert 0 < out[1][3] <= 64
    lines = out[1][1].split(b"\r\n")[:-1]
    assert len(lines) == out[1][2] >= out[1][3] and all(l.startswith(b"!AIVDM,") for l in lines)
    for l in lines:
        x_ = 0
        for ch in l[1:l.index(b"*")]:
            x_ ^= ch
        assert l.endswith(b"*%02X" % x_)
    # the calls after it: all their frames (the sequence digits may differ: the overflowed call accepted fewer)
    assert out[2][3] == want[2][2] and out[3][3] == want[3][2]


@pytest.mark.gpu
@pytest.mark.parametrize("n_ch,call,pipeline", [(1, 2560, 1), (33, 1280, 1), (70, 2560, 0), (5, 77, 1)])
def test_streamed_delivery_odd_shapes_reset_and_teardown(n_ch, call, pipeline):
    """Streaming with one channel, a channel count that is no multiple of a K3 block, the pipeline off, and
    77-sample calls; a reset in the middle (the stream starts afresh, nothing of before is handed out); a
    batch destroyed with calls in flight."""
    import torch
    from gnuais_amd import ReceiverBatch, synth
    total = call * 14
    x = np.stack([synth.make_stream(total, seed=101, channel=c, occupancy=0.95)[0] for c in range(n_ch)], axis=1)
    xd = torch.from_numpy(x).cuda()
    a, b = ReceiverBatch(n_ch, max_len=call), ReceiverBatch(n_ch, max_len=call)
    b.set_option("pipeline", pipeline)
    depth = b.stream_depth

    def both(lo, hi, seq):
        want, got = [], []
        for i in range(lo, hi):
            a.run(xd[i * call:(i + 1) * call])
            want.append(a.drain_nmea(seq))
            b.run(xd[i * call:(i + 1) * call], sync=False)
            got.append(b.stream_nmea())
        for _ in range(depth):
            got.append(b.stream_nmea())
        return want, got[depth:]

    seq = np.zeros(n_ch, dtype=np.uint8)
    want, out = both(0, 6, seq)
    assert [(g[0], g[1], g[2]) for g in out] == [(w[0], w[1], w[2]) for w in want]
    a.reset(); b.reset()
    seq[:] = 0
    first = b.stream_nmea()
    assert first[2] == -1                                   # nothing from before the reset
    want, out = both(6, 14, seq)
    got = [first] + out
    got = [g for g in got if g[2] >= 0]
    assert [(g[0], g[1], g[2]) for g in got[-len(want):]] == [(w[0], w[1], w[2]) for w in want]
    if call >= 1280:
        assert sum(w[2] for w in want) > 0
    for i in range(3):                                       # leave work in flight
        b.run(xd[i * call:(i + 1) * call], sync=False)
        b.stream_nmea()
    b.close()


@pytest.mark.gpu
def test_device_message_layer_edge_cases():
    """Nothing queued, one channel, too small buffers, a streaming batch: the drain-type device calls say what
    they must and consume nothing they should not."""
    import ctypes as C
    import torch
    from gnuais_amd import ReceiverBatch, lib, synth
    b = ReceiverBatch(1, max_len=4 * 1280)
    seq = np.zeros(1, dtype=np.uint8)
    assert b.drain_messages(seq) == (b"", b"", 0, 0, 0)
    assert len(b.fold_vessels()) == 0
    x = synth.make_stream(4 * 1280, seed=103, channel=0, occupancy=1.0)[0].reshape(-1, 1)
    b.run(torch.from_numpy(x).cuda())
    n = b.pending_frames()
    assert n >= 2
    nl, tl, ns, nln, nf = C.c_size_t(0), C.c_size_t(0), C.c_int(0), C.c_int(0), C.c_int(0)
    small = np.zeros(16, dtype=np.uint8)
    rc = b._lib.gnuais_batch_drain_messages(b._h, seq.ctypes.data, None, small.ctypes.data, small.size, C.byref(nl),
                                            C.byref(ns), small.ctypes.data, small.size, C.byref(tl), C.byref(nln), C.byref(nf))
    assert rc == lib.E_ARG and b.pending_frames() == n            # nothing consumed
    tab = b.fold_vessels()
    assert b.pending_frames() == n and len(tab) >= 1
    nm, tx, n_sent, n_lines, n_frames = b.drain_messages(seq)
    assert n_frames == n and n_sent >= n_lines == tx.count(b"\n") > 0 and b.pending_frames() == 0
    b.stream_nmea()                                               # now streaming
    rc = b._lib.gnuais_batch_drain_messages(b._h, seq.ctypes.data, None, small.ctypes.data, small.size, C.byref(nl),
                                            C.byref(ns), small.ctypes.data, small.size, C.byref(tl), C.byref(nln), C.byref(nf))
    assert rc == lib.E_ARG and b"streaming" in b._lib.gnuais_last_error()


@pytest.mark.gpu
def test_streamed_text_larger_than_the_pinned_buffer():
    """The pinned text buffers start at a fraction of the worst case and grow with the traffic: calls whose
    text does not fit (long two-sentence frames, a small frame ring) are fetched from the device copy at
    hand-out time and the buffers grow -- the bytes are the drained path's all the same."""
    from gnuais_amd import ReceiverBatch, synth
    rng = np.random.default_rng(107)
    n_ch = 6
    a = ReceiverBatch(n_ch, max_len=48000, frame_capacity=4096)
    b = ReceiverBatch(n_ch, max_len=48000, frame_capacity=4096)
    seq = np.zeros(n_ch, dtype=np.uint8)
    want, got = [], []
    for call in range(5):
        streams = []
        for c in range(n_ch):
            parts = [np.zeros(8, dtype=np.uint8)]
            for k in range(0 if call == 0 else 600):                # ~3600 frames of 53 bytes: ~0.5 MB of text
                body = bytearray(rng.integers(0, 256, 53, dtype=np.uint8).tobytes())
                body[0] = (int(rng.integers(1, 25)) << 2) | (body[0] & 3)
                parts.append(synth.hdlc_frame_bits(bytes(body), training_bits=24))
                parts.append(np.zeros(3, dtype=np.uint8))
            streams.append(np.concatenate(parts).astype(np.uint8))
        a.decode_bits(streams)
        want.append(a.drain_nmea(seq))
        b.decode_bits(streams)
        got.append(b.stream_nmea())
    depth = b.stream_depth
    for _ in range(depth):
        got.append(b.stream_nmea())
    out = got[depth:]
    assert max(len(w[0]) for w in want) > 4096 * 32 + 65536        # larger than the initial pinned buffer
    for i, (w, g) in enumerate(zip(want, out)):
        assert g[2] == w[2] and g[1] == w[1] and g[0] == w[0], i
