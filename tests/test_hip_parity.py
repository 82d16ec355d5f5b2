"""Parity of the HIP path (through the C ABI) against the golden vectors and the
CPU oracle.  Bit-exact everywhere: filter floats are compared as raw 32-bit
patterns (north_star allows 1e-5 relative; the kernels reproduce the reference
rounding sequence, so the tolerance used here is 0)."""
import os

import numpy as np
import pytest

from gnuais_amd import params, synth
from oracle_lib import FRAME_DTYPE, Oracle

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FSM_KEYS = ("state", "nstartsign", "antallpreamble", "antallenner", "bitstuff", "last", "bufferpos")


def load(name):
    return np.load(os.path.join(G, name + ".npz"))


def frames_of(raw):
    return np.frombuffer(np.ascontiguousarray(raw).tobytes(), dtype=FRAME_DTYPE)


def batch(*a, **k):
    from gnuais_amd import ReceiverBatch
    return ReceiverBatch(*a, **k)


def dev(x):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def fsm_rows(b):
    f = b.fsm_state()
    return [[int(r[k]) for k in FSM_KEYS] for r in f]


def oracle_fsm_rows(o, n, saturate=True):
    rows = []
    for c in range(n):
        h = o.hdlc(c)
        r = [h[k] for k in FSM_KEYS]
        if saturate:
            r[2] = min(r[2], 15)
        rows.append(r)
    return rows


# ---------------------------------------------------------------- FIR (K1)

@pytest.mark.parametrize("variant", [0])
def test_fir_known_answers_bit_exact(variant):
    g = load("fir_kat")
    for k in g.files:
        if not k.startswith("x_"):
            continue
        x = g[k]
        b = batch(1, max_len=int(x.size))
        b.set_option("fir_variant", variant)
        y = b.filter(dev(x[:, None])).cpu().numpy()[:, 0]
        assert np.array_equal(y.view(np.uint32), g["y_" + k[2:]]), (k, variant)
        # filter_run_buf's return value: peak positive sample of the call
        assert int(b.maxval()[0]) == int(max(0, x.max()))
        b.close()


def test_fir_192k_generic_taps_bit_exact():
    g = load("fir_kat")
    x = g["x_noise_full"]
    b = batch(1, taps=g["taps192"].view(np.float32), max_len=int(x.size))
    y = b.filter(dev(x[:, None])).cpu().numpy()[:, 0]
    assert np.array_equal(y.view(np.uint32), g["y192_noise_full"])


@pytest.mark.parametrize("variant", [0])
def test_fir_many_channels_chunked_vs_oracle(variant):
    """N not a multiple of 64, chunk lengths not multiples of 32, state carried."""
    rng = np.random.default_rng(21)
    n_ch, total = 150, 5000
    x = rng.integers(-32768, 32768, (total, n_ch)).astype(np.int16)
    x[:, 3] = 0
    x[:, 4] = 32767
    x[:, 5] = -32768
    o = Oracle(n_ch)
    want = o.run(x, want_filtered=True)["filtered"]
    b = batch(n_ch, max_len=2048)
    b.set_option("fir_variant", variant)
    b.set_option("fir_T", 256)
    got = []
    pos = 0
    for n in (1, 31, 32, 33, 2048, 700, 5, 1000, 1150):
        got.append(b.filter(dev(x[pos:pos + n])).cpu().numpy())
        pos += n
    assert pos == total
    got = np.concatenate(got, axis=0)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(b.history(), np.stack([o.history(c) for c in range(n_ch)]))


def test_fir_asymmetric_and_negative_taps_vs_oracle():
    """Non-symmetric tables take the unshared-product kernel; negative taps make
    -0 products (x = 0), which must not change any sum; an odd-sized table takes
    the generic kernel."""
    rng = np.random.default_rng(23)
    x = rng.integers(-32768, 32768, (3000, 66)).astype(np.int16)
    x[100:400, 7] = 0
    for taps in (np.concatenate([[0, 0], rng.normal(0, 0.3, 32), [0, 0]]).astype(np.float32),
                 np.concatenate([[0, 0], -np.abs(rng.normal(0, 0.3, 32)), [0, 0]]).astype(np.float32),
                 rng.normal(0, 0.3, 20).astype(np.float32)):
        o = Oracle(66, taps=taps)
        want = o.run(x, want_filtered=True)["filtered"]
        b = batch(66, taps=taps, max_len=3000)
        got = b.filter(dev(x)).cpu().numpy()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        b.close()


# ---------------------------------------------------------------- full chain

@pytest.mark.parametrize("name", ["chain_48k", "chain_192k", "chain_long"])
def test_full_chain_golden(name):
    g = load(name)
    x = g["x"]
    n_ch = x.shape[1]
    b = batch(n_ch, taps=g["taps"].view(np.float32), pllinc=int(g["pllinc"]),
              max_len=int(x.shape[0]))
    b.run(dev(x))
    bits = b.last_bits()
    for c in range(n_ch):
        assert np.array_equal(bits[c], g[f"bits{c}"]), c
    assert b.drain_frames().tobytes() == frames_of(g["frames"]).tobytes()
    cnt = b.counters()
    assert np.array_equal(np.stack([cnt["receivedframes"], cnt["lostframes"], cnt["lostframes2"]],
                                   axis=1), g["counters"])
    p = b.pll_state()
    assert np.array_equal(np.stack([p["pll"], p["prev"], p["lastbit"]], axis=1), g["pll"])
    want_fsm = g["fsm"].copy()
    want_fsm[:, 2] = np.minimum(want_fsm[:, 2], 15)
    assert fsm_rows(b) == want_fsm.tolist()
    assert np.array_equal(b.maxval(), g["maxval"])


def run_both(x, chunks, n_ch, taps=None, pllinc=0, fir_T=None, pll_variant=0, options=None):
    o = Oracle(n_ch, taps=taps, pllinc=pllinc)
    b = batch(n_ch, taps=taps, pllinc=pllinc, max_len=max(chunks))
    if fir_T:
        b.set_option("fir_T", fir_T)
    for k, v in (options or {}).items():
        b.set_option(k, v)
    b.set_option("pll_variant", pll_variant)      # 0: by channel count (the time-parallel form up to 1536 channels)
    pos = 0
    gbits = [[] for _ in range(n_ch)]
    obits = [[] for _ in range(n_ch)]
    for n in chunks:
        seg = x[pos:pos + n]
        pos += n
        r = o.run(seg, want_bits=True)
        b.run(dev(seg))
        lb = b.last_bits()
        for c in range(n_ch):
            gbits[c].append(lb[c])
            obits[c].append(r["bits"][c])
    assert pos == x.shape[0]
    for c in range(n_ch):
        assert np.array_equal(np.concatenate(gbits[c]), np.concatenate(obits[c])), c
    assert b.drain_frames().tobytes() == o.frames().tobytes()
    cnt = b.counters()
    assert np.array_equal(np.stack([cnt["receivedframes"], cnt["lostframes"], cnt["lostframes2"]],
                                   axis=1), o.counters())
    p = b.pll_state()
    assert [(int(a), int(bb), int(cc)) for a, bb, cc in zip(p["pll"], p["prev"], p["lastbit"])] == \
        [o.pll(c) for c in range(n_ch)]
    assert fsm_rows(b) == oracle_fsm_rows(o, n_ch)
    return o, b


@pytest.mark.parametrize("pll_variant", [7, 8])      # 7: the time-parallel form (pll_tp.hip); 8: one recurrence wave + three helpers (pll_h3.hip)
def test_chain_vs_oracle_ragged_chunks(pll_variant):
    n_ch, total = 70, 30 * 1280
    x = np.stack([synth.make_stream(total, seed=31, channel=c,
                                    sigma=(500.0, 1000.0, 3000.0, 6000.0, 20000.0)[c % 5])[0]
                  for c in range(n_ch)], axis=1)
    chunks = [1020] * 10 + [1, 31, 33, 4096, 4095, 7, 10000]
    chunks.append(total - sum(chunks))
    o, b = run_both(x, chunks, n_ch, fir_T=512, pll_variant=pll_variant)
    assert o.counters()[:, 0].sum() > 300


@pytest.mark.parametrize("lpw", [8, 16, 24, 32, 64])
@pytest.mark.parametrize("flag2", [0, 1])
def test_chain_vs_oracle_deframer_widths_and_flag_forms(lpw, flag2):
    """The deframer is built for 8 / 16 / 32 / 64 channels per workgroup (LDS offsets as immediates) and for any other
    width with the offsets computed; K1s gathers its flags with one alignbit per output (fir_flag2 = 1) or with the
    subtract + two alignbits of rounds 1-4.  Ragged calls over channels at five noise levels, everything == oracle."""
    n_ch, total = 70, 24 * 1280
    x = np.stack([synth.make_stream(total, seed=37, channel=c,
                                    sigma=(500.0, 1000.0, 3000.0, 6000.0, 20000.0)[c % 5])[0]
                  for c in range(n_ch)], axis=1)
    chunks = [1020] * 6 + [1, 33, 4096, 2047, 10000]
    chunks.append(total - sum(chunks))
    o, b = run_both(x, chunks, n_ch, fir_T=512, options={"hdlc_lpw": lpw, "fir_flag2": flag2})
    assert o.counters()[:, 0].sum() > 200


@pytest.mark.parametrize("pll_variant", [7, 8])
def test_chain_vs_oracle_noise_only_and_extremes(pll_variant):
    rng = np.random.default_rng(33)
    total = 40000
    cols = [rng.normal(0, s, total) for s in (30, 300, 3000, 12000)]
    cols.append(rng.integers(-32768, 32768, total).astype(np.float64))
    cols.append(np.zeros(total))
    cols.append(np.full(total, 32767.0))
    cols.append(np.where(np.arange(total) // 5 % 2, 9000.0, -9000.0))   # endless 0101... training
    x = np.clip(np.rint(np.stack(cols, axis=1)), -32768, 32767).astype(np.int16)
    run_both(x, [total], x.shape[1], pll_variant=pll_variant)
    run_both(x, [2049, 255, 257, total - 2561], x.shape[1], pll_variant=pll_variant)   # block / segment edges


def test_time_parallel_pll_walks_out_of_its_window():
    """pll_tp.hip: a block's map is tabulated on a window of +-64 nudges around the chunk's first value; a channel whose
    phase diffuses further inside a chunk (long stretches of loud noise: one nudge per transition, either way) ends the
    chunk where the value leaves the window, and the next chunk is centred on it.  48 000 samples of noise at several
    levels, with and without messages, one call and ragged calls, the 192 kHz parameter set too: bits, frames,
    counters, PLL carry == oracle."""
    rng = np.random.default_rng(77)
    total = 48000
    cols = [rng.normal(0, s, total) for s in (200, 1000, 5000, 20000)]
    cols += [synth.make_stream(total, seed=91, channel=c, sigma=(1000.0, 9000.0)[c % 2])[0].astype(np.float64) for c in range(4)]
    x = np.clip(np.rint(np.stack(cols, axis=1)), -32768, 32767).astype(np.int16)
    run_both(x, [total], x.shape[1], pll_variant=7)
    run_both(x, [16384, 300, 256, 257, total - 17197], x.shape[1], pll_variant=7)
    run_both(x[:40000], [40000], x.shape[1], taps=params.taps_192k(), pllinc=params.PLLINC_192K, pll_variant=7)


@pytest.mark.parametrize("pll_variant", [7, 8])
def test_pll_with_a_sign_change_at_every_sample(pll_variant):
    """A table that passes the input through (one tap) and inputs that change sign at every sample, at every second,
    third ... sample, in bursts between silences, and at random: up to 256 transitions and as many nudges per 256-sample
    block.  The time-parallel form's value then leaves its +-64 window inside ONE block, chunk after chunk (each chunk
    still advances by a block: the worst case of its walk); the lane-per-channel form's 128-sample blocks hold their
    maximum of transitions.  Several pllinc (slices every 4.7 to 20 samples).  == oracle, bit for bit."""
    rng = np.random.default_rng(123)
    total = 12000
    taps = np.zeros(9, dtype=np.float32)
    taps[4] = 1.0
    t = np.arange(total)
    cols = [np.where(t % 2, 900.0, -900.0),                                   # a transition at every sample
            np.where(t // 2 % 2, 900.0, -900.0), np.where(t // 3 % 2, 900.0, -900.0),
            np.where(t // 7 % 2, 900.0, -900.0),
            np.where((t // 700) % 2, np.where(t % 2, 900.0, -900.0), 0.0),     # bursts of them between silences
            np.where((t // 1500) % 2, np.where(t % 2, 900.0, -900.0), np.where(t // 5 % 2, 900.0, -900.0)),
            rng.choice([-500.0, 500.0], total),                                # a fair coin per sample
            np.where(rng.random(total) < 0.9, np.where(t % 2, 300.0, -300.0), 300.0)]
    x = np.stack(cols, axis=1).astype(np.int16)
    for pllinc in (0, 0x10000 // 20, 14000):
        run_both(x, [total], x.shape[1], taps=taps, pllinc=pllinc, pll_variant=pll_variant)
        run_both(x, [4097, 255, 1, 2048, total - 6401], x.shape[1], taps=taps, pllinc=pllinc, pll_variant=pll_variant)


def test_chain_vs_oracle_digital_silence_patterns():
    """All-zero stretches take K1s's all-windows-zero shortcut; everything around its edges
    (a lone nonzero sample 1..45 samples before / 1..30 after a silent word, silence that
    starts or ends mid-word, silence across call boundaries, tiny signals that keep y_c inside
    the ambiguity band) must still give the reference's bits."""
    rng = np.random.default_rng(41)
    total = 3 * 4096 + 777
    cols = [np.zeros(total)]
    for k in range(1, 48):                       # one nonzero sample, various alignments and signs
        v = np.zeros(total)
        v[1000 + 37 * k + k] = (1 if k % 2 else -1) * (1 + (k % 5) * 8000)
        v[5000 + k] = -3
        cols.append(v)
    for k in range(8):                           # silence <-> noise transitions at odd offsets
        v = rng.normal(0, 2000, total)
        v[700 + 13 * k: 2500 + 29 * k] = 0
        v[4096 - k: 4096 + 40 + k] = 0           # across the first call boundary
        v[9000 + k:] = 0
        cols.append(v)
    for k in range(8):                           # +-1 / +-2 dither: y_c stays ambiguous, y_ref does not vanish
        cols.append(rng.integers(-1 - k % 2, 2 + k % 2, total).astype(np.float64))
    x = np.clip(np.rint(np.stack(cols, axis=1)), -32768, 32767).astype(np.int16)
    run_both(x, [4096, 4096, 4096, 777], x.shape[1], fir_T=512)
    run_both(x, [total], x.shape[1])

@pytest.mark.gpu
@pytest.mark.parametrize("kernel", ["scalar12", "packed40", "packed48"])
def test_open_signs_settled_in_bulk(kernel):
    """K1s notes the outputs whose sign its central sum cannot certify and settles them lane-parallel after the segment
    (eight per lane; more are settled on the spot, all lanes together).  Sparse +-1 / +-2 dither makes most outputs of a
    word open without the word being silent, at densities from one per segment to nearly all: lists that stay short, lists
    that overflow in every word, bits patched in registers and bits patched in memory, channel counts that leave lanes
    of the last wave without a channel, segment and call boundaries.  Decisions == (reference float > 0), sample for
    sample, and the chain behind them == oracle."""
    rng = np.random.default_rng(77)
    n_ch, total = 70, 9000
    taps = params.taps_192k() if kernel != "scalar12" else None
    pllinc = params.PLLINC_192K if kernel != "scalar12" else 0
    opts = {"fir_pk_taps": 48} if kernel == "packed48" else {}      # the 192 kHz table: 40 central taps unless told otherwise
    cols = []
    for c in range(n_ch):
        p = (0.0005, 0.003, 0.02, 0.1, 0.4, 0.9)[c % 6]
        v = np.where(rng.random(total) < p, rng.integers(1, 3, total) * rng.choice([-1, 1], total), 0)
        if c % 7 == 3:
            v[2000:2600] = rng.normal(0, 4000, 600)          # a live stretch between the quiet ones
        cols.append(v)
    x = np.stack(cols, axis=1).astype(np.int16)
    o = Oracle(n_ch, taps=taps, pllinc=pllinc)
    want = (o.run(x, want_filtered=True)["filtered"] > 0).T.astype(np.uint8)
    xd = dev(x)
    for chunks in ([total], [4096, 3000, 1, 1903], [511] * 17 + [313]):
        b = batch(n_ch, taps=taps, pllinc=pllinc, max_len=max(chunks))
        for k, v in opts.items():
            b.set_option(k, v)
        assert b.info("sign_exact") == 1 and b.info("sign_central_taps") == int(kernel[-2:])
        got, pos = [], 0
        for n in chunks:
            b.run(xd[pos:pos + n])
            got.append(b.last_signs(n))
            pos += n
        got = np.concatenate(got, axis=1)
        assert np.array_equal(got, want), (chunks[0], np.argwhere(got != want)[:5])
    run_both(x, [4096, 4904], n_ch, taps=taps, pllinc=pllinc, options=opts)


def test_more_channel_groups_than_cus():
    """N/64 > 256: the PLL stage then shares CUs between workgroups (its LDS reservation is
    sized per CU); a sample of channels from the first, a middle and the last group is compared
    with the oracle."""
    n_ch, total = 259 * 64 + 17, 2 * 2048 + 100
    base = np.stack([synth.make_stream(total, seed=43, channel=c, sigma=(800.0, 4000.0)[c % 2])[0]
                     for c in range(48)], axis=1)
    x = np.ascontiguousarray(np.tile(base, (1, (n_ch + 47) // 48))[:, :n_ch])
    b = batch(n_ch, max_len=2048 + 100)
    pick = np.r_[0:40, 8000:8040, n_ch - 40:n_ch]
    o = Oracle(len(pick))
    gb = [[] for _ in pick]
    ob = [[] for _ in pick]
    for seg in (x[:2048], x[2048:4096], x[4096:]):
        r = o.run(np.ascontiguousarray(seg[:, pick]), want_bits=True)
        b.run(dev(seg))
        lb = b.last_bits()
        for i, c in enumerate(pick):
            gb[i].append(lb[c])
            ob[i].append(r["bits"][i])
    for i in range(len(pick)):
        assert np.array_equal(np.concatenate(gb[i]), np.concatenate(ob[i])), pick[i]
    p = b.pll_state()
    assert [(int(p["pll"][c]), int(p["prev"][c]), int(p["lastbit"][c])) for c in pick] == \
        [o.pll(i) for i in range(len(pick))]
    fr = b.drain_frames()
    sel = fr[np.isin(fr["channel"], pick)]
    remap = {int(c): i for i, c in enumerate(pick)}
    sel = sel.copy()
    sel["channel"] = [remap[int(c)] for c in sel["channel"]]
    assert sel.tobytes() == o.frames().tobytes()


@pytest.mark.parametrize("n_taps,shape", [(24, "tri"), (60, "gauss"), (37, "gauss"), (130, "gauss"), (36, "zero_edge")])
def test_chain_vs_oracle_other_symmetric_tables(n_taps, shape):
    """Tables other than the two the reference ships: the sign-exact slicer picks 12 or 48 central
    taps from its error bound (or leaves the table to the generic kernel: odd length differences,
    more than 128 effective taps), and the bits must not change."""
    k = np.arange(n_taps, dtype=np.float64)
    mid = (n_taps - 1) / 2.0
    if shape == "tri":
        t = 0.05 * (1.0 - np.abs(k - mid) / (mid + 1))
    elif shape == "zero_edge":
        t = 0.3 * np.exp(-((k - mid) ** 2) / (2 * 4.0 ** 2))
        t[:3] = 0.0
        t[-3:] = 0.0
    else:
        t = 0.25 * np.exp(-((k - mid) ** 2) / (2 * (n_taps / 9.0) ** 2))
    taps = t.astype(np.float32)
    assert np.array_equal(taps, taps[::-1])
    total = 3 * 2048 + 300
    x = np.stack([synth.make_stream(total, seed=37, channel=c, sigma=(600.0, 2500.0, 9000.0)[c % 3])[0]
                  for c in range(9)], axis=1)
    x[:, 8] = 0                                                  # and a silent channel
    run_both(x, [2048, 2048, 2048, 300], 9, taps=taps)


@pytest.mark.parametrize("pk_taps", [0, 48])             # central taps of the packed slicer: 40 (default for this table) or 48
def test_chain_192k_vs_oracle(pk_taps):
    total = 6 * 5120
    x = np.stack([synth.make_stream(total, seed=35, channel=c, sps=20, sigma=1500.0,
                                    occupancy=0.8)[0] for c in range(5)], axis=1)
    run_both(x, [4096, total - 4096], 5, taps=params.taps_192k(), pllinc=params.PLLINC_192K, options={"fir_pk_taps": pk_taps})


@pytest.mark.parametrize("pk_taps", [0, 48])
def test_chain_192k_sparse_impulses_and_ragged_ends(pk_taps):
    """Impulses in digital silence under the 144-tap table: nearly every sample is undecidable from
    the central taps, the silence shortcut must see every sample a window touches -- also in a last
    word that ends in its first half (found by scripts/fuzz_parity.py, seed 218)."""
    rng = np.random.default_rng(218)
    total = 96 + 1020 + 9999 + 1020 + 3000
    cols = []
    for c in range(6):
        x = np.zeros(total, dtype=np.int16)
        at = rng.integers(0, total, total // (40, 97, 300)[c % 3])
        x[at] = rng.integers(-32768, 32768, len(at))
        cols.append(x)
    cols.append(np.zeros(total, dtype=np.int16))
    x = np.stack(cols, axis=1)
    for chunks in ([96, 1020, 9999, 1020, 3000], [4111, 4113, 4127, total - 12351]):   # lengths = 15, 17, 31 mod 32
        run_both(x, chunks, x.shape[1], taps=params.taps_192k(), pllinc=params.PLLINC_192K, options={"fir_pk_taps": pk_taps})


@pytest.mark.parametrize("n_ch", [64, 192])
def test_chain_192k_on_the_matrix_pipe(n_ch):
    """fir_sign_mfma.hip: with whole groups of 64 channels and calls longer than the head the packed kernel keeps (640
    outputs), the 144-tap table's slicer runs everything behind that head as an integer Toeplitz product on the matrix pipe.  Messages at several noise
    levels, noise alone from 3 to 20 000 rms, full-scale random samples, digital silence with sparse impulses, silence
    that ends and begins inside a segment, constant levels: bits, frames, counters, PLL carry == oracle; calls of one
    and of several segments, ragged, and short calls in between (the packed kernel alone)."""
    rng = np.random.default_rng(4242)
    total = 4 * 1920 + 777
    cols = []
    for c in range(n_ch):
        k = c % 16
        if k < 6:
            v = synth.make_stream(total, seed=57, channel=c, sps=20, sigma=(0.0, 300.0, 1500.0, 6000.0, 20000.0, 1000.0)[k],
                                  occupancy=0.8)[0].astype(np.float64)
        elif k < 10:
            v = rng.normal(0, (3.0, 40.0, 1000.0, 20000.0)[k - 6], total)
        elif k == 10:
            v = rng.integers(-32768, 32768, total).astype(np.float64)
        elif k == 11:
            v = np.zeros(total)
            at = rng.integers(0, total, total // 90)
            v[at] = rng.integers(-32768, 32768, len(at))
        elif k == 12:
            v = rng.normal(0, 2000.0, total)
            v[2500:5200] = 0                                          # silence inside segments
            v[6100:6140] = 0
        elif k == 13:
            v = np.zeros(total)
        elif k == 14:
            v = np.full(total, (32767.0, -32768.0, 1.0, -1.0)[(c // 16) % 4])
        else:
            v = np.where(rng.random(total) < 0.02, rng.integers(-2, 3, total), 0).astype(np.float64)   # +-1 / +-2 dither in silence
        cols.append(v)
    x = np.clip(np.rint(np.stack(cols, axis=1)), -32768, 32767).astype(np.int16)
    kw = dict(taps=params.taps_192k(), pllinc=params.PLLINC_192K)
    b = batch(n_ch, max_len=total, **kw)
    assert b.info("sign_matrix_pipe") == 1
    b.close()
    run_both(x, [total], n_ch, **kw)
    run_both(x, [1921, 1920, 100, 3841, total - 7782], n_ch, **kw)
    # around the head the packed kernel keeps (640 outputs for this table): calls that end on it, one short of it, one past it
    run_both(x, [640, 641, 639, 768, 1281, 2560, total - 6529], n_ch, **kw)
    run_both(x, [total], n_ch, options={"fir_pk_taps": 48}, **kw)
    run_both(x, [total], n_ch, options={"fir_mfma": 0}, **kw)


def test_shards_equal_whole():
    """channels are independent -- two half batches == one batch."""
    n_ch, total = 128, 8 * 1280
    x = np.stack([synth.make_stream(total, seed=37, channel=c)[0] for c in range(n_ch)], axis=1)
    whole = batch(n_ch, max_len=total)
    whole.run(dev(x))
    fw = whole.drain_frames()
    parts = []
    for s in range(2):
        h = batch(n_ch // 2, max_len=total)
        h.run(dev(x[:, s * 64:(s + 1) * 64]))
        f = h.drain_frames()
        f["channel"] += s * 64
        parts.append(f)
    assert np.concatenate(parts).tobytes() == fw.tobytes()


# ---------------------------------------------------------------- deframer (K2b) + CRC

@pytest.mark.parametrize("name", ["random_p50", "random_p70", "random_p30", "alternating",
                                  "crafted", "mixed"])
def test_deframer_golden_bits(name):
    g = load("deframer_bits")
    bits = np.unpackbits(g["bits_" + name])[: int(g["n_" + name])]
    b = batch(1, max_len=4096)      # small bit ring: forces several decode chunks
    b.decode_bits([bits])
    cnt = b.counters()[0]
    assert [int(cnt["receivedframes"]), int(cnt["lostframes"]), int(cnt["lostframes2"])] == \
        g["counters_" + name].tolist()
    assert b.drain_frames().tobytes() == frames_of(g["frames_" + name]).tobytes()
    want = g["fsm_" + name].tolist()
    want[2] = min(want[2], 15)
    assert fsm_rows(b)[0] == want


def test_deframer_random_vs_oracle_many_channels():
    rng = np.random.default_rng(41)
    n_ch = 96
    streams = []
    for c in range(n_ch):
        parts = []
        for i in range(40):
            parts.append((rng.random(int(rng.integers(0, 150))) < rng.random()).astype(np.uint8))
            if rng.random() < 0.7:
                n = int(rng.choice([0, 1, 11, 21, 21, 40, 53, 54]))
                fb = synth.hdlc_frame_bits(bytes(rng.integers(0, 256, n, dtype=np.uint8)),
                                           training_bits=int(rng.integers(0, 40)))
                if rng.random() < 0.2:
                    fb[int(rng.integers(0, fb.size))] ^= 1
                parts.append(fb)
        streams.append(np.concatenate(parts).astype(np.uint8))
    o = Oracle(n_ch)
    for c in range(n_ch):
        o.decode_bits(c, streams[c])
    b = batch(n_ch, max_len=48000)
    b.decode_bits(streams)
    assert b.drain_frames().tobytes() == o.frames().tobytes()
    cnt = b.counters()
    assert np.array_equal(np.stack([cnt["receivedframes"], cnt["lostframes"], cnt["lostframes2"]],
                                   axis=1), o.counters())
    assert fsm_rows(b) == oracle_fsm_rows(o, n_ch)


def test_deframer_densest_possible_frames():
    """Frames opened as fast as the state machine allows (16 alternating bits, flag, a few data bits,
    flag: one candidate per ~32-70 bits, ten times real traffic): every one of them is checked and
    counted, none is dropped for lack of candidate slots."""
    flag = np.array([0, 1, 1, 1, 1, 1, 1, 0], dtype=np.uint8)
    alt = (np.arange(16) & 1).astype(np.uint8)
    rng = np.random.default_rng(44)
    streams = []
    for reps, datalen in ((300, 0), (300, 30), (250, 40), (150, 64)):
        data = rng.integers(0, 2, datalen).astype(np.uint8)
        for i in range(4, datalen):
            if data[i - 4:i].all():
                data[i] = 0
        streams.append(np.tile(np.concatenate([alt, flag, data, flag]), reps).astype(np.uint8))
    o = Oracle(len(streams))
    for c, st in enumerate(streams):
        o.decode_bits(c, st)
    b = batch(len(streams), max_len=48000)
    b.decode_bits(streams)
    assert b.drain_frames().tobytes() == o.frames().tobytes()
    cnt = b.counters()
    got = np.stack([cnt["receivedframes"], cnt["lostframes"], cnt["lostframes2"]], axis=1)
    assert np.array_equal(got, o.counters())
    assert got[0, 2] == 300 and got[1, 1] == 300


def test_crc16_device_known_answers():
    from gnuais_amd import crc16_batch
    g = load("crc16")
    data, pos, msgs = g["data"].tobytes(), 0, []
    for n in g["lens"]:
        msgs.append(data[pos:pos + n])
        pos += n
    got = crc16_batch(msgs)
    assert np.array_equal(got, g["crc"])
    assert int(crc16_batch([b"123456789"])[0]) == 0x906E


def test_crc16_bits_is_protodec_calculate_crc_in_one_call():
    """gnuais_crc16_bits(): the cells of d->buffer (one bit each, the first cell of a byte its LEAST significant bit,
    protodec.c:138-143) -> the CRC-16/X-25 register over the packed bytes (0x0f47 for a good frame + FCS, :166) and the
    cells back byte by byte MOST significant bit first (d->rbuffer, :150-162).  Against the oracle's CRC and numpy's bit
    packing for every golden message with its FCS appended, lengths 1 .. 64 bytes, and cells that are not 0 / 1 (the
    reference shifts whatever the cell holds)."""
    import ctypes as C
    from gnuais_amd import lib as _lib
    from oracle_lib import crc16_x25
    L = _lib.load()
    g = load("crc16")
    data, pos, msgs = g["data"].tobytes(), 0, []
    for n in g["lens"]:
        msgs.append(data[pos:pos + n])
        pos += n
    rng = np.random.default_rng(9)
    msgs += [bytes(rng.integers(0, 256, n, dtype=np.uint8)) for n in (1, 2, 7, 55, 62)]
    for m in msgs:
        if len(m) > 62 or len(m) == 0:
            continue
        fcs = crc16_x25(m)
        framed = m + bytes([fcs & 0xFF, fcs >> 8])
        cells = np.unpackbits(np.frombuffer(framed, dtype=np.uint8), bitorder="little").astype(np.uint8)
        out = np.full(8 * len(m), 7, dtype=np.uint8)
        crc = C.c_uint16(0)
        assert L.gnuais_crc16_bits(0, cells.ctypes.data, len(framed), C.byref(crc), out.ctypes.data, len(out)) == 0
        assert crc.value == 0x0F47
        assert np.array_equal(out, np.unpackbits(np.frombuffer(m, dtype=np.uint8), bitorder="big"))
        crc2 = C.c_uint16(0)
        assert L.gnuais_crc16_bits(0, cells.ctypes.data, len(m), C.byref(crc2), None, 0) == 0
        assert crc2.value == fcs
    odd = np.array([3, 0, 0, 0, 0, 0, 0, 2] * 2, dtype=np.uint8)      # cell << i, narrowed to a byte: 0x03 | 0x00 (2 << 7 falls out)
    crc = C.c_uint16(0)
    out = np.zeros(16, dtype=np.uint8)
    assert L.gnuais_crc16_bits(0, odd.ctypes.data, 2, C.byref(crc), out.ctypes.data, 16) == 0
    assert crc.value == crc16_x25(bytes([3, 3])) and out.tolist() == [0, 0, 0, 0, 0, 0, 1, 1] * 2
    assert L.gnuais_crc16_bits(0, odd.ctypes.data, 65, C.byref(crc), None, 0) != 0         # more than 64 bytes: refused


@pytest.mark.parametrize("opts", [dict(nbuf=2), dict(nbuf=7), dict(nbuf=5, hdlc_lpw=64), dict(hdlc_lpw=8)])
def test_scheduling_options_leave_every_result_alone(opts):
    """The host-side knobs -- hand-off depth, channels per deframer wave -- only move work around: bits, frames, counters,
    PLL carry and peaks equal the oracle's over ragged calls, queued without a sync in between."""
    import torch
    n_ch, total = 130, 24 * 1280
    x = np.stack([synth.make_stream(total, seed=57, channel=c, sigma=(800.0, 2500.0)[c % 2])[0] for c in range(n_ch)], axis=1)
    chunks = [4096, 1020, 33, 7000, 5000, 1, 300]
    chunks.append(total - sum(chunks))
    o = Oracle(n_ch)
    b = batch(n_ch, max_len=max(chunks))
    for k, v in opts.items():
        b.set_option(k, v)
    stream = torch.cuda.current_stream().cuda_stream
    pos = 0
    xs = []
    for n in chunks:
        seg = x[pos:pos + n]
        pos += n
        o.run(seg)
        xs.append(dev(seg))
        b.run(xs[-1], stream=stream, sync=False)               # no sync: the calls pile up in the stage pipeline
    b.sync()
    assert b.drain_frames().tobytes() == o.frames().tobytes()
    cnt = b.counters()
    assert np.array_equal(np.stack([cnt["receivedframes"], cnt["lostframes"], cnt["lostframes2"]], axis=1), o.counters())
    p = b.pll_state()
    assert [(int(a), int(bb), int(cc)) for a, bb, cc in zip(p["pll"], p["prev"], p["lastbit"])] == [o.pll(c) for c in range(n_ch)]
    assert np.array_equal(b.maxval(), np.where(x[-chunks[-1]:].max(axis=0) > 0, x[-chunks[-1]:].max(axis=0), 0).astype(np.int16))
    assert np.array_equal(b.history(), x[-36:].T)


# ---------------------------------------------------------------- full size (BASELINE C3)

# (BASELINE's full sizes, every channel: tests/test_hip_fullsize.py)


# ---------------------------------------------------------------- randomised soak (short form)

def test_randomised_soak_short():
    """scripts/fuzz_parity.py for a few dozen seeds: random tables, channel counts, call lengths down
    to one sample, inputs on the slicer's threshold; then a few pipelined cases (calls queued
    asynchronously, hand-off buffers reused).  The script itself runs for as long as it is given."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(G), "..", "scripts"))
    import fuzz_parity
    for seed in list(range(1000, 1030)) + [218, 219]:
        res = fuzz_parity.one_case(seed)
        assert res.startswith("ok"), (seed, res)
    # the 192 kHz table in every case: whole 64-channel groups (64 / 128 / 192 of the script's choices) put the matrix-pipe
    # slicer under random call lengths, levels and threshold-hugging inputs
    os.environ["TABLE"] = "192k"
    try:
        for seed in range(5000, 5024):
            res = fuzz_parity.one_case(seed)
            assert res.startswith("ok"), (seed, res)
    finally:
        del os.environ["TABLE"]
    for seed in (3, 4, 5):
        res = fuzz_parity.pipelined_case(seed)
        assert res.startswith("ok"), (seed, res)
    for seed in range(2000, 2012):
        res = fuzz_parity.deframer_case(seed)
        assert res.startswith("ok"), (seed, res)


# ---------------------------------------------------------------- API behaviour

def test_autotune_keeps_results_and_resets_state():
    """gnuais_batch_autotune() re-binds stages to streams and resets the batch: what follows must
    be what a fresh batch produces."""
    g = load("chain_48k")
    x = g["x"]
    n_ch = x.shape[1]
    big = np.ascontiguousarray(np.tile(x, (1, 64))[:, :128])      # enough channels to launch every stage
    b = batch(128, max_len=x.shape[0])
    ms = b.autotune(dev(big))
    assert ms > 0 or os.environ.get("GNUAIS_PIPELINE") == "0"     # (one stream: nothing to assign, 0 is returned)
    assert int(b.counters()["receivedframes"].sum()) == 0          # reset
    b.run(dev(big))
    fr = b.drain_frames()
    first = fr[fr["channel"] < n_ch]
    assert first.tobytes() == frames_of(g["frames"]).tobytes()


def test_two_batches_interleaved_on_two_streams():
    """Two independent batches in one process, calls interleaved asynchronously on two caller
    streams: each must produce what it produces alone."""
    import torch
    total = 4 * 1280
    xa = np.stack([synth.make_stream(total, seed=45, channel=c, sigma=1500.0)[0] for c in range(70)], axis=1)
    xb = np.stack([synth.make_stream(total, seed=46, channel=c, sigma=4000.0)[0] for c in range(130)], axis=1)
    oa, ob = Oracle(70), Oracle(130)
    ba, bb = batch(70, max_len=1280), batch(130, max_len=1280)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    da, db = dev(xa), dev(xb)
    torch.cuda.synchronize()
    for i in range(0, total, 1280):
        oa.run(xa[i:i + 1280])
        ob.run(xb[i:i + 1280])
        ba.run(da[i:i + 1280], stream=sa.cuda_stream, sync=False)
        bb.run(db[i:i + 1280], stream=sb.cuda_stream, sync=False)
    assert ba.drain_frames().tobytes() == oa.frames().tobytes()
    assert bb.drain_frames().tobytes() == ob.frames().tobytes()
    for bt, o, n in ((ba, oa, 70), (bb, ob, 130)):
        p = bt.pll_state()
        assert [(int(a), int(b_), int(c_)) for a, b_, c_ in zip(p["pll"], p["prev"], p["lastbit"])] == \
            [o.pll(c) for c in range(n)]


def test_frame_ring_overflow_is_loud_and_recoverable():
    """More CRC-valid frames between two drains than `frame_capacity`: the drain says so
    (GNUAIS_E_OVERFLOW), what it does deliver are real frames, the per-channel counters (kept on the
    device, protodec.c:1103) stay exact, and the next span is whole again."""
    import ctypes as C
    from gnuais_amd import lib
    total = 10 * 1280
    x = np.stack([synth.make_stream(2 * total, seed=47, channel=c, occupancy=1.0)[0] for c in range(40)], axis=1)
    o = Oracle(40)
    b = batch(40, max_len=total, frame_capacity=64)
    o.run(x[:total])
    b.run(dev(x[:total]))
    want = o.frames()
    assert len(want) > 200
    out = np.zeros(4096, dtype=FRAME_DTYPE)
    got = C.c_int()
    rc = b._lib.gnuais_batch_drain_frames(b._h, out.ctypes.data, int(out.size), C.byref(got))
    assert rc == lib.E_OVERFLOW and b"overflow" in b._lib.gnuais_last_error()
    assert 0 < got.value <= 64
    have = {bytes(f.tobytes()) for f in want}
    assert all(bytes(f.tobytes()) in have for f in out[: got.value])
    cnt = b.counters()
    assert np.array_equal(np.stack([cnt["receivedframes"], cnt["lostframes"], cnt["lostframes2"]], axis=1),
                          o.counters())
    # a span that fits is delivered whole and in order
    o.clear_frames()
    o.run(x[total:total + 1280])
    b.run(dev(x[total:total + 1280]))
    assert b.drain_frames().tobytes() == o.frames().tobytes()


def test_argument_errors_are_loud():
    from gnuais_amd import lib
    b = batch(4, max_len=100)
    x = dev(np.zeros((101, 4), dtype=np.int16))
    with pytest.raises(lib.GnuaisError) as e:
        b.run(x)
    assert e.value.code == lib.E_ARG
    with pytest.raises(lib.GnuaisError):
        batch(0)


def test_reset_restores_initial_state():
    g = load("chain_48k")
    x = g["x"]
    b = batch(2, max_len=int(x.shape[0]))
    b.run(dev(x))
    b.drain_frames()
    b.reset()
    b.run(dev(x))
    assert b.drain_frames().tobytes() == frames_of(g["frames"]).tobytes()


@pytest.mark.parametrize("hdlc_variant", [1, 0])
def test_protodec_reset_and_frame_cells_vs_oracle(hdlc_variant):
    """gnuais_batch_protodec_reset() = protodec_reset() (protodec.c:87-100) on every decoder, in the middle of frames:
    state, counters and every later frame as the oracle's; gnuais_batch_frame_bits() = d->buffer[0 .. bufferpos)
    (protodec.h:52, protodec.c:1019) of every channel that is inside a frame, after every ragged call."""
    n_ch, total = 70, 24 * 1280
    x = np.stack([synth.make_stream(total, seed=47, channel=c, occupancy=0.9, sigma=(800.0, 3000.0)[c % 2])[0]
                  for c in range(n_ch)], axis=1)
    b = batch(n_ch, max_len=8192)
    b.set_option("hdlc_variant", hdlc_variant)
    o = Oracle(n_ch)
    chunks = [777, 1020, 333, 2048, 1500, 913] * 4
    chunks.append(total - sum(chunks))
    assert chunks[-1] > 0
    pos = inside = 0
    for i, n in enumerate(chunks):
        b.run(dev(x[pos:pos + n]))
        o.run(x[pos:pos + n])
        pos += n
        assert fsm_rows(b) == oracle_fsm_rows(o, n_ch)
        for c in range(n_ch):
            if o.hdlc(c)["state"] in (4, 5):
                got = b.frame_bits(c)
                assert got is not None and np.array_equal(got, o.frame_cells(c)), (i, c)
                inside += 1
        if i % 5 == 2:
            b.protodec_reset()
            o.protodec_reset()
            assert fsm_rows(b) == oracle_fsm_rows(o, n_ch)
    assert inside > 500
    assert b.drain_frames().tobytes() == o.frames().tobytes()
    assert np.array_equal(np.stack([b.counters()[k] for k in ("receivedframes", "lostframes", "lostframes2")], axis=1),
                          o.counters())
    assert o.counters()[:, 0].sum() > 300


def test_dropin_receiver_run_matches_golden(tmp_path):
    """The C drop-in (init_receiver/receiver_run/free_receiver over the C ABI) driven
    like src/ais.c drives the reference: stereo raw file, 1020-frame chunks."""
    import subprocess
    exe = os.path.join(os.path.dirname(G), "c", "dropin_main.bin")
    assert os.path.exists(exe), "build it with __graft_entry__.build()"
    g = load("chain_48k")
    raw = tmp_path / "stereo.raw"
    g["x"].astype("<i2").tofile(raw)
    out = subprocess.check_output([exe, str(raw)]).decode().splitlines()
    want = frames_of(g["frames"])
    lines = [l for l in out if l.startswith("ch ")]
    got = sorted((l.split()[1], int(l.split()[3]), l.split()[5]) for l in lines)
    exp = sorted(("AB"[int(f["channel"])], int(f["nbits"]),
                  bytes(f["payload"][: f["nbits"] // 8]).hex()) for f in want)
    assert got == exp
    # per-buffer order: receiver A's frames before receiver B's, each in time order
    cnt = g["counters"]
    assert out[-2] == f"A: received {cnt[0][0]} lost {cnt[0][1]} lost2 {cnt[0][2]}"
    assert out[-1] == f"B: received {cnt[1][0]} lost {cnt[1][1]} lost2 {cnt[1][2]}"
