"""Randomised parity soak: the HIP chain against the C oracle on random configurations for a given
number of seconds -- channel counts, call lengths (down to 1 sample), tables (the reference's, the
192 kHz one, random symmetric and asymmetric ones), pllinc, signal levels from near-silence to
clipping, and inputs built to sit on the slicer's decision threshold (tiny amplitudes, long runs of
0 / +-1, sparse impulses), where the sign-exact slicer has to fall back to the exact sum.
    python scripts/fuzz_parity.py [seconds] [first_seed]        (PIPE=1: pipelined mode, DEFRAMER=1:
    deframer only, TABLE=192k: the 144-tap parameter set in every case)
Prints one line per case; exits non-zero at the first mismatch, naming the seed."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch                                                        # noqa: E402
from gnuais_amd import ReceiverBatch, nmea_from_frames, params, synth   # noqa: E402
from oracle_lib import Oracle                                       # noqa: E402

FSM_KEYS = ("state", "nstartsign", "antallpreamble", "antallenner", "bitstuff", "last", "bufferpos")


def table(rng):
    kind = rng.integers(0, 10)
    if os.environ.get("TABLE") == "192k":
        kind = 5
    if kind < 5:
        return None, 0, "ref"
    if kind == 5:
        return params.taps_192k(), params.PLLINC_192K, "192k"
    n = int(rng.integers(8, 140))
    k = np.arange(n, dtype=np.float64)
    mid = (n - 1) / 2.0
    t = rng.uniform(0.05, 0.9) * np.exp(-((k - mid) ** 2) / (2 * rng.uniform(0.8, n / 6.0 + 1) ** 2))
    if kind == 6:
        t *= rng.choice([-1.0, 1.0], n)                          # sign changes, still symmetric or not
        t = (t + t[::-1]) / 2 if rng.integers(0, 2) else t
    if kind == 7:
        t = t + rng.normal(0, 0.01, n)                           # asymmetric: exact kernels only
    if kind == 8:
        t[: int(rng.integers(0, 4))] = 0.0
        t[n - int(rng.integers(0, 4)):] = 0.0
    pllinc = int(rng.choice([0, 0x10000 // 5, 13500, 0x10000 // 7, 0x10000 // 20, 9000]))
    return t.astype(np.float32), pllinc, f"rand{kind}/{n}"


def column(rng, total, sps):
    kind = rng.integers(0, 9)
    if kind <= 3:
        sigma = float(rng.choice([0.0, 1.0, 30.0, 500.0, 1000.0, 3000.0, 9000.0, 25000.0]))
        return synth.make_stream(total, seed=int(rng.integers(1, 1 << 30)), channel=int(rng.integers(0, 999)),
                                 sps=sps, sigma=sigma, occupancy=float(rng.uniform(0.1, 1.0)),
                                 amplitude=float(rng.choice([3.0, 40.0, 1200.0, 12000.0, 30000.0])))[0]
    if kind == 4:
        return np.zeros(total, dtype=np.int16)
    if kind == 5:                                                 # tiny values: y hovers around 0
        return rng.integers(-2, 3, total).astype(np.int16)
    if kind == 6:                                                 # sparse impulses in silence
        x = np.zeros(total, dtype=np.int16)
        at = rng.integers(0, total, max(1, total // 97))
        x[at] = rng.integers(-32768, 32768, len(at))
        return x
    if kind == 7:                                                 # silence / signal / silence
        x = rng.normal(0, 800, total)
        a, b = sorted(rng.integers(0, total, 2))
        x[a:b] = 0
        return np.clip(np.rint(x), -32768, 32767).astype(np.int16)
    return rng.integers(-32768, 32768, total).astype(np.int16)   # full-scale noise


def one_case(seed):
    rng = np.random.default_rng(seed)
    taps, pllinc, tname = table(rng)
    sps = 20 if tname == "192k" else 5
    n_ch = int(rng.choice([1, 2, 3, 63, 64, 65, 128, 130, 192, 257]))     # whole groups of 64: long tables on the matrix pipe
    total = int(rng.integers(1, 30000))
    x = np.stack([column(rng, total, sps) for _ in range(n_ch)], axis=1)
    chunks = []
    left = total
    while left:
        n = int(min(left, rng.choice([1, 2, 35, 36, 37, 95, 96, 97, 1020, 4096, 9999, 30000])))
        chunks.append(n)
        left -= n
    o = Oracle(n_ch, taps=taps, pllinc=pllinc)
    b = ReceiverBatch(n_ch, taps=taps, pllinc=pllinc, max_len=max(chunks))
    if os.environ.get("FIR_VARIANT"):
        b.set_option("fir_variant", int(os.environ["FIR_VARIANT"]))
    if rng.integers(0, 3) == 0:
        b.set_option("fir_T", int(rng.choice([96, 128, 256, 512, 2048])))
    if rng.integers(0, 3) == 0:
        b.set_option("fir_pk_taps", 48)                          # long tables: 48 central taps instead of 40
    if rng.integers(0, 4) == 0:
        b.set_option("fir_mfma", 0)                              # ... and the packed kernel for every segment
    b.set_option("pll_variant", int((int(os.environ["PLL_VARIANT"]) if os.environ.get("PLL_VARIANT") else rng.choice([0, 7, 8]))))
    host_input = rng.integers(0, 4) == 0          # gnuais_batch_run_host: the drop-in's entry point
    reset_at = int(rng.integers(0, len(chunks))) if rng.integers(0, 6) == 0 else -1
    pos = 0
    for i, n in enumerate(chunks):
        if i == reset_at:                         # init_receiver() again: everything back to zero
            b.drain_frames()
            b.reset()
            o.reset()
            o.clear_frames()
        seg = np.ascontiguousarray(x[pos:pos + n])
        pos += n
        r = o.run(seg, want_bits=True)
        b.run(seg if host_input else torch.from_numpy(seg).cuda())
        lb = b.last_bits()
        if not np.array_equal(b.maxval(), r["maxval"]):
            return f"maxval differs, call of {n} at {pos - n}"
        for c in range(n_ch):
            if not np.array_equal(lb[c], r["bits"][c]):
                return f"bits differ, channel {c}, call of {n} at {pos - n}"
    if b.drain_frames().tobytes() != o.frames().tobytes():
        return "frames differ"
    cnt = b.counters()
    if not np.array_equal(np.stack([cnt["receivedframes"], cnt["lostframes"], cnt["lostframes2"]], axis=1),
                          o.counters()):
        return "counters differ"
    p = b.pll_state()
    if [(int(a), int(bb), int(cc)) for a, bb, cc in zip(p["pll"], p["prev"], p["lastbit"])] != \
            [o.pll(c) for c in range(n_ch)]:
        return "pll state differs"
    f = b.fsm_state()
    for c in range(n_ch):
        h = o.hdlc(c)
        want = [h[k] for k in FSM_KEYS]
        want[2] = min(want[2], 15)
        if [int(f[c][k]) for k in FSM_KEYS] != want:
            return f"fsm state differs, channel {c}"
    return f"ok   {tname:10s} n_ch {n_ch:3d} total {total:5d} calls {len(chunks):4d} frames {int(o.counters()[:, 0].sum())}"


def pipelined_case(seed):
    """Many channels, calls queued asynchronously (several in flight on the stage streams, hand-off
    buffers reused), optional stream autotune / deframer width / timing events; compared at the end."""
    rng = np.random.default_rng(seed)
    taps, pllinc, tname = (None, 0, "ref") if rng.integers(0, 4) else (params.taps_192k(), params.PLLINC_192K, "192k")
    sps = 20 if tname == "192k" else 5
    n_ch = int(rng.choice([64, 300, 1024, 2500, 4096]))
    n_calls = int(rng.integers(2, 14))
    lens = [int(rng.choice([1020, 4096, 20000, 7777, 333, 48000 // 4])) for _ in range(n_calls)]
    total = sum(lens)
    k = 48
    base = np.stack([column(rng, total, sps) for _ in range(k)], axis=0)
    x = synth.tile_channels(base, n_ch)
    xd = torch.from_numpy(x).cuda()
    o = Oracle(n_ch, taps=taps, pllinc=pllinc)
    b = ReceiverBatch(n_ch, taps=taps, pllinc=pllinc, max_len=max(lens))
    opts = []
    if rng.integers(0, 3) == 0:
        b.autotune(xd[: max(lens)].contiguous(), torch.cuda.current_stream().cuda_stream)
        opts.append("autotune")
    pv = int((int(os.environ["PLL_VARIANT"]) if os.environ.get("PLL_VARIANT") else rng.choice([0, 7, 8])))
    b.set_option("pll_variant", pv)
    opts.append(f"pll{pv}")
    if rng.integers(0, 3) == 0:
        lpw = int(rng.choice([1, 4, 16, 64]))
        b.set_option("hdlc_lpw", lpw)
        opts.append(f"lpw{lpw}")
    if rng.integers(0, 3) == 0:
        b.set_timing(True)
        b.set_option("timing_stride", int(rng.choice([1, 3])))
        opts.append("timing")
    stream = torch.cuda.current_stream().cuda_stream
    pos = 0
    for n in lens:
        b.run(xd[pos:pos + n], stream=stream, sync=False)
        pos += n
    o.run(x, threads=32)
    got = b.drain_frames()
    if got.tobytes() != o.frames().tobytes():
        return f"frames differ ({len(got)} vs {len(o.frames())}) {opts}"
    cnt = b.counters()
    if not np.array_equal(np.stack([cnt["receivedframes"], cnt["lostframes"], cnt["lostframes2"]], axis=1),
                          o.counters()):
        return f"counters differ {opts}"
    p = b.pll_state()
    if [(int(a), int(bb), int(cc)) for a, bb, cc in zip(p["pll"], p["prev"], p["lastbit"])] != \
            [o.pll(c) for c in range(n_ch)]:
        return f"pll state differs {opts}"
    return f"ok   {tname:5s} n_ch {n_ch:4d} calls {lens} frames {len(got)} {' '.join(opts)}"


def deframer_case(seed):
    """protodec_decode() alone: adversarial bit streams (flags back to back, endless training, frames
    of every length up to and beyond the 449-bit buffer, bit errors, stuffing at the edges, runs of
    ones), fed in several calls of random size so that every state crosses a call boundary."""
    rng = np.random.default_rng(seed)
    n_ch = int(rng.choice([1, 5, 64, 96, 200]))
    flag = np.array([0, 1, 1, 1, 1, 1, 1, 0], dtype=np.uint8)
    streams = []
    for c in range(n_ch):
        parts = []
        for _ in range(int(rng.integers(1, 40))):
            kind = rng.integers(0, 9)
            if kind == 0:
                parts.append((rng.random(int(rng.integers(0, 300))) < rng.random()).astype(np.uint8))
            elif kind == 1:
                parts.append(np.tile(flag, int(rng.integers(1, 6))))
            elif kind == 2:
                parts.append((np.arange(int(rng.integers(0, 80))) & 1).astype(np.uint8))
            elif kind == 3:
                parts.append(np.ones(int(rng.integers(0, 40)), dtype=np.uint8))
            elif kind == 4:
                parts.append(np.zeros(int(rng.integers(0, 40)), dtype=np.uint8))
            elif kind == 5 and rng.integers(0, 3) == 0:                   # frames as dense as they get
                data = (rng.random(int(rng.integers(0, 60))) < 0.4).astype(np.uint8)
                cyc = np.concatenate([(np.arange(16) & 1).astype(np.uint8), flag, data, flag])
                parts.append(np.tile(cyc, int(rng.integers(1, 120))))
            else:
                n = int(rng.choice([0, 1, 2, 3, 11, 20, 21, 22, 40, 52, 53, 54, 55, 60, 80]))
                body = bytes(rng.integers(0, 256, n, dtype=np.uint8)) if rng.integers(0, 4) else bytes([0xff] * n)
                fb = synth.hdlc_frame_bits(body, training_bits=int(rng.integers(0, 40)),
                                           stuff=bool(rng.integers(0, 8)))
                if rng.integers(0, 5) == 0:
                    fb[int(rng.integers(0, fb.size))] ^= 1
                if rng.integers(0, 6) == 0:
                    fb = fb[: int(rng.integers(0, fb.size))]               # cut off
                parts.append(fb)
        streams.append(np.concatenate(parts).astype(np.uint8) if parts else np.zeros(0, dtype=np.uint8))
    o = Oracle(n_ch)
    b = ReceiverBatch(n_ch, max_len=48000)
    b2 = ReceiverBatch(n_ch, max_len=48000)                    # the same frames, formatted on the device
    if rng.integers(0, 2):
        b.set_option("hdlc_lpw", int(rng.choice([1, 2, 8, 32, 64])))
    pos = [0] * n_ch
    while any(pos[c] < len(streams[c]) for c in range(n_ch)):
        piece = []
        for c in range(n_ch):
            n = int(rng.choice([0, 1, 7, 8, 9, 31, 32, 33, 100, 449, 1000, 5000]))
            piece.append(streams[c][pos[c]:pos[c] + n])
            pos[c] += n
        for c in range(n_ch):
            o.decode_bits(c, piece[c])
        b.decode_bits(piece)
        b2.decode_bits(piece)
    got_frames = b.drain_frames()
    if got_frames.tobytes() != o.frames().tobytes():
        return "frames differ"
    seq_h = (np.arange(n_ch) % 10).astype(np.uint8)
    seq_d = seq_h.copy()
    if rng.integers(0, 2):
        text_h = nmea_from_frames(got_frames, seq_h)
        text_d, _, nf = b2.drain_nmea(seq_d)
        if text_h != text_d or nf != len(got_frames) or not np.array_equal(seq_h, seq_d):
            return f"device NMEA differs from the host formatter ({len(text_d)} vs {len(text_h)} bytes)"
    else:                                                      # sentences + stdout lines + vessel table
        from gnuais_amd import messages_from_frames, vessels_from_frames
        chanid = bytes(65 + int(v) for v in rng.integers(0, 26, n_ch)) if rng.integers(0, 2) else None
        tab_d = b2.fold_vessels()
        tab_h = vessels_from_frames(got_frames)
        if tab_d.tobytes() != tab_h.tobytes():
            return f"device vessel table differs from the host fold ({len(tab_d)} vs {len(tab_h)} entries)"
        nm_h, tx_h = messages_from_frames(got_frames, seq_h, chanid)
        nm_d, tx_d, _, _, nf = b2.drain_messages(seq_d, chanid)
        if nm_h != nm_d or tx_h != tx_d or nf != len(got_frames) or not np.array_equal(seq_h, seq_d):
            return f"device messages differ from the host formatter ({len(tx_d)} vs {len(tx_h)} bytes of lines)"
    cnt = b.counters()
    if not np.array_equal(np.stack([cnt["receivedframes"], cnt["lostframes"], cnt["lostframes2"]], axis=1),
                          o.counters()):
        return "counters differ"
    f = b.fsm_state()
    for c in range(n_ch):
        h = o.hdlc(c)
        want = [h[k] for k in FSM_KEYS]
        want[2] = min(want[2], 15)
        if [int(f[c][k]) for k in FSM_KEYS] != want:
            return f"fsm state differs, channel {c}: {[int(f[c][k]) for k in FSM_KEYS]} vs {want}"
    return f"ok   deframer n_ch {n_ch:3d} bits {sum(len(s_) for s_ in streams)} frames {int(o.counters()[:, 0].sum())}"


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds:
        res = (pipelined_case(seed) if os.environ.get("PIPE") else deframer_case(seed) if os.environ.get("DEFRAMER")
               else one_case(seed))
        print(f"seed {seed}: {res}", flush=True)
        if not res.startswith("ok"):
            sys.exit(1)
        seed += 1
        n += 1
    print(f"{n} cases, all bit-exact, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
