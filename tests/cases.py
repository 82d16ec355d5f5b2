"""Deterministic input builders shared by the golden-vector generator and the
parity tests (CPU oracle and HIP path see exactly the same integers)."""
from __future__ import annotations

import numpy as np

from gnuais_amd import synth


def chain_48k(n_slots: int = 5):
    """2 channels (A: sigma 1000, B: sigma 3000), every slot occupied except one."""
    n = n_slots * 1280

    def pick(skip):
        def f(rng, slot):
            return None if slot == skip else synth.random_position_report(rng)
        return f

    a, _ = synth.make_stream(n, seed=1001, channel=0, sigma=1000.0, payloads=pick(2))
    b, _ = synth.make_stream(n, seed=1001, channel=1, sigma=3000.0, payloads=pick(1))
    return np.stack([a, b], axis=1)


def chain_192k(n_slots: int = 3):
    n = n_slots * 5120
    x, _ = synth.make_stream(n, seed=1002, channel=0, sps=20, sigma=1000.0, occupancy=1.0)
    return x[:, None].copy()


def long_messages():
    """Type-5 style 424-bit payloads (two slots), a 432-bit one that the deframer must
    drop (> 426 bits, protodec.c:1024-1026), a 1-byte and a 0-byte payload."""
    rng0 = np.random.default_rng(77)
    plan = {0: 53, 2: 54, 4: 1, 5: 0, 6: 21, 7: 52}

    def f(rng, slot):
        if slot not in plan:
            return None
        return bytes(rng0.integers(0, 256, plan[slot], dtype=np.uint8))

    x, _ = synth.make_stream(8 * 1280, seed=1003, channel=0, sigma=500.0, payloads=f)
    return x[:, None].copy()


def fir_kats():
    """FIR known-answer inputs (SURVEY.md 'hard parts'): single +-1 / +-full-scale
    samples in silence (only the subnormal and tiny taps contribute around the
    edges of the response), full-scale random noise and an alternating full-scale
    pattern (both sensitive to FMA contraction and to summation order)."""
    out = {}
    for name, v in (("imp_p1", 1), ("imp_m1", -1), ("imp_max", 32767), ("imp_min", -32768)):
        x = np.zeros(120, dtype=np.int16)
        x[40] = v
        out[name] = x
    rng = np.random.default_rng(5)
    out["noise_full"] = rng.integers(-32768, 32768, 4096).astype(np.int16)
    alt = np.empty(512, dtype=np.int16)
    alt[0::2] = 32767
    alt[1::2] = -32768
    out["alternating"] = alt
    out["ramp"] = np.arange(-300, 300, dtype=np.int16)
    return out


def _stuffed(payload: bytes, good_crc=True, stuff=True):
    bits = synth.hdlc_frame_bits(payload, stuff=stuff)
    if not good_crc:
        bits = bits.copy()
        bits[24 + 8 + 3] ^= 1
    return bits


def fsm_bit_cases():
    """Bit sequences for protodec_decode parity (one byte per bit)."""
    rng = np.random.default_rng(11)
    cases = {}
    cases["random_p50"] = rng.integers(0, 2, 30000, dtype=np.uint8)
    cases["random_p70"] = (rng.random(30000) < 0.7).astype(np.uint8)
    cases["random_p30"] = (rng.random(30000) < 0.3).astype(np.uint8)
    cases["alternating"] = (np.arange(5000) & 1).astype(np.uint8)
    frames = []
    for n in (21, 21, 1, 2, 53, 54, 60, 0, 12):
        frames.append(_stuffed(bytes(rng.integers(0, 256, n, dtype=np.uint8))))
        frames.append(rng.integers(0, 2, int(rng.integers(0, 40)), dtype=np.uint8))
    frames.append(_stuffed(bytes(rng.integers(0, 256, 21, dtype=np.uint8)), good_crc=False))
    frames.append(_stuffed(bytes([0xFF] * 21)))                      # heavy stuffing
    frames.append(_stuffed(bytes([0xFF] * 21), stuff=False))         # abort: 7+ ones
    # payload length not a multiple of 8: drop 3 bits before the closing flag
    odd = _stuffed(bytes(rng.integers(0, 256, 21, dtype=np.uint8)))
    frames.append(np.concatenate([odd[:-11], odd[-8:]]))
    # back-to-back frames sharing nothing, then a frame preceded by a short preamble
    frames.append(_stuffed(bytes(rng.integers(0, 256, 21, dtype=np.uint8))))
    frames.append(synth.hdlc_frame_bits(bytes(rng.integers(0, 256, 21, dtype=np.uint8)),
                                        training_bits=10))
    frames.append(synth.hdlc_frame_bits(bytes(rng.integers(0, 256, 21, dtype=np.uint8)),
                                        training_bits=40))
    cases["crafted"] = np.concatenate(frames).astype(np.uint8)
    # frames embedded in biased noise, many of them
    parts = []
    for i in range(60):
        parts.append((rng.random(int(rng.integers(5, 200))) < 0.5).astype(np.uint8))
        parts.append(_stuffed(bytes(rng.integers(0, 256, int(rng.choice([11, 21, 21, 21, 53])),
                                                 dtype=np.uint8)), good_crc=(i % 7 != 3)))
    cases["mixed"] = np.concatenate(parts).astype(np.uint8)
    return cases


def crc_cases():
    rng = np.random.default_rng(3)
    c = [b"123456789", b"", b"\x00", b"\xff\xff", bytes(range(58))]
    c += [bytes(rng.integers(0, 256, int(n), dtype=np.uint8)) for n in rng.integers(1, 58, 20)]
    return c


def nmea_frames(seed=51, n_channels=5, n_random=600):
    """Frame records for the message-layer (row f1) tests: every payload length class around
    the 6-bit padding and the 61-character split, AIS types inside and outside 1..24, several
    channels interleaved so that the rolling sequence digit wraps."""
    import numpy as np
    from oracle_lib import FRAME_DTYPE
    rng = np.random.default_rng(seed)
    lens = [0, 6, 8, 16, 38, 40, 96, 160, 168, 200, 256, 312, 360, 366, 368, 372, 376, 408, 416, 424, 425, 426]
    types = [0, 1, 2, 3, 4, 5, 6, 9, 18, 19, 21, 24, 25, 31, 63]
    rows = []
    for k in range(n_random):
        nbits = lens[k % len(lens)] if k < 3 * len(lens) else int(rng.integers(0, 54)) * 8
        bits = rng.integers(0, 2, 53 * 8).astype(np.uint8)
        t = types[(k * 7) % len(types)] if k % 3 else int(rng.integers(0, 64))
        for i in range(6):
            bits[i] = (t >> (5 - i)) & 1
        bits[nbits - nbits % 8:] = 0               # the record keeps whole bytes only (protodec.c:133)
        f = np.zeros(1, dtype=FRAME_DTYPE)[0]
        f["channel"] = int(rng.integers(0, n_channels))
        f["end_bit"] = k
        f["payload"] = np.packbits(bits)
        f["flags"] = 1
        f["nbits"] = nbits
        rows.append(f)
    # binary messages of DAC 1 (types 6 and 8) with every function identifier the reference names,
    # among them the two it decodes further (11 weather, 40 persons on board)
    for k, fi in enumerate(list(range(0, 6)) + [11, 16, 17, 21, 24, 29, 30, 40, 41, 63] * 2):
        t = 6 if k % 2 else 8
        nbits = (360, 200, 168, 424)[k % 4]
        bits = rng.integers(0, 2, 53 * 8).astype(np.uint8)
        for i in range(6):
            bits[i] = (t >> (5 - i)) & 1
        at = 72 if t == 6 else 40
        for i in range(10):
            bits[at + i] = (1 >> (9 - i)) & 1          # DAC = 1
        for i in range(6):
            bits[at + 10 + i] = (fi >> (5 - i)) & 1
        bits[nbits:] = 0
        f = np.zeros(1, dtype=FRAME_DTYPE)[0]
        f["channel"] = k % n_channels
        f["end_bit"] = n_random + k
        f["payload"] = np.packbits(bits)
        f["flags"] = 1
        f["nbits"] = nbits
        rows.append(f)
    for t in range(1, 25):                                 # and every accepted type at least once
        nbits = (168, 424, 256)[t % 3]
        bits = rng.integers(0, 2, 53 * 8).astype(np.uint8)
        for i in range(6):
            bits[i] = (t >> (5 - i)) & 1
        bits[nbits:] = 0
        f = np.zeros(1, dtype=FRAME_DTYPE)[0]
        f["channel"] = t % n_channels
        f["end_bit"] = n_random + 100 + t
        f["payload"] = np.packbits(bits)
        f["flags"] = 1
        f["nbits"] = nbits
        rows.append(f)
    return np.array(rows, dtype=FRAME_DTYPE), n_channels


def range_frames(seed=61, n_channels=6, n_random=1500, own_channel=False):
    """Position reports (types 1-3, 4, 18) for the range statistics (row f4): crafted fixes at the
    plausibility limits of update_range() (range.c:34-39), the 0/0 "no position", antipodes, and
    random fields (about a third of which are implausible); other types in between.  With
    own_channel every frame sits on a channel of its own, so best_range[] is the distance of
    each single fix rather than a maximum."""
    import numpy as np
    from oracle_lib import FRAME_DTYPE
    rng = np.random.default_rng(seed)
    where = {1: (89, 61), 2: (89, 61), 3: (89, 61), 4: (107, 79), 18: (85, 57)}   # (lat, lon) bit offsets
    crafted = [(0, 0), (599, 599), (600, 0), (-599, -600), (53400000, 0), (53400001, 0), (-53400000, 1),
               (53399999, 108006000), (0, 108006001), (0, -108006001), (12345678, 108005999),
               (35982000, 6186000), (-35982000, -101814000), (36000000, 6000000), (36000060, 6000060),
               (67108863, 134217727), (-67108864, -134217728), (54600000, 108000000)]
    rows = []
    for k in range(n_random + 5 * len(crafted)):
        if k < 5 * len(crafted):
            t = (1, 2, 3, 4, 18)[k % 5]
            lat, lon = crafted[k // 5]
        else:
            t = (1, 2, 3, 4, 18, 5, 8, 19)[int(rng.integers(0, 8))]
            lat, lon = int(rng.integers(-2 ** 26, 2 ** 26)), int(rng.integers(-2 ** 27, 2 ** 27))
            if k % 3 == 0:                                  # near the station: small distances
                lat, lon = 36000000 + int(rng.integers(-3000, 3000)), 6000000 + int(rng.integers(-3000, 3000))
        bits = rng.integers(0, 2, 53 * 8).astype(np.uint8)
        for i in range(6):
            bits[i] = (t >> (5 - i)) & 1
        if t in where:
            la, lo = where[t]
            for i in range(27):
                bits[la + i] = ((lat & (2 ** 27 - 1)) >> (26 - i)) & 1
            for i in range(28):
                bits[lo + i] = ((lon & (2 ** 28 - 1)) >> (27 - i)) & 1
        nbits = 168 if k % 11 else 96                       # a short record: fields read past its end
        bits[nbits:] = 0
        f = np.zeros(1, dtype=FRAME_DTYPE)[0]
        f["channel"] = k if own_channel else int(rng.integers(0, n_channels))
        f["end_bit"] = k
        f["payload"] = np.packbits(bits)
        f["flags"] = 1
        f["nbits"] = nbits
        rows.append(f)
    return np.array(rows, dtype=FRAME_DTYPE), (len(rows) if own_channel else n_channels)


RANGE_STATIONS = [(60.0, 10.0), (-33.9, 151.2), (0.0, 0.0), (89.5, -179.5), (95.0, 10.0), (-200.0, -200.0)]


def vessel_frames(seed=71, n_channels=4, n=1500, n_mmsi=500):
    """Frames of every type that reaches the position cache (1-5, 18, 19, 24 parts A/B and the
    unused part numbers, binary messages with and without DAC 1 / FI 40), drawn over a small
    pool of MMSIs so that entries are overwritten field group by field group in every order;
    names with trailing blanks, '@' and short records among them."""
    import numpy as np
    from oracle_lib import FRAME_DTYPE
    rng = np.random.default_rng(seed)
    pool = [int(v) for v in rng.integers(1, 2 ** 30, n_mmsi)] + [0, 2 ** 30 - 1, 1]
    types = [1, 2, 3, 4, 5, 18, 19, 24, 24, 6, 8, 9, 21]
    rows = []
    for k in range(n):
        t = types[int(rng.integers(0, len(types)))]
        bits = rng.integers(0, 2, 53 * 8).astype(np.uint8)

        def put(at, width, value):
            for i in range(width):
                bits[at + i] = (value >> (width - 1 - i)) & 1
        put(0, 6, t)
        put(8, 30, pool[int(rng.integers(0, len(pool)))])
        if t in (6, 8):
            at = 72 if t == 6 else 40
            put(at, 10, 1 if k % 4 else int(rng.integers(0, 1024)))
            put(at + 10, 6, 40 if k % 3 else int(rng.integers(0, 64)))
        if t in (5, 19, 24) and k % 5 == 0:            # text that ends in blanks / '@' (both decode to ' ')
            at = {5: 112, 19: 143, 24: 40}[t]
            for c in range(int(rng.integers(0, 21)), 20):
                put(at + 6 * c, 6, 32 if c % 2 else 0)
        nbits = (424 if t == 5 else 312 if t == 19 else 168) if k % 9 else int(rng.integers(5, 53)) * 8
        bits[nbits:] = 0
        f = np.zeros(1, dtype=FRAME_DTYPE)[0]
        f["channel"] = int(rng.integers(0, n_channels))
        f["end_bit"] = k
        f["payload"] = np.packbits(bits)
        f["flags"] = 1
        f["nbits"] = nbits
        rows.append(f)
    return np.array(rows, dtype=FRAME_DTYPE), n_channels
