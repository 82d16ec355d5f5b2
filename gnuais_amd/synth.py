"""Synthetic AIS baseband generator (SURVEY.md section 8d recipe).

What gnuais consumes is the FM-discriminator output of a marine VHF receiver
(reference src/receiver.c:109-111 slices its sign): NRZI levels shaped by the
GMSK Gaussian filter (BT = 0.4), 9600 baud, 5 samples per bit at 48 kHz
(20 at 192 kHz), int16.  This module builds such streams deterministically:

  * time is cut into AIS slots of 256 bit times (1280 samples at 48 kHz);
  * a slot is occupied with probability `occupancy`; an occupied slot carries
    one HDLC frame: 24-bit 0101... training, flag 0x7E, payload + CRC-16/X-25
    (FCS low byte first, every byte LSB first), bit-stuffed, flag 0x7E;
  * NRZI (0 toggles the level, 1 holds it), Gaussian pulse shaping, amplitude
    12000, additive white Gaussian noise, round to nearest, clamp to int16.

The generator is host-side benchmark/test plumbing; it is not part of the
receive chain and never runs inside a timed region.
"""
from __future__ import annotations

import numpy as np

SEED = 0x41495321          # "AIS!"
SLOT_BITS = 256            # 26.67 ms at 9600 baud
START_OFFSET_BITS = 8      # frame starts 8 bit times (40 samples @48k) into the slot


def crc16_x25(data: bytes) -> int:
    """CRC-16/X-25 as HDLC uses it (reflected 0x8408, init 0xffff, xorout 0xffff)."""
    crc = 0xFFFF
    for byte in data:
        crc ^= byte
        for _ in range(8):
            crc = (crc >> 1) ^ 0x8408 if crc & 1 else crc >> 1
    return crc ^ 0xFFFF


def payload_from_bits(bits) -> bytes:
    """AIS payload bits (MSB first, the order ITU-R M.1371 numbers them) -> bytes."""
    bits = np.asarray(bits, dtype=np.uint8)
    assert bits.size % 8 == 0
    return np.packbits(bits).tobytes()


def random_position_report(rng: np.random.Generator) -> bytes:
    """168-bit payload whose first 6 bits are a message type in {1,2,3} and whose
    bits 8..37 are a random 30-bit MMSI; everything else random."""
    bits = rng.integers(0, 2, 168, dtype=np.uint8)
    mtype = int(rng.integers(1, 4))
    bits[0:6] = [(mtype >> (5 - i)) & 1 for i in range(6)]
    return payload_from_bits(bits)


def hdlc_frame_bits(payload: bytes, training_bits: int = 24, stuff: bool = True) -> np.ndarray:
    """On-air bit sequence (before NRZI) of one AIS burst."""
    fcs = crc16_x25(payload)
    body = payload + bytes([fcs & 0xFF, fcs >> 8])
    raw = np.unpackbits(np.frombuffer(body, dtype=np.uint8), bitorder="little")
    out = [i & 1 for i in range(training_bits)]          # 0101...
    flag = [0, 1, 1, 1, 1, 1, 1, 0]
    out += flag
    ones = 0
    for b in raw:
        out.append(int(b))
        if b:
            ones += 1
            if ones == 5 and stuff:
                out.append(0)
                ones = 0
        else:
            ones = 0
    out += flag
    return np.asarray(out, dtype=np.uint8)


def gaussian_kernel(sps: int, bt: float = 0.4) -> np.ndarray:
    sigma = np.sqrt(np.log(2.0)) / (2.0 * np.pi * bt) * sps
    half = int(np.ceil(4 * sigma))
    t = np.arange(-half, half + 1, dtype=np.float64)
    k = np.exp(-0.5 * (t / sigma) ** 2)
    return k / k.sum()


def nrzi_levels(bits: np.ndarray, start_level: int = 1) -> np.ndarray:
    """0 -> toggle, 1 -> hold.  Returns +-1 per bit."""
    toggles = (np.asarray(bits) == 0).astype(np.int64)
    lev = (start_level + np.cumsum(toggles)) & 1
    return lev.astype(np.float64) * 2.0 - 1.0


def make_stream(n_samples: int, seed: int = SEED, channel: int = 0, sps: int = 5,
                amplitude: float = 12000.0, sigma: float = 1000.0, occupancy: float = 0.5,
                payloads=None, bt: float = 0.4):
    """One channel of synthetic baseband.

    Returns (int16[n_samples], list of (slot, payload bytes) actually placed).
    `payloads`: optional callable(rng, slot) -> bytes | None overriding the
    default random type-1/2/3 reports.
    """
    rng = np.random.default_rng([seed, channel])
    slot_len = SLOT_BITS * sps
    n_slots = (n_samples + slot_len - 1) // slot_len
    level = np.zeros(n_slots * slot_len + 64 * sps, dtype=np.float64)
    placed = []
    for slot in range(n_slots):
        if payloads is not None:
            payload = payloads(rng, slot)
        else:
            payload = random_position_report(rng) if rng.random() < occupancy else None
        if payload is None:
            continue
        bits = hdlc_frame_bits(payload)
        lev = np.repeat(nrzi_levels(bits, start_level=int(rng.integers(0, 2))), sps)
        start = slot * slot_len + START_OFFSET_BITS * sps
        end = min(start + lev.size, level.size)
        level[start:end] = lev[: end - start]
        placed.append((slot, payload))
    shaped = np.convolve(level, gaussian_kernel(sps, bt), mode="same")[:n_samples]
    noise = rng.normal(0.0, sigma, n_samples) if sigma > 0 else 0.0
    x = np.rint(amplitude * shaped + noise)
    return np.clip(x, -32768, 32767).astype(np.int16), placed


def make_base_streams(n_base: int, n_samples: int, **kw):
    """[n_base][n_samples] int16 + per-stream placed payload lists."""
    streams = np.empty((n_base, n_samples), dtype=np.int16)
    placed = []
    for c in range(n_base):
        streams[c], p = make_stream(n_samples, channel=c, **kw)
        placed.append(p)
    return streams, placed


def rotation_of(channel: int, n_samples: int) -> int:
    """Channel c replays base stream c mod K circularly rotated by this much."""
    return (channel * 7919) % n_samples


def tile_channels(base: np.ndarray, n_channels: int) -> np.ndarray:
    """Host-side tiling -> interleaved [n_samples][n_channels] int16 (the layout
    the reference consumes, receiver.c:102,107).  The HIP path has a device-side
    equivalent (gnuais_tile_channels) for benchmark-sized inputs."""
    k, n = base.shape
    out = np.empty((n, n_channels), dtype=np.int16)
    for c in range(n_channels):
        out[:, c] = np.roll(base[c % k], -rotation_of(c, n))
    return out
