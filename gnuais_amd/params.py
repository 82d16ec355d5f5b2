"""Receive-chain parameters (reference src/receiver.c:39-49,69,84).

The 48 kHz values are the reference's hard-coded ones.  The 192 kHz set is
builder-defined (the reference supplies none): the same
Gaussian low-pass stretched 4x in time (144 taps) and a PLL increment for 20
samples per bit; the unmodified reference functions are generic in both.
"""
from __future__ import annotations

import numpy as np

PLL_INC_DIV = 16                  # receiver.c:84 INC
PLLINC_48K = 0x10000 // 5         # receiver.c:69
PLLINC_192K = 0x10000 // 20
N_TAPS_48K = 36                   # receiver.c:50 COEFFS_L
N_TAPS_192K = 144

# receiver.c:39-49: symmetric table, first half (the double literals round to
# fp32: k = 0,1 -> 0.0f, k = 2 -> subnormal 0x00000069)
_TAP_HALF_48K = (
    2.5959e-55, 2.9479e-49, 1.4741e-43, 3.2462e-38, 3.1480e-33, 1.3443e-28,
    2.5280e-24, 2.0934e-20, 7.6339e-17, 1.2259e-13, 8.6690e-11, 2.6996e-08,
    3.7020e-06, 2.2355e-04, 5.9448e-03, 6.9616e-02, 3.5899e-01, 8.1522e-01,
)


def taps_48k() -> np.ndarray:
    half = np.asarray(_TAP_HALF_48K, dtype=np.float64).astype(np.float32)
    return np.concatenate([half, half[::-1]])


def taps_192k() -> np.ndarray:
    """144-tap Gaussian: 0.25 * 0.9032 * exp(-(k - 71.5)^2 / (2 (4 * 1.1043)^2))."""
    k = np.arange(N_TAPS_192K, dtype=np.float64)
    t = 0.25 * 0.9032 * np.exp(-((k - 71.5) ** 2) / (2.0 * (4.0 * 1.1043) ** 2))
    return t.astype(np.float32)
