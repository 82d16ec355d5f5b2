"""gnuais_amd -- MI355X-native batch AIS receive chain (gnuais's per-sample hot
path: FIR -> slicer/PLL/NRZI -> HDLC deframe + CRC-16) behind a C ABI.

  include/gnuais_hip.h       the C ABI (drop-in boundary)
  gnuais_amd/csrc/           hand-written gfx950 HIP kernels + C-ABI host code
  gnuais_amd/receiver.py     host-side mirror of the reference receiver interface
  gnuais_amd/synth.py        synthetic AIS baseband generator (bench/test input)
"""
from . import params, synth  # noqa: F401

__all__ = ["params", "synth", "ReceiverBatch"]


def __getattr__(name):
    if name in ("ReceiverBatch", "crc16_batch", "tile_channels", "nmea_from_frames", "messages_from_frames", "range_from_frames", "vessels_from_frames", "VESSEL_DTYPE"):
        from . import receiver
        return getattr(receiver, name)
    if name == "ReceiverNode":
        from .shard import ReceiverNode
        return ReceiverNode
    raise AttributeError(name)
