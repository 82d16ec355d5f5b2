"""File -> device chain -> what gnuais would print / send, end to end.

  python scripts/decode_file.py capture.wav            # RIFF/WAVE, any channel count
  python scripts/decode_file.py capture.raw --raw 2    # bare int16 frames, as the reference reads them
  ... --text   prints the reference's stdout lines instead of the bare NMEA sentences
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--raw", type=int, default=0, metavar="CHANNELS")
    ap.add_argument("--text", action="store_true")
    ap.add_argument("--call", type=int, default=48000, help="frames per device call")
    a = ap.parse_args()
    import torch
    from gnuais_amd import ReceiverBatch, io, messages_from_frames
    x = io.read_raw(a.path, a.raw) if a.raw else io.read_wav(a.path)[1]
    n_ch = x.shape[1]
    b = ReceiverBatch(n_ch, max_len=a.call)
    seq = np.zeros(n_ch, dtype=np.uint8)
    for part in io.chunks(x, a.call):
        b.run(torch.from_numpy(np.ascontiguousarray(part)).cuda())
        nmea, text = messages_from_frames(b.drain_frames(), seq)
        sys.stdout.write((text if a.text else nmea).decode("ascii", "replace"))
    c = b.counters()
    sys.stderr.write(f"{int(c['receivedframes'].sum())} frames, {int(c['lostframes'].sum())} CRC errors, "
                     f"{n_ch} channels, {x.shape[0]} samples per channel\n")


if __name__ == "__main__":
    main()
