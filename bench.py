#!/usr/bin/env python3
"""bench.py -- throughput of the batch AIS receive chain on MI355X.

Metric (BASELINE.json): Msamples/s demodulated (+ valid-CRC AIS msgs/s) over an
N-channel 48 kHz batch.  Workload at every GPU count: BASELINE config C3 per GPU
-- 16384 synthetic GMSK channels x 48000 samples (1 s at 48 kHz), full chain
FIR -> slicer/PLL/NRZI -> HDLC deframe + CRC-16, int16 input resident in HBM.
(C3 rather than configs[1]: the metric counts valid-CRC messages, which only the
full chain produces; configs[1] stops before the deframer.)  One step = one pass
of the chain over the batch.  Channels shard embarrassingly over GPUs (weak
scaling, no data-path collective); torch.distributed is used only for the
barrier and the max-over-ranks time.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s measured)
N_SIMD = 1024                  # 256 CUs x 4
# SIMD time one 64-channel wave of the sign-exact slicer needs per sample, from the measured
# per-instruction issue costs (scripts/ubench/valu_rate): 19 two-operand ops (6 mul, 12 add, the
# peak's max) x 1.04 ns + 4 three-operand-class ops (|y| - eps, two alignbits, spill / misc) x 1.9 ns;
# 22.6 VALU instructions per wave-sample measured (profiles/r01_pmc_sq_counters.json)
K1S_NS_PER_WAVE_SAMPLE = 27.4
TIMING_STRIDE = 4              # per-kernel events on every 4th call of the timed region
                               # per SIMD x 1024 SIMDs, unfused v_mul_f32/v_add_f32


def effective_cpus():
    """Host CPUs this process may actually use: affinity mask, capped by the cgroup CPU quota
    (the GPU box's container shows 256 logical CPUs and a quota of 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def cpu_baseline(x_host, n_sample_ch, total, x_wide=None):
    """The reference's own code (oracle/_ref, kind 'reference') -- or the C
    restatement (kind 'port') when the prebuilt reference is absent -- timed on a
    bounded sample of the same workload on this host, single thread."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    xs = np.ascontiguousarray(x_host[:, :n_sample_ch])
    sample = f"{n_sample_ch} of the bench's channels x {total} samples, 1020-sample chunks"
    if oracle_lib.have_reference():
        ref = oracle_lib.reference()
        ref.add_receivers(n_sample_ch)
        t = time.perf_counter()
        got = ref.lib.ref_bench_run(xs.ctypes.data, total, 1020)
        dt = time.perf_counter() - t
        kind = "reference"
    else:
        o = oracle_lib.Oracle(n_sample_ch)
        t = time.perf_counter()
        o.run(xs)
        dt = time.perf_counter() - t
        got = int(o.counters()[:, 0].sum())
        kind = "port"
    res = {"value": n_sample_ch * total / dt / 1e6, "unit": "Msamples/s", "cores": 1,
           "kind": kind, "sample": sample, "seconds": round(dt, 2), "msgs": int(got)}
    # context: the same work on every host core (the reference itself is single-threaded by
    # design; this is the C restatement with channels partitioned over pthreads)
    if x_wide is not None:
        cores = effective_cpus()
        o = oracle_lib.Oracle(x_wide.shape[1])
        o.run(x_wide[:1020], threads=cores)                       # thread start-up, page faults
        o = oracle_lib.Oracle(x_wide.shape[1])
        t = time.perf_counter()
        o.run(x_wide, threads=cores)
        dt = time.perf_counter() - t
        res["all_cores"] = {"value": x_wide.shape[1] * total / dt / 1e6, "unit": "Msamples/s", "cores": cores,
                            "kind": "port", "sample": f"{x_wide.shape[1]} of the bench's channels x {total} samples, "
                            "interleaved input as the reference reads it, channels partitioned over threads",
                            "seconds": round(dt, 2)}
        # and the CPU's best case: the same channels de-interleaved first (planar, unit stride) --
        # not how the reference reads its input
        xp = np.ascontiguousarray(x_wide.T)
        o = oracle_lib.Oracle(x_wide.shape[1])
        o.run_planar(xp[:, :1020], cores)
        o = oracle_lib.Oracle(x_wide.shape[1])
        t = time.perf_counter()
        o.run_planar(xp, cores)
        dt = time.perf_counter() - t
        res["all_cores_planar"] = {"value": x_wide.shape[1] * total / dt / 1e6, "unit": "Msamples/s", "cores": cores,
                                   "kind": "port", "sample": f"{x_wide.shape[1]} of the bench's channels x {total} "
                                   "samples, de-interleaved beforehand (planar), channels partitioned over threads",
                                   "seconds": round(dt, 2), "msgs": int(o.counters()[:, 0].sum())}
    return res


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes
    (profiles/r01_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this
    same command; FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 correction)."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as f:
            return float(json.load(f)["bytes_per_launch"][kernel])
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--channels", type=int, default=16384, help="channels per GPU")
    ap.add_argument("--len", type=int, default=48000, help="samples per channel per step")
    ap.add_argument("--base", type=int, default=256, help="distinct base streams")
    ap.add_argument("--cpu-channels", type=int, default=8192,
                    help="channels of the batch the single-core CPU baseline runs (about 13 s of CPU work)")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # under torch.distributed.run (even with one rank) go through RCCL for the barrier
    # and the reductions; a bare `python bench.py` needs no process group
    use_dist = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)

    from gnuais_amd import ReceiverBatch, synth, tile_channels

    n_ch, total = args.channels, args.len
    # synthetic input: base streams on the host once, tiled on the device
    base, _ = synth.make_base_streams(args.base, total, seed=synth.SEED + rank)
    x = tile_channels(torch.from_numpy(base).to(device), n_ch)
    torch.cuda.synchronize()

    b = ReceiverBatch(n_ch, max_len=total, device=local)
    stream = torch.cuda.current_stream(device).cuda_stream

    def step():
        b.run(x, stream=stream, sync=False)
        b.discard_frames(stream)

    # one-time calibration (untimed, before the warm-up): which internal stream serves which stage.
    # The hardware queue a stream gets depends on what the process created before and decides up
    # to 1.7x of the pipeline's speed; the library measures it on this input.
    b.autotune(x, stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()

    # isolated kernel durations (one call at a time, nothing overlapping): context only
    b.set_timing(True)
    iso = {k: [] for k in b.KERNELS}
    for _ in range(3):
        step()
        t = b.last_timing()
        for k in iso:
            iso[k].append(t[k])
    torch.cuda.synchronize()
    rx0 = b.total_received()

    # timed region: K steps, asynchronous; the library records HIP events around every
    # kernel on the stream it is launched on (event ring), read back after the region
    b.set_timing(True)
    b.set_option("timing_stride", TIMING_STRIDE)    # the event records themselves cost stream time
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    rx1 = b.total_received()
    live = b.mean_timing()
    b.set_timing(False)
    b.set_option("timing_stride", 1)
    msgs = float(rx1 - rx0)
    # context, outside the timed region: delivering one step's frames to the host as NMEA text,
    # formatted on the device (row f1)
    post = None
    if rank == 0:
        b.discard_frames(stream)
        step_frames_seq = np.zeros(n_ch, dtype=np.uint8)
        for _ in range(2):                          # the first use allocates the text / scratch buffers
            b.run(x, stream=stream, sync=True)
            t_post = time.perf_counter()
            text, n_sent, n_fr = b.drain_nmea(step_frames_seq)
            t_post = time.perf_counter() - t_post
        post = {"what": "gnuais_batch_drain_nmea of one step's frames (device formatter + D2H of the text)",
                "frames": n_fr, "sentences": n_sent, "text_bytes": len(text), "ms": t_post * 1e3,
                "frames_per_s": n_fr / t_post}
    if use_dist:
        from gnuais_amd.shard import reduce_bench
        dt, msgs, _ = reduce_bench(dist, device, dt, msgs, float(n_ch * total * args.steps))
    msgs_per_step = msgs / args.steps

    if rank == 0:
        samples = float(world) * n_ch * total * args.steps
        value = samples / dt / 1e6
        kavg = {k: float(live[k]) for k in b.KERNELS}
        kiso = {k: float(np.mean(v)) for k, v in iso.items()}
        # the roofline object describes the FIR/slicer launch: 98 % of the chain's algorithmic bytes and
        # the largest share of its issued instructions (the PLL launch can take as long, on 256 SIMDs)
        dom = "fir_slice"
        # algorithmic bytes of one launch: every int16 sample read once
        # by K1; K2a/K2b consume K1's 1-bit/sample and ~0.2-bit/sample streams
        alg = {"fir_slice": n_ch * total * 2.0, "pll_edges": n_ch * total / 8.0, "pll_phase": n_ch * total * 0.15 * 2.0,
               "hdlc_deframe": n_ch * total * 0.2 / 8.0,
               "hdlc_crc": msgs_per_step * 80.0}
        ach = alg[dom] / (kavg[dom] * 1e-3) / 1e9
        out = {
            "metric": "Msamples/s demodulated (full chain, N-channel 48 kHz batch)",
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 FIR on int16 samples; u32 PLL/HDLC/CRC", "data": "synthetic",
            "config": {"workload": "C3: 16384 ch x 48000 samples @48 kHz per GPU, full chain "
                                   "incl. HDLC/CRC-16",
                       "channels_per_gpu": n_ch, "samples_per_channel": total,
                       "parallelism": f"channels sharded over {world} GPU(s), no collectives"},
            "valid_crc_msgs_per_s": msgs / dt,
            "x_realtime_channels": value / 0.048,
            "kernel_ms": kavg, "kernel_ms_isolated": kiso, "kernel_ms_calls": int(live["calls"]),
            "post_stage": post,
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": pmc_traffic(dom),
                         "algorithmic_bytes_per_launch": alg[dom],
                         "achieved_isolated": alg[dom] / (kiso[dom] * 1e-3) / 1e9,
                         # K1s is VALU-issue bound, not HBM bound: 27.4 ns of SIMD time per
                         # sample and wave (see K1S_NS_PER_WAVE_SAMPLE)
                         "valu_floor_ms": K1S_NS_PER_WAVE_SAMPLE * 1e-6 * (n_ch / 64.0) * total / N_SIMD,
                         "valu_frac_isolated": (K1S_NS_PER_WAVE_SAMPLE * 1e-6 * (n_ch / 64.0) * total
                                                / N_SIMD) / kiso["fir_slice"]},
        }
        if world == 1 and not args.no_cpu:
            wide = n_ch
            out["cpu_baseline"] = cpu_baseline(x[:, : args.cpu_channels].cpu().numpy(),
                                               args.cpu_channels, total,
                                               np.ascontiguousarray(x[:, :wide].cpu().numpy()))
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
