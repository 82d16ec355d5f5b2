"""N>1 path on CPU: two gloo processes shard the channel axis, each runs the CPU
oracle on its own block (standing in for its GPU's batch), and the reductions the
benchmark uses (max time, summed messages/samples) are checked against the
unsharded run.  No data-path collective exists to test: shards are independent."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from gnuais_amd import synth
    from gnuais_amd.shard import reduce_bench, shard_range
    from oracle_lib import Oracle
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank,
                            world_size=world)
    n_ch, total = 12, 6 * 1280
    x = np.stack([synth.make_stream(total, seed=51, channel=c)[0] for c in range(n_ch)], axis=1)
    lo, hi = shard_range(n_ch, world, rank)
    o = Oracle(hi - lo)
    o.run(np.ascontiguousarray(x[:, lo:hi]))
    f = o.frames()
    f["channel"] += lo
    dist.barrier()
    t, msgs, samples = reduce_bench(dist, torch.device("cpu"), 1.0 + rank, float(len(f)),
                                    float((hi - lo) * total))
    gathered = [None] * world
    dist.all_gather_object(gathered, f.tobytes())
    if rank == 0:
        q.put((t, msgs, samples, b"".join(gathered)))
    dist.destroy_process_group()


def test_two_rank_sharding_matches_unsharded():
    sys.path.insert(0, ROOT)
    from gnuais_amd import synth
    from gnuais_amd.shard import shard_range
    from oracle_lib import Oracle
    assert [shard_range(10, 4, r) for r in range(4)] == [(0, 2), (2, 5), (5, 7), (7, 10)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    t, msgs, samples, blob = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n_ch, total = 12, 6 * 1280
    x = np.stack([synth.make_stream(total, seed=51, channel=c)[0] for c in range(n_ch)], axis=1)
    o = Oracle(n_ch)
    o.run(x)
    whole = o.frames()
    assert t == 2.0                              # max over ranks
    assert msgs == len(whole) and samples == n_ch * total
    assert blob == whole.tobytes()               # rank-ordered shards = channel-major order
