"""CPU, world_size 2, gloo: the N>1 path of bench.py is embarrassingly parallel: every rank
owns distinct windows, there is no data-path collective, the timed region is bracketed by barriers, elapsed is
the MAX over ranks and the reported value is whole-job audio-seconds / that time."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from whisper_rs_amd import synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    batch, steps = 2, 3
    audio = synth.make_audio(batch, seed=bench.window_seed(rank), n_samples=1600)   # the rank's own windows
    elapsed_local = 0.010 * (rank + 1)                                               # rank 1 is the slow one
    elapsed = bench.max_over_ranks(elapsed_local, world, device=None)
    q.put((rank, float(np.abs(audio).sum()), elapsed, bench.whole_job_xrt(steps, batch, world, elapsed)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_aggregation():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, sum0, e0, v0), (r1, sum1, e1, v1) = res
    assert sum0 != sum1, "ranks must decode different windows"
    assert abs(e0 - 0.020) < 1e-9 and abs(e1 - 0.020) < 1e-9, "elapsed must be the max over ranks on every rank"
    assert abs(v0 - (3 * 2 * 30.0 * 2) / 0.020) < 1e-6 and v0 == v1
