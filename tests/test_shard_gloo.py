"""World-size-2 gloo test (CPU) of the multi-process path: recipe slices per
rank follow Recipe::read's -B/-I rule and partition the recipe, the model
broadcast delivers rank 0's parameters bit for bit, statistics aggregate."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_lines, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aaltoasr_amd import shard, synth
    first, count = shard.rank_slice(n_lines, world, rank)
    mean, var, off, idx, w = synth.make_model(D=5, G=24, S=3, comps=8, seed=77 if rank == 0 else 78)
    got = shard.broadcast_model(dict(mean=mean, var=var, mix_off=off, mix_idx=idx, mix_w=w), src=0)
    ref = synth.make_model(D=5, G=24, S=3, comps=8, seed=77)
    same = all(np.array_equal(got[k], v) for k, v in zip(["mean", "var", "mix_off", "mix_idx", "mix_w"], ref))
    frames, secs = shard.aggregate(100 * (rank + 1), 0.5 + rank)
    q.put((rank, first, count, same, frames, secs))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_lines", [7, 10, 1])
def test_two_ranks_partition_and_broadcast(capi, oracle, n_lines):
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_lines, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    text = "\n".join("audio=a%d lna=l%d" % (i, i) for i in range(n_lines)) + "\n"
    covered = []
    for rank, first, count, same, frames, secs in res:
        want = oracle.recipe_read(text, world, rank + 1)
        assert count == len(want)
        if count:
            assert want[0].audio_path == "a%d" % first
        covered += list(range(first, first + count))
        assert same
        assert frames == 300 and secs == 1.5
    if n_lines >= world:
        assert covered == list(range(n_lines))
