"""The N>1 protocol of bench.py on the CPU (gloo, world_size 2): barrier-bracketed timed region, MAX over
ranks, whole-job value = all ranks' units / that time, one shard (seed) per rank, no data-path collective."""
import os
import socket
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import bench


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rsgnet_amd.weights import synth_input
    x = synth_input(2, (192, 256), seed=rank)               # each rank owns its shard
    delay = 0.01 * (rank + 1)                                # rank 1 is the slow one

    def step():
        time.sleep(delay)

    wall, dev_ms = bench.timed_steps(step, 5, dist, None)
    count = [0]

    def counted():
        count[0] += 1
        step()

    cold = bench.settle_clocks(counted, 1, 3, dist, None, 5.0, block=2, patience=1 + rank)
    out[rank] = (wall, float(x.sum()), count[0], cold["settle_blocks"])
    dist.destroy_process_group()


def test_two_rank_protocol_reports_max_over_ranks():
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        (w0, s0, n0, c0), (w1, s1, n1, c1) = out[0], out[1]
    # the untimed clock-settling blocks: the ranks stop TOGETHER (rank 1 is the patient one here), so that none of them waits,
    # cooling down, at the barrier of the timed region
    assert c0 == c1 >= 2 and n0 == n1 == 1 + 3 + 2 * c0
    assert abs(w0 - w1) < 1e-9            # every rank ends up with the same (max) time
    assert w0 >= 5 * 0.02 * 0.95          # ... which is the slow rank's
    assert s0 != s1                        # shards differ (seed = rank)
    value = world * 2 * 5 / w0             # whole-job units / max time, as bench.py computes it
    assert 0 < value < world * 2 * 5 / (5 * 0.02) * 1.05
