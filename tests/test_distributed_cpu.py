"""world_size-2 gloo test of the N>1 path used by bench.py: frame sharding by index and the gather of the
fixed-size CvarMarker result blocks to rank 0 (on the GPUs the same code runs over RCCL)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, batch, ok, per_frame=None):
    sys.path.insert(0, ROOT)
    import opencv_ar_amd as oa
    from opencv_ar_amd import sharding as S
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    K = per_frame or S.MAX_MARKERS   # records per frame in the gathered block (bench.py gathers 8)
    markers = np.zeros((batch, K), oa.MARKER_DTYPE)
    counts = np.zeros(batch, np.int32)
    for i in range(batch):
        g = S.frame_of(rank, world, i)
        counts[i] = g % 4
        for k in range(min(counts[i], K)):
            markers[i, k]["templateId"] = k
            markers[i, k]["markerId"] = g
            markers[i, k]["glMatrix"][:] = g + 0.5 * k
    block = torch.from_numpy(np.concatenate([markers.view(np.uint8).reshape(-1), counts.view(np.uint8)]))
    assert block.numel() == S.block_bytes(batch, K)
    blocks = S.gather_blocks(block, rank, world, dist)
    if rank == 0:
        res = S.unpack(blocks, batch, oa.MARKER_DTYPE, K)
        assert sorted(res) == list(range(world * batch))
        for g, (c, m) in res.items():
            assert c == g % 4 and len(m) == min(c, K)   # the count is the frame's full count even when the block keeps fewer
            for k in range(len(m)):
                assert m[k]["markerId"] == g and m[k]["templateId"] == k and m[k]["glMatrix"][3] == g + 0.5 * k
        ok.value = 1
    else:
        assert blocks is None
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_gather_world2():
    ok = mp.get_context("spawn").Value("i", 0)
    mp.spawn(_worker, args=(2, 29517, 5, ok), nprocs=2, join=True)
    assert ok.value == 1


def test_shard_and_gather_world2_narrow_blocks():
    """blocks of 2 records per frame: counts above 2 still arrive, the records are the frames' first two"""
    ok = mp.get_context("spawn").Value("i", 0)
    mp.spawn(_worker, args=(2, 29519, 5, ok, 2), nprocs=2, join=True)
    assert ok.value == 1


def test_frame_sharding_covers_every_frame_once():
    sys.path.insert(0, ROOT)
    from opencv_ar_amd import sharding as S
    for world in (1, 2, 4, 8):
        seen = sorted(S.frame_of(r, world, i) for r in range(world) for i in range(6))
        assert seen == list(range(6 * world))


def test_stream_sharding_covers_every_stream_once():
    sys.path.insert(0, ROOT)
    from opencv_ar_amd.tracking import streams_of
    for world in (1, 2, 3, 8):
        for n in (1, 5, 8, 13):
            owned = [streams_of(r, world, n) for r in range(world)]
            assert sorted(s for o in owned for s in o) == list(range(n))
            assert max(len(o) for o in owned) - min(len(o) for o in owned) <= 1
