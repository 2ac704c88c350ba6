"""Frame sharding and result gathering for N GPUs of one node (one process per GPU, torch.distributed).

Frames are independent work units: rank r owns global frames r, r+N, r+2N, ...; the only exchange is
one gather of the fixed-size result block per batch to rank 0 (backend "nccl" = RCCL over xGMI on the GPUs,
"gloo" in the CPU tests).  Result block per rank: B x per_frame CvarMarker records (184 B) then B int32 counts; per_frame
is MAX_MARKERS unless the caller knows a tighter bound for its frames (the counts are the full counts either way, so a frame
with more markers than its block keeps is seen on rank 0)."""
import numpy as np
import torch

MARKER_BYTES, MAX_MARKERS = 184, 64


def frame_of(rank, world, local_index):
    """global frame index processed at position local_index of rank's batch"""
    return rank + world * local_index


def block_bytes(batch, per_frame=MAX_MARKERS):
    return batch * per_frame * MARKER_BYTES + 4 * batch


def gather_blocks(local_block, rank, world, dist=None):
    """local_block: uint8 tensor [block_bytes(B)] on the rank's device.  Returns on rank 0 a list of `world`
    tensors (rank order), elsewhere None."""
    if dist is None or not dist.is_initialized():   # single process, no process group: nothing to exchange
        assert world == 1
        return [local_block]
    if local_block.is_cuda and dist.get_backend() == "gloo":   # rehearsal of the N > 1 flow without RCCL: stage through the host
        local_block = local_block.cpu()
    out = [torch.empty_like(local_block) for _ in range(world)] if rank == 0 else None
    dist.gather(local_block, out, dst=0)
    return out


def unpack(blocks, batch, marker_dtype, per_frame=MAX_MARKERS):
    """blocks (rank order) -> dict global_frame -> (count, markers[:min(count, per_frame)]) on rank 0"""
    world = len(blocks)
    res = {}
    for r, blk in enumerate(blocks):
        raw = blk.cpu().numpy()
        markers = raw[:batch * per_frame * MARKER_BYTES].view(marker_dtype).reshape(batch, per_frame)
        counts = raw[batch * per_frame * MARKER_BYTES:].view(np.int32)
        for i in range(batch):
            res[frame_of(r, world, i)] = (int(counts[i]), markers[i, :min(int(counts[i]), per_frame)].copy())
    return res
