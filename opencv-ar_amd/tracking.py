"""Stateful tracking over video streams (reference: cvarArMultRegistration's `markers` in/out argument,
/root/reference/src/opencvar.cpp:619, 635-668 -- ARTest keeps one vector<CvarMarker> per camera across frames).

A stream is sequential (frame t needs the markers of frame t-1), streams are independent: one batch = the frame of the
same time step of every stream, batch lane s = stream s, and lane s's `prev` is what lane s returned one step
earlier.  Across GPUs, streams (not frames) are sharded: rank r owns the streams `streams_of(r, world, n)`."""
import numpy as np


def streams_of(rank, world, n_streams):
    """Streams owned by `rank`: contiguous blocks, sizes differing by at most one."""
    base, extra = divmod(n_streams, world)
    first = rank * base + min(rank, extra)
    return list(range(first, first + base + (1 if rank < extra else 0)))


class StreamTracker:
    """Carries each stream's markers from one call to the next, like the reference's caller does."""

    def __init__(self, detector, n_streams):
        self.det = detector
        self.prev = [[] for _ in range(n_streams)]

    def reset(self, stream=None):
        for s in (range(len(self.prev)) if stream is None else [stream]):
            self.prev[s] = []

    def step(self, frames, grey_in_place=False):
        """frames: uint8 [n_streams, H, W, 3] in host memory, the next frame of every stream.
        Returns (markers [n_streams, MAX_MARKERS], counts [n_streams]); the state advances."""
        assert len(frames) == len(self.prev)
        markers, counts = self.det.detect_host(frames, grey_in_place=grey_in_place, prev=self.prev)
        self.prev = [[markers[s, k].copy() for k in range(min(int(counts[s]), markers.shape[1]))] for s in range(len(self.prev))]
        return markers, counts
