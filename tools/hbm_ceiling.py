#!/usr/bin/env python3
"""What a plain streaming kernel reaches on this GPU: device-to-device copies and a read-only reduction of 2 GiB, best of 20.
The roofline in bench.py is priced against the nominal 8 TB/s; this is the practical ceiling the same launch sizes see."""
import torch, time
n = 2 << 30
a = torch.empty(n, dtype=torch.uint8, device="cuda")
b = torch.empty(n, dtype=torch.uint8, device="cuda")
a.random_(0, 255)
def best(fn, reps=20):
    fn(); torch.cuda.synchronize()
    t = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        t.append(e0.elapsed_time(e1) * 1e-3)
    return min(t)
t = best(lambda: b.copy_(a))
print(f"copy 2 GiB (read + write): {2 * n / t / 1e9:.0f} GB/s")
a32 = a.view(torch.int32)
t = best(lambda: a32.sum())
print(f"read-only sum of 2 GiB: {n / t / 1e9:.0f} GB/s")
t = best(lambda: b.zero_())
print(f"write-only fill of 2 GiB: {n / t / 1e9:.0f} GB/s")
