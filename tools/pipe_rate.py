#!/usr/bin/env python3
"""frames/s through the C ABI's own scheduler (ocvar_hip_pipe_detect_device: four contexts, gate 2) on the benchmark's
frames -- one call per N resident frames, so that a C/C++ caller's number stands beside bench.py's (whose contexts keep
running across its steps; a pipe call drains at its end).   python tools/pipe_rate.py [frames] [chunk] [contexts] [gate]"""
import os
import sys
import time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # every context's stream on its own hardware queue; before torch touches HIP
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import helpers as H
import opencv_ar_amd as oa

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
nctx = int(sys.argv[3]) if len(sys.argv) > 3 else 4
gate = int(sys.argv[4]) if len(sys.argv) > 4 else 2
cfg = H.synth_config(3)
uniq = 256
base = np.stack([H.synth_frame(cfg, i)[0] for i in range(uniq)])
W, Hh = cfg.width, cfg.height
d = torch.from_numpy(base).cuda().repeat((n + uniq - 1) // uniq, 1, 1, 1)[:n].contiguous()
torch.cuda.synchronize()
pipe = oa.Pipe(W, Hh, chunk_frames=chunk, n_contexts=nctx, gate_width=gate)
pipe.set_templates(oa.load_templates([os.path.join(oa.TEMPLATE_DIR, x + ".png") for x in H.TEMPLATE_ORDER]))
pipe.set_camera(oa.default_camera(W, Hh))
pipe.detect_device(d.data_ptr(), W, Hh, min(n, 4 * chunk), max_per_frame=8)   # warm-up
best = 0.0
for rep in range(3):
    t0 = time.perf_counter()
    m, c = pipe.detect_device(d.data_ptr(), W, Hh, n, max_per_frame=8)
    dt = time.perf_counter() - t0
    best = max(best, n / dt)
    print(f"pipe: {n} frames, chunk {chunk}, {nctx} contexts, gate {gate}: {n / dt:.0f} frames/s ({1e3 * dt:.1f} ms), markers per frame {c.mean():.2f}", flush=True)
print(f"best {best:.0f} frames/s")

# streaming form: the pipeline never drains (chunks out of the same device array, round and round).  Contexts that start together
# stay in lock-step -- all in their binarise kernels, then all in their followers --, so the first n_contexts chunks are
# (i + 1) / n_contexts of a chunk, as bench.py and ocvar_hip_pipe_detect_device stagger theirs.
pipe.set_result_limit(8)
n_chunks, per = int(os.environ.get("PIPE_CHUNKS", "48")), n // chunk
for stagger in (False, True):
    sizes = [chunk * (i + 1) // nctx if stagger and i < nctx else chunk for i in range(n_chunks)]
    for rep in range(2):
        t0 = time.perf_counter()
        sub = done = 0
        while done < n_chunks:
            while sub < n_chunks and pipe.submit(d.data_ptr() + (sub % per) * chunk * W * Hh * 3, W, Hh, sizes[sub], tag=sub):
                sub += 1
            tag, m, c = pipe.collect(chunk, 8)
            assert tag == done and len(c) == sizes[done]
            done += 1
        dt = time.perf_counter() - t0
        print(f"pipe, streaming (submit / collect), first chunks {'staggered' if stagger else 'full'}: {n_chunks} chunks of <= {chunk} frames, {nctx} contexts, "
              f"gate {gate}: {sum(sizes) / dt:.0f} frames/s", flush=True)
