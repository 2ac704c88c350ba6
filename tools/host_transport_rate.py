#!/usr/bin/env python3
"""Frames in host memory -> markers in host memory through ocvar_hip_detect_host (PCIe-inclusive rate)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as H
import opencv_ar_amd as oa
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
cfg = H.synth_config(3)
base = np.stack([H.synth_frame(cfg, i)[0] for i in range(16)])
frames = np.concatenate([base] * (B // 16))
tpls = oa.load_templates([os.path.join(oa.TEMPLATE_DIR, n + ".png") for n in H.TEMPLATE_ORDER])
cam = oa.default_camera(cfg.width, cfg.height)
det = oa.Detector(cfg.width, cfg.height, max_batch=64)
det.set_templates(tpls); det.set_camera(cam)
import torch
pinned = torch.empty(frames.shape, dtype=torch.uint8, pin_memory=True).numpy()   # hipHostMalloc: page-locked by the CALLER
for label, make in (("pageable buffer (staged through the library's page-locked buffers)", lambda: frames.copy()),
                    ("buffer page-locked by the caller (used in place)", lambda: (pinned.__setitem__(slice(None), frames), pinned)[1])):
    for grey in (False, True):
        det.detect_host(make(), grey_in_place=grey)
        work = make()
        t0 = time.perf_counter()
        m, c = det.detect_host(work, grey_in_place=grey)
        dt = time.perf_counter() - t0
        print(f"detect_host {B} frames 1920x1080, {label}, grey_in_place={grey}: {B / dt:.0f} frames/s, {B * frames[0].nbytes / dt / 1e9:.1f} GB/s host->device"
              f"{' + back' if grey else ''}, markers/frame {c.mean():.2f}")
