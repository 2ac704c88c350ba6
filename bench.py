#!/usr/bin/env python3
"""bench.py -- frames/s of the AR-marker detection path (BASELINE.json metric) on N MI355X.

A "step" is one pass of the whole hot path (12 launches: binarise, follower tiers 1-3, order/crops, binarise crops,
follower tiers 1, 2 (two phases), 3 on the crops, decode, dedupe+pose, and the copy-out of the CvarMarker arrays) over one batch of
synthetic frames per GPU.
Workload at every N: BASELINE.json configs[2] -- 1920x1080, 16 planted markers/frame, templates 2x2/3x3/4x4.
Frames are resident in HBM before the timed region.  For N > 1 (one process per GPU, torchrun) frames are
sharded by frame, there is no data-path collective; the per-rank CvarMarker arrays are gathered to rank 0 over
RCCL once per step (inside the timed region).  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

# Multi-process GPU work on this pool (one rank per GPU, RCCL over xGMI): the host driver only supports dmabuf IPC; without
# this RCCL's cross-process buffer registration fails with "hipIpcGetMemHandle: invalid argument".  The ROCm runtime reads
# it when it initialises, i.e. it must be in the environment before the first HIP call of the process (torch is imported
# below).  The image exports it already; a caller's shell that dropped it gets it back here.  (One-rank runs do not need it.)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# The contexts' streams must not share hardware queues: the ROCm runtime multiplexes a process's HIP streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4), and with the null stream holding one of them the fourth context's kernels queue
# behind another stream's (measured: 166 k frames/s instead of 200 k for four contexts on the runtime's default).  Read by the
# runtime when it initialises, like the variable above; INTEGRATION.md tells C/C++ callers the same.  12: four context streams, the
# null stream, and -- in multi-rank runs -- the gather stream and RCCL's own streams each get a queue (one-GPU run: 5 .. 12 queues all give
# ~195 k; with RCCL initialised 6 / 8 / 12 queues give 155 / 182 / 195 k).  Five contexts and RCCL (round 3, one-rank rehearsal with
# the per-step gather): 12 / 16 / 24 queues give 213.6 / 213.9 / 218.8 k against 232 k without the gather -- 24 for multi-rank runs.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24" if int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("OCVAR_BENCH_FORCE_DIST") else "12")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def cpu_baseline(frames, tpls, cam, budget_s=10.0):
    """The oracle (CPU restatement of the reference path, oracle/) timed on a bounded sample of the same frames on this
    box's host cores: all hardware threads, frame-parallel (std::thread inside liboracle.so), and one thread -- the
    reference itself is single-threaded.  Reported beside the GPU number, never part of it."""
    import helpers as H
    lib = H.oracle()
    lib.orc_registration_throughput.restype = C.c_longlong
    lib.orc_registration_throughput.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_int,
                                                C.c_void_p, C.c_int, C.c_double, C.c_longlong, C.c_void_p, C.c_void_p]
    frames = np.ascontiguousarray(frames)
    n, h, w, _ = frames.shape

    def run(threads, budget):
        sec, used = C.c_double(0), C.c_int(0)
        done = lib.orc_registration_throughput(frames.ctypes.data, n, w, h, 3 * w, 3 * w * h, C.byref(tpls), len(tpls),
                                               C.byref(cam), threads, budget, 1 << 40, C.byref(sec), C.byref(used))
        return done, sec.value, used.value

    d1, s1, _ = run(1, min(6.0, budget_s))
    dn, sn, cores = run(0, budget_s)
    return {"value": round(dn / sn, 3), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{dn} runs over {n} of the benchmark's {w}x{h} frames in {sn:.1f} s, oracle (CPU restatement; the reference "
                      f"is not runnable without OpenCV), frame-parallel over {cores} host threads",
            "one_thread": {"value": round(d1 / s1, 3), "unit": "frames/s", "cores": 1,
                           "sample": f"{d1} frames in {s1:.1f} s, one thread (the reference is single-threaded)"}}


def call_latency(config_id, calls=60):
    """One frame per call through the reference's own entry point (cvarArMultRegistration of libopencv-ar.so, host frame
    in, markers out, frame greyed in place): the C++ sample samples/artest_latency.cpp as a child process."""
    import subprocess
    import opencv_ar_amd as oa
    exe = os.path.join(os.path.dirname(oa.LIB_DIR), "bin", "artest_latency")
    try:
        out = subprocess.run([exe, oa.TEMPLATE_DIR, str(config_id), str(calls)], capture_output=True, text=True, timeout=300)
        return json.loads(out.stdout.strip().splitlines()[-1])
    except Exception as e:   # reported, never fatal for the throughput line
        return {"error": repr(e)}


def check_results(det, d_frames_ptr, cfg, truths, W, Hh, n):
    """Sanity check outside the timed region (results, not just speed): in the first n frames every pre-dedupe candidate whose
    code matched a template sits on a planted marker, and most planted markers are decoded (a rotated marker's code is not
    always read back: the reference samples a (w+2)x(h+2) patch; the oracle decodes 13-18 of these frames' 16).  The
    CvarMarker arrays themselves keep at most one marker per template and frame: the `||` dedupe, opencvar.cpp:784-785."""
    det.detect_device(d_frames_ptr, W, Hh, n)
    for f in range(n):
        planted = [t["corner"] for t in truths[f] if not t["occluded"]]
        hits = [np.array(c.square).reshape(4, 2) for c in det.debug_candidates(f) if c.orient > 0]
        assert len(hits) >= 0.6 * len(planted), f"frame {f}: {len(hits)} decoded markers, {len(planted)} planted"
        for sq in hits:
            off = min(max(np.abs(tc - sq[k]).sum(axis=1).min() for k in range(4)) for tc in planted)
            assert off <= 12, f"frame {f}: decoded square {sq.tolist()} is {off} px off every planted marker"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8192, help="frames per GPU per step (over all streams)")
    ap.add_argument("--streams", type=int, default=5, help="independent detector contexts / HIP streams per GPU; the batch is split over them")
    ap.add_argument("--gate", type=int, default=2, help="at most this many binarise kernels of the contexts run at once (ocvar_hip_gate_create; 0: no gate)")
    ap.add_argument("--unique", type=int, default=256, help="distinct synthetic frames per GPU (tiled to the batch on the device)")
    ap.add_argument("--config", type=int, default=3, help="BASELINE.json config id (3 = headline)")
    ap.add_argument("--tune", action="append", default=[], metavar="KNOB=VALUE",
                    help="result-invariant launch parameter for every context (ocvar_hip_set_tuning: crop_phases, mid_steps, mid_blocks, "
                         "long_blocks, short_blocks, min_units); experiments only -- the default run sets none")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the one-frame-per-call latency leg (child process)")
    ap.add_argument("--no-check", action="store_true", help="skip the result check before the warm-up (profile runs: keeps every launch of a kernel full-size)")
    args = ap.parse_args()

    import torch
    import helpers as H
    import opencv_ar_amd as oa
    from opencv_ar_amd import sharding as S

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU; there is no CPU path"
    # OCVAR_BENCH_BACKEND=gloo rehearses the N > 1 control flow on a box with fewer GPUs than ranks (ranks share devices,
    # the gather is staged through the host); the driver's runs use the default: one GPU per rank, RCCL.
    backend = os.environ.get("OCVAR_BENCH_BACKEND", "nccl")
    device_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    dist = None
    # OCVAR_BENCH_FORCE_DIST=1 (under torchrun with one rank): initialise RCCL and run the per-step gather with world 1 -- what
    # the presence of the communicator and its streams costs the four contexts can then be measured on a one-GPU box
    force_dist = bool(os.environ.get("OCVAR_BENCH_FORCE_DIST"))
    if world > 1 or force_dist:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend=backend)

    cfg = H.synth_config(args.config)
    names = ["2x2-01"] if args.config in (1, 2) else None
    W, Hh, B = cfg.width, cfg.height, args.batch
    uniq = max(1, min(args.unique, B))
    # frame index space is sharded by frame: rank r owns frames r, r+world, ...
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(min(32, os.cpu_count() or 1)) as ex:   # the generator is C (ctypes releases the GIL)
        made = list(ex.map(lambda i: H.synth_frame(cfg, S.frame_of(rank, world, i), names), range(uniq)))
    base = np.stack([m[0] for m in made])
    truths = [m[1] for m in made]
    # setup side through the product's own host library (cvarLoadTemplateTag on the PNGs, cvarReadCamera(NULL) +
    # cvarCameraScale); the oracle is only touched by the cpu_baseline leg below
    tpl_list = oa.load_templates([os.path.join(oa.TEMPLATE_DIR, n + ".png") for n in (names or H.TEMPLATE_ORDER)])
    camera = oa.default_camera(W, Hh)
    # The batch is split over `streams` independent contexts, each with its own HIP stream and workspace, so that the
    # latency-bound kernels of one sub-batch (border following) overlap the streaming kernels of another.  A gate shared by
    # the contexts keeps at most two binarise kernels running at once (include/ocvar_hip.h: left alone the contexts drift --
    # none of them in a binarise kernel a quarter of the time, three of them a sixth of the time).
    NS = max(1, min(args.streams, B))
    sub = [B // NS + (1 if i < B % NS else 0) for i in range(NS)]
    offs = np.concatenate([[0], np.cumsum(sub)]).astype(int)
    dets, streams = [], []
    gate = oa.Gate(args.gate, device_index) if args.gate > 0 else None
    for i in range(NS):
        det = oa.Detector(W, Hh, max_batch=sub[i], device=device_index)
        det.set_templates(tpl_list)
        det.set_camera(camera)
        if gate is not None:
            det.set_gate(gate)
        if args.tune:
            det.set_tuning(**{kv.split("=")[0]: int(kv.split("=")[1]) for kv in args.tune})
        det.set_result_limit(8)   # what collect(8) takes: 3 MB instead of 24 MB of marker records per 2048-frame launch
        dets.append(det)
        # the context's own stream (ocvar_hip_enqueue with stream NULL); wrapped for the event waits of the gather below
        streams.append(torch.cuda.ExternalStream(det.stream_ptr()))
    det = dets[0]
    # the batch is the distinct frames tiled, built on the device (the host only ever holds `uniq` frames)
    d_frames = torch.from_numpy(base).cuda().repeat((B + uniq - 1) // uniq, 1, 1, 1)[:B].contiguous()
    torch.cuda.synchronize()   # the detectors run on their own streams: the tiling copy (torch's stream) must have finished
    frame_bytes = W * Hh * 3
    # result blocks for the RCCL gather, double-buffered: launches enqueued during step k write buffer (k+1)%2 while the
    # gather at the end of step k reads buffer k%2 (filled by the launches that were collected during step k)
    # (GATHER_K records per frame travel: a stateless frame yields at most one marker per template -- the `||` dedupe,
    # opencvar.cpp:784-785 --, the gathered counts are the frames' full counts and are checked against GATHER_K in the warm-up)
    GATHER_K = 8
    d_res = [torch.zeros(S.block_bytes(B, GATHER_K), dtype=torch.uint8, device="cuda") for _ in range(2)]
    nbytes_m = B * GATHER_K * S.MARKER_BYTES

    # K steps as a software pipeline.  Every context runs its sub-batches back to back on its own stream; the contexts are
    # staggered by 1/NS of a sub-batch (context i's first launch covers (i+1)/NS of its sub-batch, its last launch the
    # rest), because contexts that start together stay in lock-step -- all four in the VALU-bound binarise kernels, then
    # all four in the latency-bound followers -- and nothing overlaps.  Per context: first + (K-1) full + rest = K
    # sub-batches; the host only ever waits for the context whose launch is oldest.
    stage_full = np.zeros(12, np.float64)   # per-kernel event times of full-size launches inside the timed region
    n_full = [0]
    ref_event = torch.cuda.Event(enable_timing=True)   # one clock for the stage stamps of all contexts
    ref_event.record()
    ref_event.synchronize()
    spans = {0: [], 5: []}   # (start, end) in ms after ref_event of every full-size launch of the two binarise kernels (timed region)
    frames_done = [0]
    last = {}

    # The gather runs on its own high-priority stream and is never waited for on the host inside a step: a launch that writes a
    # result buffer again waits (stream-ordered) for the gather that last read it.  (Waiting on the host after every gather
    # cost 5 %: the collective sits in a hardware queue behind a context's kernels, and meanwhile nothing is collected or
    # re-enqueued.)
    gathering = world > 1 or force_dist
    dist_skip = os.environ.get("OCVAR_BENCH_DIST_SKIP", "") if force_dist else ""   # one-rank rehearsal only: "blocks", "gather" or both -- what the parts of the distributed path cost
    gather_host_s = [0.0]   # host time spent issuing the gathers (the contexts are not re-enqueued meanwhile)
    gather_stream = torch.cuda.Stream(priority=-1) if gathering else None
    gather_done = [None, None]

    def enqueue(i, n, buf, skip=0):
        """context i detects frames [offs[i] + skip, offs[i] + skip + n) of the batch"""
        o = int(offs[i]) + skip
        if gather_done[buf] is not None:
            streams[i].wait_event(gather_done[buf])
        dets[i].enqueue_device(d_frames.data_ptr() + o * frame_bytes, W, Hh, n)
        if gathering and "blocks" not in dist_skip:
            dets[i].results_to_device(d_res[buf].data_ptr() + o * GATHER_K * S.MARKER_BYTES,
                                      d_res[buf].data_ptr() + nbytes_m + 4 * o, streams[i].cuda_stream, per_frame=GATHER_K)
        last[i] = n

    def collect(i, timed):
        m, c = dets[i].collect(8)
        frames_done[0] += last[i]
        if timed and last[i] == sub[i]:
            stage_full[:] += dets[i].stage_ms()
            n_full[0] += 1
            st = dets[i].stage_stamps(ref_event.cuda_event)
            for k in spans:
                spans[k].append((float(st[k]), float(st[k + 1])))
        return m, c

    def run(K, timed):
        first = [max(1, ((i + 1) * sub[i]) // NS) for i in range(NS)]
        parts = [None] * NS
        for i in range(NS):
            enqueue(i, first[i], 0)
        for k in range(K):
            for i in range(NS):
                parts[i] = collect(i, timed)
                n = sub[i] if k < K - 1 else sub[i] - first[i]
                if n > 0:   # the last, partial launch covers the frames the first one left out
                    enqueue(i, n, (k + 1) % 2, skip=0 if k < K - 1 else first[i])
                else:
                    last.pop(i)
            if gathering and "gather" not in dist_skip:   # one gather of the ranks' result blocks per step (the launches that filled this buffer are complete)
                t_g = time.perf_counter()
                with torch.cuda.stream(gather_stream):
                    blocks = S.gather_blocks(d_res[k % 2], rank, world, dist)
                    gather_done[k % 2] = torch.cuda.Event()
                    gather_done[k % 2].record(gather_stream)
                gather_host_s[0] += time.perf_counter() - t_g
                if rank == 0 and not timed:   # warm-up only: the gathered blocks decode and no frame has more markers than a block keeps
                    gather_stream.synchronize()
                    got = S.unpack(blocks, B, oa.MARKER_DTYPE, GATHER_K)
                    assert max(c for c, _ in got.values()) <= GATHER_K, "a frame has more markers than the gathered block keeps"
        for i in list(last):
            parts[i] = collect(i, timed)
            last.pop(i)
        if gathering:
            gather_stream.synchronize()   # every gather of this run has arrived
        return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])

    if not args.no_check:
        check_results(dets[0], d_frames.data_ptr(), cfg, truths, W, Hh, min(8, uniq, sub[0]))
    if args.warmup > 0:
        markers, counts = run(args.warmup, False)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    frames_done[0] = 0
    t0 = time.perf_counter()
    markers, counts = run(args.steps, True)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    assert frames_done[0] == B * args.steps, (frames_done[0], B, args.steps)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # Outside the timed region: per-kernel durations of launches that run alone on the GPU (one context, its
    # sub-batch).  With several streams the timed region's launches overlap each other, so their event-to-event
    # durations include time shared with other kernels; these isolated durations are what a roofline of the kernel itself
    # should be read against.  Reported separately, never mixed into `value`.
    iso = np.zeros(12, np.float64)
    for _ in range(3):
        dets[0].enqueue_device(d_frames.data_ptr(), W, Hh, sub[0])
        dets[0].collect(8)
        iso += dets[0].stage_ms() / 3

    if rank == 0:
        K = args.steps
        fps = world * B * K / dt
        stage_ms = stage_full / max(1, n_full[0])   # mean over the full-size launches of the timed region
        crop_pixels = sum(float(d.counters()[4]) for d in dets) / B
        n_out = float(counts.sum()) / max(1, len(counts))
        # algorithmic bytes per frame = 6*W*H + sum of crop areas + 184 per output marker
        alg_frame = 6.0 * W * Hh + crop_pixels + 184.0 * n_out
        # Per-kernel mean launch duration (HIP events on the launch streams, timed region) and algorithmic bytes per
        # launch; one launch covers one sub-batch of Bs frames.
        Bs = B / NS
        WH = float(W * Hh)
        kernels = [
            ("binarise_frames_kernel", 0, 5.0 * WH * Bs),          # 3 B BGR read + 1 B grey + 1 B mask per pixel
            ("follow_kernel<frames,1>", 1, 1.0 * WH * Bs / 16),     # touches masks near borders only; bound: latency
            ("follow_mid_kernel<frames>", 2, 1.0 * WH * Bs / 16),
            ("follow_long_kernel<frames>", 3, 1.0 * WH * Bs / 16),
            ("binarise_crops_kernel", 5, 2.0 * crop_pixels * Bs),   # grey crop read + mask write
            ("follow_kernel<crops,1>", 6, 1.0 * crop_pixels * Bs / 16),
            ("follow_mid_kernel<crops>", 7, 1.0 * crop_pixels * Bs / 16),
            ("follow_long_kernel<crops>", 8, 1.0 * crop_pixels * Bs / 16),
        ]
        # the dominant kernel: the one that moves the most algorithmic bytes (the follower tiers are latency-bound walks whose
        # byte figure is nominal; with several contexts in flight their launches are the ones the GPU time-slices, so their
        # launch durations say how the contexts share the GPU, not what the kernels do)
        dom = max(kernels, key=lambda k: k[2])
        # `roofline` is the contract's figure: algorithmic bytes per launch / the kernel's mean launch duration (HIP events on
        # its stream, timed region; agrees with the rocprofv3 --stats average of the same command).  The contexts' launches of
        # this kernel overlap each other, though (the gate lets two binarise kernels run at once), and a launch's own duration
        # counts the time it shared with another launch of the SAME kernel twice.  `while_running` therefore adds what the
        # kernel moved while it was on the GPU at all: the bytes of all its launches over the union of their [start, end]
        # intervals (all contexts' events on one clock, ocvar_hip_stage_stamps).  Time shared with OTHER kernels stays in both.
        def union_ms(iv):
            iv = sorted(iv)
            tot, cur_a, cur_b = 0.0, None, None
            for a, b in iv:
                if cur_b is None or a > cur_b:
                    if cur_b is not None:
                        tot += cur_b - cur_a
                    cur_a, cur_b = a, b
                else:
                    cur_b = max(cur_b, b)
            return tot + ((cur_b - cur_a) if cur_b is not None else 0.0)
        iv = spans.get(dom[1], [])
        busy_ms = union_ms(iv) if iv else stage_ms[dom[1]] * max(1, n_full[0])
        excl_ms = busy_ms / max(1, len(iv))        # per launch, overlap with launches of the same kernel shared out
        ach = dom[2] / (stage_ms[dom[1]] * 1e-3) / 1e9      # the contract's figure: bytes per launch / mean launch duration
        ach_busy = dom[2] / (excl_ms * 1e-3) / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "r03_traffic.json")
        if os.path.exists(tf):   # HBM bytes per launch from rocprofv3 PMC passes of this kernel at this launch size (tools/collect_traffic.py)
            t = json.load(open(tf)).get(dom[0].split("<")[0] if dom[0].startswith("binarise") else dom[0])
            if t and t.get("width") == W and t.get("height") == Hh and t.get("frames_per_launch") == int(Bs):
                traffic = round(t["hbm_bytes_per_frame"] * Bs)
        out = {
            "metric": "frames/sec at 1920x1080, 16 markers/frame; 1/2/4/8 MI355X", "value": round(fps, 2), "unit": "frames/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / K, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"configs[{args.config - 1}]: {W}x{Hh}, {cfg.grid_x * cfg.grid_y} markers/frame, templates "
                                   f"{'2x2' if names else '2x2/3x3/4x4 x 4 rotations'}, batch {B} frames/GPU/step "
                                   f"({uniq} distinct) in {NS} stream(s)" + (f", at most {args.gate} binarise kernels at once" if gate is not None else "") + ", stateless", "frames_per_step_per_gpu": B, "streams": NS,
                       "library": oa.build_info(), "tuning": args.tune or "defaults",
                       "parallelism": f"frame-sharded x{world}" + ((f", RCCL gather of CvarMarker arrays ({GATHER_K} records per frame + counts)" if backend == "nccl" else f", {backend} rehearsal of the gather (ranks share devices)") if world > 1 else ""),
                       "gather_host_ms_per_step": round(1e3 * gather_host_s[0] / max(1, args.steps + args.warmup), 3) if gathering else None},
            "roofline": {"bound": "hbm", "kernel": dom[0], "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "launch_ms": round(float(stage_ms[dom[1]]), 4), "alg_bytes_per_launch": round(dom[2]),
                         "while_running": {"launches": len(iv), "kernel_on_gpu_ms": round(float(busy_ms), 3), "ms_per_launch": round(float(excl_ms), 4),
                                           "achieved": round(ach_busy, 2), "frac": round(ach_busy / HBM_PEAK_GBS, 5),
                                           "note": "bytes of all full-size launches of this kernel in the timed region / time at least one of them was running "
                                                   "(the contexts' launches overlap each other: launch_ms counts time two of them ran side by side twice)"}},
            "pipeline_roofline": {"bound": "hbm", "achieved": round(alg_frame * fps / world / 1e9, 2), "peak": HBM_PEAK_GBS,
                                  "unit": "GB/s", "frac": round(alg_frame * fps / world / 1e9 / HBM_PEAK_GBS, 5),
                                  "alg_bytes_per_frame": round(alg_frame)},
            "binarise_roofline": {"kernel": "binarise_frames_kernel", "bound": "hbm", "achieved": round(5.0 * WH * Bs / (stage_ms[0] * 1e-3) / 1e9, 2),
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(5.0 * WH * Bs / (stage_ms[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                  "launch_ms": round(float(stage_ms[0]), 4)},
            "stage_ms": {k: round(float(v), 4) for k, v in zip(oa.STAGE_NAMES, stage_ms)},
            "isolated_launch_ms": {k: round(float(v), 4) for k, v in zip(oa.STAGE_NAMES, iso)},
            "binarise_roofline_isolated": {"kernel": "binarise_frames_kernel", "bound": "hbm", "frames_per_launch": sub[0],
                                           "achieved": round(5.0 * WH * sub[0] / (iso[0] * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": round(5.0 * WH * sub[0] / (iso[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                           "note": "same kernel, launches not overlapped by other streams (outside the timed region)"},
            "markers_per_frame": round(n_out, 3),
        }
        if not args.no_cpu_baseline and world == 1:   # (the CPU leg and the per-call latency leg: rank 0 of the one-GPU run only)
            tpls = (H.Template * len(tpl_list))(*[H.Template.from_buffer_copy(bytes(t)) for t in tpl_list])
            out["cpu_baseline"] = cpu_baseline(base[:min(uniq, 256)], tpls, H.Camera.from_buffer_copy(bytes(camera)))
        if not args.no_latency and world == 1:
            lat = call_latency(args.config)
            out["latency_ms"] = lat.get("median_ms")
            out["latency"] = lat
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
