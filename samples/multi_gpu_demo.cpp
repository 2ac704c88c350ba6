// multi_gpu_demo.cpp -- C++ caller of include/ocvar_multi.h: shards a batch of synthetic frames over the GPUs of this
// node (frame f -> GPU f mod N), gathers the CvarMarker arrays to GPU 0 over RCCL, and checks the result against the
// single-GPU entry point of include/ocvar_hip.h on the same frames.  Prints one JSON line.
//
//   multi_gpu_demo <template_dir> [n_devices=all] [config_id=3] [frames_per_device=8] [timed_rounds=5]
#include "ocvar_multi.h"
#include "ocvar_synth.h"
#include "opencvar/opencvar.h"
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

int main(int argc, char** argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s <template_dir> [n_devices] [config_id] [frames_per_device] [timed_rounds]\n", argv[0]);
        return 2;
    }
    const std::string dir = argv[1];
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        std::fprintf(stderr, "no HIP device\n");
        return 1;
    }
    const int N = argc > 2 && std::atoi(argv[2]) > 0 ? std::atoi(argv[2]) : ndev;
    const int config_id = argc > 3 ? std::atoi(argv[3]) : 3;
    const int per_dev = argc > 4 ? std::atoi(argv[4]) : 8;
    const int rounds = argc > 5 ? std::atoi(argv[5]) : 5;
    static const unsigned char g22[16] = {0,0,0,0, 0,255,0,0, 0,0,255,0, 0,0,0,0};
    static const unsigned char g33[25] = {0,0,0,0,0, 0,255,255,255,0, 0,255,255,0,0, 0,255,0,255,0, 0,0,0,0,0};
    static const unsigned char g44[36] = {0,0,0,0,0,0, 0,255,0,255,255,0, 0,0,255,255,255,0, 0,0,255,255,255,0,
                                          0,255,0,255,255,0, 0,0,0,0,0,0};
    const char* names[3] = {"2x2-01.png", "3x3-01.png", "4x4-01.png"};
    const OcvarSynthTemplate grids[3] = {{g22, 4, 4}, {g33, 5, 5}, {g44, 6, 6}};
    const int n_tpl = config_id <= 2 ? 1 : 3;
    std::vector<OcvarTemplate> tpl(n_tpl);
    for (int i = 0; i < n_tpl; i++)
        if (!cvarLoadTemplateTag(reinterpret_cast<CvarTemplate*>(&tpl[i]), (dir + "/" + names[i]).c_str())) {
            std::fprintf(stderr, "cannot load %s/%s\n", dir.c_str(), names[i]);
            return 1;
        }
    OcvarSynthConfig cfg;
    ocvar_synth_config(config_id, &cfg);
    CvarCamera cam;
    cvarReadCamera(NULL, &cam);
    cvarCameraScale(&cam, cfg.width, cfg.height);

    const int F = N * per_dev;
    const size_t fb = (size_t)cfg.width * cfg.height * 3;
    std::vector<unsigned char> frames(fb * F);
    for (int f = 0; f < F; f++) ocvar_synth_frame(&cfg, (unsigned long long)f, grids, n_tpl, frames.data() + fb * f, cfg.width * 3, nullptr, 0);

    OcvarMulti* m = nullptr;
    int rc = ocvar_multi_create(&m, nullptr, N, cfg.width, cfg.height, per_dev);
    if (rc) {
        std::fprintf(stderr, "ocvar_multi_create failed (%d): %s\n", rc, ocvar_multi_last_error(m));
        return 1;
    }
    if ((rc = ocvar_multi_set_templates(m, tpl.data(), n_tpl)) || (rc = ocvar_multi_set_camera(m, reinterpret_cast<OcvarCamera*>(&cam)))) {
        std::fprintf(stderr, "setup failed (%d): %s\n", rc, ocvar_multi_last_error(m));
        return 1;
    }
    std::vector<OcvarMarker> mk((size_t)F * OCVAR_MAX_MARKERS), ref((size_t)F * OCVAR_MAX_MARKERS);
    std::vector<int> cn(F), cref(F);
    rc = ocvar_multi_detect_host(m, frames.data(), cfg.width, cfg.height, cfg.width * 3, fb, F, mk.data(), cn.data(), OCVAR_MAX_MARKERS);
    if (rc) {
        std::fprintf(stderr, "ocvar_multi_detect_host failed (%d): %s\n", rc, ocvar_multi_last_error(m));
        return 1;
    }
    // the same frames through the single-GPU entry point
    OcvarHip* one = nullptr;
    if ((rc = ocvar_hip_create(&one, 0, cfg.width, cfg.height, F)) || (rc = ocvar_hip_set_templates(one, tpl.data(), n_tpl)) ||
        (rc = ocvar_hip_set_camera(one, reinterpret_cast<OcvarCamera*>(&cam))) ||
        (rc = ocvar_hip_detect_host(one, frames.data(), cfg.width, cfg.height, cfg.width * 3, fb, F, 0, nullptr, nullptr, ref.data(), cref.data(),
                                    OCVAR_MAX_MARKERS))) {
        std::fprintf(stderr, "single-GPU reference run failed (%d): %s\n", rc, one ? ocvar_hip_last_error(one) : "");
        return 1;
    }
    long long total = 0;
    int mismatches = 0;
    for (int f = 0; f < F; f++) {
        total += cn[f];
        if (cn[f] != cref[f]) { mismatches++; continue; }
        const int k = cn[f] < OCVAR_MAX_MARKERS ? cn[f] : OCVAR_MAX_MARKERS;
        if (std::memcmp(&mk[(size_t)f * OCVAR_MAX_MARKERS], &ref[(size_t)f * OCVAR_MAX_MARKERS], (size_t)k * sizeof(OcvarMarker)) != 0) mismatches++;
    }
    // Stateful: the F frames are F video streams; three time steps (the scene of stream s at step t is frame (s + t) % F, so every
    // stream sees new quads next to tracked ones).  Each stream's markers stay on its device between the calls
    // (ocvar_multi_track_host); the check is the single-GPU entry point called per step with the previous step's markers handed
    // back by the host -- the reference's own calling pattern, samples/ARTest.cpp:57 (`markers` persists across frames).
    int track_mismatches = 0;
    {
        std::vector<unsigned char> step_frames(fb * F);
        std::vector<OcvarMarker> prev((size_t)F * OCVAR_MAX_MARKERS), tk((size_t)F * OCVAR_MAX_MARKERS), tref((size_t)F * OCVAR_MAX_MARKERS);
        std::vector<int> prev_n(F, 0), tn(F), trefn(F);
        for (int t = 0; t < 3; t++) {
            for (int s2 = 0; s2 < F; s2++) std::memcpy(step_frames.data() + fb * s2, frames.data() + fb * ((s2 + (t ? 1 : 0)) % F), fb);
            rc = ocvar_multi_track_host(m, step_frames.data(), cfg.width, cfg.height, cfg.width * 3, fb, F, t == 0, tk.data(), tn.data(), OCVAR_MAX_MARKERS);
            if (rc) {
                std::fprintf(stderr, "ocvar_multi_track_host failed (%d): %s\n", rc, ocvar_multi_last_error(m));
                return 1;
            }
            rc = ocvar_hip_detect_host(one, step_frames.data(), cfg.width, cfg.height, cfg.width * 3, fb, F, 0, t ? prev.data() : nullptr,
                                       t ? prev_n.data() : nullptr, tref.data(), trefn.data(), OCVAR_MAX_MARKERS);
            if (rc) {
                std::fprintf(stderr, "single-GPU tracked reference run failed (%d): %s\n", rc, ocvar_hip_last_error(one));
                return 1;
            }
            for (int f = 0; f < F; f++) {
                if (tn[f] != trefn[f]) { track_mismatches++; continue; }
                const int k = tn[f] < OCVAR_MAX_MARKERS ? tn[f] : OCVAR_MAX_MARKERS;
                if (std::memcmp(&tk[(size_t)f * OCVAR_MAX_MARKERS], &tref[(size_t)f * OCVAR_MAX_MARKERS], (size_t)k * sizeof(OcvarMarker)) != 0) track_mismatches++;
            }
            prev = tref;
            for (int f = 0; f < F; f++) prev_n[f] = trefn[f] < OCVAR_MAX_MARKERS ? trefn[f] : OCVAR_MAX_MARKERS;
        }
        mismatches += track_mismatches;
    }
    ocvar_hip_destroy(one);
    // narrow blocks: a caller that keeps 2 markers per frame gathers 2 records per frame; the counts stay the full counts
    {
        const int K = 2;
        std::vector<OcvarMarker> mk2((size_t)F * K);
        std::vector<int> cn2(F);
        rc = ocvar_multi_detect_host(m, frames.data(), cfg.width, cfg.height, cfg.width * 3, fb, F, mk2.data(), cn2.data(), K);
        if (rc) {
            std::fprintf(stderr, "ocvar_multi_detect_host (2 per frame) failed (%d): %s\n", rc, ocvar_multi_last_error(m));
            return 1;
        }
        for (int f = 0; f < F; f++) {
            if (cn2[f] != cref[f]) { mismatches++; continue; }
            const int k = cn2[f] < K ? cn2[f] : K;
            if (std::memcmp(&mk2[(size_t)f * K], &ref[(size_t)f * OCVAR_MAX_MARKERS], (size_t)k * sizeof(OcvarMarker)) != 0) mismatches++;
        }
    }

    // timing with the shards resident on the devices (frames d, d+N, ... on device d)
    std::vector<uint8_t*> d_bgr(N, nullptr);
    std::vector<int> n_local(N, per_dev);
    bool hip_ok = true;
    for (int d = 0; d < N; d++) {
        hip_ok &= hipSetDevice(d) == hipSuccess && hipMalloc((void**)&d_bgr[d], fb * per_dev) == hipSuccess;
        for (int i = 0; hip_ok && i < per_dev; i++)
            hip_ok &= hipMemcpy(d_bgr[d] + fb * i, frames.data() + fb * (d + (size_t)N * i), fb, hipMemcpyHostToDevice) == hipSuccess;
    }
    if (!hip_ok) {
        std::fprintf(stderr, "device staging failed\n");
        return 1;
    }
    double best = 1e30;
    for (int r = 0; r < rounds + 1; r++) {
        const auto t0 = std::chrono::steady_clock::now();
        rc = ocvar_multi_detect_device(m, d_bgr.data(), cfg.width, cfg.height, cfg.width * 3, fb, n_local.data(), mk.data(), cn.data(), OCVAR_MAX_MARKERS);
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (rc) {
            std::fprintf(stderr, "ocvar_multi_detect_device failed (%d): %s\n", rc, ocvar_multi_last_error(m));
            return 1;
        }
        if (r > 0 && dt < best) best = dt;
    }
    for (int f = 0; f < F; f++) mismatches += cn[f] != cref[f];
    for (int d = 0; d < N; d++)
        if (hipSetDevice(d) == hipSuccess) (void)hipFree(d_bgr[d]);
    ocvar_multi_destroy(m);
    std::printf("{\"devices\": %d, \"frames\": %d, \"width\": %d, \"height\": %d, \"markers_total\": %lld, \"mismatches_vs_single_gpu\": %d, \"tracked_steps\": 3, \"tracked_mismatches\": %d, "
                "\"gather\": \"ncclGather of [frames][%d] CvarMarker + counts to device 0\", \"frames_per_s_device_resident\": %.1f}\n",
                N, F, cfg.width, cfg.height, total, mismatches, track_mismatches, OCVAR_MAX_MARKERS, F / best);
    return mismatches ? 1 : 0;
}
