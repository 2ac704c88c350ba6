// artest_latency.cpp -- latency of ONE call of cvarArMultRegistration through lib/libopencv-ar.so, the way the reference
// is used (one call per displayed frame, /root/reference/samples/ARTest.cpp:57, /root/reference/src/opencvar.cpp:619):
// frame in host memory -> markers in host memory, frame greyed in place.  Prints one JSON line.
//
//   artest_latency <template_dir> [config_id=3] [calls=60]
//
// config 3 = 1920x1080 with 16 planted markers and the three shipped templates (BASELINE.json configs[2]); config 2 =
// 640x480 with 4 markers and the 2x2 template.  Every call gets a fresh copy of the frame (the call greys it) and an
// empty marker vector (stateless); the first calls (context creation, first launches) are reported separately.
#include "opencvar/opencvar.h"
#include "ocvar_synth.h"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

int main(int argc, char** argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s <template_dir> [config_id] [calls]\n", argv[0]);
        return 2;
    }
    const std::string dir = argv[1];
    const int config_id = argc > 2 ? std::atoi(argv[2]) : 3;
    const int calls = argc > 3 ? std::max(3, std::atoi(argv[3])) : 60;
    static const unsigned char g22[16] = {0,0,0,0, 0,255,0,0, 0,0,255,0, 0,0,0,0};
    static const unsigned char g33[25] = {0,0,0,0,0, 0,255,255,255,0, 0,255,255,0,0, 0,255,0,255,0, 0,0,0,0,0};
    static const unsigned char g44[36] = {0,0,0,0,0,0, 0,255,0,255,255,0, 0,0,255,255,255,0, 0,0,255,255,255,0,
                                          0,255,0,255,255,0, 0,0,0,0,0,0};
    const char* names[3] = {"2x2-01.png", "3x3-01.png", "4x4-01.png"};
    const OcvarSynthTemplate grids[3] = {{g22, 4, 4}, {g33, 5, 5}, {g44, 6, 6}};
    const int n_tpl = config_id <= 2 ? 1 : 3;
    vector<CvarTemplate> templates;
    for (int i = 0; i < n_tpl; i++) {
        CvarTemplate t;
        if (!cvarLoadTemplateTag(&t, (dir + "/" + names[i]).c_str())) {
            std::fprintf(stderr, "cannot load %s/%s\n", dir.c_str(), names[i]);
            return 1;
        }
        templates.push_back(t);
    }
    OcvarSynthConfig cfg;
    ocvar_synth_config(config_id, &cfg);
    std::vector<unsigned char> frame((size_t)cfg.width * cfg.height * 3), work(frame.size());
    const int planted = ocvar_synth_frame(&cfg, 0, grids, n_tpl, frame.data(), cfg.width * 3, nullptr, 0);
    IplImage img;
    std::memset(&img, 0, sizeof img);
    img.nSize = sizeof img;
    img.nChannels = 3;
    img.depth = IPL_DEPTH_8U;
    img.width = cfg.width;
    img.height = cfg.height;
    img.widthStep = cfg.width * 3;
    img.imageSize = img.widthStep * img.height;
    img.imageData = img.imageDataOrigin = (char*)work.data();
    CvarCamera camera;
    cvarReadCamera(NULL, &camera);
    cvarCameraScale(&camera, cfg.width, cfg.height);

    std::vector<double> ms;
    vector<CvarMarker> markers;
    int n_markers = 0;
    double first_ms = 0;
    for (int k = 0; k < calls + 3; k++) {
        std::memcpy(work.data(), frame.data(), frame.size());
        markers.clear();
        const auto t0 = std::chrono::steady_clock::now();
        n_markers = cvarArMultRegistration(&img, &markers, templates, &camera);
        const double dt = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (k == 0) first_ms = dt;
        if (k >= 3) ms.push_back(dt);   // the first calls create the context and load the code objects
    }
    std::sort(ms.begin(), ms.end());
    std::printf("{\"entry\": \"cvarArMultRegistration (libopencv-ar.so), host frame in, markers out, frame greyed in place\", "
                "\"width\": %d, \"height\": %d, \"planted_markers\": %d, \"templates\": %d, \"markers_out\": %d, \"calls\": %d, "
                "\"median_ms\": %.4f, \"min_ms\": %.4f, \"p90_ms\": %.4f, \"first_call_ms\": %.2f}\n",
                cfg.width, cfg.height, planted, n_tpl, n_markers, (int)ms.size(), ms[ms.size() / 2], ms.front(),
                ms[(ms.size() * 9) / 10], first_ms);
    return n_markers > 0 ? 0 : 1;
}
