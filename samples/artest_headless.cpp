// artest_headless.cpp -- what the reference's samples/ARTest.cpp does per displayed frame, without a webcam or
// GLUT window: load a template (cvarLoadTemplateTag), set up the camera (cvarReadCamera(NULL) + cvarCameraScale),
// flip nothing (synthetic frames are generated bottom-up already), call cvarArMultRegistration twice (the
// second call exercises the tracking stage with the first call's markers) and print the CvarMarker records.
// Links against lib/libopencv-ar.so exactly as ARTest links against the reference's library.
//
//   artest_headless <template.png> [config_id=1] [frame_index=0]
#include "opencvar/opencvar.h"
#include "opencvar/acmath.h"
#include "ocvar_synth.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

int main(int argc, char** argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s <template.png> [config_id] [frame_index]\n", argv[0]);
        return 2;
    }
    const int config_id = argc > 2 ? std::atoi(argv[2]) : 1;
    const unsigned long long frame_index = argc > 3 ? std::strtoull(argv[3], nullptr, 10) : 0;
    CvarTemplate tpl;
    if (!cvarLoadTemplateTag(&tpl, argv[1])) {
        std::fprintf(stderr, "cannot load template %s\n", argv[1]);
        return 1;
    }
    std::printf("template %dx%d codes %llx %llx %llx %llx\n", tpl.width, tpl.height, (unsigned long long)tpl.code[0],
                (unsigned long long)tpl.code[1], (unsigned long long)tpl.code[2], (unsigned long long)tpl.code[3]);
    vector<CvarTemplate> templates;
    templates.push_back(tpl);

    OcvarSynthConfig cfg;
    ocvar_synth_config(config_id, &cfg);
    // The frame generator needs the template's pixel grid (incl. its 1-px black frame).  The public API only
    // exposes the code, so the three shipped patterns (template/{2x2,3x3,4x4}-01.png) are kept here by size.
    const int gw = tpl.width + 2, gh = tpl.height + 2;
    static const unsigned char g22[16] = {0,0,0,0, 0,255,0,0, 0,0,255,0, 0,0,0,0};
    static const unsigned char g33[25] = {0,0,0,0,0, 0,255,255,255,0, 0,255,255,0,0, 0,255,0,255,0, 0,0,0,0,0};
    static const unsigned char g44[36] = {0,0,0,0,0,0, 0,255,0,255,255,0, 0,0,255,255,255,0, 0,0,255,255,255,0,
                                          0,255,0,255,255,0, 0,0,0,0,0,0};
    if (tpl.width != tpl.height || tpl.width < 2 || tpl.width > 4) {
        std::fprintf(stderr, "this sample only plants the shipped 2x2 / 3x3 / 4x4 patterns\n");
        return 1;
    }
    std::vector<unsigned char> grid(tpl.width == 2 ? g22 : tpl.width == 3 ? g33 : g44,
                                    (tpl.width == 2 ? g22 : tpl.width == 3 ? g33 : g44) + gw * gh);
    OcvarSynthTemplate st = {grid.data(), gw, gh};
    std::vector<unsigned char> bgr((size_t)cfg.width * cfg.height * 3);
    ocvar_synth_frame(&cfg, frame_index, &st, 1, bgr.data(), cfg.width * 3, nullptr, 0);

    IplImage img;
    std::memset(&img, 0, sizeof img);
    img.nSize = sizeof img;
    img.nChannels = 3;
    img.depth = IPL_DEPTH_8U;
    img.width = cfg.width;
    img.height = cfg.height;
    img.widthStep = cfg.width * 3;
    img.imageSize = img.widthStep * img.height;
    img.imageData = img.imageDataOrigin = (char*)bgr.data();

    CvarCamera camera;
    cvarReadCamera(NULL, &camera);
    cvarCameraScale(&camera, cfg.width, cfg.height);

    vector<CvarMarker> markers;
    for (int pass = 0; pass < 2; pass++) {
        const int n = cvarArMultRegistration(&img, &markers, templates, &camera);
        std::printf("pass %d: %d marker(s)\n", pass, n);
        for (size_t i = 0; i < markers.size(); i++) {
            const CvarMarker& m = markers[i];
            std::printf("  marker %zu template %d id %d score %.1f square", i, m.templateId, m.markerId, m.score);
            for (int k = 0; k < 4; k++) std::printf(" (%.0f,%.0f)", m.square[k].x, m.square[k].y);
            std::printf("\n    gl");
            for (int k = 0; k < 16; k++) std::printf(" %.6f", m.glMatrix[k]);
            std::printf("\n");
        }
    }
    // greyed in place: all three channels equal
    size_t bad = 0;
    for (size_t i = 0; i < bgr.size(); i += 3) bad += (bgr[i] != bgr[i + 1] || bgr[i] != bgr[i + 2]);
    std::printf("frame greyed in place: %s\n", bad ? "NO" : "yes");
    return 0;
}
