/*
 * ocvar_synth.h -- deterministic synthetic AR frames.
 *
 * The reference ships no test images; its only input is a webcam (samples/ARTest.cpp:122).  This
 * generator renders frames of the shape BASELINE.json's configs name, on the CPU only, so the
 * oracle and the HIP path always see identical bytes.
 */
#ifndef OCVAR_SYNTH_H
#define OCVAR_SYNTH_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int width, height;     /* frame size */
    int grid_x, grid_y;    /* markers are planted on a jittered grid_x x grid_y grid */
    int side_min, side_max;/* marker side in pixels (incl. the 1-cell black frame) */
    int rot_mode;          /* 0: {0,90,180,270} + U(-10,10) deg; 1: U[0,360) deg; 2: upright, no jitter */
    int corner_jitter_pct; /* each corner displaced U(-p,p) % of side (perspective jitter), 0 = off */
    int occlude_pct;       /* percentage of markers with one corner covered by a background rectangle */
    int textured;          /* 0: constant 220 background; 1: gradient + noise */
} OcvarSynthConfig;

typedef struct {
    double corner[8];      /* image corners of the marker's outer (black frame) quad, marker order TL,TR,BR,BL */
    int template_index;
    int quadrant;          /* rotation quadrant 0..3 drawn for this marker */
    int occluded;
    int pad;
} OcvarSynthMarker;

/* One template image: full PNG pixels incl. the 1-px frame (w x h, 8-bit, row-major, top row first). */
typedef struct { const uint8_t* pixels; int w, h; } OcvarSynthTemplate;

void ocvar_synth_config(int config_id /* 1,2,3,5 = BASELINE.json configs */, OcvarSynthConfig* out);

/* Renders frame `frame_index` (seed 0x0C0A2013 + frame_index) into bgr (8UC3, equal channels).
 * truth may be NULL; returns the number of planted markers. */
int ocvar_synth_frame(const OcvarSynthConfig* cfg, uint64_t frame_index, const OcvarSynthTemplate* templates,
                      int n_templates, uint8_t* bgr, int stride, OcvarSynthMarker* truth, int max_truth);

#ifdef __cplusplus
}
#endif
#endif
