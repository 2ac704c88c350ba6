// synth.cpp -- deterministic synthetic frames for the BASELINE.json configs (see include/ocvar_synth.h).
//
// Frames mimic what samples/ARTest.cpp:44-45 hands to cvarArMultRegistration: an 8UC3 BGR image,
// bottom-up (so a marker appears as the vertical flip of its template PNG), here with
// equal channels.  CPU only; both the oracle and the HIP path consume the bytes produced here.
#include "ocvar_synth.h"
#include <cmath>
#include <cstring>
#include <vector>

namespace {

struct Rng {
    uint64_t s;
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    double range(double a, double b) { return a + (b - a) * uni(); }
};

inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// 3x3 homography unit square -> quad, and its inverse
bool square_to_quad(const double* q, double* H) {
    // closed form (Heckbert): maps (0,0),(1,0),(1,1),(0,1) to q0..q3
    double x0 = q[0], y0 = q[1], x1 = q[2], y1 = q[3], x2 = q[4], y2 = q[5], x3 = q[6], y3 = q[7];
    double dx1 = x1 - x2, dx2 = x3 - x2, dx3 = x0 - x1 + x2 - x3;
    double dy1 = y1 - y2, dy2 = y3 - y2, dy3 = y0 - y1 + y2 - y3;
    double den = dx1 * dy2 - dx2 * dy1;
    if (den == 0) return false;
    double g = (dx3 * dy2 - dx2 * dy3) / den, h = (dx1 * dy3 - dx3 * dy1) / den;
    H[0] = x1 - x0 + g * x1;
    H[1] = x3 - x0 + h * x3;
    H[2] = x0;
    H[3] = y1 - y0 + g * y1;
    H[4] = y3 - y0 + h * y3;
    H[5] = y0;
    H[6] = g;
    H[7] = h;
    H[8] = 1;
    return true;
}

bool invert3(const double* S, double* M) {
    double d = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
    if (d == 0) return false;
    d = 1 / d;
    M[0] = (S[4] * S[8] - S[5] * S[7]) * d;
    M[1] = (S[2] * S[7] - S[1] * S[8]) * d;
    M[2] = (S[1] * S[5] - S[2] * S[4]) * d;
    M[3] = (S[5] * S[6] - S[3] * S[8]) * d;
    M[4] = (S[0] * S[8] - S[2] * S[6]) * d;
    M[5] = (S[2] * S[3] - S[0] * S[5]) * d;
    M[6] = (S[3] * S[7] - S[4] * S[6]) * d;
    M[7] = (S[1] * S[6] - S[0] * S[7]) * d;
    M[8] = (S[0] * S[4] - S[1] * S[3]) * d;
    return true;
}

struct MarkerDraw {
    double Hinv[9];
    const OcvarSynthTemplate* tpl;
    double quiet;      // quiet-zone width in marker units (0 = none)
    int occl_corner;   // -1 none, else 0..3
    double occl_size;  // side of the occluding square in marker units
};

inline int background(const OcvarSynthConfig* c, uint64_t seed, int x, int y) {
    if (!c->textured) return 220;
    long long W = c->width, H = c->height;
    long long num = 48 * ((long long)x * H + (long long)y * W - W * H);
    long long g = num >= 0 ? num / (W * H) : -((-num + W * H - 1) / (W * H));
    int noise = (int)(mix64(seed ^ ((uint64_t)((long long)y * W + x) * 0x9E3779B97F4A7C15ull)) % 25) - 12;
    int v = 128 + (int)g + noise;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// sample class: -1 = background, else 0..255 value
inline int sample(const MarkerDraw& m, double px, double py) {
    double w = m.Hinv[6] * px + m.Hinv[7] * py + m.Hinv[8];
    double u = (m.Hinv[0] * px + m.Hinv[1] * py + m.Hinv[2]) / w;
    double v = (m.Hinv[3] * px + m.Hinv[4] * py + m.Hinv[5]) / w;
    if (u < -m.quiet || v < -m.quiet || u >= 1 + m.quiet || v >= 1 + m.quiet) return -1;
    if (u < 0 || v < 0 || u >= 1 || v >= 1) return 255;
    if (m.occl_corner >= 0) {
        double cu = (m.occl_corner == 1 || m.occl_corner == 2) ? 1 - u : u;
        double cv = (m.occl_corner >= 2) ? 1 - v : v;
        if (cu < m.occl_size && cv < m.occl_size) return -1;
    }
    int gw = m.tpl->w, gh = m.tpl->h;
    int cx = (int)(u * gw), cy = (int)(v * gh);
    if (cx >= gw) cx = gw - 1;
    if (cy >= gh) cy = gh - 1;
    // marker = vertical flip of the template image
    return m.tpl->pixels[(size_t)(gh - 1 - cy) * gw + cx] > 100 ? 255 : 0;
}

}  // namespace

extern "C" void ocvar_synth_config(int id, OcvarSynthConfig* c) {
    memset(c, 0, sizeof *c);
    switch (id) {
    case 1:
        *c = OcvarSynthConfig{640, 480, 1, 1, 120, 120, 2, 0, 0, 0};
        break;
    case 2:
        *c = OcvarSynthConfig{640, 480, 2, 2, 96, 144, 0, 0, 0, 0};
        break;
    case 5:
        *c = OcvarSynthConfig{3840, 2160, 8, 8, 120, 170, 1, 8, 20, 0};
        break;
    case 3:
    default:
        *c = OcvarSynthConfig{1920, 1080, 4, 4, 120, 170, 1, 0, 0, 0};
        break;
    }
}

extern "C" int ocvar_synth_frame(const OcvarSynthConfig* cfg, uint64_t frame_index, const OcvarSynthTemplate* templates,
                                 int n_templates, uint8_t* bgr, int stride, OcvarSynthMarker* truth, int max_truth) {
    const int W = cfg->width, H = cfg->height;
    uint64_t seed = 0x0C0A2013ull + frame_index;
    Rng rng{seed};
    for (int y = 0; y < H; y++) {
        uint8_t* row = bgr + (size_t)y * stride;
        for (int x = 0; x < W; x++) {
            uint8_t v = (uint8_t)background(cfg, seed, x, y);
            row[3 * x] = row[3 * x + 1] = row[3 * x + 2] = v;
        }
    }
    if (n_templates <= 0) return 0;
    const double PI = 3.14159265358979323846;
    const double quiet_px = 12.0;
    int cw = W / cfg->grid_x, ch = H / cfg->grid_y;
    int n = 0;
    for (int gy = 0; gy < cfg->grid_y; gy++)
        for (int gx = 0; gx < cfg->grid_x; gx++, n++) {
            int ti = (int)((n + frame_index) % (uint64_t)n_templates);
            double side = rng.range(cfg->side_min, cfg->side_max + 1e-9);
            int quadrant;
            double theta;
            if (cfg->rot_mode == 0) {
                quadrant = (int)(rng.next() % 4);
                theta = 90.0 * quadrant + rng.range(-10, 10);
            } else if (cfg->rot_mode == 1) {
                theta = rng.range(0, 360);
                quadrant = ((int)floor(theta / 90.0)) & 3;
            } else {
                theta = 0;
                quadrant = 0;
            }
            double jit[8];
            for (int k = 0; k < 8; k++) jit[k] = cfg->corner_jitter_pct ? rng.range(-1, 1) * cfg->corner_jitter_pct / 100.0 : 0.0;
            bool occl = cfg->occlude_pct > 0 && (int)(rng.next() % 100) < cfg->occlude_pct;
            int occl_corner = (int)(rng.next() % 4);
            double occl_frac = rng.range(0.15, 0.30);
            double jx = rng.uni(), jy = rng.uni();
            // corners relative to the centre; shrink until the quiet-zone bbox fits the grid cell
            double rel[8], hx = 0, hy = 0;
            for (int it = 0; it < 32; it++) {
                double c = cos(theta * PI / 180), s = sin(theta * PI / 180);
                const double base[8] = {-0.5, -0.5, 0.5, -0.5, 0.5, 0.5, -0.5, 0.5};
                hx = hy = 0;
                for (int k = 0; k < 4; k++) {
                    double u = (base[2 * k] + jit[2 * k]) * side, v = (base[2 * k + 1] + jit[2 * k + 1]) * side;
                    rel[2 * k] = c * u - s * v;
                    rel[2 * k + 1] = s * u + c * v;
                    hx = fmax(hx, fabs(rel[2 * k]));
                    hy = fmax(hy, fabs(rel[2 * k + 1]));
                }
                hx += quiet_px + 2;
                hy += quiet_px + 2;
                if (hx <= cw / 2.0 && hy <= ch / 2.0) break;
                side *= 0.95;
            }
            double slx = fmax(0.0, cw / 2.0 - hx), sly = fmax(0.0, ch / 2.0 - hy);
            double cx = gx * cw + cw / 2.0 + (2 * jx - 1) * slx, cy = gy * ch + ch / 2.0 + (2 * jy - 1) * sly;
            double quad[8];
            for (int k = 0; k < 4; k++) {
                quad[2 * k] = cx + rel[2 * k];
                quad[2 * k + 1] = cy + rel[2 * k + 1];
            }
            MarkerDraw md;
            double Hm[9];
            if (!square_to_quad(quad, Hm) || !invert3(Hm, md.Hinv)) continue;
            md.tpl = &templates[ti];
            md.quiet = cfg->textured ? quiet_px / side : 0.0;
            md.occl_corner = occl ? occl_corner : -1;
            md.occl_size = sqrt(occl_frac);
            int bx0 = (int)floor(cx - hx), bx1 = (int)ceil(cx + hx), by0 = (int)floor(cy - hy), by1 = (int)ceil(cy + hy);
            if (bx0 < 0) bx0 = 0;
            if (by0 < 0) by0 = 0;
            if (bx1 > W) bx1 = W;
            if (by1 > H) by1 = H;
            for (int y = by0; y < by1; y++) {
                uint8_t* row = bgr + (size_t)y * stride;
                for (int x = bx0; x < bx1; x++) {
                    int c00 = sample(md, x + 0.0625, y + 0.0625), c10 = sample(md, x + 0.9375, y + 0.0625);
                    int c01 = sample(md, x + 0.0625, y + 0.9375), c11 = sample(md, x + 0.9375, y + 0.9375);
                    int bg = row[3 * x];
                    int val;
                    if (c00 == c10 && c00 == c01 && c00 == c11) {
                        val = c00 < 0 ? bg : c00;
                    } else {
                        int sum = 0;
                        for (int sy = 0; sy < 4; sy++)
                            for (int sx = 0; sx < 4; sx++) {
                                int c = sample(md, x + (sx + 0.5) * 0.25, y + (sy + 0.5) * 0.25);
                                sum += c < 0 ? bg : c;
                            }
                        val = (sum + 8) >> 4;
                    }
                    row[3 * x] = row[3 * x + 1] = row[3 * x + 2] = (uint8_t)val;
                }
            }
            if (truth && n < max_truth) {
                memcpy(truth[n].corner, quad, sizeof quad);
                truth[n].template_index = ti;
                truth[n].quadrant = quadrant;
                truth[n].occluded = occl;
                truth[n].pad = 0;
            }
        }
    return n;
}
