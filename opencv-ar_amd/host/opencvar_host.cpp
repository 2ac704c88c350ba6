// opencvar_host.cpp -- host C++ mirror of the reference's public API (include/opencvar/opencvar.h), same
// names, argument meaning and error behaviour as /root/reference/src/opencvar.cpp, on top of the C ABI of
// include/ocvar_hip.h.  cvarArMultRegistration (reference 619-807) and cvarFindSquares (156-223) run on the
// MI355X; everything else here is setup-side or tiny per-quad host arithmetic.
//
// Deliberate differences: no exception ever crosses the C boundary (failures return
// 0 / an empty result and print one line to stderr); the reference's leaks (D12) are not reproduced;
// cvarDrawSquares draws aliased 1-px lines instead of anti-aliased ones (display aid).
#include "opencvar/opencvar.h"
#include "opencvar/acmath.h"
#include "ocvar_hip.h"
#include "../csrc/decode_core.h"
#include "../csrc/pose_core.h"
#include "../csrc/tail_core.h"

#include <zlib.h>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <map>
#include <string>

static_assert(sizeof(CvarCamera) == sizeof(OcvarCamera) && sizeof(CvarCamera) == 248, "CvarCamera layout");
static_assert(sizeof(CvarTemplate) == sizeof(OcvarTemplate) && sizeof(CvarTemplate) == 48, "CvarTemplate layout");
static_assert(sizeof(CvarMarker) == sizeof(OcvarMarker) && sizeof(CvarMarker) == 184, "CvarMarker layout");

static_assert(offsetof(CvSeq, total) == 40 && offsetof(CvSeq, elem_size) == 44 && offsetof(CvSeq, first) == 88 && sizeof(CvSeq) == 96,
              "CvSeq layout of OpenCV 2.x/3.x (types_c.h)");

namespace {

// The sequence cvarFindSquares hands out: a CvSeq header with OpenCV's layout over one element block owned by this
// library (the reference allocates it in the caller's CvMemStorage; a caller only reads it, through `total`, cvGetSeqElem
// or the cvarGet* accessors below).  Only public CvSeq fields are used, so this compiles against real OpenCV headers too.
struct OwnedSeq {
    CvSeq seq;
    CvSeqBlock block;
    std::vector<CvPoint> pts;
    void seal() {
        std::memset(&seq, 0, sizeof seq);
        std::memset(&block, 0, sizeof block);
        seq.flags = 0x42990000 | 12;   // CV_SEQ_MAGIC_VAL | CV_SEQ_ELTYPE_POINT (CV_32SC2)
        seq.header_size = (int)sizeof(CvSeq);
        seq.total = (int)pts.size();
        seq.elem_size = (int)sizeof(CvPoint);
        block.prev = block.next = &block;
        block.count = seq.total;
        block.data = reinterpret_cast<signed char*>(pts.data());
        seq.first = pts.empty() ? nullptr : &block;
        seq.block_max = seq.ptr = block.data + sizeof(CvPoint) * pts.size();
    }
};

// element i of a point sequence, walking the blocks as cvGetSeqElem does (works on OpenCV-made sequences as well)
const CvPoint* seq_point(const CvSeq* seq, int i) {
    const CvSeqBlock* b = seq->first;
    while (b && i >= b->count) {
        i -= b->count;
        b = b->next;
        if (b == seq->first) return nullptr;
    }
    return b ? reinterpret_cast<const CvPoint*>(b->data + (size_t)i * seq->elem_size) : nullptr;
}

std::mutex g_mu;
OcvarHip* g_ctx = nullptr;
int g_w = 0, g_h = 0;
int g_maxq = OCVAR_MAX_QUADS;   // squares per frame the current context has room for (grows when a frame needs more)
// Sequences handed out by cvarFindSquares, per CvMemStorage* the caller passed.  In the reference a sequence lives in the
// caller's storage until that storage is cleared or released (opencvar.cpp:168, 621-630, 803-804).  OpenCV is not part of
// this build, so the storage is an opaque key here: a key's sequences stay alive until cvarReleaseSquares(key) -- the
// counterpart of cvClearMemStorage / cvReleaseMemStorage for this library -- or until the key has collected more than
// SEQS_PER_STORAGE of them, when the oldest goes (one registration of the reference creates 1 + quads x templates sequences in
// its storage: hundreds at most).  Different storages never evict each other's sequences.
constexpr size_t SEQS_PER_STORAGE = 4096;
std::map<const void*, std::deque<std::unique_ptr<OwnedSeq>>> g_seqs;
std::vector<CvarTemplate> g_templates;        // what the context currently holds: unchanged arguments are not uploaded again
CvarCamera g_camera;
bool g_have_camera = false;

OcvarHip* context_for(int w, int h, int maxq = 0) {
    if (maxq <= 0) maxq = g_maxq;
    if (g_ctx && w <= g_w && h <= g_h && maxq <= g_maxq) return g_ctx;
    if (g_ctx) {   // keep what the old context had room for
        w = w > g_w ? w : g_w;
        h = h > g_h ? h : g_h;
    }
    if (g_ctx) ocvar_hip_destroy(g_ctx);
    g_ctx = nullptr;
    g_templates.clear();
    g_have_camera = false;
    const int dev = std::getenv("OCVAR_DEVICE") ? std::atoi(std::getenv("OCVAR_DEVICE")) : 0;
    int rc = ocvar_hip_create_ex(&g_ctx, dev, w < 64 ? 64 : w, h < 64 ? 64 : h, 1, maxq);
    if (rc != OCVAR_OK) {
        std::fprintf(stderr, "opencvar: no MI355X context (%d): %s\n", rc, g_ctx ? ocvar_hip_last_error(g_ctx) : "no gfx950 device");
        if (g_ctx) ocvar_hip_destroy(g_ctx);
        g_ctx = nullptr;
        return nullptr;
    }
    g_w = w < 64 ? 64 : w;
    g_h = h < 64 ? 64 : h;
    g_maxq = maxq;
    return g_ctx;
}

// ---- minimal PNG reader (8-bit grey / grey+alpha / RGB / RGBA / palette, non-interlaced) over zlib ----
uint32_t be32(const unsigned char* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

bool read_png_gray(const char* path, std::vector<unsigned char>& gray, int& w, int& h) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return false;
    std::vector<unsigned char> file;
    unsigned char buf[4096];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + n);
    std::fclose(f);
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (file.size() < 33 || std::memcmp(file.data(), sig, 8) != 0) return false;
    size_t pos = 8;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<unsigned char> idat, plte;
    w = h = 0;
    while (pos + 12 <= file.size()) {
        const uint32_t len = be32(&file[pos]);
        const char* type = (const char*)&file[pos + 4];
        if (pos + 12 + len > file.size()) return false;
        const unsigned char* data = &file[pos + 8];
        if (!std::memcmp(type, "IHDR", 4) && len >= 13) {
            w = (int)be32(data);
            h = (int)be32(data + 4);
            depth = data[8];
            ctype = data[9];
            interlace = data[12];
        } else if (!std::memcmp(type, "PLTE", 4)) {
            plte.assign(data, data + len);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + len;
    }
    if (w <= 0 || h <= 0 || w > 16384 || h > 16384 || depth != 8 || interlace != 0) return false;
    const int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!ch) return false;
    const size_t stride = (size_t)w * ch;
    std::vector<unsigned char> raw((stride + 1) * h);
    uLongf out_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw.size()) return false;
    std::vector<unsigned char> img(stride * h);
    for (int y = 0; y < h; y++) {
        const unsigned char* in = &raw[(stride + 1) * y];
        const int ft = in[0];
        unsigned char* cur = &img[stride * y];
        const unsigned char* up = y ? &img[stride * (y - 1)] : nullptr;
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= (size_t)ch ? cur[i - ch] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)ch) ? up[i - ch] : 0;
            int pred = 0;
            if (ft == 1) pred = a;
            else if (ft == 2) pred = b;
            else if (ft == 3) pred = (a + b) >> 1;
            else if (ft == 4) {
                const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            } else if (ft != 0) return false;
            cur[i] = (unsigned char)(in[1 + i] + pred);
        }
    }
    gray.resize((size_t)w * h);
    for (size_t i = 0; i < (size_t)w * h; i++) {
        const unsigned char* p = &img[i * ch];
        int r, g, b;
        if (ctype == 0 || ctype == 4) r = g = b = p[0];
        else if (ctype == 3) {
            if ((size_t)p[0] * 3 + 2 >= plte.size()) return false;
            r = plte[p[0] * 3]; g = plte[p[0] * 3 + 1]; b = plte[p[0] * 3 + 2];
        } else { r = p[0]; g = p[1]; b = p[2]; }
        // cvLoadImage(GRAYSCALE) of a colour PNG: same fixed-point luma as BGR2GRAY
        gray[i] = (unsigned char)((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14);
    }
    return true;
}

// ---- camera file (opencvar.cpp:53-71) ---------------------------------------------------------------------------
// The reference reads an OpenCV FileStorage document: `imageSize` (sequence of two ints), `cameraMatrix` and
// `distCoeffs` (!!opencv-matrix maps with rows/cols/dt/data).  This reader understands the YAML 1.0 subset that
// cv::FileStorage writes for those three nodes (what OpenCV's calibration sample produces) and, for XML files,
// the equivalent <imageSize>, <cameraMatrix><data>, <distCoeffs><data> elements.
bool slurp(const char* path, std::string& out) {
    FILE* fp = std::fopen(path, "rb");
    if (!fp) return false;
    char buf[4096];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, fp)) > 0) out.append(buf, n);
    std::fclose(fp);
    return true;
}

// numbers of the bracketed / element list that follows position `at`, up to `closer`
bool numbers_until(const std::string& s, size_t at, char closer, std::vector<double>& out) {
    const size_t end = s.find(closer, at);
    if (end == std::string::npos) return false;
    size_t i = at;
    while (i < end) {
        if (std::isdigit((unsigned char)s[i]) || s[i] == '-' || s[i] == '+' || s[i] == '.') {
            char* stop = nullptr;
            const double v = std::strtod(s.c_str() + i, &stop);
            if (stop == s.c_str() + i) { i++; continue; }
            out.push_back(v);
            i = (size_t)(stop - s.c_str());
        } else {
            i++;
        }
    }
    return true;
}

// position just behind a top-level YAML key ("name:" at the start of a line) or an XML opening tag ("<name")
size_t find_node(const std::string& s, const char* name, bool xml) {
    const std::string key = xml ? std::string("<") + name : std::string(name) + ":";
    size_t at = 0;
    while ((at = s.find(key, at)) != std::string::npos) {
        if (xml || at == 0 || s[at - 1] == '\n') return at + key.size();
        at += key.size();
    }
    return std::string::npos;
}

bool read_node(const std::string& s, const char* name, bool xml, bool matrix, std::vector<double>& out) {
    size_t at = find_node(s, name, xml);
    if (at == std::string::npos) return false;
    if (xml) {
        if (matrix) {
            at = s.find("<data>", at);
            if (at == std::string::npos) return false;
            at += 6;
        } else {
            at = s.find('>', at);
            if (at == std::string::npos) return false;
            at++;
        }
        return numbers_until(s, at, '<', out);
    }
    if (matrix) {
        at = s.find("data:", at);
        if (at == std::string::npos) return false;
    }
    at = s.find('[', at);
    if (at == std::string::npos) return false;
    return numbers_until(s, at + 1, ']', out);
}

bool read_camera_file(const char* path, CvarCamera* c) {
    std::string s;
    if (!slurp(path, s)) return false;
    const bool xml = s.compare(0, 5, "<?xml") == 0;
    std::vector<double> size, K, d;
    if (!read_node(s, "imageSize", xml, false, size) || size.size() < 2) return false;
    if (!read_node(s, "cameraMatrix", xml, true, K) || K.size() < 9) return false;
    if (!read_node(s, "distCoeffs", xml, true, d) || d.empty()) return false;
    c->width = (int)size[0];
    c->height = (int)size[1];
    for (int i = 0; i < 9; i++) c->cameraMatrix[i] = K[i];
    for (int i = 0; i < 5; i++) c->distCoeffs[i] = i < (int)d.size() ? d[i] : 0.0;   // the reference copies 5 regardless
    return c->width > 0 && c->height > 0;
}

bool image_ok(const IplImage* img) {
    return img && img->imageData && img->depth == IPL_DEPTH_8U && img->nChannels == 3 && img->width >= 16 && img->height >= 16 &&
           img->widthStep >= 3 * img->width;
}

}  // namespace

extern "C" {

void cvarCameraProjection(CvarCamera* c, double* p, int glstyle) {
    const double nearplane = 0.1, farplane = 5000.0;  // opencvar.cpp:111-112
    std::memset(p, 0, sizeof(double) * 16);
    p[0] = 2. * c->cameraMatrix[0] / c->width;
    p[5] = 2. * c->cameraMatrix[4] / c->height;
    p[2] = 2. * (c->cameraMatrix[2] / c->width) - 1.;
    p[6] = 2. * (c->cameraMatrix[5] / c->height) - 1.;
    p[10] = -(farplane + nearplane) / (farplane - nearplane);
    p[11] = -2. * farplane * nearplane / (farplane - nearplane);
    p[14] = -1;
    if (glstyle) acMatrixTranspose(p);
}

int cvarReadCamera(const char* filename, CvarCamera* c) {
    if (filename) {
        if (!read_camera_file(filename, c)) return 0;   // opencvar.cpp:54-55: a file that cannot be opened returns 0
    } else {
        c->width = 640;
        c->height = 480;
        const double K[9] = {500, 0, c->width / 2.0, 0, 500, c->height / 2.0, 0, 0, 1};
        std::memcpy(c->cameraMatrix, K, sizeof K);
        std::memset(c->distCoeffs, 0, sizeof c->distCoeffs);
    }
    cvarCameraProjection(c, c->glProjection, 0);
    acMatrixTranspose(c->glProjection);
    return 1;
}

void cvarCameraScale(CvarCamera* c, int width, int height) {
    const double ru = (double)width / c->width, rv = (double)height / c->height;
    c->cameraMatrix[0] *= ru;
    c->cameraMatrix[4] *= rv;
    c->cameraMatrix[2] *= ru;
    c->cameraMatrix[5] *= rv;
    c->width = width;
    c->height = height;
    cvarCameraProjection(c, c->glProjection, 0);
    acMatrixTranspose(c->glProjection);
}

void cvarGlMatrix(double* modelview, CvMat* rotate3, CvMat* translate) {
    ocvar::gl_from_pose(rotate3->data.db, translate->data.db, modelview);
}

void cvarSquareInit(CvMat* mat, double ratio) {
    const double v[12] = {-ratio, -1, 0, ratio, -1, 0, ratio, 1, 0, -ratio, 1, 0};
    std::memcpy(mat->data.db, v, sizeof v);
}

void cvarReverseSquare(CvPoint2D32f sq[4]) {
    const CvPoint2D32f t = sq[1];
    sq[1] = sq[3];
    sq[3] = t;
}

void cvarSquareToMatrix(CvPoint2D32f* points, CvarCamera* cam, double* modelview, double ratio) {
    ocvar::CameraRec c;
    std::memcpy(&c, cam, sizeof c);
    ocvar::square_to_glmatrix(&points[0].x, c, ratio, modelview);
}

void cvarFindCamera(CvarCamera* cam, CvMat* objPts, CvMat* imgPts, double* modelview) {
    // The reference hands arbitrary object points to cvFindExtrinsicCameraParams2; this library's solver is the planar
    // one for the marker rectangle (+-ratio, +-1, 0) that cvarSquareInit produces -- the only use in the reference
    // (opencvar.cpp:529-536).  Anything else is refused loudly instead of being solved as if it were that rectangle.
    if (!cam || !objPts || !imgPts || !modelview || !objPts->data.db || !imgPts->data.db) return;
    const double* o = objPts->data.db;
    const double ratio = std::fabs(o[0]);
    const double want[12] = {-ratio, -1, 0, ratio, -1, 0, ratio, 1, 0, -ratio, 1, 0};
    for (int i = 0; i < 12; i++)
        if (o[i] != want[i]) {
            std::fprintf(stderr, "opencvar: cvarFindCamera: object points are not the marker rectangle of cvarSquareInit; not solved\n");
            return;
        }
    CvPoint2D32f p[4];
    for (int i = 0; i < 4; i++) {
        p[i].x = (float)imgPts->data.db[2 * i];
        p[i].y = (float)imgPts->data.db[2 * i + 1];
    }
    cvarSquareToMatrix(p, cam, modelview, ratio);
}

void cvarLoadTag(CvarTemplate* tpl, long long int bit, int width, int height, double scale) {
    tpl->width = width;
    tpl->height = height;
    tpl->scale = scale;
    for (int i = 0; i < 4; i++) {
        tpl->code[i] = bit;
        acBitRotate(&tpl->code[i], i, width, height);
    }
}

int cvarLoadTemplateTag(CvarTemplate* tpl, const char* filename, double scale) {
    std::vector<unsigned char> g;
    int w = 0, h = 0;
    if (!filename || !read_png_gray(filename, g, w, h) || w < 3 || h < 3 || (w - 2) * (h - 2) > 64) return 0;
    // opencvar.cpp:291-301: inner (w-2)x(h-2) -> 8UC1 image (widthStep = align4) -> >100 -> flip vertically ->
    // acArray2DToBit reading with stride = width (the reference's stride quirk; padding bytes read as 0)
    const int iw = w - 2, ih = h - 2, ws = (iw + 3) & ~3;
    std::vector<unsigned char> buf((size_t)ws * ih + (size_t)iw * ih, 0);
    for (int y = 0; y < ih; y++)
        for (int x = 0; x < iw; x++) buf[(size_t)y * ws + x] = g[(size_t)(ih - 1 - y + 1) * w + x + 1] > 100 ? 1 : 0;
    long long bit = 0;
    acArray2DToBit(buf.data(), iw, ih, &bit);
    cvarLoadTag(tpl, bit, iw, ih, scale);
    return 1;
}

void cvarSquare(CvPoint2D32f* s, int width, int height, int ccw) {
    const float W = (float)(width - 1), H = (float)(height - 1);
    s[0].x = 0; s[0].y = 0;
    s[2].x = W; s[2].y = H;
    if (ccw) { s[1].x = 0; s[1].y = H; s[3].x = W; s[3].y = 0; }
    else     { s[1].x = W; s[1].y = 0; s[3].x = 0; s[3].y = H; }
}

void cvarRotSquare(CvPoint2D32f* src, int rot) { ocvar::rot_square(&src[0].x, rot); }

CvRect cvarSquare2Rect(CvPoint2D32f pt[4]) {
    int x = 50000, y = 50000, x2 = -50000, y2 = -50000;
    for (int i = 0; i < 4; i++) {
        if (pt[i].x < x) x = (int)pt[i].x;
        if (pt[i].x > x2) x2 = (int)pt[i].x;
        if (pt[i].y < y) y = (int)pt[i].y;
        if (pt[i].y > y2) y2 = (int)pt[i].y;
    }
    CvRect r = {x, y, x2 - x, y2 - y};
    return r;
}

int cvarTrack(CvPoint2D32f pt1[4], CvPoint2D32f pt2[4]) { return ocvar::track_square(&pt1[0].x, &pt2[0].x); }

void cvarReleaseSquares(CvMemStorage* storage) {   // extension (not in the reference's header): see g_seqs
    std::lock_guard<std::mutex> lock(g_mu);
    g_seqs.erase(storage);
}

CvSeq* cvarFindSquares(IplImage* img, CvMemStorage* storage) {
    std::lock_guard<std::mutex> lk(g_mu);
    std::unique_ptr<OwnedSeq> seq(new OwnedSeq());
    if (image_ok(img)) {
        OcvarHip* ctx = context_for(img->width, img->height);
        if (ctx) {
            // The hot path calls this on images whose three channels are equal (after the in-place grey), where the grey
            // value is the channel value.  A colour image is greyed with the BGR2GRAY weights first; the reference filters
            // the three channels separately and greys afterwards (opencvar.cpp:175-180), which can differ by rounding --
            // said once on stderr.
            std::vector<unsigned char> gray((size_t)img->width * img->height);
            bool colour = false;
            for (int y = 0; y < img->height; y++) {
                const unsigned char* row = (const unsigned char*)img->imageData + (size_t)y * img->widthStep;
                for (int x = 0; x < img->width; x++) {
                    const unsigned b = row[3 * x], g = row[3 * x + 1], r = row[3 * x + 2];
                    colour |= b != g || g != r;
                    gray[(size_t)y * img->width + x] = (unsigned char)((b * 1868u + g * 9617u + r * 4899u + 8192u) >> 14);
                }
            }
            static bool told = false;
            if (colour && !told) {
                told = true;
                std::fprintf(stderr, "opencvar: cvarFindSquares: colour image greyed before the pyramid filter (the reference filters per channel)\n");
            }
            // the reference's square list is unbounded (opencvar.cpp:187-214): an image with more squares than the context has
            // room for is run again on a larger one (256 -> 1024 -> OCVAR_MAX_QUADS_EX); beyond that the failure is reported
            std::vector<int> quads;
            int n = 0, rc = OCVAR_OK;
            for (;;) {
                quads.assign((size_t)g_maxq * 8, 0);
                rc = ocvar_hip_find_squares(ctx, gray.data(), img->width, img->height, img->width, quads.data(), g_maxq, &n);
                if (rc != OCVAR_E_CAPACITY || !(ocvar_hip_capacity_flags(ctx) & 4) || g_maxq >= OCVAR_MAX_QUADS_EX) break;
                ctx = context_for(img->width, img->height, g_maxq * 4 < OCVAR_MAX_QUADS_EX ? g_maxq * 4 : OCVAR_MAX_QUADS_EX);
                if (!ctx) break;
            }
            if (!ctx) rc = OCVAR_E_HIP;
            if (rc == OCVAR_OK) {
                for (int i = 0; i < 4 * n; i++) seq->pts.push_back(CvPoint{quads[2 * i], quads[2 * i + 1]});
            } else {
                std::fprintf(stderr, "opencvar: cvarFindSquares failed: %s\n", ctx ? ocvar_hip_last_error(ctx) : "no context");
            }
        }
    }
    seq->seal();
    seq->seq.storage = storage;
    auto& mine = g_seqs[storage];
    mine.push_back(std::move(seq));
    if (mine.size() > SEQS_PER_STORAGE) mine.pop_front();
    return &mine.back()->seq;
}

int cvarGetSquare(CvSeq* squares, CvPoint2D32f* points) {
    int res = 0;
    for (int i = 0; squares && i + 3 < squares->total; i += 4, res++)
        for (int j = 0; j < 4; j++) {
            const CvPoint* p = seq_point(squares, i + j);
            if (!p) return res;
            points[j].x = (float)p->x;
            points[j].y = (float)p->y;
        }
    return res;
}

int cvarGetAllSquares(CvSeq* squares, vector<CvPoint2D32f>* pts) {
    int res = 0;
    for (int i = 0; squares && i + 3 < squares->total; i += 4, res++)
        for (int j = 0; j < 4; j++) {
            const CvPoint* p = seq_point(squares, i + j);
            if (!p) return res;
            pts->push_back(CvPoint2D32f{(float)p->x, (float)p->y});
        }
    return res;
}

int cvarCompareSquare(IplImage* img, CvPoint2D32f* points) {
    CvSeq* sq = cvarFindSquares(img, nullptr);
    int match = 0;
    for (int i = 0; i + 3 < sq->total; i += 4)
        for (int j = 0; j < 4; j++)
            for (int k = 0; k < 4; k++) {
                const CvPoint* p = seq_point(sq, i + k);
                const double dx = points[j].x - p->x, dy = points[j].y - p->y;
                if (std::sqrt(dx * dx + dy * dy) < 10) match++;
            }
    return match;
}

int cvarDrawSquares(IplImage* img, CvSeq* squares) {
    // the reference draws anti-aliased green polylines; this draws 1-px aliased ones (display aid)
    int res = 0;
    if (!image_ok(img) || !squares) return 0;
    for (int i = 0; i + 3 < squares->total; i += 4, res++)
        for (int e = 0; e < 4; e++) {
            const CvPoint *pa = seq_point(squares, i + e), *pb = seq_point(squares, i + ((e + 1) & 3));
            if (!pa || !pb) return res;
            const CvPoint a = *pa, b = *pb;
            const int steps = std::max(std::abs(b.x - a.x), std::abs(b.y - a.y));
            for (int s = 0; s <= steps; s++) {
                const int x = a.x + (steps ? (int)std::lround((double)(b.x - a.x) * s / steps) : 0);
                const int y = a.y + (steps ? (int)std::lround((double)(b.y - a.y) * s / steps) : 0);
                if (x < 0 || y < 0 || x >= img->width || y >= img->height) continue;
                unsigned char* p = (unsigned char*)img->imageData + (size_t)y * img->widthStep + 3 * x;
                p[0] = 0; p[1] = 255; p[2] = 0;
            }
        }
    return res;
}

void cvarInvertPerspective(IplImage* input, IplImage* output, CvPoint2D32f* src, CvPoint2D32f* dst) {
    // general src->dst quad mapping (reference 510-516); per channel, same fixed-point sampling as the device path
    if (!input || !output || !input->imageData || !output->imageData || input->nChannels != output->nChannels) return;
    double A[8][9];
    for (int i = 0; i < 4; i++) {
        const double sx = src[i].x, sy = src[i].y, dx = dst[i].x, dy = dst[i].y;
        const double r0[9] = {sx, sy, 1, 0, 0, 0, -sx * dx, -sy * dx, dx}, r1[9] = {0, 0, 0, sx, sy, 1, -sx * dy, -sy * dy, dy};
        std::memcpy(A[i], r0, sizeof r0);
        std::memcpy(A[i + 4], r1, sizeof r1);
    }
    for (int c = 0; c < 8; c++) {
        int piv = c;
        for (int r = c + 1; r < 8; r++)
            if (std::fabs(A[r][c]) > std::fabs(A[piv][c])) piv = r;
        if (A[piv][c] == 0) return;
        for (int k = 0; k < 9; k++) std::swap(A[c][k], A[piv][k]);
        for (int r = c + 1; r < 8; r++) {
            const double f = A[r][c] / A[c][c];
            for (int k = c; k < 9; k++) A[r][k] -= f * A[c][k];
        }
    }
    float m32[9];
    for (int c = 7; c >= 0; c--) {
        double s = A[c][8];
        for (int k = c + 1; k < 8; k++) s -= A[c][k] * A[k][8];
        A[c][8] = s / A[c][c];
    }
    for (int i = 0; i < 8; i++) m32[i] = (float)A[i][8];
    m32[8] = 1.f;
    double M[9];
    ocvar::invert_map(m32, M);
    const int cn = input->nChannels;
    std::vector<unsigned char> plane((size_t)input->width * input->height);
    for (int ch = 0; ch < cn; ch++) {
        for (int y = 0; y < input->height; y++)
            for (int x = 0; x < input->width; x++)
                plane[(size_t)y * input->width + x] = (unsigned char)input->imageData[(size_t)y * input->widthStep + cn * x + ch];
        for (int y = 0; y < output->height; y++)
            for (int x = 0; x < output->width; x++)
                output->imageData[(size_t)y * output->widthStep + cn * x + ch] =
                    (char)ocvar::warp_sample(plane.data(), input->width, input->height, input->width, M, x, y);
    }
}

int cvarArMultRegistration(IplImage* image, vector<CvarMarker>* markers, vector<CvarTemplate> templates, CvarCamera* camera) {
    if (!markers) return 0;
    std::lock_guard<std::mutex> lk(g_mu);
    if (!image_ok(image) || !camera || templates.empty() || templates.size() > OCVAR_MAX_TEMPLATES) {
        std::fprintf(stderr, "opencvar: cvarArMultRegistration: unsupported arguments (need 8UC3 image >= 16x16, 1..%d templates)\n",
                     OCVAR_MAX_TEMPLATES);
        markers->clear();
        return 0;
    }
    if (markers->size() > OCVAR_MAX_MARKERS) {
        std::fprintf(stderr, "opencvar: cvarArMultRegistration: %zu markers carried in, this build tracks at most %d per frame\n",
                     markers->size(), OCVAR_MAX_MARKERS);
        markers->clear();
        return 0;
    }
    std::vector<OcvarMarker> prev(OCVAR_MAX_MARKERS);
    int n_prev = (int)markers->size();
    if (n_prev) std::memcpy(prev.data(), markers->data(), n_prev * sizeof(OcvarMarker));
    std::vector<OcvarMarker> out(OCVAR_MAX_MARKERS);
    int count = 0, rc = OCVAR_OK;
    OcvarHip* ctx = context_for(image->width, image->height);
    for (;;) {
        if (!ctx) {
            markers->clear();
            return 0;
        }
        // templates and camera are uploaded only when they differ from what the context holds (a caller passes the same
        // ones frame after frame, ARTest.cpp:57)
        const bool same_t = g_templates.size() == templates.size() &&
                            std::memcmp(g_templates.data(), templates.data(), templates.size() * sizeof(CvarTemplate)) == 0;
        const bool same_c = g_have_camera && std::memcmp(&g_camera, camera, sizeof(CvarCamera)) == 0;
        if ((!same_t && ocvar_hip_set_templates(ctx, reinterpret_cast<const OcvarTemplate*>(templates.data()), (int)templates.size()) != OCVAR_OK) ||
            (!same_c && ocvar_hip_set_camera(ctx, reinterpret_cast<const OcvarCamera*>(camera)) != OCVAR_OK)) {
            std::fprintf(stderr, "opencvar: %s\n", ocvar_hip_last_error(ctx));
            g_templates.clear();
            g_have_camera = false;
            markers->clear();
            return 0;
        }
        g_templates = templates;
        g_camera = *camera;
        g_have_camera = true;
        rc = ocvar_hip_detect_host(ctx, (uint8_t*)image->imageData, image->width, image->height, image->widthStep,
                                   (size_t)image->widthStep * image->height, 1, 1, n_prev ? prev.data() : nullptr,
                                   n_prev ? &n_prev : nullptr, out.data(), &count, OCVAR_MAX_MARKERS);
        // The reference's square list is unbounded (opencvar.cpp:187-214).  A frame with more squares (or candidates) than
        // the context has room for fails before the caller's image is touched; it is run again on a larger context
        // (256 -> 1024 -> OCVAR_MAX_QUADS_EX squares), which is then kept.
        if (rc != OCVAR_E_CAPACITY || !(ocvar_hip_capacity_flags(ctx) & 4) || g_maxq >= OCVAR_MAX_QUADS_EX) break;
        ctx = context_for(image->width, image->height, g_maxq * 4 < OCVAR_MAX_QUADS_EX ? g_maxq * 4 : OCVAR_MAX_QUADS_EX);
    }
    markers->clear();
    if (rc != OCVAR_OK) {
        std::fprintf(stderr, "opencvar: detection failed (%d): %s\n", rc, ocvar_hip_last_error(ctx));
        return 0;
    }
    markers->resize(count);   // (more than OCVAR_MAX_MARKERS would have failed above with OCVAR_E_CAPACITY)
    if (count) std::memcpy(markers->data(), out.data(), count * sizeof(OcvarMarker));
    return (int)markers->size();
}

}  // extern "C"
