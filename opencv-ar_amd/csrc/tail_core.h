// tail_core.h -- per-frame sequential tail of cvarArMultRegistration: tracking association and the greedy
// candidate elimination.  Both are order-dependent by definition in the reference, so one lane per frame
// replays them verbatim on the device (frames are independent, so a batch still fills the chip).
//
// Replaces /root/reference/src/opencvar.cpp:592-617 (cvarTrack), 635-668 (tracking loop with its
// erase-while-iterating behaviour), 780-792 (|| dedupe) and 795-801 (output selection).
#pragma once
#include "hd.h"
#include <math.h>

namespace ocvar {

struct MarkerRec {  // == CvarMarker (include/opencvar/opencvar.h), 184 bytes
    double glMatrix[16];
    int templateId;
    int markerId;
    double score;
    float square[8];
    double aspectRatio;
};

// cvarTrack: all four corners within 20 px under one cyclic shift; on success pt1 takes pt2's corners.
OCVAR_HD int track_square(float* pt1, const float* pt2) {
    for (int j = 0; j < 4; j++) {
        int res = 0;
        for (int i = 0; i < 4; i++) {
            const int k = (i + j) & 3;
            const double dx = (double)pt1[2 * i] - (double)pt2[2 * k], dy = (double)pt1[2 * i + 1] - (double)pt2[2 * k + 1];
            if (sqrt(dx * dx + dy * dy) < 20) res++;
        }
        if (res == 4) {
            float t[8];
            for (int i = 0; i < 4; i++) {
                const int k = (i + j) & 3;
                t[2 * i] = pt2[2 * k];
                t[2 * i + 1] = pt2[2 * k + 1];
            }
            for (int i = 0; i < 8; i++) pt1[i] = t[i];
            return 1;
        }
    }
    return 0;
}

// Tracking loop: squares (n quads of 4 float points, list order) is compacted in place exactly as
// vector::erase does; reserve receives marker indices (duplicates possible).  Returns the new quad count.
OCVAR_HD int track_markers(MarkerRec* markers, int n_markers, float* squares, int n_quads, int* reserve, int max_reserve,
                           int* n_reserve) {
    int nr = 0;
    for (int i = 0; i < n_markers; i++) {
        for (int j = 0; j < n_quads; j++) {
            if (track_square(markers[i].square, squares + 8 * j)) {
                if (nr < max_reserve) reserve[nr] = i;
                nr++;
                for (int k = 8 * j; k < 8 * (n_quads - 1); k++) squares[k] = squares[k + 8];
                n_quads--;
                // the reference's loop index is not corrected after the erase: the next quad is skipped
            }
        }
    }
    *n_reserve = nr;
    return n_quads;
}

// Greedy elimination (opencvar.cpp:780-792): markerId[i] = -1 marks a loser.
OCVAR_HD void dedupe(int* markerId, const int* templateId, const double* score, int n) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j < i; j++)
            if (markerId[i] == markerId[j] || templateId[i] == templateId[j]) {
                if (score[i] > score[j])
                    markerId[j] = -1;
                else
                    markerId[i] = -1;
            }
}

}  // namespace ocvar
