/* Stand-in for <opencv/highgui.h>: the opencvar public header includes it but uses nothing from it. */
#ifndef OCVAR_SHIM_OPENCV_HIGHGUI_H
#define OCVAR_SHIM_OPENCV_HIGHGUI_H
#include "cv.h"
#endif
