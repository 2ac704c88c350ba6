/*
 * opencvar/acmath.h -- math helpers of the opencvar library (3-vectors, 4x4 matrices, quaternions and the
 * bit <-> grid codec of the marker codes).  API-compatible with the reference header
 * (/root/reference/include/opencvar/acmath.h:52-221): same names, argument meaning and C linkage.
 * Matrices are double[16]; functions documented "row-major" in the reference keep that meaning.
 * acMatrixTranslate is declared by the reference but never defined there; it is defined here.
 */
#pragma once

#ifdef ACDLL
#define AC_DLL __declspec(dllexport)
#else
#define AC_DLL
#endif

struct AcPointi { int x; int y; };
struct AcPointf { double x; double y; };

extern "C" {

AC_DLL void acVectorPrint(double* v);
AC_DLL void acVectorAdd(double* v1, double* v2, double* vOut);
AC_DLL void acVectorDeduct(double* v1, double* v2, double* vOut);
AC_DLL void acVectorCrossProduct(double* v1, double* v2, double* product);
/* normal of the triangle v1,v2,v3 (not normalised) */
AC_DLL void acVectorNormal(double* v1, double* v2, double* v3, double* normal);
AC_DLL double acVectorMagnitude(double* v);
AC_DLL void acVectorNormalise(double* vIn, double* vOut);
/* unit normal of the triangle v1,v2,v3 */
AC_DLL void acVectorNormal2(double* v1, double* v2, double* v3, double* nv);

AC_DLL double acRad2Deg(double rad);
AC_DLL double acDeg2Rad(double deg);

/* OpenGL-style (column-major) rotation / translation / scale matrices */
AC_DLL void acMatrixRotate(double deg, double x, double y, double z, double* m);
AC_DLL void acMatrixTranslate(double x, double y, double z, double* m);
AC_DLL void acMatrixScale(double x, double y, double z, double* m);
AC_DLL void acMatrixIdentity(double* m);
/* row `row` of m1 times column `col` of m2 (row-major 4x4) */
AC_DLL double acMatrixDotProduct(double* m1, double* m2, int col, int row);
AC_DLL void acMatrixMultiply(double* m1, double* m2, double* mOut);
AC_DLL void acMatrixPrint(double* m);
AC_DLL void acMatrixTranspose(double* m);

/* q = (w, x, y, z) */
AC_DLL void acMatrixToQuaternion(double* m, double* q);
AC_DLL void acQuaternionToMatrix(double* q, double* m);

/* cosine of the angle pt1-pt0-pt2 */
AC_DLL double acAngle(AcPointi* pt1, AcPointi* pt2, AcPointi* pt0);
AC_DLL double acCalcLength(AcPointf pt1, AcPointf pt2);

AC_DLL void acMatrix4Invert(double m[]);
AC_DLL double acMatrix4GetDeterminant(double m[]);
/* m (row-major 4x4) -> translation t[3], scale s[3], rotation r[16] */
AC_DLL void acMatrixDecompose(double m[], double t[], double s[], double r[]);

/* rot: 1/2/3 = 90/180/270 degrees clockwise */
AC_DLL void acArray2DRotateub(unsigned char* arr, int w, int h, int rot);
AC_DLL void acArray2DPrintub(unsigned char* arr, int w, int h);
/* binary grid -> code: rows top to bottom, each row right to left, first bit read becomes most significant */
AC_DLL void acArray2DToBit(unsigned char* arr, int w, int h, long long int* bit);
AC_DLL void acBitToArray2D(long long int bit, unsigned char* arr, int w, int h);
AC_DLL void acBitRotate(long long int* bit, int rot, int w, int h);

}  // extern "C"
