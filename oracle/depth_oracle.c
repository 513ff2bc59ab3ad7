/*
 * depth_oracle.c -- CPU oracle for the step right after the matcher:
 *     left_disp /= 16.;                                              (estimator.cpp:75)
 *     reprojectImageTo3D(left_disp, xyz, Q, true, CV_32F);           (estimator.cpp:76)
 *     calc_depth(xyz, left_disp, filter_out, img, obj_boundings, u); (estimator.cpp:77, 206-263)
 *
 * TEST INFRASTRUCTURE, PARITY UNPINNED -- see rtdm_oracle.h.  calc_depth is the reference's own code and is
 * restated from estimator.cpp:206-263 (mean of Z over the pixels of each bounding box whose mask byte is
 * non-zero and whose Z is neither the missing-value marker 10000 nor beyond it; result * unit / 10).  The two
 * OpenCV calls are restated from their published behaviour:
 *   Mat /= 16.  on CV_16S    -> convertTo(-1, 1/16): round-half-to-even of d/16, saturated to short
 *   reprojectImageTo3D       -> [X Y Z W]^T = Q [x y d 1]^T in double, point = (X/W, Y/W, Z/W) stored as float;
 *                               handleMissingValues: pixels whose disparity equals the image's minimum get Z = 10000
 * Floating point: Z is a float computed from a double quotient; the mean is a double sum in row-major order.
 * The HIP path sums in a different (fixed) order, so parity is stated with a tolerance (tests: 1e-9 relative).
 */
#include "rtdm_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>

static inline int round_half_even_div16(int d)
{
    /* d/16 with ties to even, exact in integers */
    int q = d >> 4, r = d & 15;           /* floor quotient, remainder 0..15 */
    if (r > 8 || (r == 8 && (q & 1))) ++q;
    return q;
}

int orc_depth_stats(const int16_t* disp16, size_t dstep_elems, int W, int H, const double Q[16],
                    const uint8_t* mask, size_t mstep, const int* regions /* n x (x,y,w,h) */, int n,
                    double calibration_unit, double* mean_cm, int* counts)
{
    if (!disp16 || !Q || !mask || (!regions && n > 0) || !mean_cm || !counts || W <= 0 || H <= 0) return ORC_ERR_BAD_SIZE;
    int mind = INT16_MAX;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const int d = round_half_even_div16(disp16[(size_t)y * dstep_elems + x]);
            if (d < mind) mind = d;
        }
    const double bigZ = 10000.0;
    for (int i = 0; i < n; ++i) {
        const int rx = regions[4 * i], ry = regions[4 * i + 1], rw = regions[4 * i + 2], rh = regions[4 * i + 3];
        if (rx < 0 || ry < 0 || rw < 0 || rh < 0 || rx + rw > W || ry + rh > H) return ORC_ERR_BAD_SIZE;
        double res = 0.0;
        int cnt = 0;
        for (int y = ry; y < ry + rh; ++y)
            for (int x = rx; x < rx + rw; ++x) {
                const int d = round_half_even_div16(disp16[(size_t)y * dstep_elems + x]);
                const double Zh = Q[8] * x + Q[9] * y + Q[10] * d + Q[11];
                const double Wh = Q[12] * x + Q[13] * y + Q[14] * d + Q[15];
                float z = (float)(Zh / Wh);
                if (d == mind) z = (float)bigZ;
                if (fabs((double)z - bigZ) < FLT_EPSILON || fabs((double)z) > bigZ || mask[(size_t)y * mstep + x] == 0) continue;
                res += (double)z;
                ++cnt;
            }
        counts[i] = cnt;
        mean_cm[i] = cnt > 0 ? (res / cnt) * calibration_unit / 10.0 : 0.0;
    }
    return ORC_OK;
}
