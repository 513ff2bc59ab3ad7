/*
 * c_caller.c -- the C ABI of include/rtdm.h driven from plain C (what a cgo / JNI / ctypes binding would do).
 *   c_caller                       prints device count and the status of a create call (no GPU: RTDM_ERR_NO_DEVICE)
 *   c_caller in.bin out.bin W H D w   in.bin = left||right (u8), out.bin = x16 disparity (s16), parameters of main.cpp:134-135
 * Build: gcc -std=c99 -O2 -Iinclude examples/c_caller.c -Lrt-depth-map_amd/lib -lrtdm_hip -Wl,-rpath,$PWD/rt-depth-map_amd/lib
 */
#include <stdio.h>
#include <stdlib.h>

#include "rtdm.h"

int main(int argc, char** argv)
{
    rtdm_bm_params p;
    rtdm_bm* bm = NULL;
    int ndev = 0, rc;
    if (argc < 7) {
        rc = rtdm_device_count(&ndev);
        rtdm_bm_default_params(&p, 64);
        printf("abi=%d devices=%d (rc %d)\n", rtdm_abi_version(), ndev, rc);
        rc = rtdm_bm_create(&p, 320, 240, 1, 0, &bm);
        printf("create: %d (%s)\n", rc, rtdm_strerror(rc));
        rtdm_bm_destroy(bm);
        return ndev > 0 ? (rc == RTDM_OK ? 0 : 1) : (rc == RTDM_ERR_NO_DEVICE ? 0 : 1);
    }
    {
        const int W = atoi(argv[3]), H = atoi(argv[4]), D = atoi(argv[5]), w = atoi(argv[6]);
        const size_t px = (size_t)W * H;
        unsigned char* in = (unsigned char*)malloc(2 * px);
        short* out = (short*)malloc(px * sizeof(short));
        FILE* f = fopen(argv[1], "rb");
        if (!in || !out || !f || fread(in, 1, 2 * px, f) != 2 * px) return 2;
        fclose(f);
        rtdm_bm_default_params(&p, D);
        p.blockSize = w;
        rc = rtdm_bm_create(&p, W, H, 1, 0, &bm);
        if (rc != RTDM_OK) { fprintf(stderr, "create: %s\n", rtdm_strerror(rc)); return 3; }
        rc = rtdm_bm_compute(bm, in, (size_t)W, in + px, (size_t)W, W, H, out, (size_t)W * 2);
        if (rc != RTDM_OK) { fprintf(stderr, "compute: %s\n", rtdm_strerror(rc)); return 4; }
        f = fopen(argv[2], "wb");
        if (!f || fwrite(out, sizeof(short), px, f) != px) return 5;
        fclose(f);
        rtdm_bm_destroy(bm);
        free(in); free(out);
    }
    return 0;
}
