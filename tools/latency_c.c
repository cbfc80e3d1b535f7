/* latency_c.c -- single-frame latency of the C ABI as a C/C++ caller sees it (Frame::ExtractORB calls operator() once per
 * frame, reference src/Frame.cc:247-253): one orb_extract per frame on pageable host buffers, no interpreter in between.
 * usage: latency_c frames.bin width height nframes nfeatures calls
 *   frames.bin = nframes raw 8-bit images (tools/latency_c.sh writes them with the synthetic generator)
 * build: make -C orb-slam2-chinesenotes_amd latency-c */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../include/orb_hip.h"

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

static int cmp_double(const void* a, const void* b)
{
    const double x = *(const double*)a, y = *(const double*)b;
    return x < y ? -1 : x > y;
}

int main(int argc, char** argv)
{
    if (argc < 7) { fprintf(stderr, "usage: %s frames.bin width height nframes nfeatures calls\n", argv[0]); return 2; }
    const int w = atoi(argv[2]), h = atoi(argv[3]), nf = atoi(argv[4]), nfeat = atoi(argv[5]), calls = atoi(argv[6]);
    const size_t bytes = (size_t)w * h;
    unsigned char* frames = (unsigned char*)malloc(bytes * nf);
    FILE* fp = fopen(argv[1], "rb");
    if (!fp || fread(frames, bytes, nf, fp) != (size_t)nf) { fprintf(stderr, "cannot read %d frames of %dx%d from %s\n", nf, w, h, argv[1]); return 1; }
    fclose(fp);
    orb_extractor_params prm = {nfeat, 1.2f, 8, 20, 7};
    orb_extractor* ex = NULL;
    if (orb_extractor_create(&prm, 0, &ex) != ORB_OK) { fprintf(stderr, "create: %s\n", orb_last_error()); return 1; }
    const int cap = orb_extractor_max_keypoints(ex);
    orb_keypoint* kps = (orb_keypoint*)malloc(sizeof(orb_keypoint) * cap);
    unsigned char* desc = (unsigned char*)malloc((size_t)ORB_DESC_BYTES * cap);
    double* t = (double*)malloc(sizeof(double) * calls);
    int n = 0;
    long total = 0;
    for (int i = 0; i < 10; i++)
        if (orb_extract(ex, frames + bytes * (i % nf), h, w, (size_t)w, kps, desc, cap, &n) != ORB_OK) { fprintf(stderr, "extract: %s\n", orb_last_error()); return 1; }
    for (int i = 0; i < calls; i++) {
        const double t0 = now_ms();
        if (orb_extract(ex, frames + bytes * (i % nf), h, w, (size_t)w, kps, desc, cap, &n) != ORB_OK) { fprintf(stderr, "extract: %s\n", orb_last_error()); return 1; }
        t[i] = now_ms() - t0;
        total += n;
    }
    qsort(t, calls, sizeof(double), cmp_double);
    printf("C caller %dx%d nfeatures=%d: median %.3f ms  p90 %.3f ms  min %.3f ms  (%.0f keypoints per frame)\n", w, h, nfeat, t[calls / 2],
           t[(int)(calls * 0.9)], t[0], (double)total / calls);
    orb_extractor_destroy(ex);
    return 0;
}
