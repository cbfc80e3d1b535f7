/* c5_loop.c -- the frame loop of BASELINE configs[4] driven from C: what a C++ caller of the C ABI (the reference's
 * Tracking::Relocalization loop, src/Tracking.cc:1471-1492, per stream frame) costs the host, next to bench.py's Python loop
 * (the interpreter's submission time was 72 % of a step).
 *
 * Per stream frame i:  orb_extract_batch_device(1 frame -> query slot i % n_slots of the feature store) on extractor handle
 * i % n_ex, then on matcher handle i % n_mt: orb_bow_query_frames_device (Frame::ComputeBoW + SearchByBoW against every
 * keyframe).  The handles' streams are coupled per slot by events, exactly as bench.py's run(): the extraction of frame j waits
 * for the search that last used slot j % n_slots, the search of frame i for the extraction of frame i.
 *
 * Built as a shared object (make -C orb-slam2-chinesenotes_amd c5-loop) and called by bench.py --config c5 through ctypes with
 * the handles and device buffers bench.py set up; returns 0 or the first failing call's code. */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "../include/orb_hip.h"

static double now_s(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

#define MAX_SLOTS 64

typedef struct c5_loop_args {
    orb_extractor** ex;         /* n_ex extractor handles */
    orb_matcher** mt;           /* n_mt matcher handles */
    orb_vocab* voc;
    const orb_featstore* store; /* keyframes in frames [0, n_kf), query slot s in frame n_kf + s * slot_stride */
    const uint8_t* d_stream;    /* n_stream frames of rows x cols, contiguous */
    const int32_t* d_kf_index;  /* [n_kf] */
    const int32_t* const* d_f_index;  /* per slot: [1] = the slot's frame index */
    int32_t* const* d_match;    /* per slot: [n_kf * cap] */
    int32_t* const* d_nmatches; /* per slot: [n_kf] */
    orb_keypoint* d_kps;        /* the store's arrays, written by the extraction */
    uint8_t* d_desc;
    int32_t* d_counts;
    int32_t n_ex, n_mt, n_slots, slot_stride, n_kf, cap, rows, cols, n_stream, levelsup, check_ori;
    float ratio;
} c5_loop_args;

/* steps i0 .. i0 + n - 1 (+ the extraction of frame i0 + n, as bench.py's run()); *submit_s = host time of the loop */
/* C5_LOOP_TIMING=1: where the host's time goes, per call site (printed by c5_loop_report) */
static double g_t[4];          /* extract, event waits, query, event records */
static long g_steps;
static int g_timing = -1;
#define TIMED(k, call) do { if (g_timing) { const double t_ = now_s(); call; g_t[k] += now_s() - t_; } else { call; } } while (0)

void c5_loop_report(void)
{
    if (g_timing > 0 && g_steps > 0)
        fprintf(stderr, "[c5_loop] host us per frame over %ld frames: extract call %.2f, stream waits %.2f, query call %.2f, event records %.2f\n",
                g_steps, 1e6 * g_t[0] / g_steps, 1e6 * g_t[1] / g_steps, 1e6 * g_t[2] / g_steps, 1e6 * g_t[3] / g_steps);
    g_t[0] = g_t[1] = g_t[2] = g_t[3] = 0.0;
    g_steps = 0;
}

int c5_loop_run(const c5_loop_args* A, int i0, int n, double* submit_s)
{
    if (g_timing < 0) g_timing = getenv("C5_LOOP_TIMING") != NULL;
    static hipEvent_t evEx[MAX_SLOTS], evMt[MAX_SLOTS];
    static int haveEvents = 0;
    if (A->n_slots > MAX_SLOTS || A->n_slots < 2 || A->n_ex < 1 || A->n_mt < 1) return ORB_ERR_INVALID;
    if (!haveEvents) {
        for (int s = 0; s < MAX_SLOTS; s++) {
            if (hipEventCreateWithFlags(&evEx[s], hipEventDisableTiming) != hipSuccess) return ORB_ERR_HIP;
            if (hipEventCreateWithFlags(&evMt[s], hipEventDisableTiming) != hipSuccess) return ORB_ERR_HIP;
        }
        haveEvents = 1;
    }
    const size_t frameB = (size_t)A->rows * A->cols;
    const double t0 = now_s();
    int rc;
#define EXTRACT(i)                                                                                                              \
    do {                                                                                                                        \
        const int s_ = (i) % A->n_slots;                                                                                        \
        const size_t f0_ = (size_t)A->n_kf + (size_t)s_ * A->slot_stride;                                                       \
        TIMED(0, rc = orb_extract_batch_device(A->ex[(i) % A->n_ex], A->d_stream + (size_t)((i) % A->n_stream) * frameB, 1, A->rows, A->cols,  \
                                      (size_t)A->cols, frameB, A->d_kps + f0_ * A->cap, A->d_desc + f0_ * A->cap * 32, A->cap,  \
                                      A->d_counts + f0_));                                                                      \
        if (rc != ORB_OK) return rc;                                                                                            \
        hipError_t e_;                                                                                                          \
        TIMED(3, e_ = hipEventRecord(evEx[s_], (hipStream_t)orb_extractor_stream(A->ex[(i) % A->n_ex])));                       \
        if (e_ != hipSuccess) return ORB_ERR_HIP;                                                                               \
    } while (0)
    EXTRACT(i0);
    for (int i = i0; i < i0 + n; i++) {
        const int j = i + 1;
        hipError_t e;
        if (j - A->n_slots >= i0) {     /* the search that last used the slot has let go of it */
            /* n_slots frames back: almost always finished long ago -- then the host sees it in the event (a read of host memory)
             * and the extractor's stream needs no wait command (a hipStreamWaitEvent is ~3 us of this ~50 us loop) */
            TIMED(1, e = hipEventQuery(evMt[j % A->n_slots]));
            if (e == hipErrorNotReady) {
                TIMED(1, e = hipStreamWaitEvent((hipStream_t)orb_extractor_stream(A->ex[j % A->n_ex]), evMt[j % A->n_slots], 0));
            }
            if (e != hipSuccess) { (void)hipGetLastError(); return ORB_ERR_HIP; }
        }
        EXTRACT(j);
        const int s = i % A->n_slots;
        orb_matcher* m = A->mt[i % A->n_mt];
        hipStream_t ms = (hipStream_t)orb_matcher_stream(m);
        TIMED(1, e = hipStreamWaitEvent(ms, evEx[s], 0));
        if (e != hipSuccess) return ORB_ERR_HIP;
        TIMED(2, rc = orb_bow_query_frames_device(m, A->voc, A->store, A->n_kf + s * A->slot_stride, 1, A->levelsup, A->d_kf_index, A->n_kf,
                                         A->d_f_index[s], A->ratio, A->check_ori, A->d_match[s], A->d_nmatches[s]));
        if (rc != ORB_OK) return rc;
        TIMED(3, e = hipEventRecord(evMt[s], ms));
        if (e != hipSuccess) return ORB_ERR_HIP;
        g_steps++;
    }
#undef EXTRACT
    if (submit_s) *submit_s = now_s() - t0;
    return ORB_OK;
}
