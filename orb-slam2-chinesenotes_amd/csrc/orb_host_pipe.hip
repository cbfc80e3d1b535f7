// orb_host_pipe.hip -- orb_extract_batch for LARGE host batches: the reference API hands over host images
// (cv::Mat, reference src/ORBextractor.cc:1084-1091, called from src/Frame.cc:262-268), so the PCIe-inclusive path
// matters next to the device-resident one.  The batch is cut into chunks that flow through three stages on three
// streams, three slots deep:
//     staging copy + H2D(chunk k)  ||  kernel chain(chunk k-1)  ||  D2H(chunk k-1) ... host-side unpacking(chunk k-2)
// Caller buffers that are already pinned (hipHostMalloc / hipHostRegister) are copied from / to directly; pageable
// ones go through the handle's pinned staging (CPU memcpys per chunk, the price of pageable memory; spread over up to
// four threads -- ORB_HOST_THREADS -- because one core copies ~20 GB/s and the chunk's 20 MB were the pipeline's period).
// The kernel chain, the scratch slabs and the status block are the handle's own (one chain at a time on its stream);
// only the device in/out buffers and the staging exist once per slot.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "orb_extractor_internal.h"
#include "orb_host_threads.h"                // the chunk pipeline and its copy threads, HIP-free (runs under TSan with a fake device)

static bool is_pinned(const void* p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();                               // plain malloc'ed memory: not an error
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

static int pipe_init(orb_extractor* h)
{
    orb_extractor::Pipe& P = h->pipe;
    if (P.ready) return ORB_OK;
    ORB_HIP_TRY(orb_stream_create(&P.h2d, h->device, 2));
    ORB_HIP_TRY(orb_stream_create(&P.d2h, h->device, 2));
    for (int s = 0; s < orb_extractor::kPipeSlots; s++) {
        ORB_HIP_TRY(hipEventCreateWithFlags(&P.evIn[s], hipEventDisableTiming));
        ORB_HIP_TRY(hipEventCreateWithFlags(&P.evK[s], hipEventDisableTiming));
        ORB_HIP_TRY(hipEventCreateWithFlags(&P.evOut[s], hipEventDisableTiming));
    }
    P.ready = true;
    return ORB_OK;
}

void orb_pipe_release(orb_extractor* h)
{
    orb_extractor::Pipe& P = h->pipe;
    for (int s = 0; s < orb_extractor::kPipeSlots; s++) {
        P.dImg[s].release(); P.dKps[s].release(); P.dDesc[s].release(); P.dCnt[s].release();
        if (P.pinIn[s]) (void)hipHostFree(P.pinIn[s]);
        if (P.pinOut[s]) (void)hipHostFree(P.pinOut[s]);
        P.pinIn[s] = P.pinOut[s] = nullptr;
        if (P.evIn[s]) (void)hipEventDestroy(P.evIn[s]);
        if (P.evK[s]) (void)hipEventDestroy(P.evK[s]);
        if (P.evOut[s]) (void)hipEventDestroy(P.evOut[s]);
        P.evIn[s] = P.evK[s] = P.evOut[s] = nullptr;
    }
    if (P.h2d) orb_stream_destroy(P.h2d, h->device);
    if (P.d2h) orb_stream_destroy(P.d2h, h->device);
    P.h2d = P.d2h = nullptr;
    P.pinInBytes = P.pinOutBytes = 0;
    P.ready = false;
}

static int ensure_pinned(void** slots, size_t* have, size_t need)
{
    const int NS = orb_extractor::kPipeSlots;
    bool all = need <= *have;
    for (int s = 0; s < NS; s++) all = all && slots[s] != nullptr;
    if (all) return ORB_OK;
    for (int s = 0; s < NS; s++) {
        if (slots[s]) (void)hipHostFree(slots[s]);
        slots[s] = nullptr;
    }
    *have = 0;
    for (int s = 0; s < NS; s++) ORB_HIP_TRY(hipHostMalloc(&slots[s], need, hipHostMallocDefault));
    *have = need;
    return ORB_OK;
}

int orb_extract_batch_pipelined(orb_extractor* h, const uint8_t* imgs, int nFrames, int rows, int cols, size_t rowStride,
                                size_t frameStride, orb_keypoint* kps, uint8_t* desc, int cap, int32_t* counts)
{
    int rc;
    if ((rc = pipe_init(h)) != ORB_OK) return rc;
    orb_extractor::Pipe& P = h->pipe;
    const size_t imgBytes = (size_t)rows * cols;
    // chunk: large enough for the kernels to fill the chip, small enough for several chunks to be in flight
    const int C = std::max(ORB_PIPE_CHUNK_MIN, std::min(64, (nFrames + 3) / 4));
    const int nChunks = (nFrames + C - 1) / C;
    const bool inPinned = is_pinned(imgs), outPinned = is_pinned(kps) && is_pinned(desc);
    const size_t kpSlab = sizeof(orb_keypoint) * (size_t)cap, dsSlab = (size_t)ORB_DESC_BYTES * cap;
    const size_t statB = orb_extractor::statInts(C) * 4, cntB = (size_t)4 * C;
    // pinned out slot: [status | counts | keypoints | descriptors] (the two slabs only for pageable caller buffers)
    const size_t outB = statB + cntB + (outPinned ? 0 : (kpSlab + dsSlab) * C);
    for (int s = 0; s < orb_extractor::kPipeSlots; s++) {
        if ((rc = P.dImg[s].ensure(imgBytes * C)) != ORB_OK || (rc = P.dKps[s].ensure(kpSlab * C)) != ORB_OK ||
            (rc = P.dDesc[s].ensure(dsSlab * C)) != ORB_OK || (rc = P.dCnt[s].ensure(cntB)) != ORB_OK)
            return rc;
    }
    if (!inPinned && (rc = ensure_pinned(P.pinIn, &P.pinInBytes, imgBytes * C)) != ORB_OK) return rc;
    if ((rc = ensure_pinned(P.pinOut, &P.pinOutBytes, outB)) != ORB_OK) return rc;
    // The scheduling (issue / retire order, slot reuse, the copy threads) is orb_pipe_run (csrc/orb_host_threads.h, HIP-free:
    // tools/tsan_host.cpp runs it against a fake device under the thread sanitizer); what follows is its device side.
    struct HipOps {
        orb_extractor* h;
        orb_extractor::Pipe& P;
        const uint8_t* imgs; orb_keypoint* kps; uint8_t* desc; int32_t* counts;
        int rows, cols, cap, C;
        size_t rowStride, frameStride, imgBytes, kpSlab, dsSlab, statB, cntB;
        bool inPinned, outPinned;
        hipStream_t cs;
        unsigned chunkSerial[orb_extractor::kPipeSlots] = {};
        HipOps(orb_extractor* hh, orb_extractor::Pipe& pp) : h(hh), P(pp) {}
        bool in_pinned() const { return inPinned; }
        bool out_pinned() const { return outPinned; }
        size_t in_bytes_per_frame() const { return imgBytes; }
        size_t out_bytes_per_frame() const { return kpSlab + dsSlab; }
        void stage_frame(int s, int f, int frame) const       // pageable input: CPU copy into this slot's pinned buffer
        {
            uint8_t* st = (uint8_t*)P.pinIn[s];
            for (int y = 0; y < (rowStride == (size_t)cols ? 1 : rows); y++)
                std::memcpy(st + imgBytes * f + (size_t)y * cols, imgs + frameStride * frame + rowStride * y,
                            rowStride == (size_t)cols ? imgBytes : (size_t)cols);
        }
        int upload(int s, int f0, int c)
        {
            const uint8_t* src = imgs + frameStride * f0;
            size_t srcRow = rowStride, srcFrame = frameStride;
            if (!inPinned) { src = (const uint8_t*)P.pinIn[s]; srcRow = cols; srcFrame = imgBytes; }
            if (srcRow == (size_t)cols && srcFrame == imgBytes) {
                ORB_HIP_TRY(hipMemcpyAsync(P.dImg[s].p, src, imgBytes * c, hipMemcpyHostToDevice, P.h2d));
            } else {
                for (int f = 0; f < c; f++) {
                    if (srcRow == (size_t)cols)
                        ORB_HIP_TRY(hipMemcpyAsync((uint8_t*)P.dImg[s].p + imgBytes * f, src + srcFrame * f, imgBytes, hipMemcpyHostToDevice, P.h2d));
                    else
                        ORB_HIP_TRY(hipMemcpy2DAsync((uint8_t*)P.dImg[s].p + imgBytes * f, cols, src + srcFrame * f, srcRow, cols, rows,
                                                     hipMemcpyHostToDevice, P.h2d));
                }
            }
            return ORB_OK;
        }
        int mark_uploaded(int s) { ORB_HIP_TRY(hipEventRecord(P.evIn[s], P.h2d)); return ORB_OK; }
        int compute_waits_upload(int s) { ORB_HIP_TRY(hipStreamWaitEvent(cs, P.evIn[s], 0)); return ORB_OK; }
        int compute_waits_download(int s) { ORB_HIP_TRY(hipStreamWaitEvent(cs, P.evOut[s], 0)); return ORB_OK; }   // this slot's outputs of chunk k - NS have left
        int extract(int s, int c)
        {
            int r = orb_extract_batch_device(h, (const uint8_t*)P.dImg[s].p, c, rows, cols, cols, imgBytes, (orb_keypoint*)P.dKps[s].p,
                                             (uint8_t*)P.dDesc[s].p, cap, (int32_t*)P.dCnt[s].p);
            if (r != ORB_OK) return r;
            chunkSerial[s] = h->batchSerial;
            // the status block is the handle's single one: it leaves on the compute stream, before the next chunk clears it
            ORB_HIP_TRY(hipMemcpyAsync(P.pinOut[s], h->dStat.p, orb_extractor::statInts(c) * 4, hipMemcpyDeviceToHost, cs));
            return ORB_OK;
        }
        int mark_computed(int s) { ORB_HIP_TRY(hipEventRecord(P.evK[s], cs)); return ORB_OK; }
        int download_waits_compute(int s) { ORB_HIP_TRY(hipStreamWaitEvent(P.d2h, P.evK[s], 0)); return ORB_OK; }
        int download(int s, int f0, int c)
        {
            uint8_t* po = (uint8_t*)P.pinOut[s];
            ORB_HIP_TRY(hipMemcpyAsync(po + statB, P.dCnt[s].p, (size_t)4 * c, hipMemcpyDeviceToHost, P.d2h));
            if (outPinned) {
                ORB_HIP_TRY(hipMemcpyAsync(kps + (size_t)cap * f0, P.dKps[s].p, kpSlab * c, hipMemcpyDeviceToHost, P.d2h));
                ORB_HIP_TRY(hipMemcpyAsync(desc + dsSlab * f0, P.dDesc[s].p, dsSlab * c, hipMemcpyDeviceToHost, P.d2h));
            } else {
                ORB_HIP_TRY(hipMemcpyAsync(po + statB + cntB, P.dKps[s].p, kpSlab * c, hipMemcpyDeviceToHost, P.d2h));
                ORB_HIP_TRY(hipMemcpyAsync(po + statB + cntB + kpSlab * C, P.dDesc[s].p, dsSlab * c, hipMemcpyDeviceToHost, P.d2h));
            }
            return ORB_OK;
        }
        int mark_downloaded(int s) { ORB_HIP_TRY(hipEventRecord(P.evOut[s], P.d2h)); return ORB_OK; }
        int wait_downloaded(int s) { ORB_HIP_TRY(hipEventSynchronize(P.evOut[s])); return ORB_OK; }
        int finish(int s, int f0, int c)
        {
            const uint8_t* po = (const uint8_t*)P.pinOut[s];
            const int keepFrames = h->lastFrames;              // orb_check_status reads the block of `c` frames
            h->lastFrames = c;
            h->hStat.assign((const int*)po, (const int*)po + orb_extractor::statInts(c));
            h->statSerial = chunkSerial[s];
            int r = orb_check_status(h);
            h->statSerial = 0;
            h->lastFrames = keepFrames;
            std::memcpy(counts + f0, po + statB, (size_t)4 * c);
            return r;
        }
        void unpack_frame(int s, int f, int frame) const
        {
            const uint8_t* po = (const uint8_t*)P.pinOut[s];
            const int n = counts[frame];
            if (n <= 0) return;
            std::memcpy(kps + (size_t)cap * frame, po + statB + cntB + kpSlab * f, sizeof(orb_keypoint) * (size_t)n);
            std::memcpy(desc + dsSlab * frame, po + statB + cntB + kpSlab * C + dsSlab * f, (size_t)ORB_DESC_BYTES * n);
        }
        void drain()                                           // leave nothing in flight behind an error
        {
            (void)hipStreamSynchronize(P.h2d);
            (void)hipStreamSynchronize(cs);
            (void)hipStreamSynchronize(P.d2h);
        }
    } ops(h, P);
    ops.imgs = imgs; ops.kps = kps; ops.desc = desc; ops.counts = counts;
    ops.rows = rows; ops.cols = cols; ops.cap = cap; ops.C = C;
    ops.rowStride = rowStride; ops.frameStride = frameStride; ops.imgBytes = imgBytes; ops.kpSlab = kpSlab; ops.dsSlab = dsSlab;
    ops.statB = statB; ops.cntB = cntB; ops.inPinned = inPinned; ops.outPinned = outPinned; ops.cs = h->stream;
    const int firstErr = orb_pipe_run(ops, nFrames, C, orb_extractor::kPipeSlots);
    if (firstErr != ORB_OK) return firstErr;
    // the device keeps the last chunk (pyramids included): frames [frameBase, frameBase + lastFrames) of this batch
    h->frameBase = (nChunks - 1) * C;
    h->lastFrames = nFrames - h->frameBase;
    h->statFetched = true;
    const int hadSticky = h->hStat.empty() ? 0 : h->hStat[orb_extractor::kStickyInts - 1];
    if (hadSticky) ORB_HIP_TRY(orb_fill_blocking(h->dStat.p, 0, orb_extractor::kStickyInts * 4, h->stream));
    return ORB_OK;
}
