// orb_host_threads.h -- the HOST-THREAD logic of the library, free of HIP: the short-lived copy threads of a host batch
// (orb_par_items), the chunk pipeline of orb_extract_batch for large host batches (orb_pipe_run: H2D(k) || kernels(k-1) ||
// D2H(k-1) || unpacking(k-2) over three slots, coupled by events), the block partition of a batch over devices
// (orb_shard_range_impl) and the one-thread-per-device fan-out with its error merge (orb_fan_out) of orb_multi_*.
//
// Everything that touches the GPU is reached through an OPS object (template parameter): csrc/orb_host_pipe.hip and
// csrc/orb_multi*.hip pass HIP-backed ops, tools/tsan_host.cpp passes a FAKE device -- streams are worker threads with FIFO
// queues, events are condition variables, copies and "kernels" are memcpys run by those threads -- so that this very code runs
// under -fsanitize=thread on a machine without a GPU (make tsan-host, part of the CPU test suite): a violation of the slot /
// event protocol (a staging buffer rewritten while its copy is still in flight, results unpacked before they have arrived)
// shows up there as a data race between a host thread and a fake DMA thread.  (The reference's own threading around this
// path: two extractor threads per stereo frame, src/Frame.cc:82-85; SURVEY section 5 asks for the host code under TSan.)
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

// contiguous block partition; blocks differ by at most one item (orbhip/shard.py frame_range is the same rule)
static inline void orb_shard_range_impl(int total, int world, int rank, int* first, int* count)
{
    if (world <= 0 || rank < 0 || rank >= world || total < 0) { if (first) *first = 0; if (count) *count = 0; return; }
    const int base = total / world, rem = total % world;
    if (count) *count = base + (rank < rem ? 1 : 0);
    if (first) *first = rank * base + std::min(rank, rem);
}

static inline int orb_host_thread_limit()
{
    static const int maxT = [] {
        const char* e = std::getenv("ORB_HOST_THREADS");
        const int t = e ? std::atoi(e) : (int)std::min(4u, std::max(1u, std::thread::hardware_concurrency()));
        return std::max(1, std::min(16, t));
    }();
    return maxT;
}

// fn(i) for i in [0, n), split over a few short-lived threads when the items are worth it (>= 2 MB per thread); the calling
// thread takes a share and joins the others before it returns.  fn must touch disjoint data per item.
template <class F>
static void orb_par_items(int n, size_t bytesPerItem, F fn)
{
    int T = (int)std::min<size_t>((size_t)orb_host_thread_limit(), std::max<size_t>(1, (size_t)n * bytesPerItem / ((size_t)2 << 20)));
    T = std::min(T, n);
    std::vector<std::thread> th;
    if (T > 1) {
        try {
            for (int t = 1; t < T; t++)
                th.emplace_back([=] { for (int i = t; i < n; i += T) fn(i); });
        } catch (...) {                                        // no more threads to be had: the caller's thread does the rest
            const int started = (int)th.size() + 1;
            for (std::thread& x : th) x.join();
            for (int t = started; t < T; t++)
                for (int i = t; i < n; i += T) fn(i);
            for (int i = 0; i < n; i += T) fn(i);
            return;
        }
    }
    for (int i = 0; i < n; i += std::max(T, 1)) fn(i);
    for (std::thread& x : th) x.join();
}

// work(r) -> return code, for every rank r in [0, W) with something to do, one thread per rank (the calling thread runs a
// rank's work itself when no thread is to be had: no exception leaves the C ABI); the ranks' error strings are thread-local
// (lastError() is called on the rank's own thread).  Returns the first failing rank, or -1.
template <class Work, class LastError>
static int orb_fan_out(int W, Work work, LastError lastError, std::vector<int>& rcs, std::vector<std::string>& errs)
{
    rcs.assign((size_t)W, 0);
    errs.assign((size_t)W, std::string());
    std::vector<std::thread> th;
    for (int r = 0; r < W; r++) {
        auto run = [r, &work, &lastError, &rcs, &errs]() {
            rcs[(size_t)r] = work(r);
            if (rcs[(size_t)r] != 0) errs[(size_t)r] = lastError();
        };
        try {
            th.emplace_back(run);
        } catch (...) {
            run();
        }
    }
    for (std::thread& t : th) t.join();
    for (int r = 0; r < W; r++)
        if (rcs[(size_t)r] != 0) return r;
    return -1;
}

// ---- the chunk pipeline of a large host batch --------------------------------------------------------------------------------
// Ops (all return 0 or an error code unless void):
//   bool  in_pinned(), out_pinned()              the caller's buffers are pinned: no staging copy in / no unpacking out
//   void  stage_frame(int s, int f, int frame)   CPU copy of batch frame `frame` to position f of slot s's pinned input
//   int   upload(int s, int f0, int c)           async H2D of the chunk's c frames (from the slot's staging, or from the
//                                                caller's pinned images at frame f0) on the copy-in stream
//   int   mark_uploaded(int s)                   event "slot s's input has arrived", recorded on the copy-in stream
//   int   compute_waits_upload(int s)            the compute stream waits for it
//   int   compute_waits_download(int s)          ... and for the slot's previous outputs to have left (chunk k - NS)
//   int   extract(int s, int c)                  the kernel chain of the chunk + its status block, on the compute stream
//   int   mark_computed(int s), download_waits_compute(int s)
//   int   download(int s, int f0, int c)         async D2H of counts (+ keypoints, descriptors) on the copy-out stream
//   int   mark_downloaded(int s)                 event recorded on the copy-out stream
//   int   wait_downloaded(int s)                 the HOST waits for it
//   int   finish(int s, int f0, int c)           status check + counts of the chunk into the caller's array
//   void  unpack_frame(int s, int f, int frame)  CPU copy of one frame's results out of the slot's pinned output
//   size_t in_bytes_per_frame(), out_bytes_per_frame()
//   void  drain()                                leave nothing in flight (after an error)
// Chunk k is issued while chunk k - 1 runs and chunk k - 2 is retired: the host copies chunk k's images into pinned staging
// BEFORE it waits for anything.  Slot s = k mod NS; a slot's staging is rewritten only after the chunk that used it NS chunks
// earlier has been RETIRED (its download event waited for), which orders every earlier use of the slot before the rewrite.
template <class Ops>
static int orb_pipe_run(Ops& ops, int nFrames, int C, int NS)
{
    const int nChunks = (nFrames + C - 1) / C;
    auto issue = [&](int k) -> int {
        const int s = k % NS, f0 = k * C, c = std::min(C, nFrames - f0);
        int rc;
        if (!ops.in_pinned())
            orb_par_items(c, ops.in_bytes_per_frame(), [&ops, s, f0](int f) { ops.stage_frame(s, f, f0 + f); });
        if ((rc = ops.upload(s, f0, c)) != 0) return rc;
        if ((rc = ops.mark_uploaded(s)) != 0) return rc;
        if ((rc = ops.compute_waits_upload(s)) != 0) return rc;
        if (k >= NS && (rc = ops.compute_waits_download(s)) != 0) return rc;
        if ((rc = ops.extract(s, c)) != 0) return rc;
        if ((rc = ops.mark_computed(s)) != 0) return rc;
        if ((rc = ops.download_waits_compute(s)) != 0) return rc;
        if ((rc = ops.download(s, f0, c)) != 0) return rc;
        return ops.mark_downloaded(s);
    };
    auto retire = [&](int k) -> int {
        const int s = k % NS, f0 = k * C, c = std::min(C, nFrames - f0);
        int rc;
        if ((rc = ops.wait_downloaded(s)) != 0) return rc;
        if ((rc = ops.finish(s, f0, c)) != 0) return rc;
        if (!ops.out_pinned())
            orb_par_items(c, ops.out_bytes_per_frame(), [&ops, s, f0](int f) { ops.unpack_frame(s, f, f0 + f); });
        return 0;
    };
    int issued = 0, firstErr = 0;
    for (int k = 0; k <= nChunks + 1; k++) {
        if (k < nChunks && firstErr == 0) {
            const int r = issue(k);
            if (r != 0) firstErr = r; else issued = k + 1;
        }
        if (k >= 2 && k - 2 < issued) {
            const int r = retire(k - 2);
            if (r != 0 && firstErr == 0) firstErr = r;
        }
    }
    if (firstErr != 0) ops.drain();
    return firstErr;
}
