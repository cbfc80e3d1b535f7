/* orb_hip.h -- C ABI of the MI355X-native ORB front-end (liborbhip.so).
 *
 * This is the drop-in boundary for ONE hot path of ORB-SLAM2 (reference:
 * Hello-Water/ORB-SLAM2-ChineseNotes): ORBextractor::operator() and the descriptor-matching
 * core of ORBmatcher.  Plain pointers and sizes only; no C++/torch types.  The C++ shims in
 * orb-slam2-chinesenotes_amd/host/ keep the reference class signatures and call these entry
 * points, so Frame.cc / Tracking.cc link unchanged.
 *
 * Every entry point cites the reference interface it replaces as file:line relative to the
 * reference root.  All functions return ORB_OK (0) or a negative orb_status; none aborts.
 * Handles are thread-compatible: distinct handles may be used concurrently from distinct host
 * threads (reference src/Frame.cc:82-85 drives two extractors from two threads); one handle
 * must not be re-entered.  Each handle owns one HIP stream and its scratch buffers.
 */
#ifndef ORB_HIP_H
#define ORB_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum orb_status {
    ORB_OK = 0,
    ORB_ERR_INVALID = -1,      /* bad argument                                             */
    ORB_ERR_HIP = -2,          /* a HIP runtime call failed (see orb_last_error)           */
    ORB_ERR_NO_DEVICE = -3,    /* no usable gfx950 device: the product has NO CPU fallback */
    ORB_ERR_CAPACITY = -4,     /* caller buffer too small for the result                   */
    ORB_ERR_UNSUPPORTED = -5,  /* image/parameter outside the supported envelope           */
    ORB_ERR_INTERNAL = -6      /* device-side overflow flag raised (never expected)        */
} orb_status;

/* Same 28-byte layout as cv::KeyPoint {pt.x, pt.y, size, angle, response, octave, class_id}
 * (what reference src/ORBextractor.cc:1148 appends to _keypoints). */
typedef struct orb_keypoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orb_keypoint;

/* The five positional arguments of ORBextractor::ORBextractor
 * (reference include/ORBextractor.h:52-53, src/ORBextractor.cc:498-501). */
typedef struct orb_extractor_params {
    int32_t nfeatures;
    float scale_factor;
    int32_t nlevels;
    int32_t ini_th_fast;
    int32_t min_th_fast;
} orb_extractor_params;

typedef struct orb_extractor orb_extractor; /* opaque */

#define ORB_MAX_LEVELS 16
#define ORB_DESC_BYTES 32

/* ---------------------------------------------------------------- life cycle ---------------
 * replaces: ORBextractor ctor/dtor, reference src/ORBextractor.cc:498-559, include/ORBextractor.h:55. */
int orb_extractor_create(const orb_extractor_params* params, int device_id, orb_extractor** out);
void orb_extractor_destroy(orb_extractor* h);

/* replaces the inline getters GetScaleFactors / GetInverseScaleFactors / GetScaleSigmaSquares /
 * GetInverseScaleSigmaSquares (reference include/ORBextractor.h:64-84) plus mnFeaturesPerLevel
 * (src/ORBextractor.cc:522-534).  Each array holds nlevels entries; any pointer may be NULL. */
int orb_extractor_get_tables(const orb_extractor* h, float* scale, float* inv_scale, float* sigma2,
                             float* inv_sigma2, int32_t* features_per_level);

/* Upper bound on keypoints per frame the handle can return (sum over levels of quota+slack). */
int orb_extractor_max_keypoints(const orb_extractor* h);

/* Replaces the copy of bit_pattern_31_ into `pattern` (reference src/ORBextractor.cc:537-539).
 * 512 (x,y) int8 points = 1024 bytes.  A new handle already holds the built-in table; the
 * batched multi-GPU mode overwrites it on every rank with the table rank 0 broadcast over
 * RCCL/xGMI.  `_device` takes a device pointer (the broadcast buffer itself). */
int orb_extractor_set_pattern(orb_extractor* h, const int8_t* pattern_xy_1024);
int orb_extractor_set_pattern_device(orb_extractor* h, const int8_t* d_pattern_xy_1024);
/* Copies the built-in table to a caller buffer of 1024 bytes (what rank 0 broadcasts). */
int orb_builtin_pattern(int8_t* pattern_xy_1024);

/* The integer taps of cv::GaussianBlur(image, Size(7, 7), 2, 2, BORDER_REFLECT_101) on CV_8U (reference
 * src/ORBextractor.cc:1129-1130; OpenCV is a third-party dependency whose version the reference does not pin): symmetric
 * kernel k0 k1 k2 k3 k2 k1 k0 in 8.8 fixed point, row pass sum(k * pixel), column pass (sum(k * row) + 32768) >> 16,
 * saturated.  Which integers OpenCV uses depends on its version:
 *   ORB_GAUSS_OPENCV_LEGACY         {18, 34, 49, 55} (sum 257): cvRound(256 * float tap) -- OpenCV 2.4 ... 3.4.1's integer
 *                                   filter engine AND the first fixed-point ("bit-exact") implementations of 3.4.2+ / 4.0+,
 *                                   which round every tap to nearest as well.  The default of a new handle.
 *   ORB_GAUSS_OPENCV_FIXEDPOINT_ED  {18, 34, 48, 56} (sum 256): the later fixed-point kernels, whose taps are rounded with
 *                                   error diffusion so that they sum to exactly 256 (getGaussianKernelFixedPoint_ED).
 * Both are restated from the builder's knowledge of OpenCV's sources, not checked against an OpenCV build (INTEGRATION.md 5;
 * tools/pin_against_opencv tells on a machine that has one).  orb_extractor_set_gaussian takes any taps in 0..255 whose
 * kernel sums to at most 257.  Synchronises the handle's stream. */
#define ORB_GAUSS_OPENCV_LEGACY 0
#define ORB_GAUSS_OPENCV_FIXEDPOINT_ED 1
int orb_gaussian_preset(int preset, int32_t* taps4);
int orb_extractor_set_gaussian(orb_extractor* h, const int32_t* taps4);

/* ---------------------------------------------------------------- extraction ---------------
 * replaces: ORBextractor::operator()(image, mask, keypoints, descriptors),
 * reference src/ORBextractor.cc:1084-1150 (mask is ignored there too, include/ORBextractor.h:59).
 * Host buffers; synchronous.  img: rows x cols u8, `stride` bytes between rows.
 * kps/desc need room for `cap` entries (orb_extractor_max_keypoints is always enough);
 * *n receives the count.  An empty image (rows==0 || cols==0 || img==NULL) returns ORB_OK with
 * *n = 0, like the reference's silent return at :1087-1088. */
int orb_extract(orb_extractor* h, const uint8_t* img, int rows, int cols, size_t stride,
                orb_keypoint* kps, uint8_t* desc32, int cap, int* n);

/* Batched-frames mode, host buffers (frames are independent: reference operator() carries no
 * state across calls except mvImagePyramid, which is overwritten).  Frame f starts at
 * imgs + f*frame_stride.  kps/desc32 hold n_frames*cap entries; counts[n_frames].
 * Batches of 16 frames or more run as a pipeline of chunks (host-to-device copy of chunk k+1 beside the kernels of
 * chunk k beside the copy back of chunk k-1); caller buffers that are pinned (orb_host_alloc, hipHostRegister) are
 * copied from / to directly, pageable ones through the handle's pinned staging.  Afterwards the device holds the
 * LAST chunk only: orb_get_pyramid* / orb_stereo_match address frames of that chunk (by their index in the batch).
 * A single frame (the reference's per-call path) is replayed as a captured HIP graph from the third call of one
 * image size on (environment variable ORB_NO_GRAPH=1 keeps it eager); results are identical. */
int orb_extract_batch(orb_extractor* h, const uint8_t* imgs, int n_frames, int rows, int cols,
                      size_t row_stride, size_t frame_stride,
                      orb_keypoint* kps, uint8_t* desc32, int cap, int32_t* counts);

/* Batched-frames mode, everything resident in HBM; asynchronous on the handle's stream.
 * d_* are device pointers with the same shapes as above.  Call orb_extractor_sync (or
 * synchronise the device) before reading results.  Device-side failures (capacity overflow)
 * are reported by the next orb_extractor_sync. */
int orb_extract_batch_device(orb_extractor* h, const uint8_t* d_imgs, int n_frames, int rows, int cols,
                             size_t row_stride, size_t frame_stride,
                             orb_keypoint* d_kps, uint8_t* d_desc32, int cap, int32_t* d_counts);
int orb_extractor_sync(orb_extractor* h);

/* replaces reads of the public member mvImagePyramid[level] (reference include/ORBextractor.h:86,
 * read by src/Frame.cc:520,611,626,633): lazy device->host copy of the interior of one level of
 * one frame of the last batch (the 19-px border of src/ORBextractor.cc:1168-1174 is not
 * materialised).  dst may be NULL to query rows/cols only. */
int orb_get_pyramid_level(orb_extractor* h, int frame, int level, uint8_t* dst, size_t dst_stride,
                          int* rows, int* cols);

/* Diagnostics for differential tests: per-level keypoint and FAST-candidate counts of one frame
 * of the last batch (nlevels entries each; either pointer may be NULL). */
int orb_get_level_counts(orb_extractor* h, int frame, int32_t* kept, int32_t* candidates);

/* Per-stage GPU time in milliseconds (HIP events on the handle's stream), averaged over the
 * batches issued since profiling was enabled (ring of the 64 most recent):
 * stages: 0 pyramid, 1 FAST cells, 2 quadtree, 3 orientation+descriptors, 4 total. */
int orb_extractor_set_profiling(orb_extractor* h, int enable);
int orb_extractor_get_stage_ms(orb_extractor* h, float* ms5);
/* How many frames each timed launch covered: the batch size of the last profiled orb_extract_batch_device call
 * (a batch is ONE launch chain on the handle's stream; host batches are chunked, see orb_extract_batch). */
int orb_extractor_profiled_frames(const orb_extractor* h);

/* The HIP stream (hipStream_t) the handle launches on, for callers that need to order their own
 * device work against it; and the reverse: make the handle's stream wait (on the device, no host
 * sync) for everything already enqueued on another hipStream_t, e.g. the stream that uploaded the
 * images or the extractor stream whose output the matcher consumes. */
void* orb_extractor_stream(orb_extractor* h);
int orb_extractor_wait_for(orb_extractor* h, void* hip_stream);

/* ---------------------------------------------------------------- matcher ------------------
 * replaces: static int ORBmatcher::DescriptorDistance(const cv::Mat&, const cv::Mat&),
 * reference src/ORBmatcher.cc:46-63.  Pure host function on two 32-byte rows. */
int orb_hamming(const uint8_t* a32, const uint8_t* b32);

/* replaces: ORBmatcher::ComputeThreeMaxima, reference src/ORBmatcher.cc:1663-1707
 * (counts of the 30 rotation-histogram bins in, three bin indices or -1 out). */
void orb_three_maxima(const int32_t* counts30, int32_t* ind3);

/* A DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned>>, reference include/Frame.h:161-162)
 * flattened to CSR: node_ids ascending, offsets[n_nodes+1], indices ascending within a node. */
typedef struct orb_featvec {
    const uint32_t* node_ids;
    const int32_t* offsets;
    const int32_t* indices;
    int32_t n_nodes;
} orb_featvec;

typedef struct orb_matcher orb_matcher; /* opaque: stream + scratch for the match kernels */
int orb_matcher_create(int device_id, orb_matcher** out);
void orb_matcher_destroy(orb_matcher* m);
int orb_matcher_sync(orb_matcher* m);

/* replaces: int ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&),
 * reference src/ORBmatcher.cc:552-687.  valid_kf[i] != 0 iff KF feature i has a MapPoint that is
 * not isBad() (:590-595).  match_f[iF] receives the KF feature index matched to F feature iF or
 * -1 (the shim maps indices back to MapPoint*).  ratio/check_ori are mfNNratio/mbCheckOrientation.
 * Host buffers, synchronous.  *nmatches receives the return value of the reference function. */
int orb_match_bow(orb_matcher* m,
                  const uint8_t* desc_kf, const float* angle_kf, const uint8_t* valid_kf, int n_kf,
                  const orb_featvec* fv_kf,
                  const uint8_t* desc_f, const float* angle_f, int n_f, const orb_featvec* fv_f,
                  float ratio, int check_ori, int32_t* match_f, int* nmatches);

/* replaces: int ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vector<MapPoint*>&),
 * reference src/ORBmatcher.cc:690-832 (both sides need valid MapPoints, strict < TH_LOW).
 * match_12[i1] receives the KF2 feature index or -1. */
int orb_match_bow_kk(orb_matcher* m,
                     const uint8_t* desc1, const float* angle1, const uint8_t* valid1, int n1,
                     const orb_featvec* fv1,
                     const uint8_t* desc2, const float* angle2, const uint8_t* valid2, int n2,
                     const orb_featvec* fv2,
                     float ratio, int check_ori, int32_t* match_12, int* nmatches);

/* replaces: int ORBmatcher::SearchForInitialization(Frame&, Frame&, vector<cv::Point2f>&,
 * vector<int>&, int windowSize), reference src/ORBmatcher.cc:1055-1180, including the
 * Frame grid query it makes (src/Frame.cc:243-259, 348-422; 64x48 grid, include/Frame.h:37-38).
 * kps are the undistorted keypoints (mvKeysUn).  grid4 = {mnMinX, mnMinY,
 * mfGridElementWidthInv, mfGridElementHeightInv} of frame 2.  prev_xy (n1 x 2 floats) is
 * vbPrevMatched, updated in place (:1175-1177).  match_12[i1] = index in frame 2 or -1. */
int orb_match_init(orb_matcher* m,
                   const orb_keypoint* kps1, const uint8_t* desc1, int n1,
                   const orb_keypoint* kps2, const uint8_t* desc2, int n2,
                   const float* grid4, float* prev_xy, int window_size,
                   float ratio, int check_ori, int32_t* match_12, int* nmatches);

/* The two tracking matchers.  The MapPoint projection itself (cv::Mat arithmetic,
 * reference src/ORBmatcher.cc:195-212 resp. Frame::isInFrustum) stays with the caller, written with the
 * reference's own expressions; one orb_proj_query per projected MapPoint carries its result:
 *   x, y        the projection handed to GetFeaturesInArea            (:96 mTrackProjX/Y, :204-205 u, v)
 *   r           the window radius                                      (:97 r*mvScaleFactors[level], :215 radius)
 *   min_level, max_level   the level range handed to GetFeaturesInArea (:98; :219-225 by bForward/bBackward)
 *   ur, er_max  stereo check: a candidate with mvuRight > 0 is skipped when |ur - mvuRight| > er_max
 *               (:113-118 mTrackProjXR, r*scale;  :238-244 u - mbf*invzc, radius)
 *   flags       bit0: query is live (:85-88 mbTrackInView && !isBad; :189-212 has MapPoint, not outlier,
 *               projects inside);  bit1: its MapPoint has Observations() > 0 (decides whether the feature it
 *               is assigned to blocks later queries, :109-111 / :233-235) */
typedef struct orb_proj_query {
    float x, y, r;
    int32_t min_level, max_level;
    float ur, er_max;
    int32_t flags;
} orb_proj_query;

/* mode 0 (best candidate, rotation histogram) replaces
 *     int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, float th, bool bMono)   src/ORBmatcher.cc:160-300
 *         (max_dist = TH_HIGH = 100, q_angle[i] = LastFrame.mvKeysUn[i].angle)
 *     int SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const set<MapPoint*>&, float th, int ORBdist)  :303-440
 *         (max_dist = ORBdist, u_right = NULL, every assigned feature blocks: flags bit1 set, q_angle = pKF->mvKeysUn[i].angle)
 *     int SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*>&, vector<MapPoint*>& vpMatched, int th)
 *         :443-550 (max_dist = TH_LOW = 50, check_ori = 0, occupied = vpMatched[idx] != NULL; KeyFrame::GetFeaturesInArea
 *         src/KeyFrame.cc:637-676 + the level test of :520-521 == level range (level-1, level))
 * mode 1 (best and second best with levels, ratio test) replaces
 *     int SearchByProjection(Frame& F, const vector<MapPoint*>&, float th)                          src/ORBmatcher.cc:73-157
 *         (ratio = mfNNratio, max_dist = TH_HIGH; q_angle unused).
 * q_desc[i] is pMP->GetDescriptor().  kps_un/desc/u_right describe the searched Frame/KeyFrame (mvKeysUn,
 * mDescriptors, mvuRight; u_right may be NULL = no stereo check); occupied[i] != 0 iff the feature already blocks
 * (F.mvpMapPoints[i] && Observations() > 0 on entry, resp. != NULL); grid4 as for orb_match_init.
 * match_cur[i] receives the index of the query now assigned to feature i, -1 if the entry was not touched, and
 * (mode 0) -2 if the rotation filter reset it to NULL.  *nmatches is the reference's return value. */
int orb_match_projection(orb_matcher* m, int mode, const orb_proj_query* queries, const uint8_t* q_desc,
                         const float* q_angle, int nq, const orb_keypoint* kps_un, const uint8_t* desc,
                         const float* u_right, const uint8_t* occupied, int n, const float* grid4, float ratio,
                         int max_dist, int check_ori, int32_t* match_cur, int* nmatches);

/* Independent best candidate per projected point (no state carried between points): the search loops of
 *     int ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, float th)                src/ORBmatcher.cc:1364-1480
 *         (chi2 = 1: reprojection gate e2*mvInvLevelSigma2[level] > 5.99 resp. 7.8 when mvuRight >= 0, :1401-1426;
 *          max_dist = TH_LOW; level range (level-1, level); the Replace/AddObservation surgery stays on the host)
 *     int ORBmatcher::Fuse(KeyFrame*, cv::Mat Scw, const vector<MapPoint*>&, float th, vector<MapPoint*>&)  :1483-1633
 *         (chi2 = 0, max_dist = TH_LOW)
 *     int ORBmatcher::SearchBySim3(KeyFrame*, KeyFrame*, vector<MapPoint*>&, s12, R12, t12, th)  :835-1025
 *         (two calls, one per direction, max_dist = TH_HIGH; the mutual-consistency loop :1008-1022 stays on the host)
 * best_idx[i] = feature index in the searched KeyFrame or -1; best_dist[i] (optional) = its distance (256 = no
 * candidate).  With chi2 = 0 the stereo check of orb_match_projection applies (disable it with er_max = +inf). */
int orb_match_projection_best(orb_matcher* m, const orb_proj_query* queries, const uint8_t* q_desc, int nq,
                              const orb_keypoint* kps_un, const uint8_t* desc, const float* u_right, int n,
                              const float* grid4, int max_dist, int chi2, const float* inv_level_sigma2, int n_levels,
                              int32_t* best_idx, int32_t* best_dist);

/* replaces: int ORBmatcher::SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12,
 * vector<pair<size_t,size_t>>& vMatchedPairs, bool bOnlyStereo), reference src/ORBmatcher.cc:1183-1359 (with
 * CheckDistEpipolarLine :1636-1650).  has_mp[i] != 0 iff GetMapPoint(i) != NULL; u_right = mvuRight (NULL = monocular);
 * F12 row-major 3x3 floats; (ex, ey) the epipole of :1193-1194 (computed by the caller with the reference's cv::Mat
 * expressions); scale_factors2 / level_sigma2_2 = pKF2->mvScaleFactors / mvLevelSigma2.  match_12[i1] = idx2 or -1
 * (the pair list of :1345-1352 is every i1 with match_12[i1] >= 0, ascending); *nmatches = return value. */
int orb_match_triangulation(orb_matcher* m, const orb_keypoint* kps1, const uint8_t* desc1, const uint8_t* has_mp1,
                            const float* u_right1, int n1, const orb_featvec* fv1, const orb_keypoint* kps2,
                            const uint8_t* desc2, const uint8_t* has_mp2, const float* u_right2, int n2,
                            const orb_featvec* fv2, const float* F12, float ex, float ey, const float* scale_factors2,
                            const float* level_sigma2_2, int n_levels, int only_stereo, int check_ori, int32_t* match_12,
                            int* nmatches);

/* Batched device-resident SearchByBoW: pair p matches keyframe kf_index[p] against frame
 * f_index[p] of a feature store that lives in HBM (the Relocalization candidate loop of
 * reference src/Tracking.cc:1471-1492 is the batch axis).  See orb_featstore below. */
typedef struct orb_featstore {
    /* all device pointers; frame f owns rows [f*cap, (f+1)*cap) */
    const uint8_t* desc;        /* [n_frames*cap][32]                       */
    const orb_keypoint* kps;    /* [n_frames*cap] (angle read from here)    */
    const uint8_t* valid;       /* [n_frames*cap] or NULL (= all valid)     */
    const int32_t* counts;      /* [n_frames]                               */
    const uint16_t* node_of;    /* [n_frames*cap] vocabulary node per feature */
    int32_t cap;
    int32_t n_frames;
    int32_t n_nodes;            /* size of the node-index space of node_of (0 = 128, the synthetic stand-in) */
    /* optional (all three or none; NULL = the matcher builds the feature vector of both sides for every pair): the
     * per-frame feature vectors as CSR, filled once per frame by orb_bow_build_csr_device -- the flattened
     * DBoW2::FeatureVector that the reference computes once per Frame / KeyFrame (src/Frame.cc:425-433) */
    const uint32_t* csr_keys;   /* [n_frames*cap]      node << 16 | feature index, grouped by node            */
    const uint16_t* csr_start;  /* [n_frames*n_nodes]  first key of the node                                  */
    const uint16_t* csr_cnt;    /* [n_frames*n_nodes]  features of the frame in the node                      */
    /* optional (ABI 4): the frame's descriptors in csr_keys order -- row p of frame f = desc of feature csr_keys[f*cap + p] & 0xFFFF --
     * filled by orb_bow_build_csr_desc_device; what orb_match_bow_query_device walks (contiguous runs per node) */
    const uint8_t* csr_desc;    /* [n_frames*cap][32] or NULL                                                  */
} orb_featstore;

/* Builds the CSR arrays above for frames [0, n_frames) of node_of / counts (same layout as in the store; pass
 * pointers offset to a frame to (re)build just that frame).  Asynchronous on the matcher's stream. */
int orb_bow_build_csr_device(orb_matcher* m, const uint16_t* d_node_of, const int32_t* d_counts, int n_frames, int cap,
                             int n_nodes, uint32_t* d_keys, uint16_t* d_start, uint16_t* d_cnt);
/* The same, and the node-sorted descriptor copy csr_desc of the same frames (d_desc: their descriptors, [n_frames*cap][32]). */
int orb_bow_build_csr_desc_device(orb_matcher* m, const uint16_t* d_node_of, const int32_t* d_counts, const uint8_t* d_desc,
                                  int n_frames, int cap, int n_nodes, uint32_t* d_keys, uint16_t* d_start, uint16_t* d_cnt,
                                  uint8_t* d_csr_desc);

/* Vocabulary stand-in of SURVEY 8d (the full DBoW2 descent is orb_bow_transform* below): 2-level k=10 tree,
 * centroids = 110 x 32 bytes (device pointer).
 * Fills node_of[f*cap + i] = 11 + 10*c1 + c2 for every feature of every frame. */
int orb_bow_assign_device(orb_matcher* m, const uint8_t* d_desc, const int32_t* d_counts, int n_frames,
                          int cap, const uint8_t* d_centroids_110x32, uint16_t* d_node_of);

/* d_match: [n_pairs][cap] int32 (F-feature -> KF-feature or -1); d_nmatches: [n_pairs].
 * A pair whose kf_index / f_index is outside [0, n_frames) or whose count is outside [0, cap] is not run: its
 * d_nmatches entry is -1 and its d_match row all -1.  Asynchronous on the matcher's stream. */
int orb_match_bow_batch_device(orb_matcher* m, const orb_featstore* store,
                               const int32_t* d_kf_index, const int32_t* d_f_index, int n_pairs,
                               float ratio, int check_ori, int32_t* d_match, int32_t* d_nmatches);

/* One query frame against many keyframes: the candidate loop of Relocalization / DetectLoop itself (reference
 * src/Tracking.cc:1471-1492 calls ORBmatcher::SearchByBoW(vpCandidateKFs[i], mCurrentFrame, ...) once per candidate,
 * src/ORBmatcher.cc:552-687) as ONE call: query q = frame d_f_index[q] of the store against keyframes d_kf_index[0..n_kf).
 * Pair p = q * n_kf + k; d_match [n_queries*n_kf][cap] and d_nmatches [n_queries*n_kf] exactly as
 * orb_match_bow_batch_device would fill them for the pair list (d_kf_index[k], d_f_index[q]) -- same results, but the query
 * side is staged once per vocabulary node instead of once per pair.  Needs the store's CSR arrays including csr_desc
 * (ORB_ERR_INVALID otherwise).  Asynchronous on the matcher's stream. */
int orb_match_bow_query_device(orb_matcher* m, const orb_featstore* store, const int32_t* d_kf_index, int n_kf,
                               const int32_t* d_f_index, int n_queries, float ratio, int check_ori, int32_t* d_match,
                               int32_t* d_nmatches);

/* Diagnostics: a device buffer of `capacity` 64-bit words in which every workgroup of the following
 * orb_match_bow_query_device calls leaves the 100 MHz device clock at its stage boundaries (8 words per workgroup: start,
 * query staged, phase 1 done, phase 2 start / done, finish start, row written; tools/qk_stamps.py).  NULL switches it off. */
int orb_matcher_set_stage_stamps(orb_matcher* m, unsigned long long* d_stamps, size_t capacity);

void* orb_matcher_stream(orb_matcher* m);
int orb_matcher_wait_for(orb_matcher* m, void* hip_stream);

/* replaces the arithmetic of void MapPoint::ComputeDistinctiveDescriptors(), reference src/MapPoint.cc:275-342, for
 * a batch of MapPoints: the observed descriptors of point p are rows offsets[p] .. offsets[p+1] of desc
 * (what :297-304 collects); best_idx[p] receives the index, inside that list, of the descriptor with the least
 * median Hamming distance to the others (first minimum; -1 for an empty list).  <= 256 observations per point. */
int orb_distinctive_descriptors(orb_matcher* m, const uint8_t* desc, const int32_t* offsets, int n_points,
                                int32_t* best_idx);
int orb_distinctive_descriptors_device(orb_matcher* m, const uint8_t* d_desc, const int32_t* d_offsets, int n_points,
                                       int32_t* d_best_idx);

/* ---------------------------------------------------------------- vocabulary tree -----------
 * The descriptor-touching part of DBoW2's TemplatedVocabulary::transform(features, BowVector&, FeatureVector&,
 * levelsup) as called by Frame::ComputeBoW (reference src/Frame.cc:425-433, levelsup = 4) and
 * KeyFrame::ComputeBoW (src/KeyFrame.cc:70): for every feature the word (leaf) it falls into and the node it
 * passes at level L - levelsup (first-minimum Hamming descent, children in stored order).  DBoW2 is a
 * third-party dependency absent from the reference tree; the host still builds the std::map containers.
 * The tree is passed flattened: node 0 = root, children of v = children[child_begin[v] .. child_begin[v+1]),
 * word_id[v] >= 0 for leaves, node_desc = 32 bytes per node (host pointers; copied to the device). */
typedef struct orb_vocab orb_vocab;
int orb_vocab_create(int device_id, const uint8_t* node_desc, const int32_t* child_begin, const int32_t* children,
                     const int32_t* word_id, int n_nodes, int L, orb_vocab** out);
void orb_vocab_destroy(orb_vocab* v);
/* number of nodes at level L - levelsup = size of the compact node-index space the batch matcher works in */
int orb_vocab_level_nodes(orb_vocab* v, int levelsup);
/* host buffers: word_of[i], node_id[i] (-1 when the descent hit a leaf above that level) */
int orb_bow_transform(orb_matcher* m, orb_vocab* v, const uint8_t* desc, int n, int levelsup, int32_t* word_of,
                      int32_t* node_id);
/* device-resident, batched like orb_bow_assign_device; any output pointer may be NULL.  d_node_of receives the
 * COMPACT index of the level node (ascending node id), the form orb_match_bow_batch_device consumes. */
int orb_bow_transform_device(orb_matcher* m, orb_vocab* v, const uint8_t* d_desc, const int32_t* d_counts, int n_frames,
                             int cap, int levelsup, int32_t* d_word_of, int32_t* d_node_id, uint16_t* d_node_of);

/* Frame::ComputeBoW of the query frames + the search, in ONE call: the body of the Relocalization loop as the reference
 * runs it per frame (src/Tracking.cc:1471-1492: mCurrentFrame.ComputeBoW(), then SearchByBoW per candidate keyframe).
 * Frames [first_query, first_query + n_queries) of the store already hold descriptors, keypoints and counts (written by
 * orb_extract_batch_device); this call fills their node_of / csr_* rows IN THE STORE (orb_bow_transform_device with
 * `levelsup`, orb_bow_build_csr_desc_device -- the store's pointers are written through) and then runs
 * orb_match_bow_query_device with d_f_index.  Same results as the three calls; two host round trips less per frame. */
int orb_bow_query_frames_device(orb_matcher* m, orb_vocab* v, const orb_featstore* store, int first_query, int n_queries,
                                int levelsup, const int32_t* d_kf_index, int n_kf, const int32_t* d_f_index, float ratio,
                                int check_ori, int32_t* d_match, int32_t* d_nmatches);


/* ---------------------------------------------------------------- stereo search -------------
 * Additional entry point (Frame.cc links unchanged and keeps its own CPU body): the whole of
 * void Frame::ComputeStereoMatches(), reference src/Frame.cc:513-699, on the pyramids that the LEFT and
 * RIGHT extractor handles hold on the device after their last orb_extract*() call (the reference reads
 * mpORBextractorLeft/Right->mvImagePyramid, :520,611,626,633).  kps/desc are mvKeys/mDescriptors and
 * mvKeysRight/mDescriptorsRight; mb, mbf are Frame::mb, Frame::mbf.  u_right/depth receive mvuRight /
 * mvDepth (n_l floats each, -1 = no match).  Both handles must be on the same device and have seen
 * images of the same size.  Host buffers, synchronous. */
int orb_stereo_match(orb_extractor* left, orb_extractor* right,
                     const orb_keypoint* kps_l, const uint8_t* desc_l, int n_l,
                     const orb_keypoint* kps_r, const uint8_t* desc_r, int n_r,
                     float mb, float mbf, float* u_right, float* depth);
/* Same with device pointers, for frame_l / frame_r of the handles' last batches; asynchronous on the
 * LEFT handle's stream (which is made to wait for the right handle's stream). */
int orb_stereo_match_device(orb_extractor* left, orb_extractor* right, int frame_l, int frame_r,
                            const orb_keypoint* d_kps_l, const uint8_t* d_desc_l, int n_l,
                            const orb_keypoint* d_kps_r, const uint8_t* d_desc_r, int n_r,
                            float mb, float mbf, float* d_u_right, float* d_depth);

/* Diagnostics of the last synchronised batch: per level, how many FAST strips overflowed their candidate queue and were
 * redone by the dense kernel (same results, more time), and how many strips a frame has on that level.  The library
 * shortens the strips of a level that overflows: at every orb_extractor_sync / host call, and -- for device pipelines
 * that never synchronise the handle -- from counters that every fourth orb_extract_batch_device leaves in pinned memory
 * behind an event and a later call picks up once they have arrived (picking them up never waits; a call that thereby
 * shortens a level's strips rebuilds the strip table, which ends in one hipStreamSynchronize of the handle's stream). */
int orb_get_fast_overflows(orb_extractor* h, int32_t* overflowed, int32_t* strips_per_frame);

/* Diagnostics: how the descriptor stage of the current geometry (set by the last extraction call's image size) is split.
 * The reference blurs each pyramid level once and then describes all of the level's keypoints (src/ORBextractor.cc:1118-1136);
 * levels first_level .. nlevels-1 run that way here (level-resident in LDS, as n_regions row tiles over those levels:
 * csrc/orb_desc_level.hip) in batches of at least ORB_DESC_LEVEL_MIN_FRAMES frames (default 24); the levels below, and every
 * level of smaller launches, run one wave per keypoint (csrc/orb_desc.hip).  first_level == nlevels: no level qualifies.
 * Results never depend on the split. */
int orb_extractor_desc_plan(const orb_extractor* h, int32_t* first_level, int32_t* n_regions);

/* Diagnostics: d_stamps (device memory, `capacity` 64-bit words, or NULL to switch off) receives, per workgroup w of the
 * level-resident descriptor kernel of every later batch, 8 words at d_stamps[8 w ..]: the 100 MHz clock at its phase
 * boundaries (start, staged, IC_Angle done, angles done, blur computed, blur written, samples done) and
 * (region << 32 | keypoints).  Workgroup w = region + n_regions * frame.  tools/dl_stamps.py prints the table. */
int orb_extractor_set_desc_stamps(orb_extractor* h, unsigned long long* d_stamps, size_t capacity);

/* Diagnostics of the pyramid kernels (k_pyr_chain): d_stamps receives, launch after launch of a later batch, 8 words per
 * workgroup (band b of frame f of a launch: 8 * (f * bands + b)): the 100 MHz clock at 0 start, 1 source rows requested and
 * stored, 2 staged (barrier), 3 + k level k of the chain written (barrier).  orb_extractor_pyr_stamp_layout tells how many
 * launches the last batch had and their bands per frame / levels per launch.  tools/pyr_stamps.py prints the table. */
int orb_extractor_set_pyr_stamps(orb_extractor* h, unsigned long long* d_stamps, size_t capacity);
/* The same for the quadtree kernel (one workgroup per (frame, level): workgroup = level * frames + frame): 0 start, 1 keys sorted,
 * 2 full passes done, 3 careful phase done, 4 keypoints emitted, word 5 = candidates | careful iterations << 16 | expandable nodes at
 * the first careful iteration << 32 | list size there << 48.  The caller sizes d_stamps for 8 * frames * levels words of the batches
 * it runs while the stamps are on; process-wide (one diagnostic user at a time).  tools/qt_stamps.py prints the table. */
int orb_extractor_set_qt_stamps(orb_extractor* h, unsigned long long* d_stamps, size_t capacity);
int orb_extractor_pyr_stamp_layout(const orb_extractor* h, int32_t* n_chains, int32_t* bands8, int32_t* steps8);
/* How many pyramid launches of the last batch took the persistent, prefetching form (k_pyr_chain_p: batches that fill the chip
 * several times over, and only with ORB_PYR_PERSIST=1: measured neutral, off by default).  In that form a stamp record belongs to a (frame, band) unit, same index. */
int orb_extractor_pyr_persistent(const orb_extractor* h, int32_t* launches);

/* The whole pyramid of device-resident frame `frame` of the last batch with ONE device-to-host copy and one
 * synchronisation (the reference keeps it in the public member mvImagePyramid, include/ORBextractor.h:86, read by
 * Frame::ComputeStereoMatches, src/Frame.cc:520,611,626,633).  Level l of the copy starts at dst + offsets[l], has
 * rows[l] x cols[l] pixels and a row pitch of pitches[l] bytes (interiors only, no 19-px border).  Call with
 * dst = NULL to learn the byte size needed (*bytes) and the level layout.  dst may be pageable or pinned
 * (orb_host_alloc). */
int orb_get_pyramid(orb_extractor* h, int frame, uint8_t* dst, size_t dst_bytes, size_t* bytes, int32_t* offsets,
                    int32_t* pitches, int32_t* rows, int32_t* cols);

/* Pinned (page-locked) host memory for callers that keep staging buffers across calls (the C++ shims): copies to
 * and from it are asynchronous and run at full PCIe rate, and orb_extract_batch skips its own staging for it. */
void* orb_host_alloc(size_t bytes);
void orb_host_free(void* p);

/* ---------------------------------------------------------------- batched frames over the GPUs of one node ----
 * BASELINE.json configs[3] / north_star: "a batched-frames mode shards independent frames across the 8 GPUs of one
 * node with RCCL broadcast of the BRIEF pattern over xGMI".  One extractor handle and one host thread per entry of
 * `devices`; frames are cut into contiguous blocks (orb_shard_range) with no data-path collective; the pattern goes
 * from devices[0] to the others with ncclBroadcast (librccl is opened at run time).  The same device may be listed
 * more than once (several handles on one GPU).  Frame.cc / Tracking.cc never see this: it is the batch entry a
 * dataset-processing caller uses instead of a loop over ORBextractor::operator() (src/ORBextractor.cc:1084-1150). */
typedef struct orb_multi orb_multi;
int orb_multi_create(const orb_extractor_params* p, const int* devices, int n_devices, orb_multi** out);
void orb_multi_destroy(orb_multi* m);
int orb_multi_devices(const orb_multi* m);
orb_extractor* orb_multi_handle(orb_multi* m, int i);          /* handle i (e.g. for orb_get_pyramid on its last chunk) */
/* host pattern -> devices[0] -> RCCL broadcast -> every handle */
int orb_multi_set_pattern(orb_multi* m, const int8_t* pattern_xy_1024);
/* orb_extract_batch over all devices: same buffers, layouts and results as the single-GPU call */
int orb_multi_extract_batch(orb_multi* m, const uint8_t* imgs, int n_frames, int rows, int cols, size_t row_stride,
                            size_t frame_stride, orb_keypoint* kps, uint8_t* desc32, int cap, int32_t* counts);
/* the partition rule: rank r of `world` owns frames [first, first + count) of `total` (host only, no device needed) */
void orb_shard_range(int total, int world, int rank, int* first, int* count);

/* BASELINE configs[4] over several GPUs: a keyframe descriptor database sharded BY KEYFRAME over `devices`
 * (contiguous blocks, orb_shard_range), the query frame replicated, no exchange between the GPUs -- every (keyframe, frame)
 * result of the candidate loop of reference src/Tracking.cc:1471-1492 / src/ORBmatcher.cc:552-687 is independent; each
 * shard's host thread writes its keyframes' rows of the caller's result arrays.  Arrays are laid out like an orb_featstore
 * (host pointers): frame f owns rows [f*cap, (f+1)*cap); valid may be NULL (= all valid); node_of = compact vocabulary
 * node per feature (orb_bow_transform*).  The same device may be listed more than once.  UNMEASURED on more than one
 * physical GPU (the build's GPU boxes have one). */
typedef struct orb_multi_db orb_multi_db;
int orb_multi_db_create(const int* devices, int n_devices, const uint8_t* desc, const orb_keypoint* kps, const uint8_t* valid,
                        const int32_t* counts, const uint16_t* node_of, int n_kf, int cap, int n_nodes, orb_multi_db** out);
void orb_multi_db_destroy(orb_multi_db* db);
int orb_multi_db_shards(const orb_multi_db* db);
/* one query frame (q_count <= cap features) against every keyframe of the database: match[kf*cap + iF] = keyframe feature
 * matched to query feature iF or -1, nmatches[kf] = the reference's return value for that keyframe */
int orb_multi_match_bow_batch(orb_multi_db* db, const uint8_t* q_desc, const orb_keypoint* q_kps, int q_count,
                              const uint16_t* q_node_of, float ratio, int check_ori, int32_t* match, int32_t* nmatches);

/* Batch of stereo pairs in ONE launch: pair p = frames (first_frame_l + p, first_frame_r + p) of the two handles' last
 * batches, its keypoints / descriptors / results at rows [p * cap, (p + 1) * cap) of the arrays orb_extract_batch_device
 * wrote, its keypoint counts read on the device from the extractors' d_counts (no host round trip between extraction
 * and search).  d_u_right / d_depth: [n_pairs * cap] floats.  Asynchronous on the LEFT handle's stream; it waits for the
 * right handle's stream first, and work issued on the RIGHT handle afterwards (the next extraction, which overwrites the
 * pyramid and the keypoints the search reads) is ordered behind the search.  The same holds for orb_stereo_match_device. */
int orb_stereo_match_batch_device(orb_extractor* left, orb_extractor* right, int first_frame_l, int first_frame_r,
                                  int n_pairs, const orb_keypoint* d_kps_l, const uint8_t* d_desc_l,
                                  const int32_t* d_counts_l, const orb_keypoint* d_kps_r, const uint8_t* d_desc_r,
                                  const int32_t* d_counts_r, int cap, float mb, float mbf, float* d_u_right, float* d_depth);

/* ---------------------------------------------------------------- misc ---------------------*/
const char* orb_last_error(void);   /* thread-local description of the last failure */
const char* orb_version(void);
/* ABI guard: the structs of this header grow between releases (orb_featstore gained the CSR pointers in 2, csr_desc in 4).  A caller
 * built against an older header would be read past the end of its struct: compare orb_abi_version() with the
 * ORB_HIP_ABI_VERSION it was compiled with and orb_sizeof_featstore() with sizeof(orb_featstore) before the first call. */
#define ORB_HIP_ABI_VERSION 4
int orb_abi_version(void);
size_t orb_sizeof_featstore(void);

#ifdef __cplusplus
}
#endif
#endif /* ORB_HIP_H */
