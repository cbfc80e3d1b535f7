// ORBextractor.cc -- drop-in body for the reference's src/ORBextractor.cc:498-559 (ctor) and
// :1084-1150 (operator()) on top of liborbhip.so.  Everything between (pyramid, FAST, quadtree,
// orientation, blur, rBRIEF) happens in HIP kernels behind orb_extract().
#include "ORBextractor.h"

#include <cassert>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

#include "orb_hip.h"

namespace ORB_SLAM2
{

static_assert(sizeof(cv::KeyPoint) == sizeof(orb_keypoint), "cv::KeyPoint must be the 28-byte POD the C ABI writes");

static int gDefaultDevice = -1;
void ORBextractor::SetDefaultDevice(int device) { gDefaultDevice = device; }
static int gGaussPreset = -1;
void ORBextractor::SetGaussianPreset(int preset) { gGaussPreset = preset; }
static int defaultDevice()
{
    if (gDefaultDevice >= 0) return gDefaultDevice;
    const char* e = std::getenv("ORB_HIP_DEVICE");
    return e ? std::atoi(e) : 0;
}

static void orbCheck(int rc, const char* what)
{
    if (rc != ORB_OK)
        throw std::runtime_error(std::string("ORBextractor(HIP): ") + what + " failed: " + orb_last_error());
}

ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
    : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST),
      minThFAST(_minThFAST), mpHandle(nullptr), mbDownloadPyramid(true), mpStage(nullptr), mnStageBytes(0), mnPyramidBytes(0)
{
    orb_extractor_params p;
    p.nfeatures = _nfeatures;
    p.scale_factor = _scaleFactor;
    p.nlevels = _nlevels;
    p.ini_th_fast = _iniThFAST;
    p.min_th_fast = _minThFAST;
    orbCheck(orb_extractor_create(&p, defaultDevice(), &mpHandle), "orb_extractor_create");
    {
        const char* e = std::getenv("ORB_HIP_GAUSS");
        const int preset = gGaussPreset >= 0 ? gGaussPreset : (e ? std::atoi(e) : 0);
        if (preset != 0) {
            int32_t taps[4];
            orbCheck(orb_gaussian_preset(preset, taps), "orb_gaussian_preset");
            orbCheck(orb_extractor_set_gaussian(mpHandle, taps), "orb_extractor_set_gaussian");
        }
    }
    mvScaleFactor.resize(nlevels);
    mvInvScaleFactor.resize(nlevels);
    mvLevelSigma2.resize(nlevels);
    mvInvLevelSigma2.resize(nlevels);
    mnFeaturesPerLevel.resize(nlevels);
    orbCheck(orb_extractor_get_tables(mpHandle, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(),
                                      mvInvLevelSigma2.data(), mnFeaturesPerLevel.data()), "orb_extractor_get_tables");
    mvImagePyramid.resize(nlevels);
}

ORBextractor::~ORBextractor()
{
    mvImagePyramid.clear();
    orb_host_free(mpStage);
    orb_extractor_destroy(mpHandle);
}

void ORBextractor::operator()(cv::InputArray _image, cv::InputArray /*_mask*/, std::vector<cv::KeyPoint>& _keypoints,
                              cv::OutputArray _descriptors)
{
    if (_image.empty())
        return;                                              // reference :1087-1088
    cv::Mat image = _image.getMat();
    assert(image.type() == CV_8UC1);                         // reference :1091

    const int cap = orb_extractor_max_keypoints(mpHandle);
    // staging layout: [keypoints cap x 28 | descriptors cap x 32 | pyramid slab]; grown on demand, kept across calls
    const size_t kpB = sizeof(orb_keypoint) * (size_t)cap, dsB = (size_t)cap * ORB_DESC_BYTES;
    auto ensureStage = [&](size_t pyrBytes) {
        const size_t need = kpB + dsB + pyrBytes;
        if (need <= mnStageBytes) return;
        for (auto& m : mvImagePyramid) m.release();          // headers over the old buffer
        orb_host_free(mpStage);
        mpStage = orb_host_alloc(need);
        mnStageBytes = mpStage ? need : 0;
        if (!mpStage) throw std::runtime_error("ORBextractor(HIP): pinned staging allocation failed");
    };
    ensureStage(mnPyramidBytes);
    orb_keypoint* kps = static_cast<orb_keypoint*>(mpStage);
    unsigned char* desc = static_cast<unsigned char*>(mpStage) + kpB;
    int n = 0;
    orbCheck(orb_extract(mpHandle, image.data, image.rows, image.cols, (size_t)image.step, kps, desc, cap, &n), "orb_extract");

    if (n == 0)
        _descriptors.release();                              // reference :1107-1108
    else {
        _descriptors.create(n, 32, CV_8U);
        cv::Mat d = _descriptors.getMat();
        if (d.isContinuous())
            std::memcpy(d.data, desc, (size_t)n * ORB_DESC_BYTES);
        else
            for (int i = 0; i < n; i++) std::memcpy(d.ptr<unsigned char>(i), desc + (size_t)i * ORB_DESC_BYTES, ORB_DESC_BYTES);
    }
    _keypoints.clear();
    _keypoints.resize(n);
    if (n)
        std::memcpy(static_cast<void*>(_keypoints.data()), kps, sizeof(orb_keypoint) * (size_t)n);

    if (mbDownloadPyramid) {
        // ONE device-to-host copy and one synchronisation for all levels; mvImagePyramid[level] are headers over the
        // staging buffer (valid until the next operator() call, like the reference's own buffers)
        size_t bytes = 0;
        std::vector<int32_t> off(nlevels), pitch(nlevels), r(nlevels), c(nlevels);
        orbCheck(orb_get_pyramid(mpHandle, 0, nullptr, 0, &bytes, off.data(), pitch.data(), r.data(), c.data()), "orb_get_pyramid");
        if (bytes != mnPyramidBytes || kpB + dsB + bytes > mnStageBytes) {
            mnPyramidBytes = bytes;
            ensureStage(bytes);
        }
        unsigned char* pyr = static_cast<unsigned char*>(mpStage) + kpB + dsB;
        orbCheck(orb_get_pyramid(mpHandle, 0, pyr, bytes, &bytes, off.data(), pitch.data(), r.data(), c.data()), "orb_get_pyramid");
        for (int level = 0; level < nlevels; ++level)
            mvImagePyramid[level] = cv::Mat(r[level], c[level], CV_8UC1, pyr + off[level], (size_t)pitch[level]);
    }
}

} //namespace ORB_SLAM
