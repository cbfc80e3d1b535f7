// ORBextractor.cc -- drop-in body for the reference's src/ORBextractor.cc:498-559 (ctor) and
// :1084-1150 (operator()) on top of liborbhip.so.  Everything between (pyramid, FAST, quadtree,
// orientation, blur, rBRIEF) happens in HIP kernels behind orb_extract().
#include "ORBextractor.h"

#include <cassert>
#include <cstring>
#include <stdexcept>
#include <string>

#include "orb_hip.h"

namespace ORB_SLAM2
{

static_assert(sizeof(cv::KeyPoint) == sizeof(orb_keypoint), "cv::KeyPoint must be the 28-byte POD the C ABI writes");

static void orbCheck(int rc, const char* what)
{
    if (rc != ORB_OK)
        throw std::runtime_error(std::string("ORBextractor(HIP): ") + what + " failed: " + orb_last_error());
}

ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
    : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST),
      minThFAST(_minThFAST), mpHandle(nullptr), mbDownloadPyramid(true)
{
    orb_extractor_params p;
    p.nfeatures = _nfeatures;
    p.scale_factor = _scaleFactor;
    p.nlevels = _nlevels;
    p.ini_th_fast = _iniThFAST;
    p.min_th_fast = _minThFAST;
    orbCheck(orb_extractor_create(&p, 0, &mpHandle), "orb_extractor_create");
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
    std::vector<orb_keypoint> kps(cap);
    std::vector<unsigned char> desc((size_t)cap * ORB_DESC_BYTES);
    int n = 0;
    orbCheck(orb_extract(mpHandle, image.data, image.rows, image.cols, (size_t)image.step, kps.data(), desc.data(),
                         cap, &n), "orb_extract");

    if (n == 0)
        _descriptors.release();                              // reference :1107-1108
    else {
        _descriptors.create(n, 32, CV_8U);
        cv::Mat d = _descriptors.getMat();
        for (int i = 0; i < n; i++)
            std::memcpy(d.ptr<unsigned char>(i), &desc[(size_t)i * ORB_DESC_BYTES], ORB_DESC_BYTES);
    }
    _keypoints.clear();
    _keypoints.resize(n);
    if (n)
        std::memcpy(static_cast<void*>(_keypoints.data()), kps.data(), sizeof(orb_keypoint) * (size_t)n);

    if (mbDownloadPyramid) {
        for (int level = 0; level < nlevels; ++level) {
            int r = 0, c = 0;
            orbCheck(orb_get_pyramid_level(mpHandle, 0, level, nullptr, 0, &r, &c), "orb_get_pyramid_level");
            mvImagePyramid[level].create(r, c, CV_8UC1);
            orbCheck(orb_get_pyramid_level(mpHandle, 0, level, mvImagePyramid[level].data,
                                           (size_t)mvImagePyramid[level].step, &r, &c), "orb_get_pyramid_level");
        }
    }
}

} //namespace ORB_SLAM
