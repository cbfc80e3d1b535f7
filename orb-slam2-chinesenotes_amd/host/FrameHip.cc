// FrameHip.cc -- replacement bodies for two member functions of the reference's ORB_SLAM2::Frame, compiled against the
// reference's own, unchanged include/Frame.h (optional bindings):
//
//   void Frame::ComputeStereoMatches()      src/Frame.cc:513-699   -> orb_stereo_match on the pyramids that the two
//                                                                     extractor handles keep on the device
//   void Frame::ComputeBoW()                src/Frame.cc:425-433   -> orb_bow_transform (the descriptor-touching part of
//                                                                     DBoW2's transform(features, BowVector&, FeatureVector&, 4));
//                                                                     the std::map containers are filled here
// Build: wrap the two definitions of src/Frame.cc in `#ifndef ORB_HIP_FRAME` and add this file.  No CPU fallback.
#include <cstdlib>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>

#include "Frame.h"
#include "ORBextractor.h"
#include "orb_hip.h"

namespace ORB_SLAM2
{

namespace
{
void check(int rc, const char* what)
{
    if (rc != ORB_OK) throw std::runtime_error(std::string("Frame(HIP): ") + what + " failed: " + orb_last_error());
}
int hipDevice()
{
    const char* e = std::getenv("ORB_HIP_DEVICE");
    return e ? std::atoi(e) : 0;
}
struct MatcherHandle {
    orb_matcher* m = nullptr;
    MatcherHandle() { check(orb_matcher_create(hipDevice(), &m), "orb_matcher_create"); }
    ~MatcherHandle() { orb_matcher_destroy(m); }
};
orb_matcher* matcher()
{
    static thread_local MatcherHandle h;
    return h.m;
}

// DBoW2 keeps its node table protected (TemplatedVocabulary::m_nodes): a derived class may read it
struct VocabularyAccess : public ORBVocabulary {
    const std::vector<Node>& nodes() const { return m_nodes; }
};
struct DeviceVocabulary {
    orb_vocab* v = nullptr;
    std::vector<double> wordWeight;              // by word id
};
// one flattened copy per vocabulary object (the reference loads ONE ORBVocabulary at start-up and shares it)
DeviceVocabulary& deviceVocabulary(ORBVocabulary* voc)
{
    static std::mutex mu;
    static std::map<ORBVocabulary*, DeviceVocabulary> cache;
    std::lock_guard<std::mutex> lock(mu);
    std::map<ORBVocabulary*, DeviceVocabulary>::iterator it = cache.find(voc);
    if (it != cache.end()) return it->second;
    const std::vector<DBoW2::Vocabulary::Node>& nodes = static_cast<VocabularyAccess*>(voc)->nodes();
    const int n = (int)nodes.size();
    std::vector<unsigned char> desc((size_t)n * 32, 0);
    std::vector<int32_t> childBegin(n + 1, 0), children, wordId(n, -1);
    DeviceVocabulary dv;
    for (int i = 0; i < n; i++) {                // node i of m_nodes has id i (DBoW2 invariant)
        if (!nodes[i].descriptor.empty()) std::memcpy(&desc[(size_t)i * 32], nodes[i].descriptor.ptr<unsigned char>(0), 32);
        for (size_t k = 0; k < nodes[i].children.size(); k++) children.push_back((int32_t)nodes[i].children[k]);
        childBegin[i + 1] = (int32_t)children.size();
        if (nodes[i].isLeaf() && i != 0) {
            wordId[i] = (int32_t)nodes[i].word_id;
            if (dv.wordWeight.size() <= nodes[i].word_id) dv.wordWeight.resize(nodes[i].word_id + 1, 0.0);
            dv.wordWeight[nodes[i].word_id] = nodes[i].weight;
        }
    }
    check(orb_vocab_create(hipDevice(), desc.data(), childBegin.data(), children.data(), wordId.data(), n, voc->getDepthLevels(), &dv.v),
          "orb_vocab_create");
    return cache[voc] = dv;
}
}  // namespace

void Frame::ComputeStereoMatches()
{
    mvuRight = std::vector<float>(N, -1.0f);      // :515-516
    mvDepth = std::vector<float>(N, -1.0f);
    if (N == 0) return;
    static_assert(sizeof(cv::KeyPoint) == sizeof(orb_keypoint), "cv::KeyPoint layout");
    check(orb_stereo_match(mpORBextractorLeft->Handle(), mpORBextractorRight->Handle(),
                           reinterpret_cast<const orb_keypoint*>(mvKeys.data()), mDescriptors.data, N,
                           reinterpret_cast<const orb_keypoint*>(mvKeysRight.data()), mDescriptorsRight.data, (int)mvKeysRight.size(), mb,
                           mbf, mvuRight.data(), mvDepth.data()), "orb_stereo_match");
}

void Frame::ComputeBoW()
{
    if (!mBowVec.empty()) return;                 // :427
    const int n = mDescriptors.rows;
    if (n == 0) return;
    DeviceVocabulary& dv = deviceVocabulary(mpORBvocabulary);
    std::vector<int32_t> word(n), node(n);
    check(orb_bow_transform(matcher(), dv.v, mDescriptors.data, n, 4, word.data(), node.data()), "orb_bow_transform");
    // DBoW2 TemplatedVocabulary::transform(features, v, fv, levelsup) for TF_IDF weighting / L1 scoring (ORBvoc):
    // v[word] += weight, fv[node].push_back(i) for features whose word has a positive weight, then v is L1-normalised
    for (int i = 0; i < n; i++) {
        const double w = (word[i] >= 0 && (size_t)word[i] < dv.wordWeight.size()) ? dv.wordWeight[word[i]] : 0.0;
        if (w > 0) {
            mBowVec.addWeight((DBoW2::WordId)word[i], w);
            mFeatVec.addFeature((DBoW2::NodeId)node[i], (unsigned int)i);
        }
    }
    mBowVec.normalizeL1();
}

}  // namespace ORB_SLAM2
