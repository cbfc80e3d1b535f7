// BowHip.h -- shared by FrameHip.cc and KeyFrameHip.cc: DBoW2's TemplatedVocabulary::transform(features, BowVector&,
// FeatureVector&, levelsup) (as called by Frame::ComputeBoW, reference src/Frame.cc:425-433, and KeyFrame::ComputeBoW,
// src/KeyFrame.cc:64-73) with the descriptor-touching part -- the tree descent -- on the GPU (orb_bow_transform).
// Written against DBoW2's real interface (Thirdparty/DBoW2/DBoW2/{BowVector,FeatureVector,TemplatedVocabulary}.h):
// BowVector::addWeight / addIfNotExist / normalize(LNorm), FeatureVector::addFeature, getWeightingType / getScoringType /
// getDepthLevels; the PROTECTED node table is read through a class derived from the vocabulary.
#pragma once
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "ORBVocabulary.h"
#include "orb_hip.h"

namespace ORB_SLAM2
{
namespace hipbow
{
inline void check(int rc, const char* what)
{
    if (rc != ORB_OK) throw std::runtime_error(std::string("BoW(HIP): ") + what + " failed: " + orb_last_error());
}
inline int hipDevice()
{
    const char* e = std::getenv("ORB_HIP_DEVICE");
    return e ? std::atoi(e) : 0;
}
struct MatcherHandle {
    orb_matcher* m = nullptr;
    MatcherHandle() { check(orb_matcher_create(hipDevice(), &m), "orb_matcher_create"); }
    ~MatcherHandle() { orb_matcher_destroy(m); }
};
inline orb_matcher* matcher()
{
    static thread_local MatcherHandle h;
    return h.m;
}

struct DeviceVocabulary {
    orb_vocab* v = nullptr;
    std::vector<double> wordWeight;              // by word id
};

// TemplatedVocabulary keeps Node and m_nodes protected: a derived class may name and read them
struct VocabularyAccess : public ORBVocabulary {
    void flatten(std::vector<unsigned char>& desc, std::vector<int32_t>& childBegin, std::vector<int32_t>& children,
                 std::vector<int32_t>& wordId, std::vector<double>& wordWeight) const
    {
        const int n = (int)m_nodes.size();
        desc.assign((size_t)n * 32, 0);
        childBegin.assign(n + 1, 0);
        children.clear();
        wordId.assign(n, -1);
        wordWeight.clear();
        for (int i = 0; i < n; i++) {            // node i of m_nodes has id i (DBoW2 invariant)
            const Node& nd = m_nodes[i];
            if (!nd.descriptor.empty()) std::memcpy(&desc[(size_t)i * 32], nd.descriptor.ptr<unsigned char>(0), 32);
            for (size_t k = 0; k < nd.children.size(); k++) children.push_back((int32_t)nd.children[k]);
            childBegin[i + 1] = (int32_t)children.size();
            if (nd.isLeaf() && i != 0) {
                wordId[i] = (int32_t)nd.word_id;
                if (wordWeight.size() <= nd.word_id) wordWeight.resize(nd.word_id + 1, 0.0);
                wordWeight[nd.word_id] = nd.weight;
            }
        }
    }
};

// one flattened copy per vocabulary object (the reference loads ONE ORBVocabulary at start-up and shares it)
inline DeviceVocabulary& deviceVocabulary(ORBVocabulary* voc)
{
    static std::mutex mu;
    static std::map<ORBVocabulary*, DeviceVocabulary> cache;
    std::lock_guard<std::mutex> lock(mu);
    std::map<ORBVocabulary*, DeviceVocabulary>::iterator it = cache.find(voc);
    if (it != cache.end()) return it->second;
    std::vector<unsigned char> desc;
    std::vector<int32_t> childBegin, children, wordId;
    DeviceVocabulary dv;
    static_cast<const VocabularyAccess*>(voc)->flatten(desc, childBegin, children, wordId, dv.wordWeight);
    check(orb_vocab_create(hipDevice(), desc.data(), childBegin.data(), children.data(), wordId.data(), (int)wordId.size(),
                           voc->getDepthLevels(), &dv.v), "orb_vocab_create");
    return cache[voc] = dv;
}

// == voc->transform(Converter::toDescriptorVector(descriptors), v, fv, levelsup), TemplatedVocabulary.h
inline void transform(ORBVocabulary* voc, const cv::Mat& descriptors, DBoW2::BowVector& v, DBoW2::FeatureVector& fv, int levelsup)
{
    v.clear();
    fv.clear();
    const int n = descriptors.rows;
    if (n == 0 || voc->empty()) return;
    DeviceVocabulary& dv = deviceVocabulary(voc);
    std::vector<int32_t> word(n), node(n);
    check(orb_bow_transform(matcher(), dv.v, descriptors.data, n, levelsup, word.data(), node.data()), "orb_bow_transform");
    // what GeneralScoring::mustNormalize answers for the vocabulary's scoring type (ScoringObject.h)
    const DBoW2::ScoringType sc = voc->getScoringType();
    const bool must = sc != DBoW2::DOT_PRODUCT;
    const DBoW2::LNorm norm = sc == DBoW2::L2_NORM ? DBoW2::L2 : DBoW2::L1;
    const DBoW2::WeightingType wt = voc->getWeightingType();
    const bool tf = wt == DBoW2::TF || wt == DBoW2::TF_IDF;
    for (int i = 0; i < n; i++) {
        const double w = (word[i] >= 0 && (size_t)word[i] < dv.wordWeight.size()) ? dv.wordWeight[word[i]] : 0.0;
        if (w > 0) {                              // (not stopped words)
            if (tf) v.addWeight((DBoW2::WordId)word[i], w);
            else v.addIfNotExist((DBoW2::WordId)word[i], w);
            fv.addFeature((DBoW2::NodeId)node[i], (unsigned int)i);
        }
    }
    if (tf && !v.empty() && !must) {              // unnecessary when normalizing
        const double nd = (double)v.size();
        for (DBoW2::BowVector::iterator vit = v.begin(); vit != v.end(); ++vit) vit->second /= nd;
    }
    if (must) v.normalize(norm);
}
}  // namespace hipbow
}  // namespace ORB_SLAM2
