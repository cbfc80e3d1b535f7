// orb_kernels.h -- launch wrappers of the gfx950 kernels (defined in the *.hip translation units).
#pragma once
#include "orb_common.h"

void orb_launch_copy_level0(hipStream_t st, const uint8_t* src, size_t rowStride, size_t frameStride,
                            uint8_t* pyr, size_t pyrSlab, int w, int h, int pitch, int nFrames, int* clr, int clrInts, const int8_t* pat8, float* patF);
bool orb_launch_resize_pair(hipStream_t st, uint8_t* pyr, size_t pyrSlab, const OrbLevelGeom& S, const OrbLevelGeom& M,
                            const OrbLevelGeom& D, const uint4* xqM, const int2* ytabM, const uint4* xqD, const int2* ytabD,
                            int nFrames);
void orb_launch_resize(hipStream_t st, uint8_t* pyr, size_t pyrSlab, const OrbLevelGeom& src,
                       const OrbLevelGeom& dst, const int2* xtab, const int2* ytab, const uint4* xq, int nFrames);
bool orb_launch_pyr_chain(hipStream_t st, const OrbPyrChain& C, const uint8_t* img, size_t rowStride, size_t frameStride,
                          uint8_t* pyr, size_t pyrSlab, const uint4* xqAll, const int2* ytAll, const int2* bandTab, int nFrames,
                          int* clr, int clrInts, const int8_t* pat8, float* patF, unsigned long long* stamps = nullptr, bool persist = false);
size_t orb_fast_lds_bytes(int pdw, int rowsMax, int candCap);
size_t orb_fast_dense_lds_bytes(int pdw, int rowsMax, int sdw);
void orb_launch_fast_strips(hipStream_t st, const OrbGeom& G, const uint8_t* pyr, size_t pyrSlab,
                            const OrbStrip* strips, int nStrips, const uint32_t* pathTab, unsigned long long* cand,
                            size_t candSlab, int* candCount, int* errFlags, int* ovfCount, int* ovfList, int iniTh, int minTh,
                            int pdw, int rowsMax, int sdw, int candCap, int nFrames, int fixedPitch, bool skipDense = false);
size_t orb_fast_p_lds_bytes(int P, int rowsMax, int candCap);
size_t orb_quadtree_lds_bytes(int sortCap, int nodeCap);
void orb_launch_quadtree(hipStream_t st, const OrbGeom& G, unsigned long long* cand, size_t candSlab,
                         const int* candCount, uint32_t* kpl, int* kpCount, int* errFlags, int sortCap,
                         int nodeCap, int nFrames, int* ovfBlock, unsigned char* globalScratch);
size_t orb_quadtree_scratch_stride(int nodeCap);
int orb_quadtree_set_stamps(unsigned long long* d_stamps, hipStream_t st);
void orb_launch_orient_desc(hipStream_t st, const OrbGeom& G, const uint8_t* pyr, size_t pyrSlab,
                            const uint32_t* kpl, const int* kpCount, const float* patternF, const uint4* angTab, const uint32_t* hbTab,
                            orb_keypoint* kps, uint8_t* desc, int cap, int32_t* counts, int* errFlags,
                            int nFrames, const int* gaussTaps4 = nullptr, int slotLimit = 0);
void orb_desc_hblur_table(uint32_t* tab768);
// level-resident descriptor stage (orb_desc_level.hip): the plan for a geometry, and the launch over the plan's regions
void orb_desc_level_plan(const OrbGeom& G, OrbDescPlan* P);
int orb_launch_desc_level(hipStream_t st, const OrbGeom& G, const OrbDescPlan& P, const uint8_t* pyr, size_t pyrSlab, const uint32_t* kpl,
                          const int* kpCount, const float* patternF, const uint4* angTab, orb_keypoint* kps, uint8_t* desc, int cap,
                          int nFrames, const int* gaussTaps4, unsigned long long* stamps = nullptr, size_t stampCap = 0);
