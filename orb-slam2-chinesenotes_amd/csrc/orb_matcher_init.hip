// orb_matcher_init.hip -- SearchForInitialization (reference src/ORBmatcher.cc:1055-1180).
#include "orb_common.h"

extern "C" int orb_match_init(orb_matcher* m, const orb_keypoint* kps1, const uint8_t* desc1, int n1,
                              const orb_keypoint* kps2, const uint8_t* desc2, int n2, const float* grid4,
                              float* prev_xy, int window_size, float ratio, int check_ori, int32_t* match_12,
                              int* nmatches)
{
    (void)m; (void)kps1; (void)desc1; (void)n1; (void)kps2; (void)desc2; (void)n2; (void)grid4; (void)prev_xy;
    (void)window_size; (void)ratio; (void)check_ori; (void)match_12; (void)nmatches;
    orb_set_error("orb_match_init: not built yet");
    return ORB_ERR_UNSUPPORTED;
}
