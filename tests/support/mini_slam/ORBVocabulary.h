// TEST DOUBLE: the reference's include/ORBVocabulary.h stand-in lives in ORBmatcher.h of this directory (one header for all the
// SLAM types the shims touch).
#pragma once
#include "ORBmatcher.h"
