// blaze.hip — fused BlazeBlock kernel (placeholder until the fused kernel lands; the planner emits
// DWCONV + CONV for BlazeBlocks unless asked for the fused op).
#include "common.h"

int fp_launch_blazeblock(const fp_op&, const float*, float*, hipStream_t) { return FP_ERR_UNSUPPORTED; }
