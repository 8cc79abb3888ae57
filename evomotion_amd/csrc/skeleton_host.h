// Host-side skeleton loader: parses the decoded fixture and derives every load-time constant the
// reference computes with GLM / Bullet when it builds the world
// (evo_motion_model/src/env/robot_walk.cpp:17-46, src/robot/skeleton.cpp:27-103, src/item.cpp:17-41,
//  src/robot/constraint.cpp:52-69,137-150, src/robot/muscle.cpp:14-68).
#pragma once
#include <string>

#include "../../include/evomotion.h"
#include "skel_const.h"

namespace evm {
// Returns EVM_OK or an EVM_E_* code with `err` filled.
int load_skeleton_constants(const char *path, const EvmEnvParams &prm, EvmSkelC &out, std::string &err);
// Lane-group sweep schedule for `nwaves` waves per 16-env workgroup (skel_const.h, EvmGSchedC); EVM_E_UNSUPPORTED when the
// skeleton's records do not fit the LDS image (the caller then keeps the 64-env tile kernel).
int build_group_schedule(const EvmSkelC &S, int nwaves, EvmGSchedC &out, std::string &err);
}
