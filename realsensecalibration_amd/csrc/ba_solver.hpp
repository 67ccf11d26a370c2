// Internal declarations shared by the solver translation units.
#pragma once
#include "../../include/rsba.h"

struct rsba_solver;

namespace rsba {
int DeviceCount();
}  // namespace rsba
