// Tiled Schur-complement accumulation (schur_impl = 1).  See DESIGN.md §Kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "ba_point_kernels.hpp"

struct rsba_solver;

namespace rsba {

class KernelTimer;

struct TiledSchur {
  int Build(int C, int P, const std::vector<int>& pt_ptr, const std::vector<int>& obs_cam);
  int Launch(rsba_solver* s, const IterParams& ip, KernelTimer& T);
  void Free();
};

}  // namespace rsba
