// SYNTAX-CHECK STAND-IN, not gtsam (see gtsam/linear/HessianFactor.h next to this file): gtsam::Values with the two
// members integration/sage_gtsam_prepass.h calls (gtsam/nonlinear/Values.h: exists(Key), at<ValueType>(Key)).
#pragma once
#include "gtsam/linear/HessianFactor.h"

namespace gtsam
{
class Values
{
public:
  bool exists(Key) const { return false; }
  template <class T> T at(Key) const { return T(); }
};
} // namespace gtsam
