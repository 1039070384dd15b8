// STAND-IN, not gtsam (see gtsam/linear/HessianFactor.h next to this file): gtsam::Values with the members
// integration/sage_gtsam_prepass.h calls (gtsam/nonlinear/Values.h: exists(Key), at<ValueType>(Key)) plus insert(), as a
// small type-erased map so that the glue header can be EXECUTED by integration/compile_check/prepass_run.cpp; at<T>() of a
// key that is absent or holds another type throws, like gtsam's ValuesKeyDoesNotExist / ValuesIncorrectType.
#pragma once
#include "gtsam/linear/HessianFactor.h"

#include <map>
#include <stdexcept>
#include <typeindex>

namespace gtsam
{
class Values
{
  struct Slot
  {
    std::type_index type = std::type_index(typeid(void));
    std::shared_ptr<void> value;
  };
  std::map<Key, Slot> slots_;

public:
  template <class T> void insert(Key k, const T &v)
  {
    if (slots_.count(k))
      throw std::invalid_argument("Values::insert: key exists");
    Slot s;
    s.type = std::type_index(typeid(T));
    s.value = std::make_shared<T>(v);
    slots_[k] = s;
  }
  bool exists(Key k) const { return slots_.count(k) != 0; }
  template <class T> const T &at(Key k) const
  {
    auto it = slots_.find(k);
    if (it == slots_.end())
      throw std::out_of_range("Values::at: key does not exist");
    if (it->second.type != std::type_index(typeid(T)))
      throw std::invalid_argument("Values::at: incorrect type");
    return *static_cast<const T *>(it->second.value.get());
  }
};
} // namespace gtsam
