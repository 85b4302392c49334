#pragma once
#include <ros/ros.h>
namespace message_filters {
template <class M0, class M1, class M2, class M3> class TimeSynchronizer {
 public:
  template <class F0, class F1, class F2, class F3> TimeSynchronizer(F0 &f0, F1 &f1, F2 &f2, F3 &f3, uint32_t queue_size);
  template <class C> void registerCallback(void (C::*cb)(const std::shared_ptr<M0 const> &, const std::shared_ptr<M1 const> &, const std::shared_ptr<M2 const> &,
                                                         const std::shared_ptr<M3 const> &), C *obj);
};
}
