#pragma once
#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <vector>
namespace ros {
struct Time { uint32_t sec = 0, nsec = 0; };
class Publisher { public: uint32_t getNumSubscribers() const; template <class M> void publish(const M &) const; };
class Subscriber {};
class NodeHandle {
 public:
  template <class T> T param(const std::string &name, const T &default_value) const;
  template <class M> Publisher advertise(const std::string &topic, uint32_t queue_size);
  template <class M, class T> Subscriber subscribe(const std::string &topic, uint32_t queue_size, void (T::*cb)(const std::shared_ptr<M const> &), T *obj);
};
}  // namespace ros
