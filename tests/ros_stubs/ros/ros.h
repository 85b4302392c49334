#pragma once
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>
namespace ros {
struct Duration { double toSec() const; };
struct WallDuration { double toSec() const; };
struct Time { uint32_t sec = 0, nsec = 0; static Time now(); Duration operator-(const Time &) const; };
struct WallTime { static WallTime now(); WallDuration operator-(const WallTime &) const; };
class Publisher { public: uint32_t getNumSubscribers() const; template <class M> void publish(const M &) const; };
class Subscriber {};
class NodeHandle {
 public:
  NodeHandle();
  explicit NodeHandle(const std::string &ns);
  NodeHandle(const NodeHandle &parent, const std::string &ns);   // roscpp: child handle in a sub-namespace
  template <class T> T param(const std::string &name, const T &default_value) const;
  std::string resolveName(const std::string &name) const;
  template <class M> Publisher advertise(const std::string &topic, uint32_t queue_size);
  template <class M, class T> Subscriber subscribe(const std::string &topic, uint32_t queue_size, void (T::*cb)(const std::shared_ptr<M const> &), T *obj);
};
void init(int &argc, char **argv, const std::string &name);
void spin();
void shutdown();
namespace this_node { const std::string &getName(); }
}  // namespace ros
namespace ros { template <class... A> void stub_log(const char *fmt, A &&...); }
#define ROS_INFO(...) ::ros::stub_log(__VA_ARGS__)
#define ROS_FATAL(...) ::ros::stub_log(__VA_ARGS__)
#define ROS_ERROR(...) ::ros::stub_log(__VA_ARGS__)
