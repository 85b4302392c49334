#pragma once
#include <image_transport/image_transport.h>
#include <sensor_msgs/Image.h>
namespace image_transport {
class SubscriberFilter { public: void subscribe(ImageTransport &it, const std::string &base_topic, uint32_t queue_size); };
}
