#pragma once
#include <std_msgs/Header.h>
namespace sensor_msgs {
struct PointField { enum { INT8 = 1, UINT8, INT16, UINT16, INT32, UINT32, FLOAT32, FLOAT64 }; std::string name; uint32_t offset = 0; uint8_t datatype = 0; uint32_t count = 0; };
struct PointCloud2 {
  std_msgs::Header header; uint32_t height = 0, width = 0; std::vector<PointField> fields; bool is_bigendian = false;
  uint32_t point_step = 0, row_step = 0; std::vector<uint8_t> data; bool is_dense = false;
};
typedef std::shared_ptr<PointCloud2 const> PointCloud2ConstPtr;
}
