// clusterer_node.cpp — the clusterer as a process of its own: a node that hosts exactly one nodelet, the plugin
// scene_flow_clusterer/scene_flow_clusterer, under the node's own name (what the reference's scene_flow_clusterer_node does,
// scene_flow_clusterer/src/clusterer_node.cpp:8-13; detect_moving_object.launch may load the plugin into a manager instead).
#include <nodelet/loader.h>
#include <ros/ros.h>

#include <string>
#include <vector>

namespace {
const char kPluginType[] = "scene_flow_clusterer/scene_flow_clusterer";

// the command line after the program name goes to the nodelet as its my_argv
nodelet::V_string arguments_after_program(int argc, char **argv) {
  nodelet::V_string out;
  for (int i = 1; i < argc; ++i) out.push_back(argv[i]);
  return out;
}
}  // namespace

int main(int argc, char **argv) {
  ros::init(argc, argv, "scene_flow_clusterer");
  const nodelet::V_string nodelet_argv = arguments_after_program(argc, argv);
  const nodelet::M_string no_remappings;
  nodelet::Loader host(/*provide_ros_api=*/false);      // one fixed plugin: no load / unload services
  if (!host.load(ros::this_node::getName(), kPluginType, no_remappings, nodelet_argv)) {
    ROS_FATAL("cannot load %s", kPluginType);
    return 1;
  }
  ros::spin();
  return 0;
}
