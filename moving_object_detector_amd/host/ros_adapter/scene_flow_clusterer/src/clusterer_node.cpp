// clusterer_node.cpp — stand-alone loader of the clusterer plugin, as the reference's scene_flow_clusterer_node
// (scene_flow_clusterer/src/clusterer_node.cpp:8-13): a node that loads scene_flow_clusterer/scene_flow_clusterer into itself.
#include <nodelet/loader.h>
#include <ros/ros.h>

#include <string>
#include <vector>

int main(int argc, char **argv) {
  ros::init(argc, argv, "scene_flow_clusterer");
  nodelet::Loader loader(false);                       // no loader services: one plugin, loaded right here
  const nodelet::M_string remappings;
  const nodelet::V_string my_argv(argv + 1, argv + argc);
  if (!loader.load(ros::this_node::getName(), "scene_flow_clusterer/scene_flow_clusterer", remappings, my_argv)) return 1;
  ros::spin();
  return 0;
}
