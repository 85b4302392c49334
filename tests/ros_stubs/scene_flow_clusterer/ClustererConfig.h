#pragma once
// what dynamic_reconfigure generates from scene_flow_clusterer/cfg/Clusterer.cfg:8-11
namespace scene_flow_clusterer { struct ClustererConfig { int cluster_size = 2500; double depth_diff = 0.15; double dynamic_speed = 0.3; int neighbor_distance = 4; }; }
