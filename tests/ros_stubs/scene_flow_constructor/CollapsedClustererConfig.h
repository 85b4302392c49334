#pragma once
// what dynamic_reconfigure generates from the adapter's scene_flow_constructor/cfg/CollapsedClusterer.cfg
namespace scene_flow_constructor { struct CollapsedClustererConfig { int cluster_size = 2500; double depth_diff = 0.15; double dynamic_speed = 0.3; int neighbor_distance = 4; }; }
