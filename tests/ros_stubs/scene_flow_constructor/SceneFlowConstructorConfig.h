#pragma once
// what dynamic_reconfigure generates from scene_flow_constructor/cfg/SceneFlowConstructor.cfg:8-9
namespace scene_flow_constructor { struct SceneFlowConstructorConfig { int dynamic_flow_diff = 5; double max_color_velocity = 1.0; }; }
