"""CPU: the two ROS-1 adapter shells (node + nodelet plugin, moving_object_detector_amd/host/ros_adapter/) still compile against
the C ABI, the host mirror and the ROS types they use.  There is no ROS in this image: `g++ -fsyntax-only` over declaration-only
stand-ins (tests/ros_stubs/) — a rot check that pins names and signatures, not behaviour.  Plugin identity as in the reference:
scene_flow_clusterer/nodelet_plugins.xml:3-4, clusterer_nodelet.cpp:5."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ADAPTER = os.path.join(ROOT, "moving_object_detector_amd", "host", "ros_adapter")


@pytest.mark.parametrize("src", ["clusterer_nodelet_ros.cpp", "scene_flow_constructor_ros.cpp"])
def test_adapter_shell_compiles(src):
    r = subprocess.run(["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "tests", "ros_stubs"),
                        os.path.join(ADAPTER, src)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-3000:]


def test_plugin_identity():
    src = open(os.path.join(ADAPTER, "clusterer_nodelet_ros.cpp")).read()
    assert "PLUGINLIB_EXPORT_CLASS(scene_flow_clusterer::ClustererNodeletRos, nodelet::Nodelet)" in src
    xml = open(os.path.join(ADAPTER, "nodelet_plugins.xml")).read()
    assert 'name="scene_flow_clusterer/scene_flow_clusterer"' in xml and 'base_class_type="nodelet::Nodelet"' in xml
    assert 'path="lib/libscene_flow_clusterer"' in xml
