"""CPU: the ROS-1 adapter shells (constructor node, clusterer nodelet + its loader node; two catkin packages under
moving_object_detector_amd/host/ros_adapter/) still compile against
the C ABI, the host mirror and the ROS types they use.  There is no ROS in this image: `g++ -fsyntax-only` over declaration-only
stand-ins (tests/ros_stubs/) — a rot check that pins names and signatures, not behaviour.  Plugin identity as in the reference:
scene_flow_clusterer/nodelet_plugins.xml:3-4, clusterer_nodelet.cpp:5."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ADAPTER = os.path.join(ROOT, "moving_object_detector_amd", "host", "ros_adapter")


SHELLS = ["scene_flow_constructor/src/scene_flow_constructor_node.cpp", "scene_flow_clusterer/src/clusterer_nodelet_ros.cpp",
          "scene_flow_clusterer/src/clusterer_node.cpp"]


@pytest.mark.parametrize("src", SHELLS)
def test_adapter_shell_compiles(src):
    # include directories as the packages' CMakeLists.txt set them: ROS (stand-ins here), include/ and the host mirror
    r = subprocess.run(["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "tests", "ros_stubs"),
                        "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "moving_object_detector_amd", "host"),
                        os.path.join(ADAPTER, src)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-3000:]


def test_constructor_node_keeps_the_reference_interface():
    """Node name, the four private topics, the synchronised subscriptions and the publish gating of scene_flow_constructor.cpp:22-63,99,114,141,144."""
    src = open(os.path.join(ADAPTER, SHELLS[0])).read()
    assert 'ros::init(argc, argv, "scene_flow_constructor")' in src and "ros::spin()" in src
    for topic in ('"depth"', '"optical_flow"', '"scene_flow"', '"synthetic_optical_flow"'):
        assert f"advertise<" in src and topic in src, topic
    for name in ('resolveName("left_image")', 'resolveName("right_image")', "getCameraInfoTopic", "TimeSynchronizer", "dynamic_reconfigure::Server"):
        assert name in src, name
    assert src.count("getNumSubscribers() > 0") >= 4
    assert "estimateOpticalFlow" in src and "estimateCameraMotion" in src and "CALL-OUT" in src
    # collapsed mode: one clustering per frame in this process, ~moving_objects published here, the 29.5 MB cloud
    # asked for only while ~scene_flow has subscribers (scene_flow_constructor.cpp:141-142)
    assert 'param("publish_moving_objects", true)' in src and '"moving_objects"' in src
    assert "pc_with_velocity_pub_.getNumSubscribers() > 0 ? new mod_host::PointCloud2() : nullptr" in src
    assert "moving_objects_pub_.getNumSubscribers() > 0 ? new mod_host::MovingObjectArray() : nullptr" in src


def test_packaging():
    """Both packages: catkin metadata that names what the shells include, the library / executable names of the reference."""
    import xml.dom.minidom
    for pkg, targets in (("scene_flow_constructor", ["add_executable(${PROJECT_NAME} "]),
                         ("scene_flow_clusterer", ["add_library(${PROJECT_NAME} ", "add_executable(${PROJECT_NAME}_node "])):
        cm = open(os.path.join(ADAPTER, pkg, "CMakeLists.txt")).read()
        assert f"project({pkg} CXX)" in cm and "MOD_SF_LIBRARY" in cm and "generate_dynamic_reconfigure_options(cfg/" in cm
        for t in targets:
            assert t in cm, t
        doc = xml.dom.minidom.parse(os.path.join(ADAPTER, pkg, "package.xml"))
        assert doc.getElementsByTagName("name")[0].firstChild.data == pkg
        deps = {d.firstChild.data for tag in ("depend", "build_depend", "exec_depend") for d in doc.getElementsByTagName(tag)}
        src = "".join(open(os.path.join(ADAPTER, pkg, "src", f)).read() for f in os.listdir(os.path.join(ADAPTER, pkg, "src")))
        for header_pkg in ("dynamic_reconfigure", "sensor_msgs", "roscpp"):
            assert header_pkg in deps, (pkg, header_pkg)
        if "moving_object_msgs/" in src:
            assert "moving_object_msgs" in deps
        if "nodelet/" in src:
            assert "nodelet" in deps and "pluginlib" in deps
        if "scene_flow_clusterer/ClustererConfig.h" in src and pkg != "scene_flow_clusterer":
            assert "scene_flow_clusterer" in deps
        if "image_transport/" in src:
            assert "image_transport" in deps and "message_filters" in deps
    xml.dom.minidom.parse(os.path.join(ADAPTER, "scene_flow_clusterer", "nodelet_plugins.xml"))


def test_plugin_identity():
    src = open(os.path.join(ADAPTER, "scene_flow_clusterer", "src", "clusterer_nodelet_ros.cpp")).read()
    assert "PLUGINLIB_EXPORT_CLASS(scene_flow_clusterer::ClustererNodeletRos, nodelet::Nodelet)" in src
    assert "mod_set_camera" not in src                       # no dummy camera: the context is sized by the cloud
    xml = open(os.path.join(ADAPTER, "scene_flow_clusterer", "nodelet_plugins.xml")).read()
    assert 'name="scene_flow_clusterer/scene_flow_clusterer"' in xml and 'base_class_type="nodelet::Nodelet"' in xml
    assert 'path="lib/libscene_flow_clusterer"' in xml
