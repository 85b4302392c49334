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
        for other in ("scene_flow_clusterer", "scene_flow_constructor"):      # a generated header of the other package needs the dependency
            if other != pkg and f"#include <{other}/" in src:
                assert other in deps, (pkg, other)
        if "image_transport/" in src:
            assert "image_transport" in deps and "message_filters" in deps
    xml.dom.minidom.parse(os.path.join(ADAPTER, "scene_flow_clusterer", "nodelet_plugins.xml"))


def test_no_dependency_cycle_with_the_reference_packages():
    """The adapter packages REPLACE CMakeLists.txt / package.xml / src of two reference packages inside the reference's workspace.  The
    reference's scene_flow_clusterer build-depends on scene_flow_constructor (scene_flow_clusterer/package.xml:18, CMakeLists.txt
    find_package: pcl_point_xyz_velocity.h), and the adapter's clusterer keeps that edge out; the constructor must not depend on the
    clusterer in return, in either manifest, or catkin cannot order the two."""
    import re
    import xml.dom.minidom

    def deps_of(path):
        doc = xml.dom.minidom.parse(path)
        tags = ("depend", "build_depend", "build_export_depend", "exec_depend", "run_depend", "buildtool_depend")
        return {d.firstChild.data.strip() for t in tags for d in doc.getElementsByTagName(t)}

    edges = {pkg: deps_of(os.path.join(ADAPTER, pkg, "package.xml")) for pkg in ("scene_flow_constructor", "scene_flow_clusterer")}
    ref = "/root/reference"
    if os.path.isdir(ref):                               # the workspace the adapter is dropped into: every reference manifest
        for name in os.listdir(ref):
            px = os.path.join(ref, name, "package.xml")
            if os.path.exists(px):
                edges.setdefault(name, set())
                if name not in ("scene_flow_constructor", "scene_flow_clusterer"):
                    edges[name] |= deps_of(px)
        assert "scene_flow_constructor" in deps_of(os.path.join(ref, "scene_flow_clusterer", "package.xml"))   # the fact the rule rests on
    # either way of replacing: both adapter packages, or only one of them next to the reference's other one
    for variant in ({}, {"scene_flow_clusterer": {"scene_flow_constructor"}}):
        g = {k: set(v) for k, v in edges.items()}
        for k, extra in variant.items():
            g[k] |= extra
        state = {}

        def visit(n, path):
            if state.get(n) == 1:
                raise AssertionError("dependency cycle: " + " -> ".join(path + [n]))
            if state.get(n) == 2 or n not in g:
                return
            state[n] = 1
            for m in sorted(g[n]):
                visit(m, path + [n])
            state[n] = 2

        for n in sorted(g):
            visit(n, [])
    cm = open(os.path.join(ADAPTER, "scene_flow_constructor", "CMakeLists.txt")).read()
    comps = re.search(r"find_package\(catkin REQUIRED COMPONENTS(.*?)\)", cm, flags=re.S).group(1).split()
    assert "scene_flow_clusterer" not in comps and "scene_flow_clusterer" not in edges["scene_flow_constructor"]
    assert "cfg/CollapsedClusterer.cfg" in cm and os.path.exists(os.path.join(ADAPTER, "scene_flow_constructor", "cfg", "CollapsedClusterer.cfg"))


def test_plugin_identity():
    src = open(os.path.join(ADAPTER, "scene_flow_clusterer", "src", "clusterer_nodelet_ros.cpp")).read()
    assert "PLUGINLIB_EXPORT_CLASS(scene_flow_clusterer::ClustererNodeletRos, nodelet::Nodelet)" in src
    assert "mod_set_camera" not in src                       # no dummy camera: the context is sized by the cloud
    xml = open(os.path.join(ADAPTER, "scene_flow_clusterer", "nodelet_plugins.xml")).read()
    assert 'name="scene_flow_clusterer/scene_flow_clusterer"' in xml and 'base_class_type="nodelet::Nodelet"' in xml
    assert 'path="lib/libscene_flow_clusterer"' in xml
