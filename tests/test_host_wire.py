"""CPU: ROS-1 wire format of the boundary messages (moving_object_detector_amd/host/ros_wire.hpp) and the ros::Time /
ros::Duration arithmetic the frame interval depends on (host/messages.hpp) — known-answer bytes, round trips, rejects."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ros_wire_known_answers_and_round_trips(tmp_path):
    exe = str(tmp_path / "ros_wire_test")
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-Wall", "-Werror", os.path.join(ROOT, "tests", "cpp", "ros_wire_test.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip() == "ok"
