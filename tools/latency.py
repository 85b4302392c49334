"""Frame-at-a-time figures of bench.py's `latency` object alone.  python tools/latency.py [sizes ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from moving_object_detector_amd import capi, synth

W, H = 1280, 720
cam, sq = synth.make_sequence(W, H, 16, seed=4)
host = {"disparity_now": sq["disparity"][1:], "disparity_prev": sq["disparity"][:-1], "flow": sq["flow"], "t": sq["t"], "q": sq["q"], "dt": sq["dt"]}
sizes = tuple(int(a) for a in sys.argv[1:]) or (1, 8, 64)
print(json.dumps(bench.latency_leg(torch.device("cuda:0"), 0, capi.camera_struct(cam), capi.params_struct(synth.Params()), host, W, H, sizes)))
