import json, sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "ast-text-analysis_amd"))
import numpy as np, bench
from east import hip_backend, synthetic
for knob in (-1, 0):
    hip_backend.load().east_hip_debug_set_text_stream(knob)
    r = bench.from_text_leg(hip_backend, synthetic, 0)
    for k, v in r.items():
        print("stream", knob, k, "wall %.3f prep %.3f build %.3f first_call %.3f" % (v["wall_ms"], v["prep_ms"], v["build_ms"], v["first_call_wall_ms"]), "prep kernels %.3f" % v["prep_kernels_total_ms"])
