"""The from_text leg of bench.py alone (raw text -> finished index, wall clock): TAG=... python tools/from_text_once.py [trace]"""
import json, sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "ast-text-analysis_amd"))
import numpy as np, bench
from east import hip_backend, synthetic
r = bench.from_text_leg(hip_backend, synthetic, 0)
for k, v in r.items():
    print(os.environ.get("TAG"), k, "wall %.3f prep %.3f build %.3f first call %.3f" % (v["wall_ms"], v["prep_ms"], v["build_ms"], v["first_call_wall_ms"]),
          "prep kernels %.3f" % v["prep_kernels_total_ms"])
    if len(sys.argv) > 1:
        print("   ", v["prep_kernels_ms"])
