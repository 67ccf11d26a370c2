import sys, os, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from realsensecalibration_amd import capi, synthetic as syn
prob = syn.make_problem(6, 40, 4, seed=106)
print("devices", capi.load().rsba_device_count(), flush=True)
got = capi.points_linearize_and_step(prob, 1e4)
print(got["cost"], got["solve_ok"], flush=True)
