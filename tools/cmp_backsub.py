import os, sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from realsensecalibration_amd import capi, synthetic as syn
prob = syn.make_problem(8, 1500, 6, seed=9, outlier_frac=0.05)
np.set_printoptions(linewidth=250, precision=12)
for huber in (0.0, 1.0):
    got, s, log = capi.solve_points(prob, capi.default_options(schur_impl=1, huber_delta=huber))
    print("huber", huber, "iters", s.num_iterations, "final", repr(s.final_cost))
    print(log[:, [1, 2, 3, 4, 5, 6, 7]])
    np.save(f"gpurun_out/params_{os.environ.get('RSBA_BACKSUB_PROJ','1')}_{huber}.npy", got)
