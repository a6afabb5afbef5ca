import importlib,sys
sys.path.insert(0,".")
import numpy as np
pkg=importlib.import_module("lc-crf-slam_amd"); wl=importlib.import_module("lc-crf-slam_amd.workloads")
for N in (100,500,640,700,800,900,1000,1024,1100):
    F=256
    pbs=[wl.slam_problem(N, seed=10+i) for i in range(16)]
    feats=[np.stack([pbs[f%16]["kernels"][k][0] for f in range(F)]) for k in range(2)]
    label=np.stack([pbs[f%16]["label"] for f in range(F)])
    b=pkg.BatchCRF(F,N,2,[2,2],[10.0,30.0]); b.set_inputs_host([N]*F,feats,label=label,conf=0.7)
    b.run(5,True); b.synchronize()
    print(N, "engine", b.engine(), "fallback frames", b.fallback_frames(), "V", b.lattice_sizes(0)[:3], b.lattice_sizes(1)[:3])
    b.close()
