#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
python3 - <<'PY'
import importlib,sys,time,json,importlib.util
sys.path.insert(0,".")
spec=importlib.util.spec_from_file_location("bench","bench.py"); b=importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
pkg=importlib.import_module("lc-crf-slam_amd"); wl=importlib.import_module("lc-crf-slam_amd.workloads")
pb=wl.bilateral_problem(100000,1)
print(json.dumps(b.c5_object_api(pkg,pb,20),indent=1))
PY
