#!/bin/bash
# tools/loop_time.sh [iterations]: avg_tick_ms of the ROS-free binaries' control loop (two controllers, debug-mode
# self-simulation, no sleep) with and without feedback gains, on the synthetic oval.  One JSON line per run.
cd "$GRAFT_REPO_ROOT" || exit 1
export PYTHONPATH=$GRAFT_REPO_ROOT
N=${1:-2000}
D=$(mktemp -d)
python3 - <<PY
import os
from autorally_amd import params as P, synthetic as S
d="$D"
os.makedirs(d+"/models"); os.makedirs(d+"/maps")
for f in os.listdir(S.MODELS_DIR):
    open(d+"/models/"+f,"wb").write(open(os.path.join(S.MODELS_DIR,f),"rb").read())
ch0,xb,yb,ppm=S.oval_track_map()
for m in ("ccrf_costmap_09_29_2017.npz","marietta_costmap_09_08_2018.npz"):
    P.save_costmap_npz(d+"/maps/"+m,ch0,xb,yb,ppm)
PY
for fb in false true; do
  echo "path_integral_nn K=1920 use_feedback_gains=$fb"
  AR_MPPI_PARAMS_PATH=$D ./autorally_amd/bin/path_integral_nn autorally_amd/host/launch/path_integral_nn.launch --rollouts 1920 --max-iter $N --no-sleep --set x_pos=0.0 --set y_pos=-10.0 --set heading=0.0 --set use_feedback_gains=$fb | tail -1 | cut -c1-330
done
for fb in false true; do
  echo "path_integral_bf K=2560 use_feedback_gains=$fb"
  AR_MPPI_PARAMS_PATH=$D ./autorally_amd/bin/path_integral_bf autorally_amd/host/launch/path_integral_bf.launch --rollouts 2560 --max-iter $N --no-sleep --set x_pos=0.0 --set y_pos=-10.0 --set heading=0.0 --set use_feedback_gains=$fb | tail -1 | cut -c1-330
done
rm -rf "$D"
