#!/usr/bin/env python3
"""Generates tests/golden/models/trained_writer_6_16_24_4.npz and tests/golden/nn_writer_golden.npz.

Run ONCE in the build container (needs /root/reference); the outputs are committed.
The shipped model files only show what the reference's READER accepts.  This script goes through the
WRITER at the end of the reference's training pipeline instead:
  /root/reference/autorally_control/src/path_integral/scripts/ml_pipeline/utils.py
    setup_model (:16-46) with a layer list that none of the shipped files has ([6, 16, 24, 4]),
    torch_model_to_npz (:68-90) -> model.npz (the file NeuralNetModel::loadParams reads,
    neural_net_model.cu:84-99), compute_state_ders (:132-152)
and records the file it wrote plus INPUTS and OUTPUTS of the model (values only).  The key scheme of
nn_writer_golden.npz is the one of nn_dynamics_golden.npz (gen_golden.py).
"""
import os
import shutil
import sys
import tempfile

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
REF = "/root/reference/autorally_control/src/path_integral"
sys.path.insert(0, os.path.join(REF, "scripts", "ml_pipeline"))

import numpy as np
import torch

import utils as ref_utils  # the reference's module (sets the default dtype to float64, :13)

HERE = os.path.dirname(os.path.abspath(__file__))
NAME = "trained_writer_6_16_24_4"
LAYERS = [6, 16, 24, 4]


def main():
    torch.manual_seed(20200820)
    model = ref_utils.setup_model(LAYERS, verbose=False)
    # an untrained network has tiny outputs: scale the initial weights so that tanh leaves its linear range
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(2.5)
    model.eval()
    with tempfile.TemporaryDirectory() as d:
        ref_utils.torch_model_to_npz(model, d)  # writes d/model.npz
        shutil.copyfile(os.path.join(d, "model.npz"), os.path.join(HERE, "models", NAME + ".npz"))
    rng = np.random.RandomState(424242)
    n = 48
    states = np.zeros((n, 7))
    states[:, 0:2] = rng.uniform(-20, 20, size=(n, 2))
    states[:, 2] = rng.uniform(-3.2, 3.2, size=n)
    states[:, 3] = rng.uniform(-0.3, 0.3, size=n)
    states[:, 4] = rng.uniform(0.0, 12.0, size=n)
    states[:, 5] = rng.uniform(-2.0, 2.0, size=n)
    states[:, 6] = rng.uniform(-2.0, 2.0, size=n)
    ctrls = np.stack([rng.uniform(-0.99, 0.99, size=n), rng.uniform(-0.99, 0.65, size=n)], axis=1)
    ders = np.zeros((n, 7))
    with torch.no_grad():
        for i in range(n):
            x = torch.tensor([states[i, 3], states[i, 4], states[i, 5], states[i, 6], ctrls[i, 0], ctrls[i, 1]])
            ders[i] = ref_utils.compute_state_ders(states[i], model(x.double()).numpy(), negate_yaw_der=True)
    dt = 1.0 / 50
    traj = np.zeros((11, 7))
    traj[0] = [1.0, -2.0, -0.4, 0.05, 6.0, -0.2, 0.1]
    tctrl = np.stack([np.linspace(-0.3, 0.3, 10), np.linspace(0.5, 0.1, 10)], axis=1)
    with torch.no_grad():
        for i in range(10):
            x = torch.tensor([traj[i, 3], traj[i, 4], traj[i, 5], traj[i, 6], tctrl[i, 0], tctrl[i, 1]])
            sd = ref_utils.compute_state_ders(traj[i], model(x.double()).numpy(), negate_yaw_der=True)
            traj[i + 1] = traj[i] + sd * dt
    out = {NAME + "/layers": np.array(LAYERS, dtype=np.int32),
           NAME + "/negate_yaw_der": np.array([1], dtype=np.int32),
           NAME + "/states": states, NAME + "/controls": ctrls, NAME + "/state_ders": ders,
           NAME + "/traj_states": traj, NAME + "/traj_controls": tctrl}
    np.savez(os.path.join(HERE, "nn_writer_golden.npz"), **out)
    z = np.load(os.path.join(HERE, "models", NAME + ".npz"))
    print({k: (z[k].shape, str(z[k].dtype)) for k in z.files}, "max |der|", np.abs(ders[:, 3:]).max(axis=0))


if __name__ == "__main__":
    main()
