#!/usr/bin/env python3
"""Generates tests/golden/nn_dynamics_golden.npz and tests/golden/models/*.npz.

Run ONCE in the build container (needs /root/reference); the outputs are committed.
It imports the reference's own fp64 Python restatement of the NN dynamics step
  /root/reference/autorally_control/src/path_integral/scripts/ml_pipeline/utils.py
    setup_model (:16-46), npz_to_torch_model (:49-65), compute_state_ders (:132-152)
and the Euler step `state + state_der * dt` of model_vehicle_dynamics.py:146, and
records INPUTS and OUTPUTS only (values; no reference source travels).

The model weight arrays are copied as data (np.savez, same keys/dtypes as the
reference files) so that tests and bench can run where /root/reference is absent.
"""
import os
import sys

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
REF = "/root/reference/autorally_control/src/path_integral"
sys.path.insert(0, os.path.join(REF, "scripts", "ml_pipeline"))

import numpy as np
import torch

import utils as ref_utils  # the reference's module

HERE = os.path.dirname(os.path.abspath(__file__))
MODELS = {
    "autorally_nnet_09_12_2018": ([6, 32, 32, 4], True),
    "gazebo_nnet_09_12_2018": ([6, 32, 32, 4], True),
    "shallow_network_08_20_2020": ([6, 32, 32, 4], False),   # params/models/README.md:20
    "wider_deeper_network_08_20_2020": ([6, 64, 64, 64, 64, 4], False),
}


def main():
    os.makedirs(os.path.join(HERE, "models"), exist_ok=True)
    rng = np.random.RandomState(20181209)
    out = {}
    for name, (layers, negate) in MODELS.items():
        src = os.path.join(REF, "params", "models", name + ".npz")
        z = np.load(src)
        np.savez(os.path.join(HERE, "models", name + ".npz"), **{k: z[k] for k in z.files})
        model = ref_utils.setup_model(layers, verbose=False)
        model = ref_utils.npz_to_torch_model(src, model)
        model.eval()
        n = 64
        # state = [x, y, yaw, roll, u_x, u_y, yaw_mder]; control = [steering, throttle]
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
                x = torch.tensor([states[i, 3], states[i, 4], states[i, 5], states[i, 6],
                                  ctrls[i, 0], ctrls[i, 1]])
                y = model(x.double()).numpy()
                ders[i] = ref_utils.compute_state_ders(states[i], y, negate_yaw_der=negate)
        # 10-step open-loop trajectory, Euler dt = 1/50 (model_vehicle_dynamics.py:146)
        dt = 1.0 / 50
        traj = np.zeros((11, 7))
        traj[0] = [0.0, 0.0, 0.3, 0.0, 4.0, 0.1, 0.0]
        tctrl = np.stack([np.linspace(0.0, 0.4, 10), np.linspace(0.2, 0.6, 10)], axis=1)
        with torch.no_grad():
            for i in range(10):
                x = torch.tensor([traj[i, 3], traj[i, 4], traj[i, 5], traj[i, 6], tctrl[i, 0], tctrl[i, 1]])
                y = model(x.double()).numpy()
                sd = ref_utils.compute_state_ders(traj[i], y, negate_yaw_der=negate)
                traj[i + 1] = traj[i] + sd * dt
        out[name + "/layers"] = np.array(layers, dtype=np.int32)
        out[name + "/negate_yaw_der"] = np.array([int(negate)], dtype=np.int32)
        out[name + "/states"] = states
        out[name + "/controls"] = ctrls
        out[name + "/state_ders"] = ders
        out[name + "/traj_states"] = traj
        out[name + "/traj_controls"] = tctrl
    # the sample quoted in SURVEY.md 8(c)
    model = ref_utils.npz_to_torch_model(os.path.join(REF, "params", "models", "autorally_nnet_09_12_2018.npz"),
                                         ref_utils.setup_model([6, 32, 32, 4], verbose=False))
    with torch.no_grad():
        out["sample_in"] = np.array([0, 1, 0, 0, 0.1, 0.5], dtype=np.float64)
        out["sample_out"] = model(torch.tensor(out["sample_in"])).numpy()
    np.savez(os.path.join(HERE, "nn_dynamics_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "nn_dynamics_golden.npz"), "sample_out", out["sample_out"])


if __name__ == "__main__":
    main()
