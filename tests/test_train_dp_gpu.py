"""GPU: the launcher loop (eonerf_code_amd/train_dp.py = train_eonerf.py:96-161,304 on the HIP path) end to end in one process:
epochs of a small synthetic ray table, the MSE -> uncertainty-loss / shadow-pass switch at epoch 2 (train_eonerf.py:139-143), StepLR
gamma 0.9 once per epoch (:64,304), the periodic checkpoint (:180-191) in the reference's format with torch.optim.Adam's step count
(one for every parameter: zero gradients still step) and the decayed learning rate.
"""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launcher_runs_epochs_switches_loss_decays_lr_and_checkpoints(tmp_path):
    # 8192 rays / 1024 per step = 8 steps per epoch; 36 steps = epochs 0..4, checkpoint at step 32 (save_freq = 4 epochs)
    cmd = [sys.executable, "-m", "eonerf_code_amd.train_dp", "--synthetic_rays", "8192", "--batch_size", "1024", "--n_images", "5",
           "--max_train_steps", "36", "--logs_dir", str(tmp_path), "--exp_name", "t"]
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "step=0" in r.stdout and "rays/s=" in r.stdout, r.stdout
    ck = os.path.join(str(tmp_path), "t", "ckpts", "epoch=4.ckpt")
    assert os.path.exists(ck), os.listdir(os.path.join(str(tmp_path), "t", "ckpts"))
    ckpt = torch.load(ck, map_location="cpu", weights_only=False)
    assert ckpt["epoch"] == 4 and torch.isfinite(ckpt["loss"]).all()
    opt = ckpt["optimizer_state_dict"]
    # four epoch ends before step 32 -> lr = 5e-4 * 0.9^4
    assert abs(opt["param_groups"][0]["lr"] - 5e-4 * 0.9 ** 4) < 1e-12
    # every parameter stepped 33 times (steps 0..32): the transient head with zero gradients until the loss switch at epoch 2
    from eonerf_code_amd.radiance_fields.eonerf import EONerfMLP
    names = [n for n, _ in EONerfMLP(5, radiometric_normalization=True).named_parameters()]
    steps = {names[i]: int(float(s["step"])) for i, s in opt["state"].items()}
    assert steps["base_mlp.hidden_layers.0.weight"] == 33, steps
    head = [v for k, v in steps.items() if k.startswith("transient_mlp")]
    assert head and all(v == 33 for v in head) and len(steps) == len(names), steps
    for v in ckpt["model_state_dict"].values():
        assert torch.isfinite(v).all()
