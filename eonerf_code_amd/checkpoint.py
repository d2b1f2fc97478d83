"""Checkpoint files in the reference's format (train_eonerf.py:180-191, eval_eonerf.py:44-75):

    torch.save({'epoch', 'occ_grid_state_dict', 'model_state_dict', 'optimizer_state_dict', 'loss'}, 'ckpts/epoch=<e>.ckpt')

`model_state_dict` has the reference's 44 keys, `optimizer_state_dict` is a torch.optim.Adam state_dict over
named_parameters() order (so the reference could resume from it), and `occ_grid_state_dict` is the inert nerfacc
OccGridEstimator state the reference's eval script insists on loading (the grid never influences a rendered value,
SURVEY.md 0)."""
import os

import torch


def occ_grid_state_dict(resolution=128):
    """state_dict() of nerfacc.OccGridEstimator(roi_aabb=[-1,-1,-1,1,1,1], resolution, levels=1) with everything marked
    occupied: the four PERSISTENT buffers of nerfacc v0.5.2 (setup_env.sh:10) -- `grid_coords` / `grid_indices` are registered
    with persistent=False there and must not appear, or eval_eonerf.py:71's strict load_state_dict rejects the file.
    (The grid is output-irrelevant in the reference)"""
    r = int(resolution)
    return {"resolution": torch.tensor([r, r, r], dtype=torch.int32),
            "aabbs": torch.tensor([[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0]]),
            "occs": torch.zeros(r ** 3),
            "binaries": torch.ones(1, r, r, r, dtype=torch.bool)}


def adam_state_dict(field, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.Adam.state_dict() equivalent built from the flat moment buffers of FusedTrainer.  EVERY parameter has a state
    entry with the global step: in the reference the transient / ambient heads receive defined zero gradients while epoch_idx < 2
    (torch.cat + slicing keeps them in the graph, sat_rendering.py:294,311-312,322), so torch.optim.Adam creates their state at
    step 1 like everybody else's."""
    by_name = {name: (off, r, c) for name, off, r, c in field._layout}
    state, ids = {}, []
    for i, (name, p) in enumerate(field.named_parameters()):
        ids.append(i)
        if step == 0:
            continue
        off, r, c = by_name[name]
        state[i] = {"step": torch.tensor(float(step)),
                    "exp_avg": exp_avg[off:off + r * c].view(p.shape).detach().cpu().clone(),
                    "exp_avg_sq": exp_avg_sq[off:off + r * c].view(p.shape).detach().cpu().clone()}
    group = {"lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": 0, "amsgrad": False, "maximize": False, "foreach": None,
             "capturable": False, "differentiable": False, "fused": None, "params": ids}
    return {"state": state, "param_groups": [group]}


def save_checkpoint(path, epoch, field, trainer=None, loss=None, grid_resolution=128):
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    ckpt = {"epoch": epoch, "occ_grid_state_dict": occ_grid_state_dict(grid_resolution),
            "model_state_dict": {k: v.detach().cpu() for k, v in field.state_dict().items()},
            "optimizer_state_dict": (adam_state_dict(field, trainer.exp_avg, trainer.exp_avg_sq, trainer.step_count, trainer.lr,
                                                     trainer.betas, trainer.eps) if trainer is not None else None),
            "loss": None if loss is None else torch.as_tensor(loss).detach().cpu()}
    torch.save(ckpt, path)
    return path


def load_checkpoint(path, field, trainer=None, map_location="cpu"):
    """Inverse of save_checkpoint; also loads checkpoints written by the reference's train_eonerf.py."""
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    field.load_state_dict(ckpt["model_state_dict"], strict=True)
    if trainer is not None and ckpt.get("optimizer_state_dict"):
        by_name = {name: (off, r, c) for name, off, r, c in field._layout}
        st = ckpt["optimizer_state_dict"]["state"]
        # parameters without a state entry (a checkpoint of a run that never stepped them) start from zero moments, whatever the
        # trainer held before
        trainer.exp_avg.zero_()
        trainer.exp_avg_sq.zero_()
        steps = [0]
        for i, (name, p) in enumerate(field.named_parameters()):
            if i in st:
                off, r, c = by_name[name]
                trainer.exp_avg[off:off + r * c].copy_(st[i]["exp_avg"].reshape(-1))
                trainer.exp_avg_sq[off:off + r * c].copy_(st[i]["exp_avg_sq"].reshape(-1))
                steps.append(int(float(st[i]["step"])))
        trainer.step_count = max(steps)      # one count for all (k_adam); a reference checkpoint carries the same value everywhere
        trainer.lr = ckpt["optimizer_state_dict"]["param_groups"][0]["lr"]
    return ckpt["epoch"]
