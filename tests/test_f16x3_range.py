"""GPU: the range guard of the fp16 x 3 export precision (include/eonerf_hip.h, eonerf_range_status).

An operand of the split precision is hi + lo fp16: exact to ~2^-21 for 2^-14 <= |v| <= 65504.  Above, hi is infinite and the value
is lost (and the NaN it breeds is turned into 0 by the next ReLU: silent); a weight matrix whose largest element is below 2^-9 lives
in the subnormal-lo regime as a whole, one above 64 amplifies the 2^-25 absolute operand error of small activations.  The kernels flag
the first, every re-pack the other two, and the Python layer repeats the export on the module's exact fp32 context -- the reference's
fp32 arithmetic (radiance_fields/eonerf.py:154-170) has no such limits, so neither may an export."""
import warnings

import pytest
import torch

from oracle import eonerf_oracle as orc

pytestmark = pytest.mark.gpu
N_IMG, R, STEP = 4, 256, 2.0 / 128


def _state(seed=5):
    sd = orc.random_state_dict(N_IMG, seed=seed, bias_scale=0.05)
    sd["sigma_layer.output_layer.bias"] += 1.0
    return sd


def _field(sd, precision="bf16", eval_precision=None):
    from eonerf_code_amd.radiance_fields.eonerf import EONerfMLP
    f = EONerfMLP(N_IMG, radiometric_normalization=True, precision=precision, eval_precision=eval_precision)
    f.load_state_dict(sd)
    return f.cuda()


def _export(f, seed=9):
    from eonerf_code_amd.sat_rendering import render_image
    from eonerf_code_amd.datasets.satellite import define_satrays_from_tensors
    rays, ts, _, u_cam, u_sun = orc.synthetic_batch(R, N_IMG, seed=seed)
    sr = define_satrays_from_tensors(rays.cuda(), ts.cuda())
    with torch.no_grad():
        res, n = render_image(f, None, sr, None, None, epoch_idx=3, chunk=R, render_step_size=STEP, noise=[(u_cam, None, u_sun)], eval=True)
    return res, n


def _scaled(sd, **factors):
    out = {k: v.clone() for k, v in sd.items()}
    for key, fac in factors.items():
        out[key.replace("__", ".")] *= fac
    return out


CASES = {
    # three layers of large (but in-range: max|w| ~ 43) weights: X_4 reaches ~1e6 (> 65504) -- only the kernels' activation probe sees it
    "activation_overflow": {"base_mlp__hidden_layers__1__weight": 400.0, "base_mlp__hidden_layers__2__weight": 400.0,
                            "base_mlp__hidden_layers__3__weight": 400.0},
    # a layer of weights ~1e-5 undone by the next one (~1e3): caught at re-pack (max|w| outside [2^-9, 64])
    "tiny_weight_layer": {"base_mlp__hidden_layers__3__weight": 1.0e-4, "base_mlp__hidden_layers__3__bias": 1.0e-4,
                          "base_mlp__hidden_layers__4__weight": 1.0e4},
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_out_of_range_export_falls_back_to_the_fp32_context_and_matches_it(case):
    sd = _scaled(_state(), **CASES[case])
    f = _field(sd)
    assert f.eval_precision == "fp16x3"
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        got, n = _export(f)
    assert any("fp16x3" in str(x.message) for x in w), "the switch is announced"
    assert f.eval_precision == "fp32"                       # for good: the next export does not try fp16x3 again
    want, n32 = _export(_field(sd, eval_precision="fp32"))
    assert n == n32
    for k in want:
        assert torch.isfinite(got[k]).all(), k
        assert torch.equal(got[k], want[k]), k               # the same fp32 kernels on the same inputs
    # and the fp32 kernels are right about this field (the oracle is scale-blind fp32 torch)
    rays, ts, _, u_cam, u_sun = orc.synthetic_batch(R, N_IMG, seed=9)
    with torch.no_grad():
        ref, _ = orc.render_rays(orc.Field(sd), orc.define_satrays_from_tensors(rays, ts), u_cam, u_sun, 3, STEP, eval=True)
    assert (got["rgb"].cpu() - ref[:, 0:3]).abs().max().item() < 1e-4
    assert (got["depth"].cpu() - ref[:, 3:4]).abs().max().item() < 1e-4


def test_in_range_field_stays_on_fp16x3_and_eval_queries_check_too():
    sd = _state()
    f = _field(sd)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        _export(f)
    assert not w and f.eval_precision == "fp16x3" and f._ctx_eval is not None
    # eval-mode field queries run on the export context and carry the same guard: a position beyond fp16's range
    f.eval()
    x = torch.tensor([[0.1, 0.2, 0.3], [7.0e4, 0.0, 0.0]], device="cuda")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        with torch.no_grad():
            s = f.query_density(x)
    assert w and f.eval_precision == "fp32" and torch.isfinite(s).all()
    f32 = _field(sd, precision="fp32")
    with torch.no_grad():
        assert (s - f32.query_density(x)).abs().max().item() <= 1e-5 * max(1.0, s.abs().max().item())


def test_a_module_whose_own_precision_is_fp16x3_raises_instead_of_returning_garbage():
    sd = _scaled(_state(), **CASES["activation_overflow"])
    f = _field(sd, precision="fp16x3")
    with pytest.raises(RuntimeError, match="fp16"):
        _export(f)
