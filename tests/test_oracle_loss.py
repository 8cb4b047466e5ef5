"""The oracle (oracle/loss.py, oracle/scheduler.py) against the reference's own outputs.

Goldens under tests/golden/ were produced by oracle/make_golden.py running the reference's
src/duwu/loss/*.py; known-answer constants come from the reference's configs / notebook
(SURVEY.md section 4).
"""
import numpy as np
import pytest
import torch

from oracle import loss as OL
from oracle.scheduler import (EulerDiscreteScheduler, cosine_logsnr, laplace_logsnr, logsnr_to_sigmas)
from tests.golden_util import load, names


def close(a, b, rtol=2e-6, atol=1e-6):
    torch.testing.assert_close(a, b, rtol=rtol, atol=atol)


def test_sigma_max_pin():
    # configs/sampling/demo_sampling.yaml:49 -> 14.6146
    s = EulerDiscreteScheduler.sdxl()
    assert abs(s.sigmas[0].item() - 14.6146) < 5e-5
    assert abs(s.sigmas[-2].item() - 0.029167533) < 1e-8
    assert s.sigmas[-1].item() == 0.0 and s.sigmas.shape == (1001,)
    assert s.timesteps[0].item() == 999.0 and s.timesteps[-1].item() == 0.0


def test_notebook_laplace_pin():
    # test_scripts/test_diffusion_scheduler.ipynb cell 2 stored output (b = 1.5 run)
    t = np.linspace(0, 1, 1002)[1:-1]
    sig, _ = logsnr_to_sigmas(laplace_logsnr(t, 0, 1.5, float(np.finfo(np.float32).eps)))
    assert f"{min(sig):.9f}" == "0.009449642" and f"{max(sig):.6f}" == "105.811714" and int(np.sum(sig >= 10)) == 23


def test_notebook_table_pin():
    # notebook cell 5 stored output
    t = np.linspace(0, 1, 1000)
    exp_l = [(8191.996, 1), (67108804.0, 5), (549755600000.0, 24), (4503591300000000.0, 50)]
    for b, (mx, cnt) in zip((0.5, 1.0, 1.5, 2.0), exp_l):
        sig, _ = logsnr_to_sigmas(laplace_logsnr(t, 0, b))
        assert np.isclose(max(sig), mx, rtol=1e-6) and int(sum(sig >= 10)) == cnt
    exp_c = [(28519397000000.0, 195), (5340355.5, 64), (30552.502, 21), (2310.9211, 7)]
    for s_, (mx, cnt) in zip((0.5, 1.0, 1.5, 2.0), exp_c):
        sig, _ = logsnr_to_sigmas(cosine_logsnr(t, s=s_))
        assert np.isclose(max(sig), mx, rtol=1e-6) and int(sum(sig >= 10)) == cnt


def test_tables_golden():
    _, d = load("tables")
    s = EulerDiscreteScheduler.sdxl()
    close(OL.all_snr(s), d["all_snr"], rtol=0, atol=0)
    close(s.sigmas, d["sigmas"], rtol=0, atol=0)


@pytest.mark.parametrize("name", names("dl_"))
def test_diffusion_loss_golden(name):
    meta, d = load(name)
    kw = dict(meta["kwargs"])
    s = EulerDiscreteScheduler.sdxl()
    o = OL.diffusion_loss(s, d["x"], d["noise"], d["timesteps"], lambda n, t: d["model_output"], **kw)
    close(o.sigmas, d["sigmas"], rtol=0, atol=0)
    close(o.noisy_latent, d["noisy"])
    close(o.target, d["target"])
    close(o.pred, d["pred"], rtol=1e-5, atol=1e-5)
    close(o.losses, d["losses"], rtol=1e-5, atol=1e-7)
    close(o.loss, d["loss"], rtol=1e-5, atol=1e-7)
    close(o.dloss_dout, d["dloss_dout"], rtol=2e-5, atol=1e-7)


@pytest.mark.parametrize("name", names("rf_"))
def test_rf_loss_golden(name):
    meta, d = load(name)
    kw = dict(meta["kwargs"])
    s = EulerDiscreteScheduler.sdxl(prediction_type=meta["prediction_type"])
    sig = OL.rf_time_to_sigma(s, d["u01"])
    close(sig, d["sigmas"], rtol=0, atol=0)
    o = OL.rectified_flow_loss(s, d["x"], d["noise"], sig, lambda n, t: d["model_output"], **kw)
    close(o.timesteps, d["timesteps"], rtol=1e-6, atol=1e-4)
    close(o.noisy_latent, d["noisy"])
    close(o.target, d["target"])
    close(o.pred, d["pred"], rtol=1e-5, atol=1e-5)
    close(o.losses, d["losses"], rtol=1e-5, atol=1e-7)
    close(o.loss, d["loss"], rtol=1e-5, atol=1e-7)
    close(o.dloss_dout, d["dloss_dout"], rtol=2e-5, atol=1e-7)


@pytest.mark.parametrize("name", names("rfut_"))
def test_rf_uniform_timestep_golden(name):
    """time_sampling_type="uniform_timestep" (reference rectified_flow.py:32-33), fixtures from the reference class."""
    meta, d = load(name)
    s = EulerDiscreteScheduler.sdxl(prediction_type=meta["prediction_type"])
    sig = OL.sigmas_for_timesteps(s, d["timesteps"])
    close(sig, d["sigmas"], rtol=0, atol=0)
    o = OL.rectified_flow_loss(s, d["x"], d["noise"], sig, lambda n, t: d["model_output"], timesteps=d["timesteps"])
    close(o.noisy_latent, d["noisy"])
    close(o.target, d["target"])
    close(o.pred, d["pred"], rtol=1e-5, atol=1e-5)
    close(o.losses, d["losses"], rtol=1e-5, atol=1e-7)
    close(o.loss, d["loss"], rtol=1e-5, atol=1e-7)
    close(o.dloss_dout, d["dloss_dout"], rtol=2e-5, atol=1e-7)


def test_sigma_to_timestep_golden():
    _, d = load("sigma_to_timestep")
    s = EulerDiscreteScheduler.sdxl()
    close(OL.sigma_to_timestep(s, d["sigmas"]), d["timesteps"], rtol=1e-6, atol=1e-4)


def test_analytic_grad_matches_autograd():
    torch.manual_seed(3)
    s = EulerDiscreteScheduler.sdxl()
    x, n = torch.randn(3, 4, 8, 8), torch.randn(3, 4, 8, 8)
    t = torch.tensor([5, 500, 999])
    for p in OL.PRED_TYPES:
        for tt in OL.PRED_TYPES:
            out = torch.randn(3, 4, 8, 8, requires_grad=True)
            o = OL.diffusion_loss(s, x, n, t, lambda a, b: out, prediction_type=p, target_type=tt)
            (g,) = torch.autograd.grad(o.loss, out)
            close(o.dloss_dout, g, rtol=1e-4, atol=1e-7)


def test_philox4x32_10_known_answers():
    """oracle/philox.py against the known-answer vectors of the Random123 distribution (kat_vectors, philox4x32 10 rounds):
    the all-zero, all-ones and pi-digits counters / keys."""
    from oracle import philox as OP

    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for c, k, want in kat:
        assert [int(v) for v in OP.philox4x32_10_words([c], [k])[0]] == list(want)
    z = OP.normal(1 << 16, 1215, 0)
    assert abs(float(z.mean())) < 2e-2 and abs(float(z.std()) - 1) < 2e-2
    t = OP.timesteps(20000, 1000, 3, 8)
    assert t.min() == 0 and t.max() == 999
