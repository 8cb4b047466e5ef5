#!/usr/bin/env python3
"""Generate golden vectors from the REFERENCE's own loss code (run in the build container).

TEST INFRASTRUCTURE.  Loads ``/root/reference/src/duwu/loss/{diffusion,rectified_flow}.py``
by file path (their only third-party import is the *name* ``diffusers.EulerDiscreteScheduler``,
diffusion.py:6), with a stub ``diffusers`` module that exposes this oracle's restated
scheduler (oracle/scheduler.py, pinned by sigma_max = 14.6146).  The reference source never
leaves /root/reference: only inputs and outputs (data) are written to tests/golden/*.npz.

    python oracle/make_golden.py            # rewrites tests/golden/

The noise / timestep draws the reference makes internally (diffusion.py:75,68-70;
rectified_flow.py:37) are reproduced by re-seeding the CPU generator and re-drawing in the
same order, and asserted equal to what the reference reports in its aux output.
"""
import importlib.util
import json
import os
import sys
import types

sys.dont_write_bytecode = True  # never write __pycache__ into the read-only reference
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle.scheduler import EulerDiscreteScheduler  # noqa: E402

REF = "/root/reference/src/duwu"
OUT = os.path.join(ROOT, "tests", "golden")


def load_reference_loss():
    stub = types.ModuleType("diffusers")
    stub.EulerDiscreteScheduler = EulerDiscreteScheduler
    sys.modules["diffusers"] = stub
    for name in ("duwu", "duwu.loss"):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    mods = {}
    for name, fn in (("duwu.loss.diffusion", "loss/diffusion.py"), ("duwu.loss.rectified_flow", "loss/rectified_flow.py")):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, fn))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        mods[name] = mod
    return mods["duwu.loss.diffusion"], mods["duwu.loss.rectified_flow"]


def load_reference_rope():
    spec = importlib.util.spec_from_file_location("_ref_rope", os.path.join(REF, "modules/rope.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class LeafUNet(torch.nn.Module):
    """Returns a fixed leaf tensor so that d loss / d model_output lands in ``out.grad``."""

    def __init__(self, out):
        super().__init__()
        self.out = out
        self.seen = None

    def forward(self, noisy, timesteps, **kw):
        self.seen = (noisy.detach().clone(), timesteps.detach().clone())
        return (self.out,)


def save(name, meta, **arrays):
    arrays = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrays.items()}
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez(os.path.join(OUT, name + ".npz"), **arrays)
    print("wrote", name, {k: v.shape for k, v in arrays.items() if k != "meta"})


def gen_diffusion(dmod, name, shape, seed, **kw):
    B = shape[0]
    torch.manual_seed(seed)
    x = torch.randn(shape)
    out = torch.randn(shape, requires_grad=True)
    sched = EulerDiscreteScheduler.sdxl()
    loss_mod = dmod.DiffusionLoss(sched, **kw)
    unet = LeafUNet(out)
    torch.manual_seed(seed + 1)
    loss, aux = loss_mod(x, unet)
    loss.backward()
    # re-draw what the reference drew (diffusion.py:75 then :68-70)
    torch.manual_seed(seed + 1)
    noise = torch.randn_like(x)
    t = torch.randint(0, sched.config.num_train_timesteps, (B,))
    assert torch.equal(t, aux.timesteps)
    assert torch.equal(unet.seen[0], aux.noisy_latent)
    sig = loss_mod.get_sigmas_for_timesteps(t)
    meta = dict(kind="diffusion", shape=list(shape), seed=seed, kwargs=kw,
                prediction_type=loss_mod.prediction_type, target_type=loss_mod.target_type)
    save(name, meta, x=x, noise=noise, timesteps=t, sigmas=sig, model_output=out.detach(),
         noisy=aux.noisy_latent, pred=aux.pred, target=aux.target, losses=aux.losses, loss=loss.detach(),
         dloss_dout=out.grad)


def gen_rf(rmod, name, shape, seed, **kw):
    B = shape[0]
    torch.manual_seed(seed)
    x = torch.randn(shape)
    noise = torch.randn(shape)
    out = torch.randn(shape, requires_grad=True)
    sched = EulerDiscreteScheduler.sdxl(**({"prediction_type": kw.pop("model_prediction_type")} if "model_prediction_type" in kw else {}))
    loss_mod = rmod.RectifiedFlowLoss(scheduler=sched, **kw)
    unet = LeafUNet(out)
    x5 = torch.stack([x, noise], dim=1)  # rectified_flow.py:50-53 injected-noise hook
    torch.manual_seed(seed + 1)
    loss, aux = loss_mod(x5, unet)
    loss.backward()
    torch.manual_seed(seed + 1)
    u01 = torch.rand(B)  # rectified_flow.py:37
    smax = sched.sigmas[0]
    time = u01 * (smax / (1 + smax))
    sig = time / (1 - time)
    assert torch.equal(loss_mod.sigma_to_timestep(sig), aux.timesteps)
    meta = dict(kind="rf", shape=list(shape), seed=seed, kwargs=kw, prediction_type=loss_mod.prediction_type)
    save(name, meta, x=x, noise=noise, u01=u01, sigmas=sig, timesteps=aux.timesteps, model_output=out.detach(),
         noisy=aux.noisy_latent, pred=aux.pred, target=aux.target, losses=aux.losses, loss=loss.detach(),
         dloss_dout=out.grad)


def gen_rf_uniform_timestep(rmod, name, shape, seed, **kw):
    """RectifiedFlowLoss(time_sampling_type="uniform_timestep") (rectified_flow.py:32-33): integer timesteps drawn as in
    DiffusionLoss (diffusion.py:64-72), sigmas from the scheduler table; noise injected through the 5-D hook."""
    B = shape[0]
    torch.manual_seed(seed)
    x = torch.randn(shape)
    noise = torch.randn(shape)
    out = torch.randn(shape, requires_grad=True)
    sched = EulerDiscreteScheduler.sdxl(**({"prediction_type": kw.pop("model_prediction_type")} if "model_prediction_type" in kw else {}))
    loss_mod = rmod.RectifiedFlowLoss(scheduler=sched, time_sampling_type="uniform_timestep", **kw)
    unet = LeafUNet(out)
    x5 = torch.stack([x, noise], dim=1)
    torch.manual_seed(seed + 1)
    loss, aux = loss_mod(x5, unet)
    loss.backward()
    torch.manual_seed(seed + 1)
    t = torch.randint(0, sched.config.num_train_timesteps, (B,))  # the only draw (noise is injected)
    assert torch.equal(t, aux.timesteps)
    sig = loss_mod.get_sigmas_for_timesteps(t)
    meta = dict(kind="rf_uniform_timestep", shape=list(shape), seed=seed, kwargs=kw, prediction_type=loss_mod.prediction_type)
    save(name, meta, x=x, noise=noise, timesteps=t, sigmas=sig, model_output=out.detach(),
         noisy=aux.noisy_latent, pred=aux.pred, target=aux.target, losses=aux.losses, loss=loss.detach(),
         dloss_dout=out.grad)


def gen_nnw(rmod, name, shape, seed, model_prediction_type="epsilon"):
    """NNWeightedRFLoss.forward (rectified_flow.py:154-203) with the toy loss-prediction module of tests/golden_util.py."""
    from tests.golden_util import ToyLossPred

    B = shape[0]
    torch.manual_seed(seed)
    x, noise = torch.randn(shape), torch.randn(shape)
    out = torch.randn(shape, requires_grad=True)
    sched = EulerDiscreteScheduler.sdxl(prediction_type=model_prediction_type)
    lp = ToyLossPred()
    loss_mod = rmod.NNWeightedRFLoss(loss_pred_module=lp, scheduler=sched)
    unet = LeafUNet(out)
    torch.manual_seed(seed + 1)
    loss, aux = loss_mod(torch.stack([x, noise], dim=1), unet)
    loss.backward()
    torch.manual_seed(seed + 1)
    u01 = torch.rand(B)
    meta = dict(kind="nnw_rf", shape=list(shape), seed=seed, prediction_type=loss_mod.prediction_type)
    save(name, meta, x=x, noise=noise, u01=u01, timesteps=aux.timesteps, model_output=out.detach(), noisy=aux.noisy_latent,
         pred=aux.pred, target=aux.target, rf_losses=aux.losses, rescaled_losses=aux.rescaled_losses,
         pred_losses=aux.pred_losses, loss_pred_losses=aux.loss_pred_losses, loss=loss.detach(), dloss_dout=out.grad,
         grad_a=lp.a.grad, grad_b=lp.b.grad, grad_c=lp.c.grad)


def main():
    os.makedirs(OUT, exist_ok=True)
    dmod, rmod = load_reference_loss()
    if "--rfut-only" in sys.argv:  # round 2 addition; leaves the round-1 fixtures untouched
        gen_rf_uniform_timestep(rmod, "rfut_s0_epsilon", (4, 4, 8, 8), 0, model_prediction_type="epsilon")
        gen_rf_uniform_timestep(rmod, "rfut_s1215_v_prediction", (4, 4, 8, 8), 1215, model_prediction_type="v_prediction")
        return
    sched = EulerDiscreteScheduler.sdxl()
    lm = dmod.DiffusionLoss(sched)
    save("tables", dict(kind="tables"), all_snr=sched.all_snr, sigmas=sched.sigmas, timesteps=sched.timesteps,
         alphas_cumprod=sched.alphas_cumprod)
    assert lm.n_diffusion_time_steps == 1000

    small = (4, 4, 8, 8)
    types_ = ["epsilon", "v_prediction", "sample", "rectified_flow"]
    for seed in (0, 1215):
        for p in types_:
            for t in types_:
                gen_diffusion(dmod, f"dl_s{seed}_{p}_to_{t}", small, seed, prediction_type=p, target_type=t)
        gen_diffusion(dmod, f"dl_s{seed}_eps_snr", small, seed, use_snr_weight=True)
        gen_diffusion(dmod, f"dl_s{seed}_eps_snr_debias", small, seed, use_snr_weight=True, use_debiased_estimation=True)
        gen_diffusion(dmod, f"dl_s{seed}_v_snr", small, seed, prediction_type="v_prediction", target_type="v_prediction",
                      use_snr_weight=True, min_snr_gamma=3.0)
        for p in types_:
            gen_rf(rmod, f"rf_s{seed}_{p}", small, seed, model_prediction_type=p)
        gen_rf(rmod, f"rf_s{seed}_eps_rescale", small, seed, rescale_image=True, rescale_noise=True)
    big = (16, 4, 32, 32)
    gen_diffusion(dmod, "dl_big_eps", big, 1215)
    gen_diffusion(dmod, "dl_big_eps_snr_debias", big, 1215, use_snr_weight=True, use_debiased_estimation=True)
    gen_rf(rmod, "rf_big_eps", big, 1215)
    gen_rf_uniform_timestep(rmod, "rfut_s0_epsilon", small, 0, model_prediction_type="epsilon")
    gen_rf_uniform_timestep(rmod, "rfut_s1215_v_prediction", small, 1215, model_prediction_type="v_prediction")

    gen_nnw(rmod, "nnw_s0_eps", small, 0)
    gen_nnw(rmod, "nnw_s1215_v", small, 1215, model_prediction_type="v_prediction")

    # sigma_to_timestep on 64 log-spaced sigmas in [1e-3, 20] (rectified_flow.py:98-129)
    rf = rmod.RectifiedFlowLoss(scheduler=EulerDiscreteScheduler.sdxl())
    sig = torch.logspace(-3, np.log10(20.0), 64)
    save("sigma_to_timestep", dict(kind="sigma_to_timestep"), sigmas=sig, timesteps=rf.sigma_to_timestep(sig))

    # sampling schedule helpers (section 8f rank 1): k_diffusion_wrapper.py and get_sigmas.py load by file path
    kd = {}
    for nm, fn in (("kdw", "sampling/k_diffusion_wrapper.py"), ("gs", "sampling/get_sigmas.py")):
        spec = importlib.util.spec_from_file_location("_ref_" + nm, os.path.join(REF, fn))
        kd[nm] = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(kd[nm])
    abar = EulerDiscreteScheduler.sdxl().alphas_cumprod
    ds = kd["kdw"].DiscreteSchedule(((1 - abar) / abar) ** 0.5, quantize=False)
    probe = torch.logspace(-2, np.log10(20.0), 48)
    tq = torch.tensor([0.0, 0.5, 17.25, 500.0, 998.75, 999.0])
    save("kdiff_schedule", dict(kind="kdiff_schedule"), probe_sigmas=probe, sigma_to_t=ds.sigma_to_t(probe),
         t_probe=tq, t_to_sigma=ds.t_to_sigma(tq), get_sigmas_20=ds.get_sigmas(20), get_sigmas_all=ds.get_sigmas(),
         rf_sigmas_16=np.ascontiguousarray(kd["gs"].get_sigmas_for_rf(16, float(ds.sigma_max))).astype(np.float64),
         rf_sigmas_8_min=np.ascontiguousarray(kd["gs"].get_sigmas_for_rf(8, 14.6146, 0.03)).astype(np.float64))

    # aggregation.py loads by file path (torch only) -- section 8f rank 4
    spec = importlib.util.spec_from_file_location("_ref_agg", os.path.join(REF, "utils/aggregation.py"))
    agg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(agg)
    torch.manual_seed(5)
    for name, n_el, shape, dtype, pad, pad_to in (("a", [2, 3, 1], (4, 5), torch.float32, 0.0, None),
                                                  ("b", [1, 4, 2, 3], (3, 5), torch.float32, -1.5, 5),
                                                  ("c", [3, 1, 2], (7,), torch.int64, 0, 4),
                                                  ("d", [2, 2, 5, 1, 1], (77, 8), torch.float32, 0.25, None)):
        tot = sum(n_el)
        emb = torch.randint(0, 2, (tot, *shape)) if dtype == torch.int64 else torch.randn(tot, *shape)
        cat = agg.aggregate_embeddings(emb, n_el, "concat", pad_value=pad, pad_to_n_elements=pad_to)
        cat_loop = agg.concat_aggregate_embeddings(emb, n_el, pad_value=pad, pad_to_n_elements=pad_to)
        assert torch.equal(cat, cat_loop)
        rec = agg.split_aggregate_embeddings(cat, n_el, shape[0])
        assert torch.equal(rec, emb)
        fst = agg.aggregate_embeddings(emb, n_el, "first")
        save("aggregation_" + name, dict(kind="aggregation", n_elements=n_el, pad_value=pad, pad_to_n_elements=pad_to,
                                         seq=shape[0]), emb=emb, cat=cat, first=fst)

    # AxialRoPE(64, 4) forward (modules/rope.py:83-108) -- "next" row (f2), importable reference
    rope = load_reference_rope()
    torch.manual_seed(0)
    m = rope.AxialRoPE(64, 4)
    pos = rope.make_axial_pos(8, 8).repeat(7, 1, 1)
    xin = torch.randn(7, 64, 4, 64)
    with torch.no_grad():
        y = m(xin, pos)
    save("axial_rope", dict(kind="axial_rope", dim=64, heads=4), x=xin, pos=pos, freqs_h=m.freqs_h.detach(),
         freqs_w=m.freqs_w.detach(), y=y)


if __name__ == "__main__":
    main()
