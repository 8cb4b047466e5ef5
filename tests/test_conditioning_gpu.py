"""Conditioning front-end on the device:
  * ``ConcatTextEncoders.forward`` bucket / concat / pad / mask / pooled assembly against fixtures produced by the
    REFERENCE's own forward (oracle/make_golden_te.py ran /root/reference/src/duwu/modules/text_encoders.py:139-264 over
    the same synthetic text models);
  * the VAE latent normalisation of trainer.py:241-244 fused into the q-sample kernel."""
import pytest
import torch

from tests.golden_util import load, names

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", names("te_"))
def test_concat_text_encoders_matches_reference_forward(name):
    from uwudiff_amd.conditioning import ConcatTextEncoders, SyntheticTextModel

    meta, d = load(name)
    models = [(SyntheticTextModel(m["hidden"], m["seed"]), m["config"]) for m in meta["models"]]
    te = ConcatTextEncoders(tokenizers=[], text_model_and_configs=models, zero_for_padding=meta["zero_for_padding"]).cuda()
    tok = {"input_ids": d["input_ids"], "attention_mask": d["attention_mask"]}
    emb, normed, pooled, mask = te([tok] * len(models))
    # the assembly only moves and masks values; the synthetic encoders' tanh / LayerNorm differ in the last bits between the
    # host (where the reference produced the fixture) and the device, hence the 1e-5 tolerance (the host-side assembly is
    # compared bit for bit in tests/test_host_logic_cpu.py)
    tol = dict(rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(emb.cpu(), d["embedding"], **tol)
    torch.testing.assert_close(normed.cpu(), d["normed"], **tol)
    torch.testing.assert_close(pooled.cpu(), d["pooled"], **tol)
    zero = d["embedding"] == 0  # padded / masked positions are exactly zero on the device too
    assert float(emb.cpu()[zero].abs().max() if zero.any() else 0.0) == 0.0
    if meta["has_mask"]:
        assert torch.equal(mask.cpu(), d["mask"])
    else:
        assert mask is None


def test_t5_style_encoder_joins_a_second_bucket():
    """SD3-like layout: two CLIP-style encoders in bucket 0, a T5-style one (no pooled output) in bucket 1.  The reference
    crashes on this path (``pooled_embedding.to`` on None, text_encoders.py:196) -- not reproduced; checked against the
    reference's stated layout (features of bucket 0 concatenated, the narrower bucket zero-padded, buckets on the sequence
    axis)."""
    from uwudiff_amd.conditioning import ConcatTextEncoders, SyntheticTextModel, SyntheticTokenizer

    ms = [SyntheticTextModel(48, 1), SyntheticTextModel(80, 2), SyntheticTextModel(64, 3, kind="t5")]
    cfgs = [dict(concat_bucket=0, use_pooled=True), dict(concat_bucket=0, use_pooled=True), dict(concat_bucket=1, need_mask=True)]
    te = ConcatTextEncoders(tokenizers=["a", "b", "c"], text_model_and_configs=list(zip(ms, cfgs))).cuda()
    tok = SyntheticTokenizer()(["a red fox", "two"])
    emb, normed, pooled, mask = te([tok] * 3)
    assert emb.shape == (2, 154, 128) and pooled.shape == (2, 128) and mask.shape == (2, 154)
    am = tok["attention_mask"].cuda()
    h2 = ms[2](tok["input_ids"].cuda(), output_hidden_states=True)[1][-1] * am[..., None]
    torch.testing.assert_close(emb[:, 77:, :64], h2, rtol=1e-6, atol=1e-6)
    assert float(emb[:, 77:, 64:].abs().max()) == 0.0
    assert torch.equal(mask[:, :77], torch.ones(2, 77, device="cuda", dtype=torch.long)) and torch.equal(mask[:, 77:], am)


def test_vae_latent_normalisation_is_fused_into_qsample():
    from uwudiff_amd.objective import DiffusionLoss
    from uwudiff_amd.scheduler import EulerDiscreteScheduler

    torch.manual_seed(0)
    x, noise = torch.randn(4, 4, 8, 8) * 3 + 1, torch.randn(4, 4, 8, 8)
    t = torch.tensor([0, 10, 500, 999])
    out = torch.randn(4, 4, 8, 8)

    class Leaf(torch.nn.Module):
        def forward(self, noisy, ts, **kw):
            self.seen = noisy
            return (out.cuda(),)

    mean, std = 0.7, 2.5
    plain, fused = DiffusionLoss(EulerDiscreteScheduler.from_pretrained("sdxl")), DiffusionLoss(EulerDiscreteScheduler.from_pretrained("sdxl"))
    fused.set_latent_normalisation(mean, std)
    u1, u2 = Leaf(), Leaf()
    plain.inject(noise=noise.cuda(), timesteps=t.cuda())
    l1, a1 = plain(((x - mean) / std).cuda(), u1)        # what the reference does: normalise, then the loss
    fused.inject(noise=noise.cuda(), timesteps=t.cuda())
    l2, a2 = fused(x.cuda(), u2)
    torch.testing.assert_close(a2.noisy_latent, a1.noisy_latent, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(a2.target, a1.target, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(l2, l1, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("kind", ["rf", "rf_rescale_image", "rf_rescale_noise", "nnw", "nnw_rescale_image", "rf_5d"])
def test_vae_latent_normalisation_in_rf_and_nnw_losses(kind):
    """Every loss sees the NORMALISED latent (reference trainer.py:241-244 normalises between the VAE and the loss), also
    NNWeightedRFLoss, and with rescale_image the per-sample std is the normalised latent's (normalise first, then rescale)."""
    from uwudiff_amd.objective import NNWeightedRFLoss, RectifiedFlowLoss
    from uwudiff_amd.scheduler import EulerDiscreteScheduler

    torch.manual_seed(1)
    B = 4
    x, noise = torch.randn(B, 4, 8, 8) * 3 + 1, torch.randn(B, 4, 8, 8)
    u, out = torch.rand(B), torch.randn(B, 4, 8, 8)
    mean, std = 0.7, 2.5

    class Leaf(torch.nn.Module):
        def forward(self, noisy, ts, **kw):
            return (out.cuda(),)

    class LossPred(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.tensor(0.3))

        def forward(self, noisy, sigmas, **kw):
            return self.w * noisy.flatten(1).mean(1) - 0.2 * sigmas.log()

    def make():
        kw = dict(scheduler=EulerDiscreteScheduler.from_pretrained("sdxl"), rescale_image="rescale_image" in kind,
                  rescale_noise="rescale_noise" in kind)
        if kind.startswith("nnw"):
            torch.manual_seed(2)
            return NNWeightedRFLoss(loss_pred_module=LossPred().cuda(), **kw)
        return RectifiedFlowLoss(**kw)

    plain, fused = make(), make()
    fused.set_latent_normalisation(mean, std)
    xin_plain, xin_fused = (x - mean) / std, x
    if kind == "rf_5d":
        xin_plain, xin_fused = torch.stack([xin_plain, noise], 1), torch.stack([x, noise], 1)
    plain.inject(noise=noise.cuda(), u01=u.cuda())
    l1, a1 = plain(xin_plain.cuda(), Leaf())  # what the reference does: normalise, then the loss
    fused.inject(noise=noise.cuda(), u01=u.cuda())
    l2, a2 = fused(xin_fused.cuda(), Leaf())
    torch.testing.assert_close(a2.noisy_latent, a1.noisy_latent, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(a2.target, a1.target, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(a2.losses, a1.losses, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(l2, l1, rtol=1e-5, atol=1e-6)


def test_trainer_vae_slot_encodes_and_normalises():
    from duwu.trainer import DMTrainer

    cfg = {"unet": {"_target_": "uwudiff_amd.dit.DiT.from_config", "config": {"depth": 1, "hidden": 128, "heads": 2, "sample_size": 8}},
           "te": None, "vae": {"_target_": "uwudiff_amd.conditioning.SyntheticVAE"}}
    tr = DMTrainer(cfg, vae_std=0.5, vae_mean=0.1, use_warm_up=False).cuda()
    batch = (torch.randn(2, 3, 64, 64).cuda(), ["", ""], [], {}, {})
    out = tr.training_step(batch, 0)
    assert out["aux_output"].noisy_latent.shape == (2, 4, 8, 8) and torch.isfinite(out["loss"])
    assert tr.loss._latent_norm == (0.1, 0.5)
