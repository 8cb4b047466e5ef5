"""CPU: structural pins of the UNet restatement and host-side layout conversion."""
import torch


def test_oracle_sdxl_parameter_count_pin():
    # public SDXL UNet size: 2.567 B parameters
    from oracle.unet import SDXL_UNET_CONFIG, UNetOracle

    with torch.device("meta"):
        m = UNetOracle(**SDXL_UNET_CONFIG)
    n = sum(p.numel() for p in m.parameters())
    assert n == 2_567_463_684


def test_product_registry_matches_oracle_names_and_sizes():
    from oracle.unet import UNetOracle
    from uwudiff_amd.unet import TINY_UNET_CONFIG, UNet2DConditionModel

    cfg = {k: v for k, v in TINY_UNET_CONFIG.items() if k != "sample_size"}
    o = UNetOracle(**cfg)
    m = UNet2DConditionModel(cfg, compute_dtype="fp32")
    sd_o, sd_m = o.state_dict(), m.state_dict()
    assert set(sd_o) == set(sd_m)
    for k in sd_o:
        assert sd_o[k].shape == sd_m[k].shape, k
    m.load_state_dict(sd_o)
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd_o[k]), k
    # init rule of reference unet_patch.py:34-45: residual-branch out layers start near zero
    m2 = UNet2DConditionModel(cfg, compute_dtype="fp32")
    sd = m2.state_dict()
    assert sd["down_blocks.0.resnets.0.conv2.weight"].abs().max() < 1e-3
    assert sd["mid_block.attentions.0.transformer_blocks.0.attn1.to_out.0.weight"].abs().max() < 1e-3
    assert sd["conv_out.weight"].abs().max() < 1e-3
    assert sd["down_blocks.0.resnets.0.conv1.weight"].abs().max() > 1e-2
