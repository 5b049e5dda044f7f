"""GPU parity of the multi-image VisualEncoder.encode path (fixture G16) and of the BASELINE configs[0] model (HF ResNet-18 encoder +
2-layer decoder) against the oracle.  Written at the end of round 1, green on the MI355X in every round-2 run (profiles/r02_*_pytest_gpu.txt),
now regular tests.  No new kernel is involved: both go through VisualEncoder.encode, ops.linear and the decoder."""
import pytest
import torch

import golden_recipes as R

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def close_bf16(got, ref):
    err = (got - ref).abs()
    return bool((err <= 3e-2 + 3e-2 * ref.abs()).all()) and err.mean().item() <= 1e-2


def test_multi_image_encode_vs_golden(golden):
    """VisualEncoder.encode on [B, N, C, H, W] with images_mask (fixture G16): features within the bf16 tolerance, mask bit-exact"""
    from vilmedic_amd.blocks.vision import VisualEncoder
    g = golden("g16_vit_multi_image")
    cfg, vp = g["cfg"], g["visual_projection"]
    enc = VisualEncoder(backbone="vit", permute="no_permute", dropout_out=0.0, visual_projection=dict(vp), **cfg).to(dev())
    enc.model.load_state_dict(R.rand_state(R.vit_shapes(cfg), g["seed"]), strict=True)
    gen = torch.Generator().manual_seed(g["seed"] + 77)
    with torch.no_grad():
        enc.visual_projection.weight.copy_(0.05 * torch.randn(vp["out_features"], vp["in_features"], generator=gen))
        enc.visual_projection.bias.copy_(0.02 * torch.randn(vp["out_features"], generator=gen))
    B, N, size = g["B"], g["N"], cfg["image_size"]
    images = R.make_images(B * N, size, seed=g["seed"]).view(B, N, 3, size, size)
    enc.eval()
    with torch.no_grad():
        feats, mask = enc.encode(images.to(dev()), g["images_mask"].to(dev()))
    assert torch.equal(mask.cpu(), g["mask"])
    assert close_bf16(feats.float().cpu(), g["features"])


def test_c1_hfresnet_rrg_vs_oracle():
    """BASELINE configs[0] shape family (hfresnet + visual_projection + 2-layer decoder, train-mode BatchNorm): loss and logits of
    the HIP path against oracle.rrg_cnn_forward on the model's own weights"""
    from oracle import torch_ref as O
    from vilmedic_amd.models import RRG
    cnn = dict(proto="VisualEncoder", backbone="hfresnet", permute="batch_first", dropout_out=0.0, layer_type="basic", embedding_size=16,
               hidden_sizes=[16, 32, 64, 128], depths=[2, 2, 2, 2], hidden_act="relu", visual_projection=dict(in_features=128, out_features=128))
    torch.manual_seed(0)
    model = RRG(decoder=dict(proto=None, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, **R.DEC_TINY), cnn=dict(cnn)).to(dev())
    st = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items() if "lm_head.decoder" not in k}
    images = R.make_images(4, 96, seed=5)
    ids, am = R.make_reports(4, 16, R.DEC_TINY["vocab_size"], seed=5)
    cnn_cfg = {k: cnn[k] for k in ("layer_type", "hidden_sizes", "depths", "hidden_act")}
    ref_loss, ref_logits = O.rrg_cnn_forward(images, ids, am, st, cnn_cfg, R.DEC_TINY, training=True)
    model.train()
    out = model(input_ids=ids.to(dev()), attention_mask=am.to(dev()), images=images.to(dev()))
    assert abs(out["loss"].item() - ref_loss.item()) <= 2e-3 * max(1.0, abs(ref_loss.item())), (out["loss"].item(), ref_loss.item())
    assert close_bf16(out["logits"].float().cpu(), ref_logits)
    out["loss"].backward()
    g = model.enc.model.embedder.embedder.convolution.weight.grad
    assert g is not None and torch.isfinite(g).all() and g.abs().sum().item() > 0
