"""Input pipeline: the numpy oracle against Pillow itself (CPU), and the HIP kernel against the oracle (GPU)."""
import numpy as np
import pytest
import torch

from oracle import image_pipeline as IP


def _rand_img(rng, h, w):
    base = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    base[: h // 3, : w // 2] = 255            # saturated and flat regions exercise the clip / rounding paths
    base[h // 2:, w // 3:] //= 7
    return base


@pytest.mark.parametrize("seed", range(6))
def test_oracle_resize_is_bit_exact_with_pillow(seed):
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(seed)
    for _ in range(6):
        h, w = int(rng.integers(17, 420)), int(rng.integers(17, 420))
        oh, ow = int(rng.integers(8, 300)), int(rng.integers(8, 300))
        img = _rand_img(rng, h, w)
        ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
        assert np.array_equal(IP.pil_bilinear_resize_u8(img, oh, ow), ref), (h, w, oh, ow)


def test_oracle_train_and_eval_chain_vs_pillow_chain():
    """Resize(256) -> crop 224 window -> flip -> ToTensor -> Normalize, and Resize((224,224)) -> ToTensor -> Normalize, built
    from Pillow ops (what torchvision.transforms does on PIL images, ref: datasets/base/ImageDataset.py:96-108)."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(11)
    mean, std = np.asarray(IP.MEAN, np.float32), np.asarray(IP.STD, np.float32)
    for h, w, top, left, flip in [(300, 411, 5, 40, True), (512, 390, 31, 0, False), (256, 256, 7, 9, True), (1024, 700, 0, 0, False)]:
        img = _rand_img(rng, h, w)
        nh, nw = IP.resized_hw(h, w, 256)
        pil = Image.fromarray(img).resize((nw, nh), Image.BILINEAR).crop((left, top, left + 224, top + 224))
        if flip:
            pil = pil.transpose(Image.FLIP_LEFT_RIGHT)
        ref = ((np.asarray(pil).astype(np.float32) / np.float32(255)) - mean) / std
        got = IP.preprocess(img, resize=256, crop=224, top=top, left=left, flip=flip)
        assert np.array_equal(got, ref.transpose(2, 0, 1))
        pil = Image.fromarray(img).resize((224, 224), Image.BILINEAR)
        ref = ((np.asarray(pil).astype(np.float32) / np.float32(255)) - mean) / std
        assert np.array_equal(IP.preprocess(img, resize=0, crop=224), ref.transpose(2, 0, 1))


@pytest.mark.gpu
def test_hip_pipeline_bit_exact_vs_oracle_train_and_eval():
    from vilmedic_amd.datasets.device_pipeline import DeviceImagePipeline
    rng = np.random.default_rng(5)
    sizes = [(300, 411), (512, 390), (256, 256), (1024, 700), (257, 900), (640, 480), (224, 224), (231, 229)]
    imgs = [_rand_img(rng, h, w) for h, w in sizes]
    train = DeviceImagePipeline("train", 256, 224, generator=torch.Generator().manual_seed(3))
    tl, flip = train.draw([im.shape[:2] for im in imgs])
    assert flip.any() and not flip.all() and tl.max() > 0
    out = train(imgs, top_left=tl, flip=flip).cpu().numpy()
    for b, im in enumerate(imgs):
        ref = IP.preprocess(im, resize=256, crop=224, top=int(tl[b, 0]), left=int(tl[b, 1]), flip=bool(flip[b]))
        assert np.array_equal(out[b], ref), (b, np.abs(out[b] - ref).max())
    ev = DeviceImagePipeline("validate", 256, 224)
    out = ev([torch.from_numpy(im) for im in imgs]).cpu().numpy()
    for b, im in enumerate(imgs):
        assert np.array_equal(out[b], IP.preprocess(im, resize=0, crop=224)), b
    # up-scaling (scale < 1) and a crop equal to the resized image (no draw for the crop, torchvision RandomCrop.get_params)
    small = [_rand_img(rng, 64, 64), _rand_img(rng, 80, 120)]
    p = DeviceImagePipeline("train", 96, 96, generator=torch.Generator().manual_seed(1))
    tl, flip = p.draw([(64, 64), (80, 120)])
    assert tuple(tl[0]) == (0, 0)
    out = p(small, top_left=tl, flip=flip).cpu().numpy()
    for b, im in enumerate(small):
        assert np.array_equal(out[b], IP.preprocess(im, resize=96, crop=96, top=int(tl[b, 0]), left=int(tl[b, 1]), flip=bool(flip[b]))), b


@pytest.mark.gpu
def test_hip_pipeline_errors_and_throughput_line(capsys):
    """bad crop windows are rejected by the C ABI; prints the measured rate of a B=64 batch of 512x512 images."""
    import time
    from vilmedic_amd._lib import VmHipError
    from vilmedic_amd.datasets.device_pipeline import DeviceImagePipeline
    p = DeviceImagePipeline("train", 256, 224)
    img = np.zeros((300, 300, 3), np.uint8)
    with pytest.raises(VmHipError):
        p([img], top_left=np.array([[40, 0]], np.int32), flip=np.zeros(1, np.uint8))      # 40 + 224 > 256
    rng = np.random.default_rng(0)
    batch = [torch.from_numpy(_rand_img(rng, 512, 512)) for _ in range(64)]
    tl, flip = p.draw([(512, 512)] * 64)
    packed, offs, sizes = p.pack(batch)
    for _ in range(3):
        out = p.run(packed, offs, sizes, tl, flip)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    algo_bytes = 64 * 512 * 512 * 3 + out.numel() * 4
    ev0.record()
    for _ in range(10):
        out = p.run(packed, offs, sizes, tl, flip)
    ev1.record(); torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / 10
    with capsys.disabled():
        print(f"\n[image pipeline] B=64 512x512 -> 224x224 (Resize 256, crop, flip, normalize): {ms * 1e3:.1f} us per batch "
              f"({64 / ms * 1e3:.0f} images/s, {algo_bytes / ms / 1e6:.1f} GB/s of algorithmic bytes = source bytes once + fp32 output)")
    assert torch.isfinite(out).all()
