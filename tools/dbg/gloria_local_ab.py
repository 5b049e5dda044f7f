"""A/B timing of GLoRIA's local loss on the GPU: the per-caption-length batched formulation (two bmm's with the feature
dimension per (caption, image) pair, repeated operands) against GLoRIALoss._local's all-pairs formulation.
    python tools/dbg/gloria_local_ab.py [B] [D] [T] [hw]"""
import sys
import time

import torch

from vilmedic_amd.blocks.losses import GLoRIALoss, cosine_similarity, gloria_attention_fn


def grouped(img, words, cap_lens, temp1=4.0, temp2=5.0, temp3=10.0):
    B = img.shape[0]
    sims = torch.empty(B, B, device=img.device, dtype=torch.float32)
    by_len = {}
    for i, T in enumerate(cap_lens):
        by_len.setdefault(T, []).append(i)
    ih, iw = img.shape[2], img.shape[3]
    for T, idxs in by_len.items():
        w = words[idxs][:, :, :T]
        n = len(idxs)
        q = w[:, None].expand(n, B, -1, T).reshape(n * B, -1, T)
        c = img[None].expand(n, B, -1, ih, iw).reshape(n * B, -1, ih, iw)
        wctx, attn = gloria_attention_fn(q, c, temp1)
        row = cosine_similarity(q.transpose(1, 2).reshape(n * B * T, -1), wctx.transpose(1, 2).reshape(n * B * T, -1)).view(n, B, T)
        sims[:, idxs] = torch.log(torch.exp(row * temp2).sum(-1)).t()
    sims = sims * temp3
    labels = torch.arange(B, device=img.device)
    return torch.nn.functional.cross_entropy(sims, labels), torch.nn.functional.cross_entropy(sims.t(), labels)


def main():
    B, D, T, hw = [int(x) for x in sys.argv[1:5]] if len(sys.argv) >= 5 else (48, 768, 32, 19)
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    img = torch.randn(B, D, hw, hw, generator=g).to(dev).requires_grad_(True)
    words = torch.randn(B, D, T, generator=g).to(dev).requires_grad_(True)
    lens = [int(x) for x in torch.randint(T // 2, T + 1, (B,), generator=g)]
    crit = GLoRIALoss()
    fns = {"grouped": lambda: grouped(img, words, lens)[:2], "all_pairs": lambda: crit._local(img, words, lens)[:2]}
    vals = {}
    for name, fn in fns.items():
        for it in range(4):
            if it == 1:
                torch.cuda.synchronize()
                torch.cuda.reset_peak_memory_stats()
                t0 = time.perf_counter()
            img.grad = words.grad = None
            l0, l1 = fn()
            (l0 + l1).backward()
        torch.cuda.synchronize()
        vals[name] = (l0.item(), l1.item(), img.grad.clone())
        print(f"{name:10s} fwd+bwd {(time.perf_counter() - t0) / 3 * 1e3:8.2f} ms   peak {torch.cuda.max_memory_allocated() / 2**30:6.2f} GiB   "
              f"loss {l0.item():.5f} {l1.item():.5f}")
    a, b = vals["grouped"], vals["all_pairs"]
    print("loss diff", abs(a[0] - b[0]), abs(a[1] - b[1]), "grad rel", ((a[2] - b[2]).norm() / a[2].norm()).item())


if __name__ == "__main__":
    main()
