"""Minimal dataset side of the batch-dict contract.  The reference's datasets (PIL / torchvision / HF
tokenizers; vilmedic/datasets/**) are a CPU input pipeline and OUT OF SCOPE; these two classes produce the same batch
dicts from synthetic or pre-tokenised tensors so that the executors and models run end to end."""
import torch
from torch.utils.data import Dataset

from .prefetch import PrefetchLoader, stack  # noqa: F401


class _IdTokenizer:
    """stand-in tokenizer: decode = space-joined ids (enough for decode drivers / SCST rewards on synthetic data)"""

    def __init__(self, vocab_size, cls=0, pad=1, sep=2):
        self.vocab_size = vocab_size
        self.cls_token, self.pad_token, self.sep_token = "[CLS]", "[PAD]", "[SEP]"
        self.vocab = {"[CLS]": cls, "[PAD]": pad, "[SEP]": sep}
        self.special = {cls, pad, sep}
        # the id attributes HF tokenizers expose (read by RRG_HF, ref: models/rrg/RRG_HF.py:73-79)
        self.cls_token_id, self.pad_token_id, self.sep_token_id, self.unk_token_id = cls, pad, sep, None

    def decode(self, ids, skip_special_tokens=True, clean_up_tokenization_spaces=False):
        ids = ids.tolist() if hasattr(ids, "tolist") else list(ids)
        return " ".join(str(i) for i in ids if not (skip_special_tokens and i in self.special))

    def get_vocab(self):
        return self.vocab


class SyntheticImSeq(Dataset):
    """images ~ N(0,1) [3,S,S]; reports: [CLS] U{3..V-1} x U{L/2..L-2} [SEP] [PAD]*   (SURVEY §8d)."""

    def __init__(self, split="train", num_samples=256, image_size=224, vocab_size=30522, tokenizer_max_len=128, seed=0, **kwargs):
        g = torch.Generator().manual_seed(seed + {"train": 0, "validate": 1, "test": 2}.get(split, 3))
        self.images = torch.randn(num_samples, 3, image_size, image_size, generator=g)
        L, V = tokenizer_max_len, vocab_size
        self.ids = torch.full((num_samples, L), 1, dtype=torch.long)
        self.mask = torch.zeros(num_samples, L, dtype=torch.long)
        for b in range(num_samples):
            n = int(torch.randint(L // 2, L - 1, (1,), generator=g))
            self.ids[b, 0] = 0
            self.ids[b, 1:n] = torch.randint(3, V, (n - 1,), generator=g)
            self.ids[b, n] = 2
            self.mask[b, :n + 1] = 1
        self.tokenizer = _IdTokenizer(V)
        self.tokenizer_max_len = L
        self.seq = self           # model code reads dl.dataset.seq.tokenizer.vocab_size (RRG.py:16)

    def __len__(self):
        return len(self.images)

    def __getitem__(self, i):
        return {"images": self.images[i], "input_ids": self.ids[i], "attention_mask": self.mask[i]}

    def get_collate_fn(self):
        def collate(items):
            return {"images": stack([x["images"] for x in items]), "images_mask": None,
                    "input_ids": torch.stack([x["input_ids"] for x in items]),
                    "attention_mask": torch.stack([x["attention_mask"] for x in items])}
        return collate


class SyntheticImLabel(Dataset):
    """images ~ N(0,1) [3,S,S] + one class label each: the batch dict of the reference's ImLabel (``images``, ``labels``)."""

    def __init__(self, split="train", num_samples=256, image_size=232, num_classes=330, seed=0, **kwargs):
        g = torch.Generator().manual_seed(seed + {"train": 0, "validate": 1, "test": 2}.get(split, 3))
        self.images = torch.randn(num_samples, 3, image_size, image_size, generator=g)
        self.labels = torch.randint(0, num_classes, (num_samples,), generator=g)
        self.num_classes = num_classes

    def __len__(self):
        return len(self.images)

    def __getitem__(self, i):
        return {"images": self.images[i], "labels": self.labels[i]}

    def get_collate_fn(self):
        def collate(items):
            return {"images": stack([x["images"] for x in items]), "labels": torch.stack([x["labels"] for x in items])}
        return collate


class SyntheticSeq2Seq(Dataset):
    """source / target token sequences with the batch dict of the reference's Seq2Seq (``input_ids``, ``attention_mask``,
    ``decoder_input_ids``, ``decoder_attention_mask``): findings -> impression shaped lengths (target ~ a quarter of source)."""

    def __init__(self, split="train", num_samples=256, vocab_size=30522, src_max_len=128, tgt_max_len=32, seed=0, **kwargs):
        g = torch.Generator().manual_seed(seed + {"train": 0, "validate": 1, "test": 2}.get(split, 3))

        def make(L, special):
            ids = torch.full((num_samples, L), 1, dtype=torch.long)
            mask = torch.zeros(num_samples, L, dtype=torch.long)
            for b in range(num_samples):
                n = int(torch.randint(L // 2, L - 1, (1,), generator=g))
                body = torch.randint(3, vocab_size, (n + 1,), generator=g)
                if special:                      # tgt: [CLS] ... [SEP]; src is tokenised without special tokens (TextDataset.py)
                    body[0], body[n] = 0, 2
                ids[b, :n + 1], mask[b, :n + 1] = body, 1
            return ids, mask
        self.src_ids, self.src_mask = make(src_max_len, False)
        self.tgt_ids, self.tgt_mask = make(tgt_max_len, True)
        tok = _IdTokenizer(vocab_size)
        self.src = self.tgt = type("_Side", (), {"tokenizer": tok})()
        self.tgt_tokenizer, self.tgt_tokenizer_max_len = tok, tgt_max_len

    def __len__(self):
        return len(self.src_ids)

    def __getitem__(self, i):
        return {"input_ids": self.src_ids[i], "attention_mask": self.src_mask[i],
                "decoder_input_ids": self.tgt_ids[i], "decoder_attention_mask": self.tgt_mask[i]}

    def get_collate_fn(self):
        return torch.utils.data.dataloader.default_collate


class TensorImSeq(SyntheticImSeq):
    """pre-processed tensors on disk: ``{root}/{split}.pt`` = dict(images, input_ids, attention_mask, vocab_size)."""

    def __init__(self, root, split="train", **kwargs):
        d = torch.load(f"{root}/{split}.pt")
        self.images, self.ids, self.mask = d["images"], d["input_ids"], d["attention_mask"]
        self.tokenizer = _IdTokenizer(int(d["vocab_size"]))
        self.tokenizer_max_len = self.ids.shape[1]
        self.seq = self


from .imseq import DeviceBatchLoader, ImSeq  # noqa: E402,F401  (file-based datasets with the reference's config keys)
from .combos import AnyDataset, ImLabel, ImSeq2Seq, ImSeqAny, ImSeqLabel, LabelDataset, Seq2Seq  # noqa: E402,F401
from .imseq import ImageDataset, TextDataset  # noqa: E402,F401
