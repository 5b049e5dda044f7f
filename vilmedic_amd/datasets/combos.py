"""The reference's remaining file-based dataset compositions, built on datasets/imseq.py's ImageDataset / TextDataset
(ref: vilmedic/datasets/{ImLabel,ImSeqLabel,Seq2Seq,ImSeq2Seq}.py, base/LabelDataset.py, base/utils.py:31-50): same
constructor keys, same on-disk files ({split}.{file}, ckpt_dir/labels.tok, ckpt_dir/vocab.{src,tgt}), same batch-dict keys.
Image-bearing ones hand decoded uint8 images to the device pipeline (``device_transform``) like ImSeq does."""
import os

import torch
from torch.utils.data import Dataset

from .imseq import ImageDataset, ImSeq, TextDataset, _DeviceImages, _same_batch_size, load_file


class Labels:
    """label vocabulary file: first line ``multi-label:<bool>``, then one label per line (base/utils.py:31-50).  The
    reference writes ``list(set(...))`` (hash order, differs between runs); written sorted here -- ids only have to agree
    with the file, which every split reads back."""

    def __init__(self, labels=None):
        if labels is not None:
            self.labels = sorted(set(l for label in labels for l in label.split(",")))
            self.multi_label = max(len(label.split(",")) for label in labels) > 1

    def dump(self, path):
        with open(path, "w") as f:
            f.write("\n".join(str(w) for w in ["multi-label:" + str(self.multi_label)] + self.labels))

    def load(self, path):
        with open(path) as f:
            self.labels = [w.strip() for w in f.readlines()]
        flag = self.labels.pop(0).split(":")[-1]
        assert flag in ("True", "False"), "Bad formatting"
        self.multi_label = flag == "True"
        self.label2idx = {l: i for i, l in enumerate(self.labels)}
        self.idx2label = {i: l for i, l in enumerate(self.labels)}
        return self


class LabelDataset(Dataset):
    def __init__(self, root=None, split=None, file=None, ckpt_dir=None, label_file=None, **kwargs):
        assert split is not None, "Argument split cant be None"
        assert not (file is None and label_file is None), "Please specify a file or a label_file"
        self.root, self.split, self.label_file = root, split, label_file
        self.labels_map = self.labels = None
        raw = None
        if file is not None:
            raw = load_file(os.path.join(root, split + "." + file))
            self.label_file = os.path.join(ckpt_dir, "labels.tok")
            if split == "train" and not os.path.exists(self.label_file):
                os.makedirs(ckpt_dir, exist_ok=True)
                Labels(raw).dump(self.label_file)
        try:
            self.labels_map = Labels().load(self.label_file)
        except FileNotFoundError:
            raise FileNotFoundError("label file does not exists, verify path or start a training")
        if raw is not None:
            self.labels = [self.get_processed_label(l) for l in raw]

    def __len__(self):
        return len(self.labels or [])

    def __getitem__(self, index):
        return {"label": self.labels[index]}

    def get_collate_fn(self):
        def collate_fn(batch):
            return {"labels": torch.stack([s["label"] for s in batch])}
        return collate_fn

    def inference(self, label):
        if not isinstance(label, list):
            label = [label]
        return self.get_collate_fn()([{"label": self.get_processed_label(l)} for l in label])

    def get_processed_label(self, label):
        try:
            classes = label.split(",")
            if not self.labels_map.multi_label:
                return torch.tensor(self.labels_map.label2idx[classes[0]]).long()
            multi_hot = torch.zeros(len(self.labels_map.idx2label))
            multi_hot[[self.labels_map.label2idx[c] for c in classes]] = 1.0
            return multi_hot
        except KeyError:                 # a label absent from the train split: ignore_index (LabelDataset.py:84-86)
            return torch.tensor(-100).long()

    def __repr__(self):
        return "LabelDataset\n{} labels".format(len(self.labels_map.labels))


class ImLabel(_DeviceImages, Dataset):
    def __init__(self, label, image, split, ckpt_dir=None, **kwargs):
        self.split = split
        self.image = ImageDataset(**dict(image), split=split)
        self.label = LabelDataset(**dict(label), split=split, ckpt_dir=ckpt_dir)
        assert len(self.image) == len(self.label)
        self.labels_map = self.label.labels_map
        self._pipeline = None

    def __getitem__(self, index):
        return {**self.image[index], **self.label[index]}

    def __len__(self):
        return len(self.image)

    def get_collate_fn(self):
        def collate_fn(batch):
            return {**self.image.get_collate_fn()(batch), **self.label.get_collate_fn()(batch)}
        return collate_fn

    def inference(self, image=None, label=None):
        """ref: datasets/ImLabel.py:29-41"""
        batch = {}
        if image is not None:
            batch.update(self.device_transform(self.image.inference(image)))
        if label is not None:
            batch.update(self.label.inference(label))
        return _same_batch_size(batch)

    def __repr__(self):
        return "ImLabel\n" + str(self.image) + "\n" + str(self.label)


class ImSeqLabel(_DeviceImages, Dataset):
    def __init__(self, seq, label, image, split, ckpt_dir=None, **kwargs):
        self.split = split
        self.imgseq = ImSeq(seq, image, split=split, ckpt_dir=ckpt_dir)
        self.label = LabelDataset(**dict(label), split=split, ckpt_dir=ckpt_dir)
        assert len(self.imgseq) == len(self.label), str(len(self.imgseq)) + "vs " + str(len(self.label))
        self.image, self.seq = self.imgseq.image, self.imgseq.seq
        self.tokenizer, self.tokenizer_max_len, self.tokenizer_args = self.seq.tokenizer, self.seq.tokenizer_max_len, self.seq.tokenizer_args
        self._pipeline = None

    def __getitem__(self, index):
        return {**self.imgseq[index], **self.label[index]}

    def __len__(self):
        return len(self.label)

    def get_collate_fn(self):
        def collate_fn(batch):
            return {**self.imgseq.get_collate_fn()(batch), **self.label.get_collate_fn()(batch)}
        return collate_fn

    def __repr__(self):
        return "ImSeqLabel\n" + str(self.imgseq) + "\n" + str(self.label)


class AnyDataset(Dataset):
    """one processed line of ``{split}.{file}`` per sample under the key ``name`` (collated as a plain list); ref: base/AnyDataset.py"""

    def __init__(self, root=None, file=None, split=None, processing=None, name=None, **kwargs):
        assert split is not None, "Argument split cannot be None"
        self.root, self.file, self.split, self.name = root, file, split, name or "any"
        self.processing = eval(processing or "lambda x: x")
        with open(os.path.join(root, split + "." + file)) as f:
            self.lines = [self.processing(line.strip()) for line in f]

    def __getitem__(self, index):
        return {self.name: self.lines[index]}

    def __len__(self):
        return len(self.lines)

    def get_collate_fn(self):
        def collate_fn(batch):
            return {self.name: [s[self.name] for s in batch]}
        return collate_fn

    def inference(self, sentences):
        raise NotImplementedError()


class ImSeqAny(_DeviceImages, Dataset):
    """ImSeq + a free-form per-sample field (ref: datasets/ImSeqAny.py)"""

    def __init__(self, seq, any, image, split, ckpt_dir=None, **kwargs):
        self.split = split
        self.imgseq = ImSeq(seq, image, split=split, ckpt_dir=ckpt_dir)
        self.any = AnyDataset(**dict(any), split=split)
        assert len(self.imgseq) == len(self.any), str(len(self.imgseq)) + "vs " + str(len(self.any))
        self.seq, self.image = self.imgseq.seq, self.imgseq.image
        self.tokenizer, self.tokenizer_max_len, self.tokenizer_args = self.seq.tokenizer, self.seq.tokenizer_max_len, self.seq.tokenizer_args
        self._pipeline = None

    def __getitem__(self, index):
        return {**self.imgseq[index], **self.any[index]}

    def __len__(self):
        return len(self.any)

    def get_collate_fn(self):
        def collate_fn(batch):
            return {**self.imgseq.get_collate_fn()(batch), **self.any.get_collate_fn()(batch)}
        return collate_fn

    def __repr__(self):
        return "ImSeqAny\n" + str(self.imgseq) + "\n{} lines".format(len(self.any))


class Seq2Seq(Dataset):
    def __init__(self, src, tgt, split, ckpt_dir=None, **kwargs):
        self.split = split
        self.src = TextDataset(**{**dict(src), "source": "src"}, split=split, ckpt_dir=ckpt_dir)
        self.tgt = TextDataset(**{**dict(tgt), "source": "tgt"}, split=split, ckpt_dir=ckpt_dir)
        self.tgt_tokenizer, self.tgt_tokenizer_max_len = self.tgt.tokenizer, self.tgt.tokenizer_max_len
        assert len(self.src) == len(self.tgt), (len(self.src), len(self.tgt))

    def __getitem__(self, index):
        return {**self.src[index], **self.tgt[index]}

    def __len__(self):
        return len(self.src)

    def get_collate_fn(self):
        def collate_fn(batch):
            tgt = self.tgt.get_collate_fn()(batch)
            tgt["decoder_input_ids"] = tgt.pop("input_ids")
            tgt["decoder_attention_mask"] = tgt.pop("attention_mask")
            return {**self.src.get_collate_fn()(batch), **tgt}
        return collate_fn

    def inference(self, src=None, tgt=None):
        """ref: datasets/Seq2Seq.py:36-51"""
        batch = {}
        if src is not None:
            batch.update(self.src.inference(src))
        if tgt is not None:
            t = self.tgt.inference(tgt)
            batch["decoder_input_ids"], batch["decoder_attention_mask"] = t["input_ids"], t["attention_mask"]
        return _same_batch_size(batch)

    def __repr__(self):
        return "Seq2Seq\n{} source / {} target sentences".format(len(self.src), len(self.tgt))


class ImSeq2Seq(_DeviceImages, Dataset):
    def __init__(self, src, tgt, image, split, ckpt_dir=None, **kwargs):
        self.split = split
        self.seq2seq = Seq2Seq(src, tgt, split, ckpt_dir)
        self.image = ImageDataset(**dict(image), split=split)
        self.src, self.tgt = self.seq2seq.src, self.seq2seq.tgt
        self.tgt_tokenizer, self.tgt_tokenizer_max_len = self.tgt.tokenizer, self.tgt.tokenizer_max_len
        assert len(self.image) == len(self.seq2seq)
        self._pipeline = None

    def __getitem__(self, index):
        return {**self.image[index], **self.seq2seq[index]}

    def __len__(self):
        return len(self.seq2seq)

    def get_collate_fn(self):
        def collate_fn(batch):
            return {**self.seq2seq.get_collate_fn()(batch), **self.image.get_collate_fn()(batch)}
        return collate_fn

    def __repr__(self):
        return "ImSeq2Seq\n" + str(self.seq2seq) + "\n" + str(self.image)
