"""File-based ImSeq dataset with the reference's config keys, feeding the device-side image pipeline.

ref: vilmedic/datasets/ImSeq.py:8-32, base/ImageDataset.py:63-150 (file list, open_image, transforms, multi-image collate),
     base/TextDataset.py:30-119 (sentence file, Vocab-built BertTokenizer, padding / truncation, collate),
     base/utils.py:16-28 (Vocab).
Differences: the worker processes only DECODE (PIL -> uint8 HWC); Resize / RandomCrop / RandomHorizontalFlip
/ ToTensor / Normalize run as one HIP kernel on the whole batch (``DeviceImagePipeline``, bit-exact with the PIL transforms), so
the host ships 3 B/pixel instead of 12.  ``DeviceBatchLoader`` applies it while iterating, so training / validation loops see
the reference's batch dict (``images`` fp32 [B,3,crop,crop] or [B,N,3,crop,crop], ``images_mask``, ``input_ids``, ``attention_mask``).
Pretrained tokenizers named by a hub id need a download; a local directory / vocab file works."""
import itertools
import os

import numpy as np
import torch
from torch.utils.data import Dataset

from .device_pipeline import DeviceImagePipeline, pack_host, packed_layout
from .prefetch import stack


def r2gen_clean_report(report):
    """the report normalisation the RRG configs name (``processing: r2gen_clean_report``; ref:
    vilmedic/datasets/base/papers/report_preprocessing.py:8-23, after R2Gen's tokenizer): bounded collapsing of ``__`` /
    double spaces / ``..`` (7 / 6 / 8 passes, as the reference chains them), enumeration markers dropped, lower-cased sentences
    stripped of punctuation and re-joined with `` . ``.  Pinned by tests/golden/g12_report_cleaning.pt."""
    import re
    t = report.replace("\n", " ")
    for _ in range(7):
        t = t.replace("__", "_")
    for _ in range(6):
        t = t.replace("  ", " ")
    for _ in range(8):
        t = t.replace("..", ".")
    t = t.replace("1. ", "")
    for k in (2, 3, 4, 5):
        t = t.replace(". %d. " % k, ". ")
    for k in (2, 3, 4, 5):
        t = t.replace(" %d. " % k, ". ")
    sents = t.strip().lower().split(". ")

    def clean(x):
        x = x.replace('"', "").replace("/", "").replace("\\", "").replace("'", "").strip().lower()
        return re.sub(r"[.,?;*!%^&_+():-\[\]{}]", "", x)      # the reference's class includes the range ')'..'-' = ) * + , -
    tokens = [clean(x) for x in sents]
    if tokens == [""]:
        return ""
    return " . ".join(tokens) + " ."


def rouge(text, use_stemmer=False):
    """ROUGE-style normalisation the RRS configs name (``processing: rouge``; ref: report_preprocessing.py:69-108, after
    google-research/rouge tokenize.py): lower-case, every run of characters outside [a-z0-9] becomes one space, empty tokens
    dropped.  Pinned by tests/golden/g12_report_cleaning.pt (use_stemmer needs nltk's Porter stemmer: not available here)."""
    import re
    if use_stemmer:
        raise NotImplementedError("rouge(use_stemmer=True) needs nltk's PorterStemmer")
    return " ".join(t for t in re.sub(r"[^a-z0-9]+", " ", text.lower()).split(" ") if t)


def ifcc_clean_report(report):
    """lower-case + nltk ``wordpunct_tokenize`` (= the regular expression ``\\w+|[^\\w\\s]+``), space-joined
    (ref: report_preprocessing.py:26-30).  nltk is absent from this image: restated from its documented pattern, not pinned."""
    import re
    return " ".join(re.findall(r"\w+|[^\w\s]+", report.lower()))


def gloria_clean_report_chexpert(report):
    """GLoRIA's report normalisation (ref: report_preprocessing.py:33-65): split on ``<digits>.`` and ``.``, keep the
    ``\\w+`` tokens (nltk RegexpTokenizer) of each lower-cased piece with more than one token, ASCII-only, space-joined.
    Restated from the documented tokenizer pattern (nltk absent here), not pinned."""
    import re
    pieces = [s for point in re.split(r"[0-9]+\.", report.replace("\n", " ")) for s in point.split(".")]
    out = []
    for piece in pieces:
        tokens = re.findall(r"\w+", piece.replace("\ufffd\ufffd", " ").lower())
        if len(tokens) <= 1:
            continue
        tokens = [t.encode("ascii", "ignore").decode("ascii") for t in tokens]
        out.append(" ".join(t for t in tokens if t))
    return " ".join(out)


PROCESSING = {"r2gen_clean_report": r2gen_clean_report, "rouge": rouge, "ifcc_clean_report": ifcc_clean_report,
              "gloria_clean_report_chexpert": gloria_clean_report_chexpert}


def load_file(path):
    with open(path, "r") as f:
        content = f.read().strip()
    return [s for s in content.split("\n")]


class Vocab:
    """ref: base/utils.py:16-28 -- specials first, then the sorted word set"""

    def __init__(self, sentences, pad_token="[PAD]", eos_token="[SEP]", bos_token="[CLS]", unk_token="[UNK]", mask_token="[MASK]"):
        words = list(itertools.chain(*sentences))
        self.words = [bos_token, pad_token, eos_token, unk_token, mask_token] + sorted(set(words))

    def dump(self, path):
        open(path, "w").write("\n".join(str(w) for w in self.words))


class _Encoding:
    def __init__(self, input_ids, attention_mask):
        self.input_ids, self.attention_mask = input_ids, attention_mask

    def __getitem__(self, k):
        return getattr(self, k)


class WordPieceTokenizer:
    """``BertTokenizer(vocab_file=..., do_basic_tokenize=False)`` as the reference builds it (base/TextDataset.py:91-92) for the
    call pattern of its collate (:99-107,113-115): whitespace split, greedy longest-match-first WordPiece with ``##``
    continuations and a 100-character word limit (hf 4.55.3 models/bert/tokenization_bert.py: WordpieceTokenizer.tokenize),
    ``[CLS] ... [SEP]``, truncation and ``[PAD]`` padding to ``max_length``.  (transformers >= 5 no longer builds a
    BertTokenizer from a bare vocab file, hence this restatement; checked against the ``tokenizers`` library's WordPiece.)"""

    def __init__(self, vocab_file, unk_token="[UNK]", sep_token="[SEP]", pad_token="[PAD]", cls_token="[CLS]", mask_token="[MASK]"):
        words = open(vocab_file, encoding="utf-8").read().split("\n")
        self.vocab = {w.rstrip("\n"): i for i, w in enumerate(words)}
        self.ids_to_tokens = {i: w for w, i in self.vocab.items()}
        self.unk_token, self.sep_token, self.pad_token, self.cls_token, self.mask_token = unk_token, sep_token, pad_token, cls_token, mask_token
        self.unk_token_id, self.sep_token_id = self.vocab[unk_token], self.vocab[sep_token]
        self.pad_token_id, self.cls_token_id = self.vocab[pad_token], self.vocab[cls_token]
        self.all_special_ids = {self.vocab[t] for t in (unk_token, sep_token, pad_token, cls_token, mask_token) if t in self.vocab}
        self.max_input_chars_per_word = 100

    @property
    def vocab_size(self):
        return len(self.vocab)

    def get_vocab(self):
        return dict(self.vocab)

    def tokenize(self, text):
        out = []
        for token in text.strip().split():
            chars = list(token)
            if len(chars) > self.max_input_chars_per_word:
                out.append(self.unk_token)
                continue
            start, pieces, bad = 0, [], False
            while start < len(chars):
                end, cur = len(chars), None
                while start < end:
                    sub = "".join(chars[start:end])
                    if start > 0:
                        sub = "##" + sub
                    if sub in self.vocab:
                        cur = sub
                        break
                    end -= 1
                if cur is None:
                    bad = True
                    break
                pieces.append(cur)
                start = end
            out.extend([self.unk_token] if bad else pieces)
        return out

    def __call__(self, texts, return_tensors="pt", padding=True, add_special_tokens=True, truncation=False, max_length=None, **kw):
        rows = []
        for t in ([texts] if isinstance(texts, str) else texts):
            ids = [self.vocab[w] for w in self.tokenize(t)]
            room = (max_length - (2 if add_special_tokens else 0)) if (truncation and max_length is not None) else None
            if room is not None:
                ids = ids[:max(room, 0)]
            rows.append(([self.cls_token_id] + ids + [self.sep_token_id]) if add_special_tokens else ids)
        width = max_length if padding == "max_length" else max(len(r) for r in rows)
        input_ids = torch.full((len(rows), width), self.pad_token_id, dtype=torch.long)
        attention_mask = torch.zeros(len(rows), width, dtype=torch.long)
        for b, r in enumerate(rows):
            input_ids[b, :len(r)] = torch.tensor(r, dtype=torch.long)
            attention_mask[b, :len(r)] = 1
        return _Encoding(input_ids, attention_mask)

    def decode(self, ids, skip_special_tokens=True, clean_up_tokenization_spaces=False):
        ids = ids.tolist() if hasattr(ids, "tolist") else list(ids)
        toks = [self.ids_to_tokens[int(i)] for i in ids if not (skip_special_tokens and int(i) in self.all_special_ids)]
        return " ".join(toks).replace(" ##", "").strip()


def _dicom_to_u8(path):
    """ref: base/ImageDataset.py:124-132 -- VOI LUT when the file carries a window, clip at 0, scale by the maximum to 0..255,
    grey replicated to three channels (what ``Image.fromarray(img).convert('RGB')`` does with an 8-bit grey image)"""
    try:
        import pydicom
        from pydicom.pixel_data_handlers.util import apply_voi_lut
    except ImportError as e:
        raise NotImplementedError("ext '.dcm' needs the pydicom package, which is not installed here") from e
    ds = pydicom.dcmread(path)
    img = (apply_voi_lut(ds.pixel_array, ds) if "WindowWidth" in ds else ds.pixel_array).astype(float)
    return grey_to_u8_rgb(img)


def grey_to_u8_rgb(img):
    """float grey image -> uint8 [H,W,3]: ``uint8(max(img,0) / img.max() * 255)`` replicated (the DICOM branch's arithmetic)"""
    img = np.uint8((np.maximum(img, 0) / img.max()) * 255.0)
    return np.repeat(img[:, :, None], 3, axis=2)


def open_image(image, ext):
    """-> uint8 [H,W,3] numpy for the device pipeline (ref: base/ImageDataset.py:111-140)"""
    if isinstance(image, np.ndarray):
        arr = image
    elif ext in (".jpg", ".jpeg", ".png"):
        from PIL import Image
        arr = np.asarray(Image.open(image).convert("RGB"))
    elif ext == ".dcm":
        arr = _dicom_to_u8(image)
    else:
        raise NotImplementedError("Image extension {} not implemented".format(ext))
    if arr.dtype != np.uint8 or arr.ndim != 3 or arr.shape[2] != 3:
        raise ValueError("decoded image must be uint8 [H,W,3], got {} {}".format(arr.dtype, arr.shape))
    return np.array(arr, order="C")      # an owned, writable copy (PIL hands out read-only buffers)


def open_tensor_image(image):
    """ext .npy / .npz: already pre-processed tensors, identity transform (ref: base/ImageDataset.py:93-94,134-139)"""
    if isinstance(image, str):
        image = np.load(image)
    return torch.from_numpy(image) if isinstance(image, np.ndarray) else image


class ImageDataset(Dataset):
    def __init__(self, root=None, file=None, split=None, image_path=None, resize=256, crop=224, ext=".jpg", multi_image=None,
                 called_by_ensemblor=None, custom_transform_train=None, custom_transform_validate=None, hf_dataset=None,
                 hf_processor=None, **kwargs):
        assert split is not None, "Argument split cant be None"
        for name, val in (("custom_transform_train", custom_transform_train), ("custom_transform_validate", custom_transform_validate),
                          ("hf_dataset", hf_dataset), ("hf_processor", hf_processor)):
            if val is not None:       # torchvision transform strings / hub datasets / HF processors: not available in this build
                raise NotImplementedError(f"ImageDataset({name}=...) is not supported by the device image pipeline")
        self.root, self.file, self.split, self.image_path = root, file, split, image_path
        self.resize, self.crop, self.ext = int(resize), int(crop), ext
        self.multi_image = multi_image or 0
        self.tensor_images = ext in (".npy", ".npz")
        file_path = os.path.join(root, split + "." + file) if file is not None else None
        if file_path is None:                         # no data file: transform only, for inference on a shipped checkpoint
            self.images = None
        elif ".npy" in file_path:                     # one array holding every sample's (pre-processed) image (ImageDataset.py:65-66)
            self.images = [[x] for x in np.load(file_path)]
        else:
            self.images = []
            for line in load_file(file_path):
                paths = []
                for p in line.split(","):
                    p = p.strip()
                    if not os.path.exists(p) and image_path and os.path.exists(os.path.join(image_path, p)):
                        p = os.path.join(image_path, p)
                    assert os.path.exists(p), f"Image path does not exist: {p}"
                    paths.append(p)
                self.images.append(paths)
        self.pipeline_split = "validate" if called_by_ensemblor else split

    def __len__(self):
        return len(self.images or [])

    def inference(self, image):
        """image path(s) / decoded arrays -> the collated batch: one sample per list entry; an entry may itself be a list (the
        sample's several images, with ``multi_image``)"""
        samples = image if isinstance(image, (list, tuple)) else [image]
        rows = []
        for smp in samples:
            items = smp if isinstance(smp, (list, tuple)) else [smp]
            rows.append({"image": [open_tensor_image(i) if self.tensor_images else open_image(i, self.ext) for i in items]})
        return self.get_collate_fn()(rows)

    def __getitem__(self, index):
        if self.tensor_images:
            return {"image": [open_tensor_image(p) for p in self.images[index]]}
        return {"image": [open_image(p, self.ext) for p in self.images[index]]}

    def get_collate_fn(self):
        def collate_fn(batch):
            n = self.multi_image if self.multi_image and self.multi_image > 1 else 1
            if self.tensor_images:                     # vilmedic_collate (ImageDataset.py:25-54): stack, pad with zero images
                if n == 1:
                    return {"images": stack([s["image"][0] for s in batch]), "images_mask": None}
                rows = []
                for s in batch:
                    cur = list(s["image"][:n])
                    cur += [cur[0].new_zeros(cur[0].size()) for _ in range(n - len(cur))]
                    rows.append(torch.stack(cur))
                images = torch.stack(rows)
                return {"images": images, "images_mask": images.sum(dim=(2, 3, 4)) != 0}
            imgs, mask = [], []
            for s in batch:
                cur = list(s["image"][:n])
                mask.append([1] * len(cur) + [0] * (n - len(cur)))
                imgs.extend(cur + [None] * (n - len(cur)))
            return {"images_u8": imgs, "images_n": n, "images_mask": torch.tensor(mask, dtype=torch.bool) if n > 1 else None}
        return collate_fn


class TextDataset(Dataset):
    def __init__(self, root=None, file=None, split=None, ckpt_dir=None, processing=None, tokenizer=None, tokenizer_max_len=None,
                 vocab_file=None, source="src", **kwargs):
        assert source in ["src", "tgt"]
        assert split is not None, "Argument split cannot be None"
        assert not (file is not None and vocab_file is not None), "You cannot mention both a data file and a vocab file"
        assert not (vocab_file is not None and tokenizer is not None), "You cannot mention both a pretrained tokenizer and a vocab file"
        assert not (source == "tgt" and tokenizer_max_len is None), "You must specify tokenizer_max_len for source tgt"
        assert file is not None or vocab_file is not None, "Either a data file or a vocab file must be specified"
        if kwargs.get("hf_dataset") is not None:
            raise NotImplementedError("TextDataset(hf_dataset=...): hub datasets cannot be fetched here (no network)")
        self.split, self.source, self.tokenizer_max_len = split, source, tokenizer_max_len
        self.processing = PROCESSING[processing] if processing in PROCESSING else eval(processing or "lambda x: x")
        # a vocab_file without a data file: tokenizer only, for inference on a shipped checkpoint (TextDataset.py:46-49,70-72)
        self.sentences = None if file is None else \
            [self.processing(s.strip()).split() for s in load_file(os.path.join(root, split + "." + file))]
        if tokenizer is not None:
            from transformers import AutoTokenizer
            self.tokenizer = AutoTokenizer.from_pretrained(tokenizer)       # a local directory works; hub ids need a download
        else:
            if vocab_file is None:
                vocab_file = os.path.join(ckpt_dir, "vocab.{}".format(source))
                if split == "train" and not os.path.exists(vocab_file):
                    os.makedirs(ckpt_dir, exist_ok=True)
                    Vocab(self.sentences).dump(vocab_file)
            self.tokenizer = WordPieceTokenizer(vocab_file)
        self.tokenizer_args = {"return_tensors": "pt", "padding": True, "add_special_tokens": True}
        if source == "src":
            self.tokenizer_args.update({"add_special_tokens": False})
        if tokenizer_max_len is not None:
            self.tokenizer_args.update({"padding": "max_length", "truncation": True, "max_length": tokenizer_max_len})

    def __len__(self):
        return len(self.sentences or [])

    def __getitem__(self, index):
        return {"{}_seq".format(self.source): " ".join(self.sentences[index])}

    def inference(self, sentences):
        """raw sentence(s) -> the collated batch (processing + tokenizer), the entry point the zoo loader's datasets expose"""
        if isinstance(sentences, str):
            sentences = [sentences]
        key = "{}_seq".format(self.source)
        return self.get_collate_fn()([{key: " ".join(self.processing(s.strip()).split())} for s in sentences])

    def get_collate_fn(self):
        def collate_fn(batch):
            seq = self.tokenizer([s["{}_seq".format(self.source)] for s in batch], **self.tokenizer_args)
            return {"input_ids": seq.input_ids, "attention_mask": seq.attention_mask}
        return collate_fn


class _DeviceImages:
    """datasets with an ``image`` ImageDataset: the collated batch carries decoded uint8 images; this turns them into the
    reference's ``images`` tensor on the device (one launch of the Resize / crop / flip / normalise kernel per batch)"""
    _pipeline = None

    def stage_batch(self, batch, ring):
        """(PrefetchLoader's thread) the decoded images of a collated batch packed into one pinned byte buffer"""
        if "images_u8" not in batch:
            return batch
        real = [im for im in batch["images_u8"] if im is not None]
        batch = dict(batch)
        batch["images_packed"] = pack_host(real, lambda n: ring.flat("images_packed", n))
        return batch

    def device_transform(self, batch):
        """decoded uint8 images of a collated batch -> the reference's ``images`` tensor, on the device"""
        if "images_u8" not in batch:
            return batch
        if self._pipeline is None:
            self._pipeline = DeviceImagePipeline(self.image.pipeline_split, self.image.resize, self.image.crop)
        batch = dict(batch)
        imgs, n = batch.pop("images_u8"), batch.pop("images_n")
        real = [im for im in imgs if im is not None]
        packed = batch.pop("images_packed", None)
        if packed is not None:                          # staged by the prefetch thread (stage_batch): already one buffer in HBM
            offs, sizes, _ = packed_layout(real)
            out = self._pipeline.run(packed, offs, sizes)
        else:
            out = self._pipeline(real)
        if n > 1:                                       # multi-image: zero images where the sample has fewer (ref vilmedic_collate)
            full = out.new_zeros(len(imgs), *out.shape[1:])
            idx = torch.tensor([i for i, im in enumerate(imgs) if im is not None], device=out.device)
            full.index_copy_(0, idx, out)
            out = full.view(len(imgs) // n, n, *out.shape[1:])
        batch["images"] = out
        return batch


def _same_batch_size(batch):
    sizes = {len(v) for k, v in batch.items() if v is not None and hasattr(v, "__len__") and k not in ("images_u8",)}
    assert len(sizes) <= 1, "elements in batch do not have the same size"
    return batch


class ImSeq(_DeviceImages, Dataset):
    def __init__(self, seq, image, split, ckpt_dir=None, **kwargs):
        self.split = split
        self.seq = TextDataset(**dict(seq), split=split, ckpt_dir=ckpt_dir)
        self.image = ImageDataset(**dict(image), split=split)
        assert len(self.image) == len(self.seq)
        self.tokenizer = self.seq.tokenizer
        self.tokenizer_max_len = self.seq.tokenizer_max_len
        self.tokenizer_args = self.seq.tokenizer_args
        self._pipeline = None

    def __getitem__(self, index):
        return {**self.image[index], **self.seq[index]}

    def __len__(self):
        return len(self.image)

    def get_collate_fn(self):
        def collate_fn(batch):
            return {**self.seq.get_collate_fn()(batch), **self.image.get_collate_fn()(batch)}
        return collate_fn

    def inference(self, seq=None, image=None):
        """ref: datasets/ImSeq.py:39-52 -- raw sentences and / or image paths -> a model-ready batch"""
        batch = {}
        if image is not None:
            batch.update(self.device_transform(self.image.inference(image)))
        if seq is not None:
            batch.update(self.seq.inference(seq))
        return _same_batch_size(batch)

    def __repr__(self):
        return "ImSeq\n{} sentences, {} image lists".format(len(self.seq), len(self.image))


class DeviceBatchLoader:
    """wraps a DataLoader whose dataset has ``device_transform``: iterating yields device-side pre-processed batches"""

    def __init__(self, loader):
        self.loader = loader
        self.dataset = loader.dataset

    def __iter__(self):
        base = self.dataset.dataset if isinstance(self.dataset, torch.utils.data.Subset) else self.dataset
        for batch in self.loader:
            yield base.device_transform(batch)

    def __len__(self):
        return len(self.loader)
