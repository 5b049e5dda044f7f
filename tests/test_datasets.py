"""File-based ImSeq dataset (reference config keys) -- host logic on the CPU; the device transform itself is covered by
tests/test_image_pipeline.py (GPU) and by the end-to-end GPU test below."""
import os

import numpy as np
import pytest
import torch


def _make_corpus(root, n=6, seed=0):
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(root, "img"), exist_ok=True)
    words = ["heart", "lungs", "clear", "no", "effusion", "the", "is", "normal", "enlarged", "small"]
    for split, cnt in (("train", n), ("validate", 3)):
        paths, sents = [], []
        for i in range(cnt):
            h, w = int(rng.integers(70, 120)), int(rng.integers(70, 120))
            p = os.path.join("img", f"{split}_{i}.png")
            Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(os.path.join(root, p))
            paths.append(p)
            sents.append(" ".join(rng.choice(words, size=int(rng.integers(3, 9)))) + ". 2. No change.")
        open(os.path.join(root, f"{split}.image.tok"), "w").write("\n".join(paths))
        open(os.path.join(root, f"{split}.report.tok"), "w").write("\n".join(sents))
    return words


def test_report_cleaning_matches_reference_fixture(golden):
    from vilmedic_amd.datasets.imseq import r2gen_clean_report
    g = golden("g12_report_cleaning")
    assert [r2gen_clean_report(r) for r in g["reports"]] == g["cleaned"]
    from vilmedic_amd.datasets.imseq import rouge
    assert [rouge(r) for r in g["reports_rouge"]] == g["rouge"]


def test_nltk_based_cleaners_follow_the_documented_token_patterns():
    """ifcc / gloria cleaners (nltk wordpunct_tokenize = ``\\w+|[^\\w\\s]+``; RegexpTokenizer(``\\w+``)); nltk is absent here, so these are
    known-answer cases worked out from the reference's code by hand."""
    from vilmedic_amd.datasets.imseq import gloria_clean_report_chexpert, ifcc_clean_report
    assert ifcc_clean_report("No acute (process). Heart's size: 3.5cm") == "no acute ( process ). heart ' s size : 3 . 5cm"
    assert gloria_clean_report_chexpert("1. No acute process.\n2. x. Stable cardiomegaly \u00e9. 12. ok") == "no acute process stable cardiomegaly"


def test_wordpiece_tokenizer_matches_the_tokenizers_library(tmp_path):
    """the WordPiece restatement against the Rust implementation HF's fast BertTokenizer runs (``tokenizers`` package)"""
    tk = pytest.importorskip("tokenizers")
    from vilmedic_amd.datasets.imseq import WordPieceTokenizer
    vocab = ["[CLS]", "[PAD]", "[SEP]", "[UNK]", "[MASK]", "heart", "##s", "##beat", "en", "##larg", "##ed", "the", "is", "a", "##b", "##c"]
    vf = tmp_path / "vocab.txt"
    vf.write_text("\n".join(vocab))
    mine = WordPieceTokenizer(str(vf))
    ref = tk.Tokenizer(tk.models.WordPiece({w: i for i, w in enumerate(vocab)}, unk_token="[UNK]", max_input_chars_per_word=100))
    ref.pre_tokenizer = tk.pre_tokenizers.WhitespaceSplit()
    texts = ["the hearts is enlarged", "heartbeat heartx abc a ab unknown", "", "x" * 101 + " the", "the  the\tis"]
    for t in texts:
        assert [mine.vocab[w] for w in mine.tokenize(t)] == ref.encode(t, add_special_tokens=False).ids, t
    enc = mine(texts[:2], padding="max_length", truncation=True, max_length=6)
    assert enc.input_ids.tolist()[0] == [0, 11, 5, 6, 12, 2] and enc.attention_mask.tolist()[1] == [1, 1, 1, 1, 1, 1]
    assert mine.decode(torch.tensor([0, 11, 5, 6, 12, 2, 1, 1])) == "the hearts is"


def test_imseq_reads_files_builds_vocab_and_collates(tmp_path):
    from vilmedic_amd.datasets import ImSeq
    root = str(tmp_path)
    _make_corpus(root)
    ds = ImSeq(seq=dict(root=root, file="report.tok", tokenizer=None, tokenizer_max_len=12, processing="r2gen_clean_report", source="tgt"),
               image=dict(root=root, file="image.tok", image_path=root, resize=64, crop=56, ext=".png"), split="train",
               ckpt_dir=os.path.join(root, "ckpt"))
    vocab = open(os.path.join(root, "ckpt", "vocab.tgt")).read().split("\n")
    assert vocab[:5] == ["[CLS]", "[PAD]", "[SEP]", "[UNK]", "[MASK]"] and vocab[5:] == sorted(set(vocab[5:]))     # base/utils.py:16-28
    assert len(ds) == 6 and ds.tokenizer_max_len == 12
    batch = ds.get_collate_fn()([ds[0], ds[1], ds[2]])
    ids, am = batch["input_ids"], batch["attention_mask"]
    assert ids.shape == (3, 12) and ids.dtype == torch.long and (ids[:, 0] == 0).all()                  # [CLS] = 0
    for b in range(3):
        n = int(am[b].sum())
        assert ids[b, n - 1] == 2 or n == 12                                                               # [SEP] = 2 unless truncated
        assert (ids[b, n:] == 1).all()                                                                     # [PAD] = 1
        words = ds.seq.sentences[b][: n - 2]
        assert [vocab[i] for i in ids[b, 1:1 + len(words)].tolist()] == words
    assert len(batch["images_u8"]) == 3 and batch["images_n"] == 1 and batch["images_mask"] is None
    assert all(im.dtype == np.uint8 and im.ndim == 3 and im.shape[2] == 3 for im in batch["images_u8"])
    # the validation split reuses the vocabulary written by the training split
    dv = ImSeq(seq=dict(root=root, file="report.tok", tokenizer=None, tokenizer_max_len=12, source="tgt"),
               image=dict(root=root, file="image.tok", image_path=root, resize=64, crop=56, ext=".png"), split="validate",
               ckpt_dir=os.path.join(root, "ckpt"))
    assert dv.tokenizer.vocab_size == ds.tokenizer.vocab_size and len(dv) == 3


def test_tensor_images_dicom_arithmetic_and_unsupported_keys(tmp_path):
    """ext .npy: pre-processed tensors pass through untouched (identity transform) with the reference's multi-image zero padding
    and sum != 0 mask (base/ImageDataset.py:25-54,93-94); the DICOM branch's grey -> uint8 arithmetic (:124-132); keys that would
    silently change the transform are rejected."""
    from vilmedic_amd.datasets.imseq import ImageDataset, grey_to_u8_rgb
    root = str(tmp_path)
    rng = np.random.default_rng(3)
    paths = []
    for i in range(5):
        np.save(os.path.join(root, f"im{i}.npy"), rng.standard_normal((3, 8, 8)).astype(np.float32))
        paths.append(os.path.join(root, f"im{i}.npy"))
    open(os.path.join(root, "train.image.tok"), "w").write("\n".join([paths[0], paths[1] + "," + paths[2], ",".join(paths[2:5])]))
    ds = ImageDataset(root=root, file="image.tok", split="train", ext=".npy", multi_image=2)
    b = ds.get_collate_fn()([ds[0], ds[1], ds[2]])
    assert b["images"].shape == (3, 2, 3, 8, 8) and b["images"].dtype == torch.float32
    assert b["images_mask"].tolist() == [[True, False], [True, True], [True, True]]
    assert torch.equal(b["images"][0, 0], torch.from_numpy(np.load(paths[0]))) and not b["images"][0, 1].any()
    assert torch.equal(b["images"][2, 1], torch.from_numpy(np.load(paths[3])))           # truncated to the first two of three
    single = ImageDataset(root=root, file="image.tok", split="train", ext=".npy")
    b1 = single.get_collate_fn()([single[0], single[2]])
    assert b1["images"].shape == (2, 3, 8, 8) and b1["images_mask"] is None
    # one .npy holding every sample (file name contains '.npy', ImageDataset.py:65-66)
    np.save(os.path.join(root, "validate.all.npy"), rng.standard_normal((4, 3, 8, 8)).astype(np.float32))
    whole = ImageDataset(root=root, file="all.npy", split="validate", ext=".npy")
    assert len(whole) == 4 and whole.get_collate_fn()([whole[1], whole[3]])["images"].shape == (2, 3, 8, 8)
    g = grey_to_u8_rgb(np.array([[-5.0, 0.0], [100.0, 400.0]]))
    assert g.dtype == np.uint8 and g.shape == (2, 2, 3) and g[:, :, 0].tolist() == [[0, 0], [63, 255]] and (g[:, :, 0] == g[:, :, 2]).all()
    with pytest.raises(NotImplementedError):
        ImageDataset(root=root, file="image.tok", split="train", ext=".npy", custom_transform_train="transforms.Compose([])")
    # relative names resolved against image_path when they do not exist as given
    os.makedirs(os.path.join(root, "sub"), exist_ok=True)
    np.save(os.path.join(root, "sub", "x.npy"), np.zeros((3, 4, 4), np.float32))
    open(os.path.join(root, "test.image.tok"), "w").write("x.npy")
    assert ImageDataset(root=root, file="image.tok", split="test", ext=".npy", image_path=os.path.join(root, "sub")).images == [[os.path.join(root, "sub", "x.npy")]]


def test_label_and_seq2seq_compositions(tmp_path):
    """ImLabel / ImSeqLabel / Seq2Seq / ImSeq2Seq: the reference's files, label-map file and batch-dict keys
    (datasets/{ImLabel,ImSeqLabel,Seq2Seq,ImSeq2Seq}.py, base/LabelDataset.py)"""
    from vilmedic_amd.datasets import ImLabel, ImSeq2Seq, ImSeqLabel, Seq2Seq
    root, ck = str(tmp_path), os.path.join(str(tmp_path), "ckpt")
    _make_corpus(root)
    single = {"train": ["a", "b", "a", "c", "b", "a"], "validate": ["c", "zz", "a"]}
    multi = {"train": ["a,b", "b", "a", "c,a", "b", "a"], "validate": ["c", "a,b", "b"]}
    for split in ("train", "validate"):
        open(os.path.join(root, f"{split}.label.tok"), "w").write("\n".join(single[split]))
        open(os.path.join(root, f"{split}.mlabel.tok"), "w").write("\n".join(multi[split]))
        n = len(single[split])
        open(os.path.join(root, f"{split}.findings.tok"), "w").write("\n".join("The heart is normal, lungs clear!" for _ in range(n)))
        open(os.path.join(root, f"{split}.impression.tok"), "w").write("\n".join("No change." for _ in range(n)))
    image = dict(root=root, file="image.tok", image_path=root, resize=64, crop=56, ext=".png")
    tr = ImLabel(label=dict(root=root, file="label.tok"), image=image, split="train", ckpt_dir=ck)
    assert open(os.path.join(ck, "labels.tok")).read().split("\n") == ["multi-label:False", "a", "b", "c"]
    b = tr.get_collate_fn()([tr[i] for i in range(4)])
    assert b["labels"].tolist() == [0, 1, 0, 2] and b["labels"].dtype == torch.long and len(b["images_u8"]) == 4
    va = ImLabel(label=dict(root=root, file="label.tok"), image=image, split="validate", ckpt_dir=ck)
    assert va.get_collate_fn()([va[i] for i in range(3)])["labels"].tolist() == [2, -100, 0]         # unseen label -> ignore_index
    os.remove(os.path.join(ck, "labels.tok"))
    seq = dict(root=root, file="report.tok", tokenizer=None, tokenizer_max_len=12, processing="r2gen_clean_report", source="tgt")
    ml = ImSeqLabel(seq=seq, label=dict(root=root, file="mlabel.tok"), image=image, split="train", ckpt_dir=ck)
    b = ml.get_collate_fn()([ml[0], ml[3]])
    assert b["labels"].tolist() == [[1.0, 1.0, 0.0], [1.0, 0.0, 1.0]] and b["input_ids"].shape == (2, 12) and len(b["images_u8"]) == 2
    assert ml.tokenizer is ml.seq.tokenizer and hasattr(ml, "device_transform")
    ck = os.path.join(root, "ckpt_s2s")          # (a ckpt_dir that already holds vocab.tgt would be reused, as in the reference)
    s2s = Seq2Seq(src=dict(root=root, file="findings.tok", tokenizer=None, tokenizer_max_len=10, processing="rouge"),
                  tgt=dict(root=root, file="impression.tok", tokenizer=None, tokenizer_max_len=6, processing="rouge"), split="train", ckpt_dir=ck)
    b = s2s.get_collate_fn()([s2s[0], s2s[1]])
    src_vocab = open(os.path.join(ck, "vocab.src")).read().split("\n")
    assert [src_vocab[i] for i in b["input_ids"][0].tolist()] == ["the", "heart", "is", "normal", "lungs", "clear"] + ["[PAD]"] * 4   # no specials
    assert b["decoder_input_ids"][0].tolist()[:4] == [0, 6, 5, 2] and b["decoder_attention_mask"][0].tolist() == [1, 1, 1, 1, 0, 0]
    assert s2s.tgt_tokenizer is s2s.tgt.tokenizer and s2s.tgt_tokenizer_max_len == 6
    i2s = ImSeq2Seq(src=dict(root=root, file="findings.tok", tokenizer=None, tokenizer_max_len=10, processing="rouge"),
                    tgt=dict(root=root, file="impression.tok", tokenizer=None, tokenizer_max_len=6, processing="rouge"),
                    image=image, split="validate", ckpt_dir=ck)
    b = i2s.get_collate_fn()([i2s[0]])
    assert set(b) == {"input_ids", "attention_mask", "decoder_input_ids", "decoder_attention_mask", "images_u8", "images_n", "images_mask"}
    from vilmedic_amd.datasets import ImSeqAny
    for split in ("train", "validate"):
        open(os.path.join(root, f"{split}.meta.tok"), "w").write("\n".join("study-%d" % i for i in range(len(single[split]))))
    isa = ImSeqAny(seq=seq, any=dict(root=root, file="meta.tok", name="study", processing="lambda x: x.upper()"), image=image, split="train",
                   ckpt_dir=os.path.join(root, "ckpt"))
    b = isa.get_collate_fn()([isa[1], isa[4]])
    assert b["study"] == ["STUDY-1", "STUDY-4"] and b["input_ids"].shape == (2, 12) and len(b["images_u8"]) == 2


@pytest.mark.gpu
def test_imseq_trains_rrg_end_to_end_through_the_device_pipeline(tmp_path):
    """image files + report text -> ImSeq -> DeviceBatchLoader (HIP resize / crop / flip / normalise) -> RRG training + beam-search
    validation through bin/train.py's Trainor."""
    from vilmedic_amd.config import executor_view, get_config
    from vilmedic_amd.executors import Trainor
    root = str(tmp_path)
    _make_corpus(root, n=16)
    cfg_path = os.path.join(root, "rrg-files.yml")
    open(cfg_path, "w").write(f"""
name: rrg_files
ckpt_dir: {root}/ckpt
dataset:
  proto: ImSeq
  image: {{root: {root}, file: image.tok, image_path: {root}, resize: 40, crop: 32, ext: .png}}
  seq: {{root: {root}, file: report.tok, tokenizer: null, tokenizer_max_len: 16, processing: r2gen_clean_report, source: tgt}}
model:
  proto: RRG
  decoder: {{proto: null, hidden_size: 128, num_attention_heads: 2, intermediate_size: 256, num_hidden_layers: 2,
            max_position_embeddings: 64, bos_token_id: 0, pad_token_id: 1, eos_token_id: 2}}
  cnn: {{proto: VisualEncoder, backbone: vit, permute: no_permute, dropout_out: 0.0, image_size: 32, patch_size: 8, hidden_size: 128,
        num_attention_heads: 2, intermediate_size: 256, num_hidden_layers: 2}}
trainor: {{optimizer: FusedAdam, optim_params: {{lr: 0.003}}, batch_size: 8, epochs: 2, early_stop: 10, early_stop_metric: training_loss, eval_start: 0}}
validator: {{batch_size: 4, beam_width: 2, splits: [validate]}}
""")
    cfg = get_config(cfg_path, [])
    t = executor_view(cfg, "trainor")
    t["validator_view"] = executor_view(cfg, "validator")
    tr = Trainor(t, seed=0)
    batch = next(iter(tr.dl))
    assert batch["images"].shape == (8, 3, 32, 32) and batch["images"].is_cuda and batch["images"].dtype == torch.float32
    tr.start()
    assert len([f for f in os.listdir(os.path.join(root, "ckpt")) if f.endswith(".pth")]) == 1


def _imseq(root, split="validate", n=6):
    from vilmedic_amd.datasets import ImSeq
    _make_corpus(root, n=n)
    ds = None
    for sp in (["train"] if split == "train" else ["train", split]):          # the training split writes the vocabulary the others read
        ds = ImSeq(seq=dict(root=root, file="report.tok", tokenizer=None, tokenizer_max_len=12, processing="r2gen_clean_report", source="tgt"),
                   image=dict(root=root, file="image.tok", image_path=root, resize=40, crop=32, ext=".png"), split=sp,
                   ckpt_dir=os.path.join(root, "ckpt"))
    return ds


def test_prefetch_thread_packs_decoded_images_into_one_staging_buffer(tmp_path):
    """ImSeq.stage_batch under the PrefetchLoader: ``images_packed`` holds every decoded image at its 16-byte aligned offset"""
    from torch.utils.data import DataLoader
    from vilmedic_amd.datasets import PrefetchLoader
    from vilmedic_amd.datasets.device_pipeline import packed_layout
    ds = _imseq(str(tmp_path), split="train")
    pl = PrefetchLoader(DataLoader(ds, batch_size=3, collate_fn=ds.get_collate_fn()), device=None)
    nb = 0
    for batch in pl:
        imgs = batch["images_u8"]
        offs, sizes, total = packed_layout(imgs)
        packed = batch["images_packed"]
        assert packed.dtype == torch.uint8 and packed.numel() == total and all(o % 16 == 0 for o in offs)
        for im, o, (h, w) in zip(imgs, offs, sizes):
            assert np.array_equal(packed[o:o + h * w * 3].numpy().reshape(h, w, 3), np.asarray(im))
        nb += 1
    assert nb == len(pl) > 0


@pytest.mark.gpu
def test_device_transform_of_a_staged_batch_equals_the_unstaged_one(tmp_path):
    from torch.utils.data import DataLoader
    from vilmedic_amd.datasets import DeviceBatchLoader, PrefetchLoader
    ds = _imseq(str(tmp_path), split="validate", n=16)     # evaluation transform: no random draws
    plain = [b["images"].clone() for b in DeviceBatchLoader(DataLoader(ds, batch_size=3, collate_fn=ds.get_collate_fn()))]
    staged = [b["images"].clone() for b in DeviceBatchLoader(PrefetchLoader(DataLoader(ds, batch_size=3, collate_fn=ds.get_collate_fn())))]
    assert len(plain) == len(staged) > 0
    for a, b in zip(plain, staged):
        assert a.is_cuda and torch.equal(a, b)
