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
