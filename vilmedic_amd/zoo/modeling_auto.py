"""AutoModel -- ref: vilmedic/zoo/modeling_auto.py:42-117: rebuild (model, dataset) from a published checkpoint directory
(``config.yml`` + one ``*.pth`` + the vocabulary / label files its dataset section names) and load the weights ``strict=True``.

The reference downloads the directory from Google Drive / the HF hub into ``~/.cache/vilmedic/zoo/models/<name>``; there is no
network here, so ``from_pretrained`` takes either a directory path or one of the zoo names and expects the files to be present
already (``VILMEDIC_ZOO_DIR`` overrides the cache root).  What is reproduced: the checkpoint wire format
({"model": state_dict, "__version__": ...}; ``module.`` prefixes and pre-1.3.2 ``enc.0.cnn.`` / ``enc.1.`` names migrated,
executors/utils.py:26-34), the dataset built with ``split='test'`` and no data files (tokenizer / label map / image transform
only), vocabulary paths re-rooted at the checkpoint directory, the model in eval mode on the GPU when there is one."""
import copy
import glob
import os

import torch

from ..config import load_yaml, wrap
from ..datasets import *  # noqa: F401,F403  (eval(proto) namespace)
from ..executors.utils import vilmedic_state_dict_versioning
from ..models import *  # noqa: F401,F403

# names the reference publishes (zoo/modeling_auto.py:16-39); the download ids are not reproduced: nothing can be fetched here
MODEL_ZOO = [
    "selfsup/gloria-chexpert", "selfsup/gloria-mimic-48", "selfsup/convirt-mimic-balanced", "selfsup/convirt-mimic",
    "selfsup/convirt-padchest-16", "selfsup/convirt-padchest-32", "selfsup/convirt-indiana-16", "selfsup/convirt-indiana-32",
    "selfsup/convirt-indiana-64", "rrg/biomed-roberta-baseline-mimic", "rrg/biomed-roberta-baseline-indiana", "rrg/baseline-padchest",
    "rrg/baseline-mimic", "rrs/biomed-roberta-baseline-mimic", "rrs/biomed-roberta-baseline-indiana", "mvqa/mvqa-imageclef",
]


def zoo_cache_dir():
    return os.environ.get("VILMEDIC_ZOO_DIR") or os.path.join(os.path.expanduser("~"), ".cache", "vilmedic", "zoo", "models")


def edit_vocab_path_in_dict(obj, keys, replace_value):
    """re-root ``vocab_file`` / ``label_file`` entries, at any depth, at the checkpoint directory (zoo/utils.py:8-15)"""
    for k, v in obj.items():
        if isinstance(v, dict):
            obj[k] = edit_vocab_path_in_dict(v, keys, replace_value)
    for key in keys:
        if obj.get(key) is not None:
            obj[key] = os.path.join(replace_value, obj[key])
    return obj


class _DL:
    """what models read from ``dl``: ``dl.dataset`` (the reference wraps the dataset in a DataLoader it never iterates)"""

    def __init__(self, dataset):
        self.dataset = dataset


class AutoModel:
    def __init__(self):
        raise EnvironmentError("AutoModel is designed to be instantiated using the "
                               "`AutoModel.from_pretrained(pretrained_model_name_or_path)` method.")

    @staticmethod
    def from_config(config):
        raise NotImplementedError()

    @staticmethod
    def from_pretrained(pretrained_model_name):
        if os.path.isdir(pretrained_model_name):
            checkpoint_dir = pretrained_model_name
        else:
            if pretrained_model_name not in MODEL_ZOO:
                raise KeyError("Unrecognized pretrained_model_name {}. Model name should be one of {} or a checkpoint directory."
                               .format(pretrained_model_name, MODEL_ZOO))
            checkpoint_dir = os.path.join(zoo_cache_dir(), pretrained_model_name)
            if not glob.glob(os.path.join(checkpoint_dir, "*.pth")):
                raise FileNotFoundError("{} is not in {}: zoo checkpoints cannot be downloaded here (no network); copy the "
                                        "checkpoint directory (config.yml, *.pth, vocabulary files) there or pass its path"
                                        .format(pretrained_model_name, checkpoint_dir))
        checkpoint = glob.glob(os.path.join(checkpoint_dir, "*.pth"))
        assert len(checkpoint) == 1, "More than one or no checkpoint found"
        state_dict = torch.load(checkpoint[0], map_location="cpu")
        try:
            config = wrap(load_yaml(os.path.join(checkpoint_dir, "config.yml")))
        except FileNotFoundError:
            raise FileNotFoundError("The file config.yml is missing")
        try:
            model_config, dataset_config = copy.deepcopy(config["model"]), copy.deepcopy(config["dataset"])
        except KeyError:
            raise KeyError("This config doesnt have a model and/or dataset key. Deprecated checkpoint of vilmedic version?")
        classname = dataset_config.pop("proto")
        dataset_config = edit_vocab_path_in_dict(dataset_config, ["vocab_file", "label_file"], checkpoint_dir)
        try:
            dataset_cls = eval(classname)
        except NameError:
            raise NameError("Dataset {} does not exist anymore. Deprecated checkpoint of vilmedic?".format(classname))
        dataset = dataset_cls(split="test", ckpt_dir=None, **dataset_config)
        classname = model_config.pop("proto")
        try:
            model_cls = eval(classname)
        except NameError:
            raise NameError("Model {} does not exists anymore. Deprecated checkpoint of vilmedic?".format(classname))
        model = model_cls(**model_config, dl=_DL(dataset), logger=None)
        model.load_state_dict(vilmedic_state_dict_versioning(state_dict["model"], state_dict.get("__version__")), strict=True)
        if torch.cuda.is_available():
            model = model.cuda()
        model.eval()
        assert hasattr(dataset, "inference"), "Dataset has not implemented an inference function"
        return model, dataset
