from .classifier import Classifier  # noqa: F401
