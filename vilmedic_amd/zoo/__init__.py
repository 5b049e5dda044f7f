from .modeling_auto import MODEL_ZOO, AutoModel  # noqa: F401
