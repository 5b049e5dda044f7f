from .visual_encoder import VisualEncoder  # noqa: F401
