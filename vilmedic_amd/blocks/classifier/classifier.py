"""ref: vilmedic/blocks/classifier/classifier.py:4-15 (Dropout + Linear head; a [B,768]x[768,330] GEMM -- a plain library GEMM)."""
from torch import Tensor
from torch.nn import Dropout, Linear, Module, Sequential


class Classifier(Module):
    def __init__(self, input_size, num_classes, dropout=0., **kwargs):
        super().__init__()
        self.classifier = Sequential(Linear(in_features=input_size, out_features=num_classes))
        self.dropout = Dropout(p=dropout)

    def forward(self, input: Tensor):
        return self.classifier(self.dropout(input))
