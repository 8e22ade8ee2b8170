"""Identity feature extractor (reference online_gp/models/stems.py:4-17).  The
learned stems (LinearStem / MLP) are plain torch.nn and out of the hot-path
scope; any torch module exposing ``input_dim`` /
``output_dim`` can be passed instead."""
import torch


class Identity(torch.nn.Module):
    def __init__(self, input_dim):
        super().__init__()
        self.input_dim = input_dim
        self.output_dim = input_dim

    def forward(self, inputs):
        return inputs

    def parameters(self, **kwargs):
        return [torch.eye(self.input_dim)]

    def modules(self):
        return []
