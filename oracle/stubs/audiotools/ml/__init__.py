import torch.nn as nn


class BaseModel(nn.Module):
    @property
    def device(self):
        return next(self.parameters()).device
