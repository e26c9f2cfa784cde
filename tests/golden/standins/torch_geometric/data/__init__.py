"""Stand-in for torch_geometric.data.Data: attribute bag with .to(device)."""
import torch


class Data(object):
    def __init__(self, x=None, edge_index=None, **kwargs):
        self.x = x
        self.edge_index = edge_index
        for k, v in kwargs.items():
            setattr(self, k, v)

    def keys(self):
        return [k for k in self.__dict__ if not k.startswith('_')]

    def to(self, device):
        for k in list(self.__dict__):
            v = self.__dict__[k]
            if torch.is_tensor(v):
                self.__dict__[k] = v.to(device)
        return self
