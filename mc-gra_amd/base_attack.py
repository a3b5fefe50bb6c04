"""BaseAttack: same surface as the reference's base class
(/root/reference/MC-GRA/base_attack.py:11-119) so that PGDAttack drops into
main.py unchanged.  Holds no compute."""
import os.path as osp

import numpy as np
import scipy.sparse as sp
import torch
from torch.nn.modules.module import Module


class BaseAttack(Module):
    """Abstract base class for target attack classes (base_attack.py:11-27).

    Parameters: model (victim), nnodes, attack_structure, attack_features, device.
    """

    def __init__(self, model, nnodes, attack_structure=True, attack_features=False, device='cpu'):
        super(BaseAttack, self).__init__()
        self.surrogate = model
        self.nnodes = nnodes
        self.attack_structure = attack_structure
        self.attack_features = attack_features
        self.device = device
        self.modified_adj = None
        self.modified_features = None
        if model is not None:                       # base_attack.py:39-42
            self.nclass = model.nclass
            self.nfeat = model.nfeat
            self.hidden_sizes = model.hidden_sizes

    def attack(self, ori_adj, n_perturbations, **kwargs):
        pass

    def check_adj(self, adj):
        """base_attack.py:61-66."""
        assert np.abs(adj - adj.T).sum() == 0, "Input graph is not symmetric"
        assert adj.tocsr().max() == 1, "Max value should be 1!"
        assert adj.tocsr().min() == 0, "Min value should be 0!"

    @staticmethod
    def _to_scipy(tensor):
        t = tensor.detach().cpu()
        if t.is_sparse:
            t = t.coalesce()
            idx, val = t.indices().numpy(), t.values().numpy()
            return sp.csr_matrix((val, (idx[0], idx[1])), shape=tuple(t.shape))
        idx = t.nonzero().t().numpy()
        val = t[idx[0], idx[1]].numpy()
        return sp.csr_matrix((val, (idx[0], idx[1])), shape=tuple(t.shape))

    def save_adj(self, root=r'/tmp/', name='mod_adj'):
        """base_attack.py:68-92."""
        assert self.modified_adj is not None, 'modified_adj is None! Please perturb the graph first.'
        name = name + '.npz'
        m = self.modified_adj
        sp.save_npz(osp.join(root, name), self._to_scipy(m) if type(m) is torch.Tensor else m)

    def save_features(self, root=r'/tmp/', name='mod_features'):
        """base_attack.py:94-119."""
        assert self.modified_features is not None, 'modified_features is None! Please perturb the graph first.'
        name = name + '.npz'
        m = self.modified_features
        sp.save_npz(osp.join(root, name), self._to_scipy(m) if type(m) is torch.Tensor else m)
