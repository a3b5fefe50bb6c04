"""Host-side helpers main.py needs around the hot path; same names and argument meaning as the reference's
utils.py (/root/reference/MC-GRA/utils.py).  None of this is on the per-step path."""
import numpy as np
import scipy.sparse as sp
import torch

# utils.py:1100-1111
Align_Parameter_Cora = {"c1": 100, "c2": 1000, "c3": 100, "c4": 10, "c5": 10, "c6": 10, "c7": 10, "c8": 0.01,
                        "c9": 1, "c10": 1}


def preprocess(adj, features, labels, preprocess_adj=False, preprocess_feature=False, onehot_feature=False,
               sparse=False, device='cpu'):
    """utils.preprocess (utils.py:50-89), dense branch: scipy matrices -> torch tensors."""
    if preprocess_adj or preprocess_feature or sparse:
        raise NotImplementedError("only the dense, un-normalised branch main.py:162 uses is provided")
    labels = torch.LongTensor(np.asarray(labels))
    if onehot_feature:
        features = torch.eye(features.shape[0])
    else:
        features = torch.FloatTensor(np.array(features.todense() if sp.issparse(features) else features))
    adj = torch.FloatTensor(np.array(adj.todense() if sp.issparse(adj) else adj))
    return adj.to(device), features.to(device), labels.to(device)


def to_tensor(adj, features, labels=None, device='cpu'):
    """utils.to_tensor (utils.py:92-120), dense inputs."""
    adj = torch.FloatTensor(np.array(adj.todense() if sp.issparse(adj) else adj))
    features = torch.FloatTensor(np.array(features.todense() if sp.issparse(features) else features))
    if labels is None:
        return adj.to(device), features.to(device)
    return adj.to(device), features.to(device), torch.LongTensor(np.asarray(labels)).to(device)


def normalize_adj_tensor(adj, sparse=False):
    """utils.normalize_adj_tensor (utils.py:211-230): the HIP op on a CUDA tensor; torch ops for the
    victim-training forward on autograd tensors (outside the hot path)."""
    if sparse:
        raise NotImplementedError("sparse branch (utils.py:214-220) is not provided")
    if adj.is_cuda and not adj.requires_grad:
        from . import engine as E
        return E.normalize_adj_tensor(adj.contiguous())
    mx = adj + torch.eye(adj.shape[0], device=adj.device)
    r = mx.sum(1).pow(-0.5)
    r[torch.isinf(r)] = 0.
    return r[:, None] * mx * r[None, :]


def accuracy(output, labels):
    """utils.accuracy (utils.py:286-308)."""
    if not isinstance(labels, torch.Tensor):
        labels = torch.LongTensor(labels)
    preds = output.max(1)[1].type_as(labels)
    return preds.eq(labels).double().sum() / len(labels)


def get_train_val_test_gcn(labels, seed=None):
    """utils.get_train_val_test_gcn (utils.py:480-519): 20 per class train, rest split val/test."""
    if seed is not None:
        np.random.seed(seed)
    labels = np.asarray(labels)
    idx = np.arange(len(labels))
    idx_train, idx_unlabeled = np.array([], dtype=int), np.array([], dtype=int)
    for i in range(labels.max() + 1):
        li = np.random.permutation(idx[labels == i])
        idx_train = np.hstack((idx_train, li[:20])).astype(int)
        idx_unlabeled = np.hstack((idx_unlabeled, li[20:])).astype(int)
    idx_unlabeled = np.random.permutation(idx_unlabeled)
    return idx_train, idx_unlabeled[:len(idx_unlabeled) // 2], idx_unlabeled[len(idx_unlabeled) // 2:]


class MutualInformation(torch.nn.Module):
    """utils.MutualInformation (utils.py:980-1049) as topology_attack.py:199-201 / :244-246 / :261-263 construct it:
    sigma = 0.4, normalize = True, num_bins = the operands' width.  forward(input1, input2) on 2-D CUDA tensors returns a
    tensor of shape [1] like the reference's (callers index it with [0]); the bins live where the operands are (the
    reference asks for device='cuda:0', utils.py:991).  Evaluated by mcgra_mutual_information (csrc/kde_kernels.hip)."""

    def __init__(self, sigma=0.4, num_bins=256, normalize=True):
        super().__init__()
        if sigma != 0.4 or not normalize:
            raise NotImplementedError("MutualInformation(sigma=0.4, normalize=True) is the form topology_attack.py uses")
        self.sigma, self.num_bins, self.normalize, self.epsilon = 2 * sigma ** 2, num_bins, normalize, 1e-10

    def forward(self, input1, input2):
        from . import engine as E
        if input1.dim() != 2 or input1.shape != input2.shape or input1.shape[1] != self.num_bins:
            raise ValueError("2-D operands of width num_bins (utils.py:995 broadcasts the bins over the last axis)")
        return E.mutual_information(input1.float(), input2.float()).reshape(1)

    getMutualInformation = forward
