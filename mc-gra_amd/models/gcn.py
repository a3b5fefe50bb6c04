"""Victim GCN and its embedding view with the reference's class surface (/root/reference/MC-GRA/models/gcn.py).
Training (fit) is plain torch autograd: it happens once, before the attack, outside the hot path.  The attack
reads only .gc[l].weight/.bias, .linear1, .nclass/.nfeat/.hidden_sizes, .nlayer from these objects."""
import math
from copy import deepcopy

import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.optim as optim
from torch.nn.parameter import Parameter

from .. import utils


class GraphConvolution(nn.Module):
    """models/gcn.py:13-51: output = adj @ (input @ weight) + bias."""

    def __init__(self, in_features, out_features, with_bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = Parameter(torch.FloatTensor(in_features, out_features))
        self.bias = Parameter(torch.FloatTensor(out_features)) if with_bias else None
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1. / math.sqrt(self.weight.size(1))
        self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.uniform_(-stdv, stdv)

    def forward(self, input, adj):
        out = adj @ (input @ self.weight)
        return out + self.bias if self.bias is not None else out


class embedding_GCN(nn.Module):
    """models/gcn.py:54-84."""

    def __init__(self, nfeat, nhid, nlayer=2, with_bias=True, device=None):
        super().__init__()
        assert device is not None, "Please specify 'device'!"
        self.device, self.nfeat, self.nlayer, self.hidden_sizes = device, nfeat, nlayer, [nhid]
        self.gc = [GraphConvolution(nfeat, nhid, with_bias)] + [GraphConvolution(nhid, nhid, with_bias)
                                                               for _ in range(nlayer - 1)]

    def forward(self, x, adj):
        for i in range(self.nlayer):
            x = F.relu(self.gc[i].to(self.device)(x, adj))
        return x

    def set_layers(self, nlayer):
        self.nlayer = nlayer


class GCN(nn.Module):
    """models/gcn.py:87-411 (forward, fit with validation, predict)."""

    def __init__(self, nfeat, nhid, nclass, nlayer=2, dropout=0.5, lr=0.01, weight_decay=5e-4, with_relu=True,
                 with_bias=True, device=None):
        super().__init__()
        assert device is not None, "Please specify 'device'!"
        self.device, self.nfeat, self.hidden_sizes, self.nclass, self.nlayer = device, nfeat, [nhid], nclass, nlayer
        self.gc = [GraphConvolution(nfeat, nhid, with_bias)] + [GraphConvolution(nhid, nhid, with_bias)
                                                               for _ in range(nlayer - 1)]
        self.gc1, self.gc2 = self.gc[0], self.gc[1]
        self.linear1 = nn.Linear(nhid, nclass, bias=with_bias)
        self.dropout, self.lr = dropout, lr
        self.weight_decay = weight_decay if with_relu else 0
        self.with_relu = with_relu

    def forward(self, x, adj):
        for i, layer in enumerate(self.gc):
            x = layer.to(self.device)(x, adj)
            if self.with_relu:
                x = F.relu(x)
            if i != len(self.gc) - 1:
                x = F.dropout(x, self.dropout, training=self.training)
        return F.log_softmax(self.linear1(x), dim=1)

    def parameters(self, recurse=True):      # gc is a plain list in the reference too; include its layers explicitly
        ps = list(super().parameters(recurse))
        for l in self.gc:
            ps += [p for p in l.parameters() if all(p is not q for q in ps)]
        return ps

    def fit(self, features, adj, labels, idx_train, idx_val=None, train_iters=200, initialize=True, verbose=True,
            normalize=True, patience=500, **kwargs):
        """models/gcn.py:182-241 with validation-based model selection (:283-322)."""
        self.device = self.gc1.weight.device
        features, adj, labels = features.to(self.device), adj.to(self.device), labels.to(self.device)
        adj_norm = utils.normalize_adj_tensor(adj) if normalize else adj
        opt = optim.Adam(self.parameters(), lr=self.lr, weight_decay=self.weight_decay)
        best_loss, best_acc, weights = 100, 0, None
        for i in range(train_iters):
            self.train()
            opt.zero_grad()
            loss = F.nll_loss(self.forward(features, adj_norm)[idx_train], labels[idx_train])
            loss.backward()
            opt.step()
            if idx_val is None:
                continue
            self.eval()
            with torch.no_grad():
                out = self.forward(features, adj_norm)
                lv = F.nll_loss(out[idx_val], labels[idx_val]).item()
                av = utils.accuracy(out[idx_val], labels[idx_val]).item()
            if verbose and i % 10 == 0:
                print('Epoch {}, training loss: {}, val acc: {}'.format(i, loss.item(), av))
            if lv < best_loss or av > best_acc:
                best_loss, best_acc = min(best_loss, lv), max(best_acc, av)
                weights = (deepcopy(self.state_dict()), [deepcopy(l.state_dict()) for l in self.gc])
        if weights is not None:
            self.load_state_dict(weights[0])
            for l, sd in zip(self.gc, weights[1]):
                l.load_state_dict(sd)
        self.eval()
