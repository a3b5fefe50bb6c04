"""GraphSAGE victim and its embedding view with the reference's class surface
(/root/reference/MC-GRA/models/graphsage.py:12-55, 57-106, 108-153).

A layer is ``cat([x, adj @ x], 1) @ weight`` with ``weight`` of shape [2*in, out] and no bias (:37-50); PGDAttack
splits the weight into its self half (rows that multiply x) and its neighbour half.  In the reference ``gc`` is a
plain list and only ``gc1 = gc[0]`` is a registered sub-module (:124-131), so fit trains the first layer and
``linear1``; deeper layers keep their initial weights.  Kept as is.  Training is torch autograd, once, before the
attack, outside the hot path.
"""
import math
from copy import deepcopy

import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.optim as optim
from torch.nn.parameter import Parameter

from .. import utils


class GraphConvolution(nn.Module):
    def __init__(self, in_features, out_features, with_bias=False):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = Parameter(torch.FloatTensor(in_features * 2, out_features))
        self.bias = Parameter(torch.FloatTensor(out_features)) if with_bias else None
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1. / math.sqrt(self.weight.size(1))
        self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.uniform_(-stdv, stdv)

    def forward(self, input, adj):
        out = torch.cat([input, adj @ input], dim=1) @ self.weight
        return out + self.bias if self.bias is not None else out


def _layers(nfeat, nhid, nlayer, with_bias):
    return [GraphConvolution(nfeat, nhid, with_bias)] + [GraphConvolution(nhid, nhid, with_bias) for _ in range(nlayer - 1)]


class embedding_graphsage(nn.Module):
    def __init__(self, nfeat, nhid, nlayer=2, with_bias=False, device=None):
        super().__init__()
        assert device is not None, "Please specify 'device'!"
        self.device, self.nfeat, self.nlayer, self.hidden_sizes, self.with_bias = device, nfeat, nlayer, [nhid], with_bias
        self.gc = _layers(nfeat, nhid, nlayer, with_bias)

    def forward(self, x, adj):
        for i in range(self.nlayer):
            x = F.relu(self.gc[i].to(self.device)(x, adj))
        return x

    def set_layers(self, nlayer):
        self.nlayer = nlayer


class graphsage(nn.Module):
    def __init__(self, nfeat, nhid, nclass, nlayer=2, dropout=0.5, lr=0.01, weight_decay=5e-4, with_relu=True,
                 with_bias=False, device=None):
        super().__init__()
        assert device is not None, "Please specify 'device'!"
        self.device, self.nfeat, self.hidden_sizes, self.nclass, self.nlayer = device, nfeat, [nhid], nclass, nlayer
        self.gc = _layers(nfeat, nhid, nlayer, with_bias)
        self.gc1 = self.gc[0]                      # the only registered layer (graphsage.py:126)
        self.gc2 = self.gc1                        # alias of it (graphsage.py:127): state_dict keys gc1.* and gc2.* as in the reference's checkpoints
        self.linear1 = nn.Linear(nhid, nclass, bias=with_bias)
        self.dropout, self.lr = dropout, lr
        self.weight_decay = weight_decay if with_relu else 0
        self.with_relu, self.with_bias = with_relu, with_bias

    def forward(self, x, adj):
        for i, layer in enumerate(self.gc):
            x = layer.to(self.device)(x, adj)
            if self.with_relu:
                x = F.relu(x)
            if i != len(self.gc) - 1:
                x = F.dropout(x, self.dropout, training=self.training)
        return F.log_softmax(self.linear1(x), dim=1)

    def fit(self, features, adj, labels, idx_train, idx_val=None, train_iters=200, initialize=True, verbose=False,
            normalize=True, patience=500, **kwargs):
        """graphsage.py:160-215 + the validation-selected weights of _train_with_val."""
        self.device = self.gc1.weight.device
        if initialize:
            for l in self.gc:
                l.reset_parameters()
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
                weights = deepcopy(self.state_dict())
        if weights is not None:
            self.load_state_dict(weights)
        self.eval()
