"""Dense "GAT" victim and its embedding view with the reference's class surface
(/root/reference/MC-GRA/models/gat.py:14-50, 157-258).

What the reference's layer computes: GraphAttentionLayer.forward builds the attention matrix (:36-43) and then
overwrites its result with ``h_prime = adj @ h`` (:44-45), so a layer is ``elu(adj @ (x @ W))`` and the heads of a
stage are concatenated (:172, :200).  The attention vector ``a`` never reaches the output; it is kept as an attribute
for surface fidelity and not evaluated here (the reference spends an N x N x 2F tensor per head on it).  ``attentions``
is a plain list of lists in the reference (:166-168), so neither ``.parameters()`` nor ``state_dict()`` see the head
weights: GAT.fit (:211-236) trains ``out_att`` only.  That behaviour is kept -- the attack consumes whatever weights the
victim ends up with.  Training is torch autograd, once, before the attack, outside the hot path; PGDAttack reads
``attentions[l][k].W``, ``out_att``, ``nclass / nfeat / hidden_sizes / nlayer``.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.optim as optim

from .. import utils


class GraphAttentionLayer(nn.Module):
    def __init__(self, in_features, out_features, dropout, alpha, concat=True, device="cpu"):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.dropout, self.alpha, self.concat = dropout, alpha, concat
        self.W = nn.Parameter(nn.init.xavier_uniform_(torch.zeros(in_features, out_features), gain=1.414).to(device))
        self.a = nn.Parameter(nn.init.xavier_uniform_(torch.zeros(2 * out_features, 1), gain=1.414).to(device))

    def forward(self, input, adj):
        h_prime = adj @ (input @ self.W)           # gat.py:45: the attention product of :44 is overwritten
        return F.elu(h_prime) if self.concat else h_prime


def _stages(nfeat, nhid, nheads, nlayer, dropout, alpha, device):
    mk = lambda fin: [GraphAttentionLayer(fin, nhid, dropout=dropout, alpha=alpha, concat=True, device=device)
                      for _ in range(nheads)]
    return [mk(nfeat)] + [mk(nhid * nheads) for _ in range(nlayer - 1)]


class embedding_gat(nn.Module):
    """gat.py:157-177; forward runs every stage whatever set_layers says (:170-174)."""

    def __init__(self, nfeat, nhid, nclass, dropout, alpha, nheads, device, nlayer=2):
        super().__init__()
        self.dropout, self.device, self.nfeat, self.hidden_sizes = dropout, device, nfeat, [nhid]
        self.nclass, self.nlayer = nclass, nlayer
        self.attentions = _stages(nfeat, nhid, nheads, nlayer, dropout, alpha, device)

    def forward(self, x, adj):
        for heads in self.attentions:
            x = F.dropout(x, self.dropout, training=self.training)
            x = torch.cat([att(x, adj) for att in heads], dim=1)
        return x

    def set_layers(self, nlayer):
        self.nlayer = nlayer


class GAT(nn.Module):
    """gat.py:180-258."""

    def __init__(self, nfeat, nhid, nclass, dropout, alpha, nheads, device, nlayer=2):
        super().__init__()
        self.dropout, self.device, self.nfeat, self.nlayer = dropout, device, nfeat, nlayer
        self.hidden_sizes, self.nclass = [nhid], nclass
        self.attentions = _stages(nfeat, nhid, nheads, nlayer, dropout, alpha, device)
        self.out_att = nn.Linear(nheads * nhid, nclass)

    def forward(self, x, adj):
        for heads in self.attentions:
            x = F.dropout(x, self.dropout, training=self.training)
            x = torch.cat([att(x, adj) for att in heads], dim=1)
        x = F.dropout(x, self.dropout, training=self.training)
        return F.log_softmax(F.elu(self.out_att(x)), dim=1)

    def fit(self, features, adj, labels, idx_train, idx_val=None, train_iters=100, verbose=False):
        opt = optim.Adam(self.parameters(), lr=0.005, weight_decay=5e-4)      # out_att only, as in the reference
        features, adj, labels = features.to(self.device), adj.to(self.device), labels.to(self.device)
        adj = utils.normalize_adj_tensor(adj)
        for epoch in range(train_iters):
            self.train()
            opt.zero_grad()
            out = self.forward(features, adj)
            loss = F.nll_loss(out[idx_train], labels[idx_train])
            loss.backward()
            opt.step()
            if verbose and idx_val is not None:
                self.eval()
                with torch.no_grad():
                    out = self.forward(features, adj)
                print('Epoch: {:04d} loss_train: {:.4f} acc_val: {:.4f}'.format(
                    epoch + 1, loss.item(), utils.accuracy(out[idx_val], labels[idx_val]).item()))
        self.eval()

    def set_layers(self, nlayer):
        self.nlayer = nlayer
