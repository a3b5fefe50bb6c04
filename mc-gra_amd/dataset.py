"""Dataset loading for main.py; same constructor and attributes as the reference's Dataset
(/root/reference/MC-GRA/dataset.py:44-70) for the graph files the README commands use.  Caller side, not hot path."""
import os.path as osp

import numpy as np
import scipy.sparse as sp

from .utils import get_train_val_test_gcn


class Dataset():
    """root/name.npz (cora, citeseer, polblogs, cora_ml; dataset.py:340-390) and the text formats of
    usair / brazil (dataset.py:230-300) / AIDS (:200-222)."""

    def __init__(self, root, name, setting='gcn', seed=None, require_mask=False):
        self.name = name.lower()
        self.setting = setting.lower()
        assert self.setting == 'gcn', "main.py uses setting='GCN' (main.py:148); nettack's LCC split is not provided"
        self.seed = seed
        self.root = osp.expanduser(osp.normpath(root))
        self.adj, self.features, self.labels = self.load_data()
        self.init_adj = sp.csr_matrix(np.zeros(self.adj.shape))            # init_matrix (dataset.py:433-437)
        self.idx_train, self.idx_val, self.idx_test = get_train_val_test_gcn(self.labels, seed=self.seed)

    def load_data(self):
        print('Loading {} dataset...'.format(self.name))
        if self.name in ('usair', 'brazil'):
            return self._load_edge_list(self.name)
        if self.name == 'aids':
            return self._load_aids()
        return self._load_npz(osp.join(self.root, self.name + '.npz'))

    def _load_npz(self, file_name):
        with np.load(file_name) as loader:
            adj = sp.csr_matrix((loader['adj_data'], loader['adj_indices'], loader['adj_indptr']), shape=loader['adj_shape'])
            if 'attr_data' in loader:
                features = sp.csr_matrix((loader['attr_data'], loader['attr_indices'], loader['attr_indptr']),
                                         shape=loader['attr_shape'])
            else:
                features = None
            labels = loader.get('labels')
        if features is None:
            features = sp.csr_matrix(np.eye(adj.shape[0]))
        adj = adj + adj.T                                                  # get_adj (dataset.py:340-361)
        adj = adj.tolil()
        adj[adj > 1] = 1
        adj.setdiag(0)
        adj = adj.astype("float32").tocsr()
        adj.eliminate_zeros()
        assert np.abs(adj - adj.T).sum() == 0, "Input graph is not symmetric"
        return adj, features, labels

    # lines of <name>_A.txt the reference reads (dataset.py:255, :295): usair's file holds 13 599, the last 17 are never seen
    # (europe: dataset.py:276; its data files are not part of the reference checkout).  Any other edge-list dataset: every line
    EDGE_LINES = {'usair': 13582, 'brazil': 1074, 'europe': 5995}

    def _load_edge_list(self, name):
        f = np.loadtxt(osp.join(self.root, name, f'{name}_lable.txt'))
        ids, labels = f[:, 0], f[:, 1]
        pos = {v: i for i, v in enumerate(ids)}
        n = len(ids)
        g = np.zeros((n, n))
        with open(osp.join(self.root, name, f'{name}_A.txt')) as fh:
            lines = fh.readlines()
            for ln in lines[:self.EDGE_LINES.get(name, len(lines))]:
                row = ln.strip().split()
                if len(row) < 2:
                    continue
                i, j = int(row[0]), int(row[1])
                g[pos[i], pos[j]] = 1
                g[pos[j], pos[i]] = 1
        return sp.csr_matrix(g), sp.csr_matrix(np.identity(n)), np.array(labels, dtype='int8')

    def _load_aids(self):
        n = 1429
        g = np.zeros((n, n))
        with open(osp.join(self.root, 'AIDS', 'AIDS_A.txt')) as fh:
            for _ in range(2948):
                i, j = [int(w) for w in fh.readline().strip().replace(',', ' ').split()]
                g[i - 1][j - 1] = 1
        feats = []
        with open(osp.join(self.root, 'AIDS', 'AIDS_node_attributes.txt')) as fh:
            for _ in range(n):
                feats.append([float(w) for w in fh.readline().strip().replace(',', ' ').split()])
        labels = np.array(np.loadtxt(osp.join(self.root, 'AIDS', 'AIDS_node_labels.txt')), dtype='int8')[:n]
        return sp.csr_matrix(g), sp.csr_matrix(feats), labels
