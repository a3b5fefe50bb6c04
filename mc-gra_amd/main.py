"""main.py entry with the reference's command line (/root/reference/MC-GRA/main.py:78-139) and its
``--mode evaluate`` flow (:141-324, :392-409): load the graph, train the victim GCN, compute the priors
H_A / Y_A, run PGDAttack on the MI355X hot path, report the recovered-adjacency AUC.

    python -m mc_gra_amd.main --dataset cora --w1 0.01 --w6 10 --w7 10 --w9 10 --w10 1000 --lr -2 \
        --useH_A --useY_A --useY --measure MSELoss            (after mcgra_loader.load())
    python mc-gra_amd/main.py ...                             (stand-alone; bootstraps the loader itself)

--arch gcn | sage | gat select the victim family as main.py:175-231 does.  Not provided (each exits with a message
naming the reference line): --mode search/baseline/gaussian/gcn_attack.

Several GPUs of one node -- one process per GPU, RCCL over xGMI:
    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 mc-gra_amd/main.py --dataset ... --measure HSIC ...
Every rank loads the graph and trains the victim on the same seed (rank 0's weights are then broadcast), and
PGDAttack.attack (main.py:298-307) runs ONE attack row-block sharded over the ranks when the configuration allows it
(mc-gra_amd/topology_attack.py); rank 0 prints and logs the result.  MCGRA_SHARED_GPU=1 puts every rank on cuda:0 over
gloo with host-staged exchanges: the test mode of a 1-GPU box.
"""
import argparse
import os
import random
import sys
from copy import deepcopy

import numpy as np

if __package__ in (None, ""):          # run as a script: load the hyphenated directory as package mc_gra_amd
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import mcgra_loader
    mcgra_loader.load()
    from mc_gra_amd import utils
    from mc_gra_amd.dataset import Dataset
    from mc_gra_amd.models.gcn import GCN, embedding_GCN
    from mc_gra_amd.models.gat import GAT, embedding_gat
    from mc_gra_amd.models.graphsage import graphsage, embedding_graphsage
    from mc_gra_amd.topology_attack import PGDAttack
else:
    from . import utils
    from .dataset import Dataset
    from .models.gcn import GCN, embedding_GCN
    from .models.gat import GAT, embedding_gat
    from .models.graphsage import graphsage, embedding_graphsage
    from .topology_attack import PGDAttack

import torch
import torch.nn.functional as F
from sklearn.metrics import auc, roc_curve


def build_parser():
    """Same flags and defaults as main.py:78-137 (+ --dataset_root, --saved_data for file locations)."""
    p = argparse.ArgumentParser()
    p.add_argument('--seed', type=int, default=15)
    p.add_argument('--epochs', type=int, default=100)
    p.add_argument('--lr', type=float, default=0.01)
    p.add_argument('--weight_decay', type=float, default=5e-4)
    p.add_argument('--hidden', type=int, default=16)
    p.add_argument('--dropout', type=float, default=0.5)
    p.add_argument('--nlayers', type=int, default=2)
    p.add_argument('--arch', type=str, choices=["gcn", "gat", "sage"], default='gcn')
    p.add_argument('--dataset', type=str, default='cora',
                   choices=['cora', 'cora_ml', 'citeseer', 'polblogs', 'pubmed', 'AIDS', 'usair', 'brazil'])
    p.add_argument('--density', type=float, default=10000000.0)
    p.add_argument('--model', type=str, default='PGD', choices=['PGD', 'min-max'])
    p.add_argument('--nlabel', type=float, default=1.0)
    p.add_argument('--iter', type=int, default=1)
    p.add_argument('--max_eval', type=int, default=100)
    p.add_argument('--log_name', type=str, default="result.txt")
    p.add_argument("--mode", type=str, default="evaluate", choices=["evaluate", "prepare", "notrain_test"])
    p.add_argument("--measure", type=str, default="HSIC", choices=["HSIC", "MSELoss", "KL", "KDE", "CKA", "DP"])
    p.add_argument("--measure2", type=str, default="HSIC")
    p.add_argument("--nofeature", action='store_true')
    p.add_argument('--weight_aux', type=float, default=0)
    p.add_argument('--weight_sup', type=float, default=1)
    for i in range(1, 11):
        p.add_argument(f'--w{i}', type=float, default=0)
    p.add_argument('--eps', type=float, default=0)
    p.add_argument('--useH_A', action='store_true')
    p.add_argument('--useY_A', action='store_true')
    p.add_argument('--useY', action='store_true')
    p.add_argument('--ensemble', action='store_true')
    p.add_argument('--add_noise', action='store_true')
    p.add_argument('--defense', action='store_true')
    p.add_argument('--dataset_root', type=str, default='./dataset')
    p.add_argument('--saved_data', type=str, default='./saved_data')
    p.add_argument('--device', type=str, default='cuda:0')
    return p


def dot_product_decode(Z, dataset):
    """main.dot_product_decode (main.py:44-55): feature_adj."""
    if dataset in ('cora', 'citeseer', 'AIDS'):
        Z = torch.matmul(Z, Z.t())
        return torch.sigmoid(torch.relu(Z - torch.eye(Z.shape[0], device=Z.device)))
    Z = F.normalize(Z, p=2, dim=1)
    Z = torch.matmul(Z, Z.t())
    return torch.relu(Z - torch.eye(Z.shape[0], device=Z.device))


def metric_pool(ori_adj, inference_adj, idx):
    """main.metric_pool (main.py:66-75)."""
    real = ori_adj[idx, :][:, idx].reshape(-1).cpu()
    pred = inference_adj[idx, :][:, idx].reshape(-1).cpu()
    fpr, tpr, _ = roc_curve(real, pred)
    return auc(fpr, tpr)


def label_adjacency(labels):
    """main.prepare (main.py:440-450): label_adj[i][j] = (labels[i] == labels[j])."""
    lab = np.asarray(labels)
    return (lab[:, None] == lab[None, :]).astype(np.float32)


def victim_tensors(m):
    """Every parameter of a victim: the registered ones and those of the layers it keeps in plain lists (models/gcn.py:44,
    graphsage.py:50, gat.py:47 -- as the reference's classes do)."""
    mods = [m] + list(getattr(m, "gc", [])) + [a for heads in getattr(m, "attentions", []) for a in heads]
    seen, out = set(), []
    for mod in mods:
        for p_ in mod.parameters():
            if id(p_) not in seen:
                seen.add(id(p_))
                out.append(p_)
    return out


def init_distributed(args):
    """Under a launcher that set WORLD_SIZE > 1 (torchrun): join the process group BEFORE anything touches the GPU and take
    this rank's device.  Returns (rank, world); (0, 1) for a plain single-process run.  args._own_group says whether the group
    was created here (run() then destroys it on the way out; a caller's group is the caller's)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args._own_group = False
    if world < 2:
        return 0, 1
    import torch.distributed as dist
    if not dist.is_initialized():
        args._own_group = True
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("MCGRA_SHARED_GPU") == "1":
            dist.init_process_group("gloo")
            args.device = "cuda:0"
        else:
            local = int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))
            args.device = f"cuda:{local}"
            dist.init_process_group("nccl", device_id=torch.device(args.device))
    torch.cuda.set_device(torch.device(args.device))
    return dist.get_rank(), dist.get_world_size()


def run(args):
    """main.py:78-324 for one configuration.  Under a launcher every rank calls this; the process group this call created is
    destroyed on the way out, also when the run fails (a rank that dies with the group alive leaves its peers in a collective)."""
    rank, world = init_distributed(args)
    try:
        return _run(args, rank, world)
    finally:
        if world > 1 and getattr(args, "_own_group", False):
            import torch.distributed as dist
            if dist.is_initialized():
                dist.destroy_process_group()


def _run(args, rank, world):
    device = torch.device(args.device)
    np.random.seed(args.seed); random.seed(args.seed); torch.manual_seed(args.seed)       # main.py:142-146
    data = Dataset(root=args.dataset_root, name=args.dataset, setting='GCN')
    adj, features, labels, init_adj = data.adj, data.features, data.labels, data.init_adj
    idx_train, idx_val, idx_test = data.idx_train, data.idx_val, data.idx_test
    random.sample(range(adj.shape[0]), int(adj.shape[0] * args.nlabel))                    # main.py:155 (consumes the RNG)
    adj, features, labels = utils.preprocess(adj, features, labels, preprocess_adj=False, onehot_feature=False)
    if args.mode == "prepare":
        if rank == 0:
            os.makedirs(args.saved_data, exist_ok=True)
            np.save(os.path.join(args.saved_data, args.dataset + ".npy"), label_adjacency(labels))
        if world > 1:      # the file is there when any rank returns (the next command of a script reads it)
            import torch.distributed as dist
            dist.barrier()
        return None
    feature_adj = dot_product_decode(features, args.dataset)
    if args.nofeature:
        feature_adj = torch.eye(*feature_adj.size())
    init_adj = torch.FloatTensor(init_adj.todense())

    nfeat, nclass = features.shape[1], labels.max().item() + 1
    if args.arch == "gcn":                                                                 # main.py:175-190
        victim_model = GCN(nfeat=nfeat, nclass=nclass, nhid=16, nlayer=args.nlayers, dropout=0.5, weight_decay=5e-4,
                           device=device).to(device)
        victim_model.fit(features, adj, labels, idx_train, idx_val, verbose=False)
        embedding = embedding_GCN(nfeat=nfeat, nhid=16, nlayer=args.nlayers, device=device)
        embedding.gc = deepcopy(victim_model.gc)
    elif args.arch == "sage":                                                              # main.py:193-210
        victim_model = graphsage(nfeat=nfeat, nclass=nclass, nhid=16, nlayer=args.nlayers, dropout=0.5,
                                 weight_decay=5e-4, device=device).to(device)
        for l in victim_model.gc:
            l.to(device)
        victim_model.fit(features, adj, labels, idx_train, idx_val, verbose=False)
        embedding = embedding_graphsage(nfeat=nfeat, nhid=16, nlayer=args.nlayers, device=device)
        embedding.gc = deepcopy(victim_model.gc)
    else:                                                                                  # main.py:213-231
        victim_model = GAT(nfeat=nfeat, nclass=nclass, nhid=16, nlayer=args.nlayers, dropout=0.5, alpha=0.1, nheads=5,
                           device=device).to(device)
        victim_model.fit(features, adj, labels, idx_train, idx_val, train_iters=getattr(args, "gat_train_iters", 200))
        embedding = embedding_gat(nfeat=nfeat, nclass=nclass, nhid=16, nlayer=args.nlayers, dropout=0.5, alpha=0.1,
                                  nheads=5, device=device)
        embedding.attentions = victim_model.attentions
    if world > 1:
        # one victim for all ranks: rank 0's parameters (training on the same seed is not guaranteed to be bit-reproducible
        # across processes); the embedding shares / copies them as main.py:190 / :210 / :231 do
        import torch.distributed as dist
        staged = str(dist.get_backend()).lower() != "nccl"
        with torch.no_grad():
            for p_ in victim_tensors(victim_model):
                t_ = p_.data.cpu() if staged else p_.data
                dist.broadcast(t_, 0)
                if staged:
                    p_.data.copy_(t_)
        if args.arch != "gat":
            embedding.gc = deepcopy(victim_model.gc)
    # main.py:233-241, call for call: the victim is in eval mode (fit() leaves it there), the freshly built embedding is
    # NOT -- embedding_gat.forward therefore applies F.dropout(0.5) and the reference's H_A priors of --arch gat carry
    # dropout noise (embedding_GCN / embedding_graphsage have no dropout); each call consumes the RNG as the reference's does
    embedding = embedding.to(device)
    with torch.no_grad():
        fd, ad = features.to(device), adj.to(device)
        embedding(fd, ad)                                                                  # H_A     main.py:235
        Y_A = victim_model(fd, ad)                                                         # main.py:236
        embedding.set_layers(1)
        embedding(fd, ad)                                                                  # H_A1    main.py:238-239
        embedding.set_layers(2)
        H_A2 = embedding(fd, ad)                                                           # main.py:240-241
        out = victim_model(fd, utils.normalize_adj_tensor(ad))
        print("Test set results:", "accuracy= {:.4f}".format(utils.accuracy(out[idx_test], labels.to(device)[idx_test]).item()))
    idx_attack = np.array(random.sample(range(adj.shape[0]), int(adj.shape[0] * args.nlabel)))   # main.py:244
    num_edges = int(0.5 * args.density * adj.sum() / adj.shape[0] ** 2 * len(idx_attack) ** 2)

    lr = 10 ** args.lr                                                                     # objective(): main.py:282-283
    weight_param = tuple(getattr(args, f"w{i}") for i in range(1, 11))
    lab_path = os.path.join(args.saved_data, args.dataset + ".npy")
    label_adj = np.load(lab_path) if os.path.exists(lab_path) else label_adjacency(labels)
    model = PGDAttack(model=victim_model, embedding=embedding, H_A=H_A2, Y_A=Y_A, nnodes=adj.shape[0],
                      loss_type='CE', device=device)
    if getattr(args, "adj_changes_init", None) is not None:      # (tests: a seeded start; adj_changes is a public attribute, :77)
        model.adj_changes = args.adj_changes_init
    model.attack(args, None, lr, 0, args.weight_sup, weight_param, feature_adj, 0, 0, 0, idx_train, idx_val,
                 idx_test, adj, features, init_adj, labels, idx_attack, num_edges, 0, epochs=args.epochs,
                 label_adj=label_adj)
    inference_adj = model.modified_adj.cpu()
    res = {"auc_attack": float(metric_pool(adj, inference_adj, idx_attack)),
           "auc_train": float(metric_pool(adj, inference_adj, idx_train)),
           "auc_all": float(metric_pool(adj, inference_adj, np.arange(adj.shape[0]))),
           "density": float(inference_adj.mean())}
    res["path"] = dict(model.history.get("path", {}), world=world)
    if rank != 0:           # every rank holds the same modified_adj; rank 0 reports
        return res
    print(f"current auc={res['auc_all']}")
    os.makedirs("./results/", exist_ok=True)
    with open(os.path.join("./results", args.log_name), "a") as f:                         # main.py:314-323
        f.write(f"current parameter: {args}\n")
        f.write(f"In attack graph: AUC={res['auc_attack']}\tIn train graph: AUC={res['auc_train']}\t"
                f"In Whole Graph: AUC={res['auc_all']}\n")
        f.write(f"current density: {res['density']}\n")
    return res


if __name__ == '__main__':
    run(build_parser().parse_args())
