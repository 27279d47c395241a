"""Synthetic PCQM4Mv2-schema mini-batches (no dataset is reachable offline).

Schema = what the reference's `padded_collate` hands to the scheme
(/root/reference/lib/data/pcqm/data.py:105-122,
lib/data/pcqm/structural_transform.py:30-45, lib/data/dataset/stack_with_pad.py).
Value ranges follow SURVEY §8(d): node features U{0..99}+1+128*f, hop counts
U{1..11}, bond features on ~10 % of pairs (+1+8*f), coords N(0, 2^2),
target N(5.69, 1.16^2) as float64 (quirk Q9).  Everything comes from
`numpy.random.default_rng(seed)` so CPU oracle runs and GPU runs see
bit-identical inputs.
"""
import numpy as np
import torch

NODE_FEATURES_OFFSET, NUM_NODE_FEATURES = 128, 9
EDGE_FEATURES_OFFSET, NUM_EDGE_FEATURES = 8, 3
HL_MEAN, HL_STD = 5.6894608, 1.1621397


def batch_seed(step, rank=0, base=1234):
    """default_rng(1234 + rank*1000 + step)."""
    return base + rank * 1000 + step


def make_batch(batch_size, max_nodes, seed, num_nodes=None, ragged=False,
               min_nodes=None, bond_density=0.10, as_torch=True):
    """Returns the collated batch dict (numpy or CPU torch tensors).

    num_nodes: explicit per-graph node counts; else all = max_nodes, or with
    ragged=True  U{min_nodes..max_nodes} with graph 0 forced to max_nodes.
    """
    rng = np.random.default_rng(seed)
    B, N = batch_size, max_nodes
    if num_nodes is None:
        if ragged:
            lo = max(1, N // 2) if min_nodes is None else min_nodes
            num_nodes = rng.integers(lo, N + 1, size=B)
            num_nodes[0] = N
        else:
            num_nodes = np.full(B, N)
    num_nodes = np.asarray(num_nodes, dtype=np.int64)
    assert num_nodes.shape == (B,) and num_nodes.max() <= N

    node_mask = (np.arange(N)[None, :] < num_nodes[:, None])
    edge_valid = node_mask[:, :, None] & node_mask[:, None, :]

    nf = rng.integers(0, 100, size=(B, N, NUM_NODE_FEATURES))
    nf = nf + 1 + NODE_FEATURES_OFFSET * np.arange(NUM_NODE_FEATURES)
    nf = np.where(node_mask[:, :, None], nf, 0).astype(np.int16)

    hops = rng.integers(1, 12, size=(B, N, N))
    hops = np.triu(hops, 1)
    hops = hops + hops.transpose(0, 2, 1)                    # symmetric, zero diagonal
    hops = np.where(edge_valid, hops, 0).astype(np.int16)

    bonded = rng.random((B, N, N)) < bond_density
    bonded = np.triu(bonded, 1)
    bonded = bonded | bonded.transpose(0, 2, 1)
    bf = rng.integers(0, 7, size=(B, N, N, NUM_EDGE_FEATURES))
    bf = np.triu(bf.transpose(0, 3, 1, 2), 1)
    bf = (bf + bf.transpose(0, 1, 3, 2)).transpose(0, 2, 3, 1)
    bf = bf + 1 + EDGE_FEATURES_OFFSET * np.arange(NUM_EDGE_FEATURES)
    fm = np.where((bonded & edge_valid)[..., None], bf, 0).astype(np.int16)

    coords = (rng.standard_normal((B, N, 3)) * 2.0).astype(np.float32)
    coords = np.where(node_mask[:, :, None], coords, 0).astype(np.float32)
    target = HL_MEAN + HL_STD * rng.standard_normal(B)       # float64 (Q9)

    batch = dict(
        num_nodes=num_nodes,
        node_mask=node_mask.astype(np.uint8),
        node_features=nf,
        distance_matrix=hops,
        feature_matrix=fm,
        target=target,
        dft_coords=coords,
    )
    if as_torch:
        batch = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in batch.items()}
    return batch
