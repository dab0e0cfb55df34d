"""SHARP_unlimited sharded one-block-per-GPU (reference: R/SHARP_unlimited.R:125-183).

Blocks are independent until the final sMetaC, and sMetaC only ever uses the per-label column means
of E1 (R/sMetaC.R:58-63), so the single exchange step is one all-gather of the per-(block, cluster)
centroid means (<= a few hundred rows x p fp64) and cluster sizes.  Every rank then runs the tiny
centroid-level sMetaC redundantly (deterministic), so no broadcast of the relabel map is needed.
One process per GPU; torch.distributed with backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the
CPU tests (where the compute callbacks are injected)."""
import math

import numpy as np


def global_reduced_dim(ncells_total):
    # p depends on the GLOBAL cell count (R/SHARP_unlimited.R:65-66) and must be agreed before any block starts
    return int(math.ceil(math.log2(ncells_total) / 0.04))


def block_owner(block_index, world_size):
    return block_index % world_size


def _gather_tables(means_list, counts_list, p, group=None, device="cpu"):
    """All-gather ragged (G_b x p) tables in GLOBAL block order: returns (means, counts, block_of_row)."""
    import torch
    import torch.distributed as dist

    active = dist.is_initialized()          # a world of ONE still runs the collectives: the RCCL path is then exercised on a 1-GPU box
    world = dist.get_world_size(group) if active else 1
    local_rows = sum(m.shape[0] for m in means_list)
    nblocks_local = len(means_list)
    # header: number of local blocks and rows of each, padded to a fixed size negotiated by one all-reduce
    sizes = torch.tensor([nblocks_local, local_rows], dtype=torch.int64, device=device)
    mx = sizes.clone()
    if active:
        dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
    max_blocks, max_rows = int(mx[0]), int(mx[1])
    hdr = torch.zeros(1 + max_blocks, dtype=torch.int64, device=device)
    hdr[0] = nblocks_local
    for i, m in enumerate(means_list):
        hdr[1 + i] = m.shape[0]
    payload = torch.zeros((max(max_rows, 1), p + 1), dtype=torch.float64, device=device)
    if local_rows:
        mm = np.concatenate(means_list, 0)
        cc = np.concatenate(counts_list, 0).astype(np.float64)     # exact below 2^53
        payload[:local_rows, :p] = torch.from_numpy(mm).to(device)
        payload[:local_rows, p] = torch.from_numpy(cc).to(device)
    if active:
        hdrs = [torch.zeros_like(hdr) for _ in range(world)]
        pls = [torch.zeros_like(payload) for _ in range(world)]
        dist.all_gather(hdrs, hdr, group=group)
        dist.all_gather(pls, payload, group=group)
    else:
        hdrs, pls = [hdr], [payload]
    hdrs = [h.cpu().numpy() for h in hdrs]
    pls = [q.cpu().numpy() for q in pls]
    return hdrs, pls


def unlimited_sharded(local_blocks, local_block_ids, ncells_per_block, run_block, merge, group=None, device="cpu",
                      N_cluster=0, minN_cluster=0, maxN_cluster=0, run_blocks=None):
    """Run this rank's blocks and combine across ranks.

    local_blocks      : this rank's block objects (passed to run_block)
    local_block_ids   : their global block indices (block b is owned by rank b % world)
    ncells_per_block  : cells of EVERY global block (needed for p and the k-range rules)
    run_block(block, p[, next_block]) -> (pred (nb,), means (G, p), counts (G,))
    run_blocks(blocks, p) -> one such triple per block: all of the rank's blocks in ONE call (device.unlimited_blocks_dev)
    merge(means, counts, ncells_total, N_cluster, minN, maxN) -> (final_id (nC,), n_final)
    Returns {global block id: final labels of that block} for the local blocks, and n_final."""
    ncells_total = int(sum(ncells_per_block))
    p = global_reduced_dim(ncells_total)
    preds, means_list, counts_list = [], [], []
    import inspect

    takes_next = len(inspect.signature(run_block).parameters) >= 3      # run_block(block, p, next_block): lets the library prepare the
    if run_blocks is not None and len(local_blocks) > 1:                # all of the rank's blocks in one library call (one pipelined batch)
        for pr, mn, cn in run_blocks(local_blocks, p):
            preds.append(pr)
            means_list.append(np.asarray(mn, np.float64).reshape(-1, p))
            counts_list.append(np.asarray(cn, np.int64))
        local_blocks = []
    for i, blk in enumerate(local_blocks):                               # rank's next block under the current one's tail
        if takes_next:
            pr, mn, cn = run_block(blk, p, local_blocks[i + 1] if i + 1 < len(local_blocks) else None)
        else:
            pr, mn, cn = run_block(blk, p)
        preds.append(pr)
        means_list.append(np.asarray(mn, np.float64).reshape(-1, p))
        counts_list.append(np.asarray(cn, np.int64))
    hdrs, pls = _gather_tables(means_list, counts_list, p, group, device)
    world = len(hdrs)
    # rebuild the tables in global block order: rank r owns blocks r, r + world, ... in that local order
    nblocks = len(ncells_per_block)
    rows_of_block = {}
    for r in range(world):
        nb_r = int(hdrs[r][0])
        off = 0
        for j in range(nb_r):
            g = int(hdrs[r][1 + j])
            rows_of_block[r + j * world] = pls[r][off:off + g]
            off += g
    assert sorted(rows_of_block) == list(range(nblocks)), "every block must be owned by rank (block % world)"
    first = np.zeros(nblocks + 1, np.int64)
    tabs = []
    for b in range(nblocks):
        tabs.append(rows_of_block[b])
        first[b + 1] = first[b] + rows_of_block[b].shape[0]
    allrows = np.concatenate(tabs, 0)
    final_id, n_final = merge(allrows[:, :p].copy(), np.rint(allrows[:, p]).astype(np.int64), ncells_total, N_cluster,
                              minN_cluster, maxN_cluster)
    out = {}
    for blk_id, pr in zip(local_block_ids, preds):
        out[blk_id] = np.asarray(final_id)[first[blk_id] + pr - 1]
    return out, int(n_final), p
