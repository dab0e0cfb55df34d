"""Worker for tests/test_dist_gloo.py: one rank of a world_size-2 gloo run of sharp_amd.dist.unlimited_sharded
with the per-block compute and the centroid merge injected from the CPU oracle (no GPU involved)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def oracle_callbacks(orc, m, K, seed):
    def run_block(X, p):
        tern = np.concatenate([orc.ranM(m, p, 50 + seed + k) for k in range(1, K + 1)], 0)
        r = orc.SHARP(X, K=K, reduced_ndim=p, tern=tern, rN_seed=seed, nthreads=2)
        pred = r["pred_clusters"]
        G = int(pred.max())
        means = np.stack([r["viE"][pred == g + 1].mean(0) for g in range(G)])
        counts = np.bincount(pred, minlength=G + 1)[1:]
        return pred, means, counts

    def merge(means, counts, ncells, N_cluster, minN, maxN):
        # centroid-level sMetaC: one row per (block, cluster); k-range rules see the TRUE cell count only through
        # floor(ncells/1e4), which is the same for the row count at this test's size (both < 3e4 -> baseN = 2)
        assert ncells < 30000
        nC = means.shape[0]
        r = orc.sMetaC(np.arange(1, nC + 1), means, minN=minN or 2, maxN=maxN or max(40, -(-ncells // 5000)))
        fid = r["tf"].copy()
        cnt = np.bincount(fid, weights=counts, minlength=fid.max() + 1)
        if ncells > 10000:
            small = [q for q in range(1, fid.max() + 1) if 0 < cnt[q] < 10]
            if small:
                fid[np.isin(fid, small)] = min(small)
                cnt = np.bincount(fid, weights=counts, minlength=fid.max() + 1)
        ids = [q for q in range(1, len(cnt)) if cnt[q] > 0]
        ids.sort(key=lambda q: str(q))
        ids.sort(key=lambda q: -cnt[q])
        mp = {q: i + 1 for i, q in enumerate(ids)}
        return np.array([mp[q] for q in fid], np.int32), len(ids)

    return run_block, merge


def main():
    rank, world, port, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    nblocks_arg = int(sys.argv[5]) if len(sys.argv) > 5 else 4
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from oracle import pyoracle as orc
    from sharp_amd import dist as sdist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    seed, m, G, nm, nb, nblocks, K = 20261003, 1500, 5, 250, 900, nblocks_arg, 3
    ncb = [nb] * nblocks
    mine = [b for b in range(nblocks) if sdist.block_owner(b, world) == rank]
    blocks = [orc.synth_fill(seed, m, b * nb, nb, G, nm) for b in mine]
    run_block, merge = oracle_callbacks(orc, m, K, 2103)
    out, nfin, p = sdist.unlimited_sharded(blocks, mine, ncb, run_block, merge, device="cpu")
    # the same with all of the rank's blocks handed over in ONE call (run_blocks: what device.unlimited_blocks_dev is on a GPU)
    calls = []

    def run_blocks(bs, p_):
        calls.append(len(bs))
        return [run_block(b, p_) for b in bs]

    out2, nfin2, p2 = sdist.unlimited_sharded(blocks, mine, ncb, run_block, merge, device="cpu", run_blocks=run_blocks)
    assert calls == ([len(blocks)] if len(blocks) > 1 else []) and nfin2 == nfin and p2 == p and all(np.array_equal(out2[b], out[b]) for b in mine)
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), blocks=np.array(mine), nfin=nfin, p=p,
             **{f"pred{b}": out[b] for b in mine})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
