"""Worker for tests/test_dist_rccl_gpu.py: a world of ONE rank on backend "nccl" (= RCCL) runs sharp_amd.dist.unlimited_sharded with
the real device callbacks, so that the collectives of the N > 1 path (all-reduce, all-gather of cuda tensors, barrier) execute
through RCCL on the single GPU of the test box."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    port, out = sys.argv[1], sys.argv[2]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK="0", WORLD_SIZE="1")
    import torch
    import torch.distributed as dist

    import sharp_amd
    from sharp_amd import device as dev
    from sharp_amd import dist as sdist

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    sharp_amd.init(0)
    seed, m, nb, nblocks, K, rs = 20261003, 1500, 5200, 3, 3, 2103
    blocks = []
    for b in range(nblocks):
        dX = torch.empty((nb, m), dtype=torch.float32, device="cuda")
        dev.synth_fill(dX, seed, b * nb, 5, 250)
        blocks.append(dX)
    torch.cuda.synchronize()
    p = sdist.global_reduced_dim(nb * nblocks)
    proj = sharp_amd.Projector(m, p, [50 + rs + k for k in range(1, K + 1)])
    res, nfin, p2 = sdist.unlimited_sharded(blocks, list(range(nblocks)), [nb] * nblocks,
                                            lambda blk, p_: dev.unlimited_block_dev(blk, p_, proj.handle, K, rs),
                                            dev.unlimited_merge, device="cuda")
    t = torch.ones(1, device="cuda")
    dist.all_reduce(t)
    dist.barrier()
    np.savez(out, pred=np.concatenate([res[b] for b in range(nblocks)]), nfin=nfin, p=p2, allreduce=float(t.item()))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
