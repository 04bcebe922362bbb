"""Child process of tests/test_distributed_nccl.py, started by torch.distributed.run (one process per GPU,
backend nccl = RCCL).  Nothing here touches the GPU before the process group is up.

    python -m torch.distributed.run --nproc-per-node N ... tests/nccl_worker.py OUT_DIR

Every rank builds its shard of the text collection through east.parallel.DistributedASTRelevanceMeasure and
saves the gathered K x D table; rank 0 also saves the table of a plain single-GPU ASTRelevanceMeasure over the
whole collection for the parent to compare bit for bit."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "ast-text-analysis_amd"), ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def inputs():
    from conftest import word_stream
    from east import utils
    rng = np.random.default_rng(4242)
    sizes = [30000, 500, 80000, 1200, 700, 40000, 25000, 64, 9000, 150000, 3000]
    texts = [word_stream(rng, s) for s in sizes]
    kps = [utils.prepare_text(word_stream(rng, int(rng.integers(4, 24))).decode()) for _ in range(60)]
    toks = texts[2].decode().split()
    kps += [utils.prepare_text(" ".join(toks[i:i + 2])) for i in range(0, 60, 3)]
    return texts, [k for k in kps if k.replace(" ", "")]


def main():
    out_dir = sys.argv[1]
    import torch
    import torch.distributed as dist
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    from east import parallel, relevance
    texts, kps = inputs()
    for n_texts in (len(texts), 1):                          # (1 text: every rank but one has an empty shard)
        for normalized in (True, False):
            m = parallel.DistributedASTRelevanceMeasure(normalized=normalized, device=local)
            m.set_text_collection(texts[:n_texts])
            table = m.relevance_table(kps)
            np.save(os.path.join(out_dir, "table_%d_%d_%d.npy" % (n_texts, int(normalized), rank)), table)
            if rank == 0:
                single = relevance.ASTRelevanceMeasure("easa", normalized, device=local)
                single.set_text_collection(texts[:n_texts])
                np.save(os.path.join(out_dir, "single_%d_%d.npy" % (n_texts, int(normalized))), single.relevance_table(kps))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
