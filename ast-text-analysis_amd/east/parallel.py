# -*- coding: utf-8 -*-
"""Multi-GPU sharding of the hot path: one process per GPU, one AST shard per GPU.

Every document is an independent annotated suffix tree (reference
east/relevance.py:41-46) and every (keyphrase, document) score is independent
(east/applications.py:43-52), so the corpus shards at document granularity with
no collective on the build path.  The only exchange step is assembling the
K x D score table: one all-gather of each rank's K x D_local block
(`torch.distributed`, backend "nccl" = RCCL over xGMI on the GPU box, "gloo"
in the CPU tests).  A single document's suffix sort does not shard.
"""
import os

import numpy as np

from east import consts
from east import hip_backend
from east import relevance
from east import utils


class LazyText(object):
    """A text of the collection that is still a file: its size is known (os.stat), its bytes are read by the rank whose
    shard holds it -- and by nobody else (`east keyphrases table` over a directory of a few thousand files would otherwise
    read the whole corpus on every rank to index an eighth of it)."""

    def __init__(self, path, reader=None):
        self.path = path
        self.size = os.stat(path).st_size
        self._reader = reader

    def __len__(self):
        return self.size

    def load(self):
        if self._reader is not None:
            return self._reader(self.path)
        with open(self.path, "rb") as f:
            return f.read()


def shard_documents(sizes, world_size):
    """Contiguous blocks of documents balanced by size (symbols or bytes).

    Returns world_size (start, end) pairs covering range(len(sizes)); blocks may
    be empty when there are fewer documents than ranks."""
    sizes = np.asarray(sizes, dtype=np.float64)
    n = len(sizes)
    csum = np.cumsum(sizes)
    total = float(csum[-1]) if n else 0.0
    bounds = [0]
    for r in range(1, world_size):
        # first document count whose cumulative size reaches r/world of the total
        cut = int(np.searchsorted(csum, total * r / world_size, side="left")) + 1 if n else 0
        bounds.append(min(max(cut, bounds[-1]), n))
    bounds.append(n)
    return [(bounds[r], bounds[r + 1]) for r in range(world_size)]


def all_gather_table(local_block, counts, group=None, assemble=True):
    """Assemble the K x D table from per-rank K x D_local blocks.

    local_block: torch tensor (K, D_local) float64 on the rank's device (cuda for
    RCCL, cpu for gloo); counts[r] = D_local of rank r.  Blocks are padded to the
    largest D_local so that a single all_gather_into_tensor moves everything.
    assemble=False: the rank takes part in the collective and returns None (it does not need the table)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    K = local_block.shape[0]
    width = max(max(counts), 1)
    padded = torch.zeros((K, width), dtype=local_block.dtype, device=local_block.device)
    padded[:, :local_block.shape[1]] = local_block
    gathered = torch.empty((world * K, width), dtype=local_block.dtype, device=local_block.device)
    dist.all_gather_into_tensor(gathered, padded.contiguous(), group=group)     # rank-major concatenation
    if not assemble:
        return None
    gathered = gathered.view(world, K, width)
    return torch.cat([gathered[r, :, :counts[r]] for r in range(world)], dim=1)


class DistributedASTRelevanceMeasure(relevance.ASTRelevanceMeasure):
    """ASTRelevanceMeasure whose text collection is sharded over the ranks of a
    torch.distributed process group.  Every rank calls the same methods with the
    same arguments; relevance_table returns the full K x D table on every rank.

    With the nccl (= RCCL) backend the rank's K x D_local block never leaves the device before the
    collective: the score kernels write it into a torch tensor (east_hip_score_resident's d_out),
    which goes straight into all_gather_into_tensor; the assembled table is copied to the host once."""

    def __init__(self, ast_algorithm=consts.ASTAlgorithm.EASA, normalized=True, device=None, group=None,
                 measure_factory=None, table_rank=None):
        """table_rank: None -- relevance_table returns the K x D table on every rank; r -- only rank r assembles it and
        copies it to the host (the CLI: rank 0 prints), the others take part in the collective and get a K x 0 array."""
        super(DistributedASTRelevanceMeasure, self).__init__(ast_algorithm, normalized, device)
        self.group = group
        self.table_rank = table_rank
        # ONE device for the index and for the collective: the explicit one, else EAST_HIP_DEVICE / LOCAL_RANK
        self.gpu = hip_backend.default_device() if device is None else int(device)
        # the local shard's measure; the factory hook exists so that the collective
        # logic can be exercised on CPU (gloo) with a stand-in scorer
        self._factory = measure_factory or (lambda: relevance.ASTRelevanceMeasure(ast_algorithm, normalized, self.gpu))

    def _on_gpu(self):
        import torch.distributed as dist
        return dist.get_backend(self.group) == "nccl"

    def set_text_collection(self, texts, language=consts.Language.ENGLISH):
        import torch.distributed as dist
        self.texts = texts
        self.language = language
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        if self._on_gpu():
            import torch
            torch.cuda.set_device(self.gpu)          # also for a rank whose shard is empty and never calls the library
        # The ranks must cut the collection at the same documents, and the sizes of texts that are still files come from
        # every rank's own os.stat (LazyText): a file that changes between two ranks' looks, or an attribute cache that
        # shows another size, would shift one rank's bounds -- columns assembled with the wrong counts, documents scored
        # twice or never, no error.  Rank 0's sizes are everybody's.
        sizes = [len(t) for t in texts]
        if world > 1:
            box = [sizes if rank == 0 else None]
            dist.broadcast_object_list(box, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
            if len(box[0]) != len(sizes):
                raise ValueError("the ranks were given collections of different lengths (%d here, %d on rank 0)" % (len(sizes), len(box[0])))
            sizes = box[0]
        self.shards = shard_documents(sizes, world)
        self.counts = [e - b for b, e in self.shards]
        b, e = self.shards[rank]
        self.local = self._factory()
        self.local.normalized = self.normalized
        if e > b:
            # (texts that are still files are read here, by the rank that indexes them)
            self.local.set_text_collection([t.load() if isinstance(t, LazyText) else t for t in texts[b:e]], language)

    def _check_queries(self, prepared_keyphrases, synonimizer):
        """What the reference raises on a query (easa.py:134 on an empty keyphrase or synonym variant, KeyError out of
        the synonym dictionary), raised on EVERY rank before anyone looks at its shard: a rank with an empty shard
        would otherwise walk into the collective alone and wait there for the others."""
        if synonimizer:
            for kp in prepared_keyphrases:                       # per word: no product over the alternatives here
                if relevance.has_empty_variant(relevance.synonym_alternatives(kp, synonimizer)):
                    raise ZeroDivisionError("float division by zero")
        elif not all(kp.replace(" ", "") for kp in prepared_keyphrases):
            raise ZeroDivisionError("float division by zero")

    def _local_block(self, prepared_keyphrases, synonimizer):
        """K x D_local block of this rank as a torch tensor on the collective's device."""
        import torch
        import torch.distributed as dist
        rank = dist.get_rank(self.group)
        K, D = len(prepared_keyphrases), self.counts[rank]
        self.local.normalized = self.normalized
        if self._on_gpu():
            dev = torch.device("cuda", self.gpu)
            if D and not synonimizer and getattr(self.local, "index", None) is not None:
                # (empty, not zeros: a fill queued on torch's stream is not ordered with the library's own stream,
                # which writes every entry of the block)
                block = torch.empty((K, D), dtype=torch.float64, device=dev)
                qs, qo = hip_backend.pack_queries([kp.replace(" ", "") for kp in prepared_keyphrases])
                self.local.index.set_keyphrases(qs, qo)
                self.local.index.score_resident(self.normalized, block.data_ptr())    # device to device, synchronised
                return block
            block = torch.zeros((K, D), dtype=torch.float64, device=dev)
            if D:
                args = (prepared_keyphrases, synonimizer) if synonimizer else (prepared_keyphrases,)
                block.copy_(torch.from_numpy(np.ascontiguousarray(self.local.relevance_table(*args))))
            return block
        if D:
            args = (prepared_keyphrases, synonimizer) if synonimizer else (prepared_keyphrases,)
            return torch.from_numpy(np.asarray(self.local.relevance_table(*args), dtype=np.float64))
        return torch.zeros((K, 0), dtype=torch.float64)

    def relevance_table(self, prepared_keyphrases, synonimizer=None):
        import torch
        import torch.distributed as dist
        self._check_queries(prepared_keyphrases, synonimizer)       # the same verdict on every rank
        # whatever else goes wrong on one rank (device error, out of memory) must not leave the others in the
        # collective: the ranks agree on an error flag first and raise together
        block, error = None, None
        try:
            block = self._local_block(prepared_keyphrases, synonimizer)
        except Exception as exc:                                    # noqa: BLE001 (re-raised below, on every rank)
            error = exc
        flag_dev = torch.device("cuda", self.gpu) if self._on_gpu() else torch.device("cpu")
        flag = torch.tensor([1 if error is not None else 0], dtype=torch.int32, device=flag_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
        if error is not None:
            raise error
        if int(flag.item()):
            raise RuntimeError("relevance_table failed on another rank of the process group")
        wanted = self.table_rank is None or dist.get_rank(self.group) == self.table_rank
        table = all_gather_table(block, self.counts, self.group, assemble=wanted)
        if not wanted:
            return np.zeros((len(prepared_keyphrases), 0), dtype=np.float64)
        return table.cpu().numpy()                       # the one copy of the assembled table to the host

    def relevance(self, keyphrase, text, synonimizer=None):
        return float(self.relevance_table([keyphrase], synonimizer)[0, text])
