"""CPU tier: the N>1 path with world_size 2 on gloo.

The collective logic (document sharding, padded all-gather of the K x D_local
blocks, table assembly) is the product's east.parallel; the per-shard scorer is
replaced by an oracle-backed stand-in through the measure_factory hook, because
the HIP scorer needs a GPU.  On the GPU box the same class runs over RCCL."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, PKG, word_stream


class OracleMeasure(object):
    """Test stand-in with the ASTRelevanceMeasure surface, scoring with the CPU oracle."""

    def __init__(self, normalized=True):
        self.normalized = normalized

    def set_text_collection(self, texts, language=None):
        from east import utils
        from oracle import easa_oracle
        self.asts = [easa_oracle.OracleEASA(utils.text_to_strings_collection(t)) for t in texts]

    def relevance_table(self, prepared):
        return np.array([[a.score(q, normalized=self.normalized, fast=True) for a in self.asts] for q in prepared])


def _make_inputs():
    rng = np.random.default_rng(99)
    sizes = [3000, 500, 8000, 1200, 700, 4000, 2500]
    texts = [word_stream(rng, s) for s in sizes]
    from east import utils
    kps = [utils.prepare_text(word_stream(rng, int(rng.integers(4, 20))).decode()) for _ in range(25)]
    return texts, [k for k in kps if k.replace(" ", "")]


def _worker(rank, world, port, n_texts, out_dir):
    for p in (PKG, ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from east import parallel
    texts, kps = _make_inputs()
    texts = texts[:n_texts]
    for normalized in (True, False):
        m = parallel.DistributedASTRelevanceMeasure(normalized=normalized,
                                                    measure_factory=lambda: OracleMeasure(normalized))
        m.set_text_collection(texts)
        table = m.relevance_table(kps)
        np.save(os.path.join(out_dir, "table_%d_%d.npy" % (int(normalized), rank)), table)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_texts", [7, 1])
def test_world_size_2_table_equals_single_process(tmp_path, n_texts):
    port = 29500 + (os.getpid() % 2000) + n_texts
    mp.spawn(_worker, args=(2, port, n_texts, str(tmp_path)), nprocs=2, join=True)
    texts, kps = _make_inputs()
    texts = texts[:n_texts]
    for normalized in (True, False):
        single = OracleMeasure(normalized)
        single.set_text_collection(texts)
        want = single.relevance_table(kps)
        for rank in range(2):
            got = np.load(os.path.join(str(tmp_path), "table_%d_%d.npy" % (int(normalized), rank)))
            assert got.shape == want.shape == (len(kps), n_texts)
            assert np.array_equal(got, want)       # every rank holds the full table, bit-equal


class FailingMeasure(OracleMeasure):
    """Scores like OracleMeasure, but its relevance_table fails -- on the rank that holds documents only."""

    def relevance_table(self, prepared):
        raise RuntimeError("device lost")


def _error_worker(rank, world, port, out_dir):
    for p in (PKG, ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from east import parallel
    texts, kps = _make_inputs()
    seen = []
    # one document, two ranks: rank 1's shard is empty.  (a) an empty keyphrase is refused on BOTH ranks, before
    # anyone enters the collective; (b) a failure inside rank 0's scorer is raised on both as well.
    m = parallel.DistributedASTRelevanceMeasure(measure_factory=lambda: OracleMeasure(True))
    m.set_text_collection(texts[:1])
    try:
        m.relevance_table(kps[:3] + [" "])
    except ZeroDivisionError:
        seen.append("zero")
    m = parallel.DistributedASTRelevanceMeasure(measure_factory=lambda: FailingMeasure(True))
    m.set_text_collection(texts[:1])
    try:
        m.relevance_table(kps[:3])
    except RuntimeError as exc:
        seen.append("own" if "device lost" in str(exc) else "other")
    # ... and the group is still usable afterwards
    m = parallel.DistributedASTRelevanceMeasure(measure_factory=lambda: OracleMeasure(True))
    m.set_text_collection(texts[:1])
    seen.append(str(m.relevance_table(kps[:3]).shape))
    with open(os.path.join(out_dir, "seen_%d.txt" % rank), "w") as f:
        f.write(",".join(seen))
    dist.barrier()
    dist.destroy_process_group()


def test_errors_are_raised_on_every_rank(tmp_path):
    """A rank with an empty shard must not be left alone in the all-gather when the other one raises."""
    port = 29500 + (os.getpid() % 2000) + 17
    mp.spawn(_error_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert open(os.path.join(str(tmp_path), "seen_0.txt")).read() == "zero,own,(3, 1)"
    assert open(os.path.join(str(tmp_path), "seen_1.txt")).read() == "zero,other,(3, 1)"


def test_all_gather_table_single_rank_roundtrip():
    from east import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(31000 + os.getpid() % 1000)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        block = torch.arange(12, dtype=torch.float64).reshape(3, 4)
        assert torch.equal(parallel.all_gather_table(block, [4]), block)
    finally:
        dist.destroy_process_group()
