"""CPU suite: the N>1 path (row shards + all-gather + merge) with world_size 2 over gloo.  HIP kernels
cannot run here, so the two injection points of ShardedFlatIndex (local index, merge function) are
served by the oracle; what is exercised is the product's sharding arithmetic, id offsets, collective
call and result assembly."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _OracleLocal:
    """Stand-in for MI355XFlatIndex on a CPU rank (same attributes ShardedFlatIndex touches)."""

    def __init__(self, metric):
        self.metric, self.id_offset, self.rows, self._torch_device = metric, 0, None, "cpu"

    def add(self, rows, total_hint=None):
        rows = np.asarray(rows, np.float32)
        self.rows = rows if self.rows is None else np.concatenate([self.rows, rows])

    @property
    def ntotal(self):
        return 0 if self.rows is None else len(self.rows)

    def reconstruct_n(self):
        return self.rows

    def search_device(self, q, k):
        from oracle import knn as ok
        X = self.rows if self.rows is not None else np.zeros((0, q.shape[1]), np.float32)
        D, I = ok.knn(X, q.numpy(), k, metric=self.metric, id_offset=self.id_offset)
        return torch.from_numpy(D), torch.from_numpy(I)


def _oracle_merge(Ds, Is, metric):
    from oracle import knn as ok
    D, I = ok.topk_merge(Ds.numpy(), Is.numpy(), metric)
    return torch.from_numpy(D), torch.from_numpy(I)


def _worker(rank, world, port, metric, n, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from viquae_amd.sharded import ShardedFlatIndex, shard_bounds
        rng = np.random.default_rng(42)
        X = rng.integers(-4, 5, (n, 12)).astype(np.float32)
        Q = rng.integers(-4, 5, (9, 12)).astype(np.float32)
        idx = ShardedFlatIndex(string_factory="Flat", metric_type=metric, local_index=_OracleLocal(metric),
                               merge_fn=_oracle_merge)
        idx.add_vectors(X)
        lo, hi = shard_bounds(n, world, rank)
        assert idx.local.id_offset == lo and (idx.local.rows is None or len(idx.local.rows) == hi - lo)
        D, I = idx.search_batch(Q, 20)
        # chunked (software-pipelined) search: several collectives in flight one after the other
        D2, I2 = idx.search_device(torch.from_numpy(Q), 20, chunk=4)
        assert np.array_equal(D2.numpy(), D) and np.array_equal(I2.numpy(), I)
        # save (every rank writes its rows into ONE file) -> load_rows (every rank reads its range back)
        path = out + ".index"
        idx.save(path)
        again = ShardedFlatIndex(string_factory="Flat", metric_type=metric, local_index=_OracleLocal(metric),
                                 merge_fn=_oracle_merge).load_rows(path)
        assert again.ntotal == n and again.local.id_offset == lo
        D3, I3 = again.search_batch(Q, 20)
        assert np.array_equal(D3, D) and np.array_equal(I3, I)
        # "*.index" is written in FAISS's own file format (IndexFlat), any other name in this build's: both load
        with open(path, "rb") as f:
            assert f.read(4) == (b"IxFI" if metric == 0 else b"IxF2")
        idx.save(out + ".rows")
        with open(out + ".rows", "rb") as f:
            assert f.read(8) == b"MQFLAT01"
        D4, I4 = ShardedFlatIndex(string_factory="Flat", metric_type=metric, local_index=_OracleLocal(metric),
                                  merge_fn=_oracle_merge).load_rows(out + ".rows").search_batch(Q, 20)
        assert np.array_equal(D4, D) and np.array_equal(I4, I)
        if rank == 0:
            np.savez(out, D=D, I=I, X=X, Q=Q)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("metric,n,world", [(0, 1000, 2), (1, 1000, 2), (0, 70, 2), (1, 40, 2), (0, 5, 2),   # 40, 5: rank 1 holds NO rows
                                            (0, 333, 3), (1, 130, 4)])                                    # odd world; 4 ranks, the last one empty
def test_world2_sharded_equals_unsharded(tmp_path, metric, n, world):
    from oracle import knn as ok
    out = str(tmp_path / "res.npz")
    mp.spawn(_worker, args=(world, _free_port(), metric, n, out), nprocs=world, join=True)
    z = np.load(out)
    D, I = ok.knn(z["X"], z["Q"], 20, metric=metric)
    assert np.array_equal(z["I"], I) and np.array_equal(z["D"], D)


# ---------------------------------------------------------------------------------------------------
# multi-process embedding: every rank embeds its contiguous block of rows, rank 0 stitches the blocks
# ---------------------------------------------------------------------------------------------------
class _FakeTokenizer:
    sep_token = "[SEP]"

    def __call__(self, texts, **kw):
        L = max(len(t) for t in texts)
        ids = torch.zeros((len(texts), L), dtype=torch.int64)
        for i, t in enumerate(texts):
            ids[i, :len(t)] = torch.tensor([ord(c) for c in t])
        return {"input_ids": ids, "attention_mask": (ids != 0).to(torch.int64)}


class _FakeEncoder:
    """A callable with the encoder's surface: the 'embedding' is a function of the text only."""

    def __call__(self, input_ids=None, attention_mask=None, **kw):
        x = input_ids.to(torch.float32)
        return {"pooler_output": torch.stack([x.sum(1), (x * x).sum(1), attention_mask.sum(1).float()], dim=1)}


def _embed_worker(rank, world, port, src, dst):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from viquae_amd.ir.embedding import dataset_embed
        out = dataset_embed(src, output_path=dst, model=_FakeEncoder(), tokenizer=_FakeTokenizer(), key="passage",
                            save_as="emb", output_key="pooler_output", map_kwargs={"batch_size": 4})
        assert len(out) == 23 and "emb" in out.column_names   # every rank gets the stitched dataset back
    finally:
        dist.destroy_process_group()


def test_world2_dataset_embed_stitches_rank_blocks_in_order(tmp_path):
    import datasets
    from viquae_amd.ir.embedding import dataset_embed
    datasets.disable_progress_bars()
    texts = ["passage number %d %s" % (i, "x" * (i % 7)) for i in range(23)]
    src = str(tmp_path / "kb")
    datasets.Dataset.from_dict({"passage": texts, "id": list(range(23))}).save_to_disk(src)
    single = dataset_embed(src, output_path=str(tmp_path / "single"), model=_FakeEncoder(), tokenizer=_FakeTokenizer(),
                           key="passage", save_as="emb", output_key="pooler_output", map_kwargs={"batch_size": 4})
    dst = str(tmp_path / "multi")
    mp.spawn(_embed_worker, args=(2, _free_port(), src, dst), nprocs=2, join=True)
    multi = datasets.load_from_disk(dst)
    assert multi["id"] == single["id"] == list(range(23))
    assert np.array_equal(np.asarray(multi["emb"], np.float32), np.asarray(single["emb"], np.float32))
    assert not [p for p in os.listdir(tmp_path) if "mq_rank" in p]   # the per-rank blocks are cleaned up
