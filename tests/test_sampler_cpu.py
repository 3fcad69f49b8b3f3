"""RankPartition == torch.utils.data.DistributedSampler (what the reference's loader uses, ref lxmert_data.py:663-665)."""
import pytest
import torch
from torch.utils.data import DistributedSampler

from xlxmert_amd.sampler import RankPartition


@pytest.mark.parametrize("n,world,drop_last,shuffle", [(103, 8, False, True), (103, 8, True, True), (64, 4, False, True),
                                                       (5, 8, False, True), (1000, 2, False, False), (17, 3, True, False)])
def test_rank_partition_equals_distributed_sampler(n, world, drop_last, shuffle):
    ds = list(range(n))
    seen = []
    for epoch in (0, 1, 7):
        per_rank = []
        for r in range(world):
            ref = DistributedSampler(ds, num_replicas=world, rank=r, shuffle=shuffle, seed=9595, drop_last=drop_last)
            ref.set_epoch(epoch)
            mine = RankPartition(n, world, r, shuffle=shuffle, seed=9595, drop_last=drop_last)
            mine.set_epoch(epoch)
            assert list(mine) == list(ref) and len(mine) == len(ref)
            per_rank.append(list(mine))
        flat = [i for p in per_rank for i in p]
        assert len({len(p) for p in per_rank}) == 1                       # same number of steps on every rank
        if not drop_last:
            assert set(flat) == set(range(n))                             # every example is drawn
        if n % world == 0:
            assert sorted(flat) == list(range(n))                         # ... exactly once: disjoint shards
        seen.append(per_rank[0])
    if shuffle and n > 8:
        assert seen[0] != seen[1]                                         # set_epoch changes the permutation


def test_batches():
    p = RankPartition(100, 4, 1, shuffle=False)
    bs = list(p.batches(8))
    assert [len(b) for b in bs] == [8, 8, 8, 1] and bs[0][:3] == [1, 5, 9]
    assert [len(b) for b in p.batches(8, drop_last=True)] == [8, 8, 8]
