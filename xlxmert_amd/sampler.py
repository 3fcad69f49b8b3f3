"""Rank partition of the training set, as the reference's loader draws it (ref x-lxmert/src/pretrain/lxmert_data.py:663-665:
`DistributedSampler(dataset)`; lxmert_pretrain.py:267-268: `sampler.set_epoch(epoch)`): every epoch one seeded permutation of
the dataset indices, padded by wrap-around to a multiple of the world size, rank r takes positions r, r+world, r+2*world, ...
Each of the N ranks therefore draws a disjoint minibatch per step and all ranks see the same number of
steps.  Restated here (same generator arithmetic as torch.utils.data.DistributedSampler: `randperm` under
`manual_seed(seed + epoch)`) so that a loader feeding PretrainStep needs nothing but this index function."""
import math

import torch


class RankPartition:
    def __init__(self, n_items, world_size=None, rank=None, shuffle=True, seed=0, drop_last=False):
        if world_size is None or rank is None:
            import torch.distributed as dist
            if not (dist.is_available() and dist.is_initialized()):
                raise RuntimeError("RankPartition needs world_size / rank or an initialised process group")
            world_size = dist.get_world_size() if world_size is None else world_size
            rank = dist.get_rank() if rank is None else rank
        if not 0 <= rank < world_size:
            raise ValueError(f"rank {rank} outside [0, {world_size})")
        self.n, self.world, self.rank = int(n_items), int(world_size), int(rank)
        self.shuffle, self.seed, self.drop_last, self.epoch = shuffle, seed, drop_last, 0
        if drop_last and self.n % self.world != 0:
            self.num_samples = math.ceil((self.n - self.world) / self.world)
        else:
            self.num_samples = math.ceil(self.n / self.world)
        self.total_size = self.num_samples * self.world

    def set_epoch(self, epoch):
        """ref lxmert_pretrain.py:267-268: a different permutation every epoch, identical on all ranks."""
        self.epoch = int(epoch)

    def indices(self):
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            idx = torch.randperm(self.n, generator=g).tolist()
        else:
            idx = list(range(self.n))
        if not self.drop_last:
            pad = self.total_size - len(idx)
            if pad <= len(idx):
                idx += idx[:pad]
            else:
                idx += (idx * math.ceil(pad / len(idx)))[:pad]
        else:
            idx = idx[:self.total_size]
        return idx[self.rank:self.total_size:self.world]

    def __iter__(self):
        return iter(self.indices())

    def __len__(self):
        return self.num_samples

    def batches(self, batch_size, drop_last=False):
        """this rank's minibatches of one epoch (lists of dataset indices), in order."""
        idx = self.indices()
        for i in range(0, len(idx), batch_size):
            b = idx[i:i + batch_size]
            if len(b) == batch_size or not drop_last:
                yield b
