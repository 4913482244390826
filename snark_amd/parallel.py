"""One Groth16 proof with its MSM term ranges sharded over the GPUs of a node (SURVEY.md 8e,
BASELINE.json configs[2]) -- one process per GPU, `torch.distributed` for the exchange.

The path partitions cleanly: sum_i k_i P_i splits over disjoint term ranges and the partial sums combine by
group addition.  Each rank keeps terms [T*g/G, T*(g+1)/G) of the five (extended) query vectors resident
(window tables included), computes the witness map h redundantly from z (7 NTTs of <= 256 MiB cost ~3 ms; an
all-to-all transpose over point-to-point xGMI would cost more), runs its five partial MSMs and contributes
ONE message of `ark355_partial_size()` bytes (960 B for BLS12-381).  RCCL has no elliptic-curve reduction
operator, so the "all-reduce of partial sums" is an all-gather of those messages followed by the local EC
additions inside `ark355_prove_combine`; with <1 KiB per rank the collective is latency-bound, far below the
~153 GB/s per xGMI link.  Every rank ends up with the same, byte-identical proof.

Independent proofs (BASELINE.json configs[4], and bench.py's default multi-GPU mode) need no collective at all:
one `Groth16` instance per rank.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from .groth16 import Groth16, Proof, ProvingKey, R1CS


class ShardedGroth16:
    """`prove` across the ranks of a torch.distributed process group (backend nccl == RCCL on GPUs)."""

    def __init__(self, groth: Groth16, group=None, device=None):
        import torch.distributed as dist
        self.g = groth
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.device = device          # torch device of the exchange buffers ("cuda:<local>" for RCCL, "cpu" for gloo)
        self._pk_handles = {}

    def load_pk_shard(self, pk: ProvingKey):
        key = id(pk)
        if key not in self._pk_handles:
            g = self.g
            self._pk_handles[key] = g.lib.pk_load(
                g.ctx, g.curve.curve_id, pk.ell, pk.w, pk.N, pk.a_query, pk.b_g1_query, pk.b_g2_query, pk.h_query,
                pk.l_query, pk.vk.alpha_g1, pk.beta_g1, pk.delta_g1, pk.vk.beta_g2, pk.vk.delta_g2,
                shard=(self.rank, self.world))
        return self._pk_handles[key]

    def prove(self, pk: ProvingKey, r1cs: R1CS, z: bytes, r: int, s: int) -> Proof:
        """All ranks call this with the same z, r, s; all return the same proof."""
        import torch
        g, cv = self.g, self.g.curve
        pkh, rh = self.load_pk_shard(pk), g.load_r1cs(r1cs)
        rb, sb = cv.fr_canon(r), cv.fr_canon(s)
        part = g.lib.prove_shard(g.ctx, cv.curve_id, pkh, rh, z, len(z) // 32, rb, sb)
        mine = torch.from_numpy(np.frombuffer(part, dtype=np.uint8).copy())
        if self.device is not None:
            mine = mine.to(self.device)
        gathered = torch.empty(self.world * mine.numel(), dtype=torch.uint8, device=mine.device)
        self.dist.all_gather_into_tensor(gathered, mine, group=self.group)
        allp = gathered.cpu().numpy().tobytes()
        a, b, c = g.lib.prove_combine(g.ctx, cv.curve_id, allp, self.world, rb, sb, g.sizes)
        return Proof(a, b, c)

    def close(self):
        for h in self._pk_handles.values():
            self.g.lib.dll.ark355_pk_free(h)
        self._pk_handles.clear()
