"""One Groth16 proof with its MSM term ranges sharded over the GPUs of a node (SURVEY.md 8e,
BASELINE.json configs[2]) -- one process per GPU; the exchange is RCCL behind the C ABI
(`ark355_comm_init` / `ark355_prove_sharded`, snark_amd/csrc/comm_impl.cuh).  This module is the thin host-side
caller a Rust shim would be as well: it only moves the 128-byte communicator id from rank 0 to the other ranks over
the channel the host already has (here `torch.distributed`'s broadcast; any IPC does).

The path partitions cleanly: sum_i k_i P_i splits over disjoint term ranges and the partial sums combine by
group addition.  Each rank keeps terms [T*g/G, T*(g+1)/G) of the five (extended) query vectors resident
(window tables included).  The witness map is sharded as well (round 4, csrc/witness_dist_impl.cuh): a rank owns 1/G
of every vector through the whole map, three all-to-all exchanges (grouped ncclSend / ncclRecv on the witness-map
stream) move the vectors between the two layouts of the four-step transform, and the rank's share of h comes out in
the order its shard of `h_query` was loaded in; world sizes that are not a power of two (or domains too small) keep
the replicated map, where every rank computes h from z itself.  The rank then runs its five partial MSMs and contributes
ONE message of `ark355_partial_size()` bytes (960 B for BLS12-381) to an all-gather issued by the library on its reduction
stream, straight from HBM; `mode=SHARD_BUCKET_RING` selects the bucket-level ring reduce-scatter instead (the literal
"all-reduce of partial bucket sums"; it keeps the replicated map).  Every rank ends up with the same, byte-identical proof.

Independent proofs (BASELINE.json configs[4], and bench.py's default multi-GPU mode) need no collective at all:
one `Groth16` instance per rank.
"""
from __future__ import annotations

import numpy as np

from ._binding import COMM_ID_BYTES, SHARD_BUCKET_RING, SHARD_WINDOW  # noqa: F401  (re-exported)
from .groth16 import Groth16, Proof, ProvingKey, R1CS, SynthesisError
from ._binding import Ark355Error


class ShardedGroth16:
    """Collective `prove` across the ranks of a torch.distributed process group.  The process group is used ONCE, to
    hand rank 0's communicator id to the others (`device`: where that 128-byte tensor lives -- "cpu" for gloo,
    "cuda:<local>" for the nccl backend); the proof's own exchange never goes through torch."""

    def __init__(self, groth: Groth16, group=None, device=None):
        import torch
        import torch.distributed as dist
        self.g = groth
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        ident = groth.lib.comm_unique_id() if self.rank == 0 else bytes(COMM_ID_BYTES)
        t = torch.from_numpy(np.frombuffer(ident, dtype=np.uint8).copy())
        if device is not None:
            t = t.to(device)
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast(t, src=src, group=group)
        self.comm = groth.lib.comm_init(groth.ctx, t.cpu().numpy().tobytes(), self.rank, self.world)

    def load_pk_shard(self, pk: ProvingKey):
        h = getattr(pk, "_ark355_shard", None)
        if h is None or h[0] is not self:
            g = self.g
            handle = g.lib.pk_load(
                g.ctx, g.curve.curve_id, pk.ell, pk.w, pk.N, pk.a_query, pk.b_g1_query, pk.b_g2_query, pk.h_query,
                pk.l_query, pk.vk.alpha_g1, pk.beta_g1, pk.delta_g1, pk.vk.beta_g2, pk.vk.delta_g2,
                shard=(self.rank, self.world))
            g._track(pk, "_ark355_shard", (self, handle), g.lib.dll.ark355_pk_free, handle)
            h = (self, handle)
        return h[1]

    def prove(self, pk: ProvingKey, r1cs: R1CS, z, r: int, s: int, mode: int = SHARD_WINDOW,
              z_device_ptr=None) -> Proof:
        """All ranks call this with the same z, r, s; all return the same proof."""
        g, cv = self.g, self.g.curve
        pkh, rh = self.load_pk_shard(pk), g.load_r1cs(r1cs)
        try:
            if z_device_ptr is not None:
                a, b, c = g.lib.prove_sharded(g.ctx, self.comm, pkh, rh, z_device_ptr, r1cs.m, cv.fr_canon(r),
                                              cv.fr_canon(s), g.sizes, mode=mode, z_is_device_ptr=True)
            else:
                a, b, c = g.lib.prove_sharded(g.ctx, self.comm, pkh, rh, z, len(z) // 32, cv.fr_canon(r),
                                              cv.fr_canon(s), g.sizes, mode=mode)
        except Ark355Error as e:
            raise SynthesisError(str(e)) from e
        return Proof(a, b, c)

    def close(self):
        if self.comm is not None:
            self.g.lib.comm_destroy(self.comm)
            self.comm = None
