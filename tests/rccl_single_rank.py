"""Helper for tests/test_gpu_parity.py::test_sharded_prove_over_rccl: the sharded prover over the REAL RCCL, behind the
C ABI (ark355_comm_init / ark355_prove_sharded), both exchange modes.  A 1-GPU box only admits world_size 1 -- the
shard is then the whole key -- but communicator creation from a broadcast id, the ncclAllGather of the partial sums
from HBM on the library's reduction stream and the combine run exactly as they do on 8 GPUs.

Second pass, policy RCCL_SELF=1 (round 5): the rank is its own peer.  The key shard is loaded in the layout of the
distributed witness map, whose three all-to-all exchanges run as grouped ncclSend / ncclRecv pairs to rank 0 itself on the
witness-map stream, and the bucket ring makes one step with itself (send, receive, EC-add kernel) per MSM -- the
point-to-point RCCL calls of an 8-GPU proof, on the one GPU there is; the proofs must not change by a byte.  With
ARK355_TRACE_RCCL_SELF=<file> the pass is repeated at 2^12 constraints for a rocprofv3 kernel trace."""
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from snark_amd import params, synthetic
    from snark_amd.groth16 import Groth16
    from snark_amd.parallel import ShardedGroth16, SHARD_BUCKET_RING, SHARD_WINDOW
    cases = [(params.BLS12_381, 300), (params.BN254, 77)]
    if os.environ.get("ARK355_RCCL_SELF_BIG"):
        cases.append((params.BLS12_381, 1 << 12))
    for self_mode in (0, 1):
        for cv, n in cases:
            r1, z = synthetic.mulchain(cv, n)
            g = Groth16(cv, device=0)
            if self_mode:
                g.lib.ctx_set_policy(g.ctx, "RCCL_SELF", 1)
            seq = iter([101, 202, 303, 404, 505])
            pk, vk = g.circuit_specific_setup(r1, lambda: next(seq), keep_trapdoor=True)
            sg = ShardedGroth16(g, device="cuda:0")
            zb = synthetic.z_to_mont_bytes(cv, z)
            closed = g.prove_closed_form(pk, z, 12345, 67890)
            for mode in (SHARD_WINDOW, SHARD_BUCKET_RING):
                assert sg.prove(pk, r1, zb, r=12345, s=67890, mode=mode) == closed, (cv.name, mode, self_mode)
            if not self_mode:
                assert g.prove(pk, r1, z, r=12345, s=67890) == closed
            sg.close()
            g.close()
        print("rccl_self %d ok" % self_mode)
    dist.destroy_process_group()
    print("rccl_ok 1")


if __name__ == "__main__":
    main()
