"""Helper for tests/test_gpu_parity.py::test_sharded_prove_over_rccl: the sharded prover over the REAL RCCL, behind the
C ABI (ark355_comm_init / ark355_prove_sharded), both exchange modes.  A 1-GPU box only admits world_size 1 -- the
shard is then the whole key and the ring has no steps -- but communicator creation from a broadcast id, the
ncclAllGather of the partial sums from HBM on the library's reduction stream and the combine run exactly as they do
on 8 GPUs."""
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
    for cv, n in ((params.BLS12_381, 300), (params.BN254, 77)):
        r1, z = synthetic.mulchain(cv, n)
        g = Groth16(cv, device=0)
        seq = iter([101, 202, 303, 404, 505])
        pk, vk = g.circuit_specific_setup(r1, lambda: next(seq), keep_trapdoor=True)
        sg = ShardedGroth16(g, device="cuda:0")
        zb = synthetic.z_to_mont_bytes(cv, z)
        whole = g.prove(pk, r1, z, r=12345, s=67890)
        closed = g.prove_closed_form(pk, z, 12345, 67890)
        for mode in (SHARD_WINDOW, SHARD_BUCKET_RING):
            assert sg.prove(pk, r1, zb, r=12345, s=67890, mode=mode) == whole == closed, (cv.name, mode)
        sg.close()
        g.close()
    dist.destroy_process_group()
    print("rccl_ok 1")


if __name__ == "__main__":
    main()
