"""Helper for tests/test_gpu_parity.py::test_sharded_prove_over_rccl: the sharded prover of snark_amd.parallel over
the real RCCL backend.  A 1-GPU box only admits world_size 1 -- the shard is then the whole key -- but the exchange
(all_gather_into_tensor of the partial sums from HBM, ark355_prove_combine) runs exactly as it does on 8 GPUs."""
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
    from snark_amd.parallel import ShardedGroth16
    for cv, n in ((params.BLS12_381, 300), (params.BN254, 77)):
        r1, z = synthetic.mulchain(cv, n)
        g = Groth16(cv, device=0)
        seq = iter([101, 202, 303, 404, 505])
        pk, vk = g.circuit_specific_setup(r1, lambda: next(seq), keep_trapdoor=True)
        sg = ShardedGroth16(g, device="cuda:0")
        proof = sg.prove(pk, r1, synthetic.z_to_mont_bytes(cv, z), r=12345, s=67890)
        whole = g.prove(pk, r1, z, r=12345, s=67890)
        closed = g.prove_closed_form(pk, z, 12345, 67890)
        assert proof == whole == closed, cv.name
        sg.close()
        g.close()
    dist.destroy_process_group()
    print("rccl_ok 1")


if __name__ == "__main__":
    main()
