"""world_size-2 CPU test of the multi-GPU path (snark_amd.parallel.ShardedGroth16): two processes over the
`gloo` backend, each holding one MSM shard of the proving key.  The device work runs on the CPU emulator build
(test infrastructure); what is under test is the sharding of term ranges, the single all-gather of the partial
sums, and `ark355_prove_combine` -- the proof must be byte-identical to the oracle's on both ranks."""
import os
import socket
import sys

import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "emul")):
            if p not in sys.path:
                sys.path.insert(0, p)
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import build_emul
        from snark_amd._binding import Lib
        from snark_amd import params, synthetic
        from snark_amd.groth16 import Groth16
        from snark_amd.parallel import ShardedGroth16
        lib = Lib(build_emul.build())
        cv = params.BLS12_381
        r1, z = synthetic.mulchain(cv, n)
        g = Groth16(cv, lib=lib)
        seq = iter([101, 202, 303, 404, 505])
        pk, vk = g.circuit_specific_setup(r1, lambda: next(seq), keep_trapdoor=True)
        sg = ShardedGroth16(g, device="cpu")
        proof = sg.prove(pk, r1, synthetic.z_to_mont_bytes(cv, z), r=12345, s=67890)
        # single-device proof of the same statement and the closed form must agree byte for byte
        whole = g.prove(pk, r1, z, r=12345, s=67890)
        closed = g.prove_closed_form(pk, z, 12345, 67890)
        ok = (proof == whole == closed)
        # the whole-key entry point must refuse a shard handle
        refused = False
        try:
            lib.prove(g.ctx, sg.load_pk_shard(pk), g.load_r1cs(r1), synthetic.z_to_mont_bytes(cv, z), r1.m,
                      cv.fr_canon(1), cv.fr_canon(2), g.sizes)
        except Exception as e:
            refused = getattr(e, "code", None) == -1
        q.put((rank, ok, refused, proof.a.hex()))
        sg.close()
        g.close()
        dist.destroy_process_group()
    except Exception as e:      # pragma: no cover
        import traceback
        q.put((rank, False, False, traceback.format_exc()))


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_prove_two_ranks_gloo(world):
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests", "emul"))
    import build_emul
    build_emul.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 7, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _, _ in res), res
    assert all(refused for _, _, refused, _ in res), res
    assert len({h for _, _, _, h in res}) == 1          # every rank holds the same proof
