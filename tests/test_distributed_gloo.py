"""Multi-process CPU tests of the multi-GPU path (one proof, MSM term ranges sharded over the ranks): N processes,
each holding one shard of the proving key, `ark355_comm_init` + `ark355_prove_sharded` behind the C ABI.  The device
work runs on the CPU emulator build of the library's sources and RCCL is the shared-memory emulation of
tests/emul/rccl_emul.cpp (test infrastructure); `gloo` carries only the 128-byte communicator id, exactly as the
nccl backend does on GPUs.  Under test: the sharding of term ranges (equal window sizes across ranks), the all-gather of
the partial sums, the bucket-level ring reduce-scatter, and the combine -- the proof must be byte-identical to the
single-device proof and to the oracle's closed form on every rank, for both exchange modes."""
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


def _make_key(n, path):
    """One key for all ranks, from the oracle's C generator (fast), with the trapdoor's scalars for the closed form."""
    import pickle
    from oracle import groth16 as G, synthetic as S
    from oracle.c import cbase
    from oracle.fields import BLS12_381 as C
    nn, ell, w, mats, z = S.mulchain_csr(C.r, n)
    td = G.Trapdoor(tau=101, alpha=202, beta=303, gamma=404, delta=505)
    pk, sc = cbase.setup_raw_c(C, nn, ell, w, mats, td)
    ints = {k: [int.from_bytes(sc[k][32 * i:32 * i + 32], "little") for i in range(ell + w)] for k in "uvw"}
    with open(path, "wb") as f:
        pickle.dump(dict(pk=pk, N=sc["N"], ell=ell, w=w, z=z, td=dict(tau=td.tau, alpha=td.alpha, beta=td.beta,
                                                                     gamma=td.gamma, delta=td.delta, **ints)), f)


def _worker(rank, world, port, n, key_path, modes, whole, q, stride_on_rank1=None):
    try:
        if stride_on_rank1 == "rccl_self":
            # world size 1, the rank as its own peer: every exchange of the distributed witness map and one ring step per MSM go
            # through grouped ncclSend / ncclRecv to rank 0 itself (policy RCCL_SELF; the GPU tier runs the same on real RCCL)
            os.environ["ARK355_RCCL_SELF"] = "1"
            stride_on_rank1 = None
        if stride_on_rank1 and rank == 1:
            # this rank alone plans its tables differently ("2": another window stride) or lays its h_query shard out for the
            # replicated witness map ("dist_wm=0")
            if stride_on_rank1 == "dist_wm=0":
                os.environ["ARK355_SHARD_DIST_WM"] = "0"
            else:
                os.environ["ARK355_TABLE_STRIDE"] = stride_on_rank1
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "emul")):
            if p not in sys.path:
                sys.path.insert(0, p)
        import pickle
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import build_emul
        from snark_amd._binding import Lib, SHARD_BUCKET_RING, SHARD_WINDOW
        from snark_amd import params, synthetic
        from snark_amd.groth16 import Groth16, ProvingKey, VerifyingKey
        from snark_amd.parallel import ShardedGroth16
        lib = Lib(build_emul.build())
        cv = params.BLS12_381
        r1, z = synthetic.mulchain(cv, n)
        with open(key_path, "rb") as f:
            K = pickle.load(f)
        assert z == K["z"]
        raw = K["pk"]
        vk = VerifyingKey(raw["alpha_g1"], raw["beta_g2"], raw["gamma_g2"], raw["delta_g2"], raw["gamma_abc_g1"])
        pk = ProvingKey(vk=vk, beta_g1=raw["beta_g1"], delta_g1=raw["delta_g1"], a_query=raw["a_query"],
                        b_g1_query=raw["b_g1_query"], b_g2_query=raw["b_g2_query"], h_query=raw["h_query"],
                        l_query=raw["l_query"], ell=K["ell"], w=K["w"], N=K["N"], trapdoor=K["td"])
        g = Groth16(cv, lib=lib)
        sg = ShardedGroth16(g, device="cpu")
        zb = synthetic.z_to_mont_bytes(cv, z)
        closed = g.prove_closed_form(pk, z, 12345, 67890)
        if stride_on_rank1:
            # every rank must refuse: the ranks compare their plans over the communicator before the first exchange
            try:
                sg.prove(pk, r1, zb, r=12345, s=67890, mode=SHARD_BUCKET_RING)
                q.put((rank, False, False, "mismatching plans were not detected"))
            except Exception as e:
                q.put((rank, "different window sizes / table strides" in str(e), True, "refused"))
            sg.close()
            g.close()
            dist.destroy_process_group()
            return
        proofs = [sg.prove(pk, r1, zb, r=12345, s=67890, mode={"window": SHARD_WINDOW, "ring": SHARD_BUCKET_RING}[m])
                  for m in modes]
        proof_w = proofs[0]
        ok = all(p == closed for p in proofs)
        if not ok:
            sys.stderr.write("[rank %d] proofs that differ from the closed form: %s\n" % (rank, [m for m, p in zip(modes, proofs) if p != closed]))
        if rank == 0 and whole:
            # single-device proof of the same statement
            ok = ok and (g.prove(pk, r1, z, r=12345, s=67890) == closed)
        # the whole-key entry point must refuse a shard handle
        refused = world == 1           # (a "shard" of one is the whole key)
        try:
            if world > 1:
                lib.prove(g.ctx, sg.load_pk_shard(pk), g.load_r1cs(r1), zb, r1.m, cv.fr_canon(1), cv.fr_canon(2), g.sizes)
        except Exception as e:
            refused = getattr(e, "code", None) == -1
        q.put((rank, ok, refused, proof_w.a.hex()))
        sg.close()
        g.close()
        dist.destroy_process_group()
    except Exception:      # pragma: no cover
        import traceback
        q.put((rank, False, False, traceback.format_exc()))


def _run(world, n, tmp_path, modes=("window", "ring"), whole=True, stride_on_rank1=None):
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests", "emul"))
    import build_emul
    build_emul.build()
    key_path = str(tmp_path / "key.pkl")
    _make_key(n, key_path)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, key_path, modes, whole, q, stride_on_rank1)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _, _ in res), res
    assert all(refused for _, _, refused, _ in res), res
    assert len({h for _, _, _, h in res}) == 1          # every rank holds the same proof


# The emulator needs about a minute per 2^10-constraint proof, so at that size every world runs what it adds: the
# bucket ring with its minimal (2) and an odd (3) ring, both exchange modes at the node's real rank count (8).
@pytest.mark.parametrize("world,n,modes", [(2, 1 << 10, ("ring",)), (3, 1 << 10, ("window",)), (8, 1 << 10, ("window", "ring"))])
def test_sharded_prove_over_the_c_abi_comm(world, n, modes, tmp_path):
    _run(world, n, tmp_path, modes=modes, whole=False)


@pytest.mark.parametrize("world", [2])
def test_sharded_prove_small_instance_all_modes(world, tmp_path):
    """Both modes, plus the single-device proof of the same statement, at a size the emulator proves in seconds."""
    _run(world, 150, tmp_path)


def test_sharded_prove_self_exchange_world_size_1(tmp_path):
    """Policy RCCL_SELF at world size 1: the key "shard" is loaded in the layout of the distributed witness map, its three
    all-to-alls and one ring step per MSM run as grouped send / receive pairs of the rank with itself; same proof bytes as
    the closed form and the single-device proof."""
    _run(1, 150, tmp_path, stride_on_rank1="rccl_self")


def test_sharded_prove_tiny_instance_with_empty_shards(tmp_path):
    """n = 3 over 8 ranks: several shards of the h query are empty and still take part in the exchange."""
    _run(8, 3, tmp_path)


@pytest.mark.parametrize("override", ["2", "dist_wm=0"], ids=["table-stride", "witness-map-layout"])
def test_sharded_prove_refuses_ranks_with_different_table_plans(tmp_path, override):
    """One rank plans its window tables with another stride, or loads its h_query shard in the layout of the replicated
    witness map (an environment override on that rank only): the bucket-level exchange would add bucket arrays of different
    shapes, the other ranks would wait in all-to-alls that rank never enters.  The ranks compare their plans over the
    communicator on every sharded proof and every one of them returns ARK355_EINVAL instead."""
    _run(2, 150, tmp_path, stride_on_rank1=override)
