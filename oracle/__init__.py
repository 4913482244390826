"""CPU oracle for the Groth16/R1CS prover hot path -- TEST INFRASTRUCTURE ONLY.

This package is the *checker*, never the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
``snark_amd`` (the product) never imports anything from here.

What it restates
----------------
* ``oracle.r1cs``      -- the in-tree semantics of ``ark_relations::gr1cs``
  (``/root/reference/relations/src/gr1cs/constraint_system.rs`` etc.), pinned against the
  reference's own golden matrices (``gr1cs/tests/circuit1.rs:28-61``, ``circuit2.rs:19-43``).
* ``oracle.fields / curves / ntt / groth16 / serialize / pairing`` -- the arithmetic that
  lives in crates that are NOT vendored under ``/root/reference`` (``ark-groth16``, ``ark-ec``,
  ``ark-poly``, ``ark-ff``, ``ark-bls12-381``, ``ark-bn254``, ``ark-serialize``; versions
  unpinned: the reference has no Cargo.lock and does not even depend on ark-groth16).  These
  follow the published algorithms.

PARITY UNPINNED at the Groth16 boundary: the reference holds no golden vector, KAT or fixture
for proofs / MSM / NTT, and it cannot be compiled here (no Rust toolchain).  What pins this
oracle instead: (1) the reference's R1CS golden matrices, (2) public curve constants
(generators, their zcash-format encodings, field moduli, roots of unity), (3) three
independent derivations that must agree byte-for-byte (step-by-step prover, trapdoor
closed form, C restatement in ``oracle/c``), (4) the pairing verification equation.
"""
