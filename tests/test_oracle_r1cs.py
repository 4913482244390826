"""The oracle's restatement of ark_relations::gr1cs against the reference's OWN golden vectors
(the only KATs the reference holds for this path: SURVEY.md 8c)."""
import json
import os

from oracle import r1cs as R, synthetic as S
from oracle.fields import BLS12_381

P = BLS12_381.r
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    """tests/golden/*.json (written by tests/golden/make_r1cs_golden.py from the reference's test sources):
    {label: [matrix]} with rows of (coefficient, column) pairs"""
    raw = json.load(open(os.path.join(GOLDEN, name)))
    return {label: [[[(c, col) for c, col in row] for row in mat] for mat in mats] for label, mats in raw.items()}


def test_circuit2_matrices_golden():
    # /root/reference/relations/src/gr1cs/tests/mod.rs:136-147 vs circuit2.rs:19-43
    cs = R.ConstraintSystem(P)
    R.circuit2(cs, 1, 1, 2)
    cs.finalize()
    assert cs.to_matrices() == golden("circuit2_matrices.json")


def test_circuit1_matrices_golden():
    # tests/mod.rs:78-103 vs circuit1.rs:28-61 (before finalize)
    cs = R.ConstraintSystem(P)
    R.circuit1(cs, [0] * 5, [0] * 8)
    assert cs.to_matrices() == golden("circuit1_matrices.json")


def test_circuit1_sat_and_non_sat():
    # tests/mod.rs:17-76
    w = [4, 2, 5, 29, 28, 10, 57, 22022]
    cs = R.ConstraintSystem(P)
    R.circuit1(cs, [1, 2, 3, 0, 1255254], w)
    cs.finalize()
    assert cs.is_satisfied()
    cs = R.ConstraintSystem(P)
    R.circuit1(cs, [4, 2, 3, 0, 1255254], w)
    assert not cs.is_satisfied()
    assert cs.which_is_unsatisfied() == "poly-predicate-A - 0"


def test_variable_ordering():
    # utils/variable.rs:206-266: Zero < One < Instance < Witness < SymbolicLc, then by index
    vs = [R.VAR_ZERO, R.VAR_ONE, R.instance(0), R.instance(5), R.witness(0), R.witness(9), R.symbolic_lc(0),
          R.symbolic_lc(3)]
    assert vs == sorted(vs)


def test_trivial_lcs_are_not_stored():
    # constraint_system.rs:480-485
    cs = R.ConstraintSystem(P)
    a = cs.new_witness_variable(lambda: 3)
    assert cs.new_lc(lambda: R.LC(P)) == R.symbolic_lc(0)
    assert cs.new_lc(lambda: R.LC(P) + a) == a
    assert cs.num_linear_combinations == 1
    v = cs.new_lc(lambda: R.LC(P) + (2, a))
    assert v == R.symbolic_lc(1) and cs.lc_assignment[1] == 6


def test_witness_only_mode_records_nothing():
    # constraint_system_ref.rs:241-243; SURVEY 3.2
    cs = R.ConstraintSystem(P)
    cs.set_mode_prove(construct_matrices=False, generate_lc_assignments=False)
    a = cs.new_witness_variable(lambda: 3)
    cs.enforce_r1cs_constraint(lambda: R.LC(P) + a, lambda: R.LC(P) + a, lambda: R.LC(P) + a)
    assert cs.num_constraints() == 0
    cs.finalize()
    assert cs.witness_assignment == [3]


def test_setup_mode_has_no_assignments():
    cs = R.ConstraintSystem(P)
    cs.set_mode_setup()
    cs.new_witness_variable(lambda: 1 / 0)      # closure must not be evaluated (constraint_system.rs:613-615)
    assert cs.witness_assignment == []
    try:
        cs.which_is_unsatisfied()
        assert False
    except R.SynthesisError:
        pass


def test_dummy_circuit_shape():
    # sr1cs/mod.rs:320-330 (128 vars / 128 constraints, a=3, b=5)
    cs = S.dummy_cs(P, 128)
    assert cs.num_constraints() == 128
    assert cs.num_instance_variables == 2 and cs.num_witness_variables == 127
    A, B, C, z, ell = S.cs_to_instance(cs)
    assert cs.is_satisfied()
    assert A[0] == [(1, 2)] and B[0] == [(1, 3)] and C[0] == [(1, 1)] and A[127] == []
    assert z[1] == 15


def test_mulchain_direct_matches_constraint_system():
    for n in (1, 2, 7, 33):
        cs = S.mulchain_cs(P, n)
        assert cs.is_satisfied()
        assert S.cs_to_instance(cs) == S.mulchain_direct(P, n)


def test_bench_lc_satisfiable_and_rows_compact():
    cs = S.bench_lc_cs(P, 40)
    A, B, C, z, ell = S.cs_to_instance(cs)
    assert R.first_unsatisfied_r1cs(A, B, C, z, P) == -1
    zb = list(z)
    zb[-1] = (zb[-1] + 1) % P
    assert R.first_unsatisfied_r1cs(A, B, C, zb, P) == 39


def test_reference_example_programs():
    """relations/examples/satisfiable.rs (assert!(cs.is_satisfied())) and non_satisfiable.rs (which_is_unsatisfied is
    Some): the fixtures the reference holds for BASELINE configs[0]."""
    from oracle import r1cs as R
    from oracle.fields import BLS12_381 as C
    cs = R.ConstraintSystem(C.r)
    final = R.example_circuit(cs, satisfiable=True)
    assert cs.num_constraints() == 8 and cs.num_instance_variables == 6 and cs.num_witness_variables == 11
    assert cs.assigned_value(final) == 198
    assert cs.is_satisfied()
    cs.finalize()
    A, B, Cm = cs.to_matrices()[R.R1CS_PREDICATE_LABEL]
    z = cs.full_assignment()
    assert z[:6] == [1, 3, 4, 6, 7, 198] and z[6:10] == [2, 5, 8, 9]
    assert R.first_unsatisfied_r1cs(A, B, Cm, z, C.r) == -1
    bad = R.ConstraintSystem(C.r)
    R.example_circuit(bad, satisfiable=False)
    assert not bad.is_satisfied()
    assert bad.which_is_unsatisfied() == "R1CS - 1"          # the first addition gate: 2 + 5 != 2 * 5
    bad.finalize()
    A2, B2, C2 = bad.to_matrices()[R.R1CS_PREDICATE_LABEL]
    assert (A2, B2, C2) == (A, B, Cm)                        # same circuit, different witness
    assert R.first_unsatisfied_r1cs(A2, B2, C2, bad.full_assignment(), C.r) == 1
