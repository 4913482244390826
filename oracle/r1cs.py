"""Restatement of ``ark_relations::gr1cs`` semantics on Python ints -- oracle (test infra).

Every function cites the reference file:line (relative to /root/reference/relations/src) it
follows.  Pinned against the reference's golden matrices in tests/test_oracle_r1cs.py.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple

# ---- Variable: utils/variable.rs:4-18,52-113 ---------------------------------------------------
ZERO_T, ONE_T, INSTANCE_T, WITNESS_T, LC_T = 0, 1, 2, 3, 4
Var = Tuple[int, int]           # (tag, index); tuple order == Variable's derived Ord (tag in top bits)

VAR_ZERO: Var = (ZERO_T, 0)
VAR_ONE: Var = (ONE_T, 0)


def instance(i) -> Var:
    return (INSTANCE_T, i)


def witness(i) -> Var:
    return (WITNESS_T, i)


def symbolic_lc(i) -> Var:
    return (LC_T, i)


def get_variable_index(v: Var, witness_offset: int) -> Optional[int]:
    """utils/variable.rs:105-113."""
    if v[0] == ONE_T:
        return 0
    if v[0] == INSTANCE_T:
        return v[1]
    if v[0] == WITNESS_T:
        return v[1] + witness_offset
    return None


class SynthesisError(Exception):
    """utils/error.rs:5-21."""


class LC:
    """LinearCombination<F>(Vec<(F, Variable)>) -- utils/linear_combination.rs:15."""

    def __init__(self, p, terms=None):
        self.p = p
        self.t: List[Tuple[int, Var]] = list(terms) if terms else []

    def copy(self):
        return LC(self.p, self.t)

    def get_var_loc(self, var):
        """utils/linear_combination.rs:174-191: linear scan never reports a hit below 6 terms."""
        if len(self.t) < 6:
            idx = 0
            for i, (_, v) in enumerate(self.t):
                if v >= var:
                    idx = i
                    break
                idx += 1
            return (False, idx)
        lo, hi = 0, len(self.t)
        while lo < hi:                      # binary_search_by_key
            mid = (lo + hi) // 2
            if self.t[mid][1] < var:
                lo = mid + 1
            elif self.t[mid][1] > var:
                hi = mid
            else:
                return (True, mid)
        return (False, lo)

    def add_term(self, coeff, var):
        """AddAssign<(F, Variable)> -- utils/linear_combination.rs:204-212."""
        found, i = self.get_var_loc(var)
        if found:
            self.t[i] = ((self.t[i][0] + coeff) % self.p, var)
        else:
            self.t.insert(i, (coeff % self.p, var))
        return self

    def __add__(self, other):
        out = self.copy()
        if isinstance(other, tuple) and len(other) == 2 and isinstance(other[1], tuple):
            return out.add_term(other[0], other[1])      # (coeff, var)
        return out.add_term(1, other)                    # Variable

    def __sub__(self, other):
        out = self.copy()
        if isinstance(other, tuple) and len(other) == 2 and isinstance(other[1], tuple):
            return out.add_term(-other[0], other[1])
        return out.add_term(-1, other)

    def compactify(self):
        """utils/linear_combination.rs:53-82 (sort by Variable, merge equal keys)."""
        if len(self.t) <= 1:
            return
        self.t.sort(key=lambda e: e[1])
        out = [self.t[0]]
        for c, v in self.t[1:]:
            if out[-1][1] == v:
                out[-1] = ((out[-1][0] + c) % self.p, v)
            else:
                out.append((c, v))
        self.t = out


def sum_vars(p, vars_):
    """lc![a, b, ...] -- utils/linear_combination.rs:28,84-92."""
    return LC(p, [(1, v) for v in vars_])


# ---- predicates: gr1cs/predicate/{mod,polynomial_constraint}.rs ------------------------------
R1CS_PREDICATE_LABEL = "R1CS"          # predicate/polynomial_constraint.rs:69


class PolynomialPredicate:
    """Sparse multivariate polynomial; satisfied iff it evaluates to 0 (polynomial_constraint.rs:46-48)."""

    def __init__(self, p, arity, terms):
        self.p, self.arity, self.terms = p, arity, terms   # terms: [(coeff, [(var_idx, exp), ...])]

    def is_satisfied(self, values):
        acc = 0
        for coeff, mono in self.terms:
            t = coeff
            for vi, e in mono:
                t = t * pow(values[vi], e, self.p) % self.p
            acc = (acc + t) % self.p
        return acc == 0


class PredicateCS:
    """PredicateConstraintSystem -- predicate/mod.rs:81-94; column-wise argument_lcs."""

    def __init__(self, predicate: PolynomialPredicate):
        self.predicate = predicate
        self.argument_lcs: List[List[Var]] = [[] for _ in range(predicate.arity)]
        self.num_constraints = 0

    @staticmethod
    def new_r1cs(p):
        """predicate/mod.rs:115-120: x0*x1 - x2."""
        return PredicateCS(PolynomialPredicate(p, 3, [(1, [(0, 1), (1, 1)]), (p - 1, [(2, 1)])]))

    def enforce_constraint(self, lcs):
        """predicate/mod.rs:156-174."""
        if len(lcs) != self.predicate.arity:
            raise SynthesisError("ArityMismatch")
        for col, v in zip(self.argument_lcs, lcs):
            col.append(v)
        self.num_constraints += 1


class ConstraintSystem:
    """gr1cs/constraint_system.rs:44-97 (fields) / :109-139 (new)."""

    def __init__(self, p):
        self.p = p
        self.num_instance_variables = 1
        self.num_witness_variables = 0
        self.num_linear_combinations = 1
        self.instance_assignment = [1]            # :121
        self.witness_assignment: List[int] = []
        self.lc_assignment = [0]
        self.lc_map: List[List[Tuple[int, Var]]] = [[]]   # LcMap with the zero LC pushed (:111)
        self.predicates: Dict[str, PredicateCS] = {}
        # SynthesisMode (gr1cs/mod.rs:74-90)
        self.setup_mode = False
        self.construct_matrices = True
        self.generate_lc_assignments = True
        self.register_predicate(R1CS_PREDICATE_LABEL, PredicateCS.new_r1cs(p))   # :136-137

    # -- modes --------------------------------------------------------------------------------
    def set_mode_setup(self):
        self.setup_mode, self.construct_matrices, self.generate_lc_assignments = True, True, False

    def set_mode_prove(self, construct_matrices=True, generate_lc_assignments=True):
        self.setup_mode = False
        self.construct_matrices = construct_matrices
        self.generate_lc_assignments = generate_lc_assignments

    def should_construct_matrices(self):
        return self.setup_mode or self.construct_matrices

    def should_generate_lc_assignments(self):
        return (not self.setup_mode) and self.generate_lc_assignments

    def register_predicate(self, label, pcs):
        self.predicates[label] = pcs

    # -- allocation: constraint_system.rs:591-617 ---------------------------------------------------
    def new_input_variable(self, f: Callable[[], int]) -> Var:
        i = self.num_instance_variables
        self.num_instance_variables += 1
        if not self.setup_mode:
            self.instance_assignment.append(f() % self.p)
        return instance(i)

    def new_witness_variable(self, f: Callable[[], int]) -> Var:
        i = self.num_witness_variables
        self.num_witness_variables += 1
        if not self.setup_mode:
            self.witness_assignment.append(f() % self.p)
        return witness(i)

    # -- LCs: constraint_system.rs:455-532 -----------------------------------------------------------
    def assigned_value(self, v: Var):
        """assignment.rs:26-35."""
        if v[0] == ZERO_T:
            return 0
        if v[0] == ONE_T:
            return 1
        if v[0] == INSTANCE_T:
            return self.instance_assignment[v[1]] if v[1] < len(self.instance_assignment) else None
        if v[0] == WITNESS_T:
            return self.witness_assignment[v[1]] if v[1] < len(self.witness_assignment) else None
        return self.lc_assignment[v[1]] if v[1] < len(self.lc_assignment) else None

    def _eval_terms(self, terms):
        acc = 0
        for c, v in terms:
            val = self.assigned_value(v)
            if val is None:
                return None
            acc = (acc + c * val) % self.p
        return acc

    def _new_lc_add_helper(self, lc: LC) -> Var:
        """constraint_system.rs:472-499."""
        t = lc.t
        if len(t) == 0 or (len(t) == 1 and t[0][1] == VAR_ZERO):
            return symbolic_lc(0)
        if len(t) == 1 and t[0][0] % self.p == 1:
            return t[0][1]
        index = self.num_linear_combinations
        self.lc_map.append(list(t))
        self.num_linear_combinations += 1
        if self.should_generate_lc_assignments():
            self.lc_assignment.append(self._eval_terms(t))
        return symbolic_lc(index)

    def _new_lc_helper(self, f) -> Var:
        """constraint_system.rs:503-519."""
        if self.should_construct_matrices() or self.should_generate_lc_assignments():
            return self._new_lc_add_helper(f())
        index = self.num_linear_combinations
        self.num_linear_combinations += 1
        return symbolic_lc(index)

    def new_lc(self, f) -> Var:
        """constraint_system.rs:523-532."""
        return self._new_lc_helper(f)

    def _new_constraint_lc(self, f) -> Var:
        """constraint_system.rs:455-461."""
        if self.should_construct_matrices():
            return self._new_lc_helper(f)
        index = self.num_linear_combinations
        self.num_linear_combinations += 1
        return symbolic_lc(index)

    def enforce_constraint(self, label, *lcs):
        """constraint_system.rs:323-353 (arity 3) and siblings."""
        if label not in self.predicates:
            raise SynthesisError("PredicateNotFound")
        if self.should_construct_matrices():
            vs = [self._new_constraint_lc(f) for f in lcs]
            self.predicates[label].enforce_constraint(vs)

    def enforce_r1cs_constraint(self, a, b, c):
        """constraint_system.rs:431-438; constraint_system_ref.rs:235-250 returns early when
        matrices are not being constructed (:241-243)."""
        if not self.should_construct_matrices():
            return
        self.enforce_constraint(R1CS_PREDICATE_LABEL, a, b, c)

    # -- counts: constraint_system.rs:210-230 ---------------------------------------------------------
    def num_constraints(self):
        return sum(p.num_constraints for p in self.predicates.values())

    # -- finalize / inlining: constraint_system.rs:691-758 -------------------------------------------
    def finalize(self):
        self.inline_all_lcs()

    def inline_all_lcs(self):
        if not self.should_construct_matrices():
            return
        if not any(v[0] == LC_T for lc in self.lc_map for (_, v) in lc):       # any_lcs_used :762-764
            return
        p = self.p
        inlined: List[List[Tuple[int, Var]]] = []
        for lc in self.lc_map:
            out = LC(p)
            for coeff, var in lc:
                if var[0] == LC_T:
                    sub = inlined[var[1]]
                    if coeff % p == 1:
                        out.t.extend(sub)
                    else:
                        out.t.extend(((coeff * c) % p, v) for (c, v) in sub
                                     if v != VAR_ZERO and c % p != 0)
                else:
                    out.t.append((coeff, var))
            out.compactify()
            inlined.append(out.t)
        self.lc_map = inlined

    # -- matrices: constraint_system.rs:768-804 -------------------------------------------------------
    def get_lc(self, v: Var):
        if v == VAR_ZERO:
            return []
        if v[0] == LC_T:
            return list(self.lc_map[v[1]])
        return [(1, v)]

    def make_row(self, terms):
        n_in = self.num_instance_variables
        row = []
        for c, v in terms:
            if c % self.p == 0 or v == VAR_ZERO:
                continue
            idx = get_variable_index(v, n_in)
            if idx is None:
                raise RuntimeError("un-inlined LC in make_row (reference panics: constraint_system.rs:800)")
            row.append((c % self.p, idx))
        return row

    def to_matrices(self):
        """BTreeMap<Label, Vec<Matrix<F>>> -- label order is sorted (BTreeMap)."""
        out = {}
        for label in sorted(self.predicates):
            pcs = self.predicates[label]
            mats = [[] for _ in range(pcs.predicate.arity)]
            for i in range(pcs.num_constraints):
                for k in range(pcs.predicate.arity):
                    mats[k].append(self.make_row(self.get_lc(pcs.argument_lcs[k][i])))
            out[label] = mats
        return out

    # -- satisfaction: constraint_system.rs:652-687, predicate/mod.rs:185-204 -----------------------
    def which_is_unsatisfied(self):
        if self.setup_mode:
            raise SynthesisError("AssignmentMissing")
        for label in sorted(self.predicates):
            pcs = self.predicates[label]
            for i in range(pcs.num_constraints):
                vals = []
                for k in range(pcs.predicate.arity):
                    v = pcs.argument_lcs[k][i]
                    val = self.assigned_value(v) if v[0] != LC_T or self.should_generate_lc_assignments() else None
                    if val is None:
                        val = self._eval_terms(self.get_lc(v))
                    vals.append(val)
                if not pcs.predicate.is_satisfied(vals):
                    return "%s - %d" % (label, i)
        return None

    def is_satisfied(self):
        return self.which_is_unsatisfied() is None

    # -- the prover's view ------------------------------------------------------------------------
    def full_assignment(self):
        """z = instance || witness (constraint_system.rs:193-206)."""
        return list(self.instance_assignment) + list(self.witness_assignment)


def mat_vec_mul(matrix, z, p):
    """utils/matrix.rs:26-36."""
    return [sum(c * z[j] for c, j in row) % p for row in matrix]


def first_unsatisfied_r1cs(A, B, C, z, p):
    """Index of the first i with <A_i,z>*<B_i,z> != <C_i,z>, or -1."""
    for i, (ra, rb, rc) in enumerate(zip(A, B, C)):
        a = sum(c * z[j] for c, j in ra) % p
        b = sum(c * z[j] for c, j in rb) % p
        c_ = sum(c * z[j] for c, j in rc) % p
        if a * b % p != c_:
            return i
    return -1


# ---- the reference's own test / example circuits --------------------------------------------------
def circuit2(cs: ConstraintSystem, a, b, c):
    """gr1cs/tests/circuit2.rs:46-60."""
    p = cs.p
    va = cs.new_input_variable(lambda: a)
    vb = cs.new_witness_variable(lambda: b)
    vc = cs.new_witness_variable(lambda: c)
    cs.enforce_r1cs_constraint(lambda: LC(p) + va, lambda: LC(p) + (2, vb), lambda: LC(p) + vc)
    d = cs.new_lc(lambda: LC(p) + va + vb)
    cs.enforce_r1cs_constraint(lambda: LC(p) + va, lambda: LC(p) + d, lambda: LC(p) + d)
    e = cs.new_lc(lambda: LC(p) + d + d)
    cs.enforce_r1cs_constraint(lambda: LC(p) + VAR_ONE, lambda: LC(p) + e, lambda: LC(p) + e)


def circuit1(cs: ConstraintSystem, x, w):
    """gr1cs/tests/circuit1.rs:64-165; x = [x1..x5], w = [w1..w8]."""
    p = cs.p
    xs = [cs.new_input_variable(lambda v=v: v) for v in x]
    ws = [cs.new_witness_variable(lambda v=v: v) for v in w]
    x1, x2, x3, x4, x5 = xs
    w1, w2, w3, w4, w5, w6, _w7, w8 = ws
    cs.register_predicate("poly-predicate-A", PredicateCS(PolynomialPredicate(
        p, 4, [(1, [(0, 1), (1, 1)]), (3, [(2, 2)]), (p - 1, [(3, 1)])])))
    cs.register_predicate("poly-predicate-B", PredicateCS(PolynomialPredicate(
        p, 3, [(7, [(1, 1)]), (1, [(0, 3)]), (p - 1, [(2, 1)])])))
    cs.register_predicate("poly-predicate-C", PredicateCS(PolynomialPredicate(
        p, 3, [(1, [(0, 1), (1, 1)]), (p - 1, [(2, 1)])])))
    L = lambda *vs: (lambda: sum_vars(p, []) if not vs else _chain(p, vs))
    cs.enforce_constraint("poly-predicate-A", L(x1), L(x2), L(x3), L(w4))
    cs.enforce_constraint("poly-predicate-B", L(x4), L(w1), L(w5))
    cs.enforce_constraint("poly-predicate-B", L(w5), L(w6), L(w8))
    cs.enforce_constraint("poly-predicate-C", L(w2), L(w3), L(w6))
    cs.enforce_constraint("poly-predicate-C", L(w5, w4), L(w8), L(x5))


def _chain(p, vs):
    lc = LC(p)
    for v in vs:
        lc = lc + v
    return lc


def dummy_circuit(cs: ConstraintSystem, a, b, num_variables, num_constraints):
    """sr1cs/mod.rs:295-318 (DummyCircuit)."""
    p = cs.p
    va = cs.new_witness_variable(lambda: a)
    vb = cs.new_witness_variable(lambda: b)
    vc = cs.new_input_variable(lambda: a * b % p)
    for _ in range(num_variables - 3):
        cs.new_witness_variable(lambda: a)
    for _ in range(num_constraints - 1):
        cs.enforce_r1cs_constraint(lambda: sum_vars(p, [va]), lambda: sum_vars(p, [vb]),
                                   lambda: sum_vars(p, [vc]))
    cs.enforce_r1cs_constraint(lambda: LC(p), lambda: LC(p), lambda: LC(p))


def example_circuit(cs: ConstraintSystem, satisfiable=True):
    """The reference's two example programs, exactly: /root/reference/relations/examples/satisfiable.rs:7-32 (inputs
    3, 4, 6, 7 and the expected result 198; witnesses 2, 5, 8, 9) with its gates (:36-150), and
    examples/non_satisfiable.rs:10-44, whose enforce_addition assigns left * right to the sum (:150-166) so that the
    second constraint -- the first addition gate -- is the one `which_is_unsatisfied` reports.
    8 constraints, 6 instance variables (with One), 11 witnesses."""
    p = cs.p
    p1 = cs.new_input_variable(lambda: 3)
    p2 = cs.new_input_variable(lambda: 4)
    p3 = cs.new_input_variable(lambda: 6)
    p4 = cs.new_input_variable(lambda: 7)
    w1 = cs.new_witness_variable(lambda: 2)
    w2 = cs.new_witness_variable(lambda: 5)
    w3 = cs.new_witness_variable(lambda: 8)
    w4 = cs.new_witness_variable(lambda: 9)
    expected = cs.new_input_variable(lambda: 198)

    def mul(left, right):
        prod = cs.new_witness_variable(lambda: cs.assigned_value(left) * cs.assigned_value(right) % p)
        cs.enforce_r1cs_constraint(lambda: LC(p) + left, lambda: LC(p) + right, lambda: LC(p) + prod)
        return prod

    def add(left, right):
        if satisfiable:
            s = cs.new_witness_variable(lambda: (cs.assigned_value(left) + cs.assigned_value(right)) % p)
        else:       # "Instead of + we use *, This is intentionally made to fail" (non_satisfiable.rs:150-156)
            s = cs.new_witness_variable(lambda: cs.assigned_value(left) * cs.assigned_value(right) % p)
        cs.enforce_r1cs_constraint(lambda: LC(p) + left + right, lambda: LC(p) + VAR_ONE, lambda: LC(p) + s)
        return s

    # subcircuit 1: (w1 + w2) * (p1 * p2)
    product = mul(p1, p2)
    sm = add(w1, w2)
    r1 = mul(sm, product)
    # subcircuit 2: p3 * p4 + w3 * w4
    product1 = mul(p3, p4)
    product2 = mul(w3, w4)
    r2 = add(product1, product2)
    final = add(r1, r2)
    cs.enforce_r1cs_constraint(lambda: LC(p) + final, lambda: LC(p) + VAR_ONE, lambda: LC(p) + expected)
    return final
