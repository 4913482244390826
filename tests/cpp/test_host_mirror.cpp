// Tests of the C++ host mirror, written to read like the reference's own tests
// (/root/reference/relations/src/gr1cs/tests/mod.rs, utils/variable.rs:206-266, sr1cs/mod.rs:320-330).
//
//   ./test_host_mirror                 CPU-only checks (no device call)
//   ./test_host_mirror --prove <curve> <circuit> <n>
//        GPU: Groth16 circuit_specific_setup + prove through libark355.so with a scripted rng; prints the
//        proof / vk bytes so that the pytest side can compare them with the oracle.
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <string>

#include "../../host_mirror/snark.hpp"

using namespace ark_relations;
using namespace ark_relations::gr1cs;
using Fr = ark_snark::Field<ark355::BlsFr>;
using LC = LinearCombination<Fr>;

#define CHECK(cond)                                                            \
  do {                                                                         \
    if (!(cond)) {                                                             \
      fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);        \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

// gr1cs/tests/circuit2.rs:46-60
template <class F>
struct Circuit2 : ConstraintSynthesizer<F> {
  F a, b, c;
  Circuit2(F a_, F b_, F c_) : a(a_), b(b_), c(c_) {}
  void generate_constraints(ConstraintSystemRef<F> cs) override {
    using L = LinearCombination<F>;
    F two = F::one() + F::one();
    Variable va = cs.new_input_variable([&] { return a; });
    Variable vb = cs.new_witness_variable([&] { return b; });
    Variable vc = cs.new_witness_variable([&] { return c; });
    cs.enforce_r1cs_constraint([&] { return L() + va; }, [&] { return L() + std::make_pair(two, vb); }, [&] { return L() + vc; });
    Variable d = cs.new_lc([&] { return L() + va + vb; });
    cs.enforce_r1cs_constraint([&] { return L() + va; }, [&] { return L() + d; }, [&] { return L() + d; });
    Variable e = cs.new_lc([&] { return L() + d + d; });
    cs.enforce_r1cs_constraint([&] { return L() + Variable::One(); }, [&] { return L() + e; }, [&] { return L() + e; });
  }
};

// sr1cs/mod.rs:276-319
template <class F>
struct DummyCircuit : ConstraintSynthesizer<F> {
  F a, b;
  size_t num_variables, num_constraints;
  DummyCircuit(F a_, F b_, size_t nv, size_t nc) : a(a_), b(b_), num_variables(nv), num_constraints(nc) {}
  void generate_constraints(ConstraintSystemRef<F> cs) override {
    using L = LinearCombination<F>;
    Variable va = cs.new_witness_variable([&] { return a; });
    Variable vb = cs.new_witness_variable([&] { return b; });
    Variable vc = cs.new_input_variable([&] { return a * b; });
    for (size_t i = 0; i < num_variables - 3; i++) cs.new_witness_variable([&] { return a; });
    for (size_t i = 0; i < num_constraints - 1; i++)
      cs.enforce_r1cs_constraint([&] { return L::sum_vars({va}); }, [&] { return L::sum_vars({vb}); }, [&] { return L::sum_vars({vc}); });
    cs.enforce_r1cs_constraint([] { return L(); }, [] { return L(); }, [] { return L(); });
  }
};

// relations/examples/satisfiable.rs:7-150 (satisfiable = true) and examples/non_satisfiable.rs:10-166, whose
// enforce_addition assigns left * right to the sum ("intentionally made to fail").
template <class F>
struct ExampleCircuit : ConstraintSynthesizer<F> {
  bool satisfiable;
  explicit ExampleCircuit(bool sat = true) : satisfiable(sat) {}
  void generate_constraints(ConstraintSystemRef<F> cs) override {
    using L = LinearCombination<F>;
    Variable p1 = cs.new_input_variable([] { return F::from_u64(3); });
    Variable p2 = cs.new_input_variable([] { return F::from_u64(4); });
    Variable p3 = cs.new_input_variable([] { return F::from_u64(6); });
    Variable p4 = cs.new_input_variable([] { return F::from_u64(7); });
    Variable w1 = cs.new_witness_variable([] { return F::from_u64(2); });
    Variable w2 = cs.new_witness_variable([] { return F::from_u64(5); });
    Variable w3 = cs.new_witness_variable([] { return F::from_u64(8); });
    Variable w4 = cs.new_witness_variable([] { return F::from_u64(9); });
    Variable expected = cs.new_input_variable([] { return F::from_u64(198); });
    auto mul = [&](Variable l, Variable r) {
      Variable prod = cs.new_witness_variable([&] { return cs.assigned_value(l) * cs.assigned_value(r); });
      cs.enforce_r1cs_constraint([&] { return L() + l; }, [&] { return L() + r; }, [&] { return L() + prod; });
      return prod;
    };
    auto add = [&](Variable l, Variable r) {
      Variable sum = cs.new_witness_variable([&] {
        return satisfiable ? cs.assigned_value(l) + cs.assigned_value(r) : cs.assigned_value(l) * cs.assigned_value(r);
      });
      cs.enforce_r1cs_constraint([&] { return L() + l + r; }, [&] { return L() + Variable::One(); }, [&] { return L() + sum; });
      return sum;
    };
    Variable product = mul(p1, p2);
    Variable sum = add(w1, w2);
    Variable r1 = mul(sum, product);
    Variable product1 = mul(p3, p4);
    Variable product2 = mul(w3, w4);
    Variable r2 = add(product1, product2);
    Variable fin = add(r1, r2);
    cs.enforce_r1cs_constraint([&] { return L() + fin; }, [&] { return L() + Variable::One(); }, [&] { return L() + expected; });
  }
};

// S2 "mulchain"  with the seed values handed in
template <class F>
struct MulChain : ConstraintSynthesizer<F> {
  F w0, w1;
  size_t n;
  MulChain(F a, F b, size_t n_) : w0(a), w1(b), n(n_) {}
  void generate_constraints(ConstraintSystemRef<F> cs) override {
    using L = LinearCombination<F>;
    std::vector<F> vals{w0, w1};
    for (size_t i = 0; i + 1 < n; i++) vals.push_back((vals[i] + vals[i + 1]) * vals[i + 1]);
    Variable x1 = cs.new_input_variable([&] { return vals[n]; });
    std::vector<Variable> ws;
    for (size_t i = 0; i < vals.size(); i++) ws.push_back(cs.new_witness_variable([&, i] { return vals[i]; }));
    for (size_t i = 0; i + 1 < n; i++)
      cs.enforce_r1cs_constraint([&, i] { return L() + ws[i] + ws[i + 1]; }, [&, i] { return L() + ws[i + 1]; },
                                 [&, i] { return L() + ws[i + 2]; });
    cs.enforce_r1cs_constraint([&] { return L() + ws[n]; }, [&] { return L() + Variable::One(); }, [&] { return L() + x1; });
  }
};

// splitmix64 stream of SURVEY.md 8d (same generator as oracle/synthetic.py)
struct SplitMix64 {
  uint64_t s;
  explicit SplitMix64(uint64_t seed) : s(seed) {}
  uint64_t next() {
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  // four little-endian u64 limbs -> integer mod r
  template <class F>
  F next_fr() {
    uint64_t l[4];
    for (int i = 0; i < 4; i++) l[i] = next();
    const F t32 = F::from_u64(1ull << 32), t64 = t32 * t32;
    F v = F::from_u64(l[3]);
    for (int i = 2; i >= 0; i--) v = v * t64 + F::from_u64(l[i]);
    return v;
  }
};

// S3 "bench-LC" (shape of relations/examples/bench.rs:13,36-56 made satisfiable): LCs of 1..=10 terms
// with random coefficients over the 10 most recent variables, c_i a fresh witness.  Same stream order as
// oracle/synthetic.py:bench_lc_cs, so the two constraint systems are identical.
template <class F>
struct BenchLc : ConstraintSynthesizer<F> {
  size_t n;
  uint64_t seed;
  BenchLc(size_t n_, uint64_t seed_ = 0x355) : n(n_), seed(seed_) {}
  void generate_constraints(ConstraintSystemRef<F> cs) override {
    using L = LinearCombination<F>;
    SplitMix64 rng(seed);
    std::vector<Variable> vars;
    std::vector<F> vals;                      // value of vars[i]
    for (int i = 0; i < 10; i++) {
      F v = rng.next_fr<F>();
      vars.push_back(cs.new_witness_variable([&] { return v; }));
      vals.push_back(v);
    }
    vars.push_back(cs.new_input_variable([] { return F::from_u64(7); }));
    vals.push_back(F::from_u64(7));
    auto rand_lc = [&](std::vector<std::pair<F, Variable>>& terms) {
      const size_t k = 1 + rng.next() % 10;
      const size_t base = vars.size() - 10;
      F acc = F::zero();
      terms.clear();
      for (size_t j = 0; j < k; j++) {
        F c = rng.next_fr<F>();
        const size_t idx = base + rng.next() % 10;
        terms.push_back({c, vars[idx]});
        acc = acc + c * vals[idx];
      }
      return acc;
    };
    std::vector<std::pair<F, Variable>> ta, tb;
    for (size_t i = 0; i < n; i++) {
      const F va = rand_lc(ta);
      const F vb = rand_lc(tb);
      const F cval = va * vb;
      const Variable cvar = cs.new_witness_variable([&] { return cval; });
      vars.push_back(cvar);
      vals.push_back(cval);
      auto mk = [](const std::vector<std::pair<F, Variable>>& t) {
        L lc;
        for (const auto& cv : t) lc = lc + cv;
        return lc;
      };
      cs.enforce_r1cs_constraint([&] { return mk(ta); }, [&] { return mk(tb); }, [&] { return L() + cvar; });
    }
  }
};

// The reference's own synthesis benchmark circuit (relations/examples/bench.rs:16-83): unit coefficients, an extra
// symbolic LC on every other constraint, three fresh witnesses per constraint.  Timing only (not satisfiable); the
// pseudo-random choices come from splitmix64 instead of the reference's StdRng.
template <class F>
struct RefBenchCircuit : ConstraintSynthesizer<F> {
  F a;
  size_t n;
  RefBenchCircuit(F a_, size_t n_) : a(a_), n(n_) {}
  void generate_constraints(ConstraintSystemRef<F> cs) override {
    using L = LinearCombination<F>;
    std::vector<Variable> vars;
    for (int i = 0; i < 3; i++) vars.push_back(cs.new_witness_variable([&] { return a; }));
    vars.reserve(3 * n + 3);
    SplitMix64 ra(0), rb(1), rc(2);
    for (size_t i = 0; i < n; i++) {
      const size_t cur = vars.size() < 10 ? vars.size() : 10, lower = vars.size() - cur;
      const size_t ka = 1 + ra.next() % 10, kb = 1 + rb.next() % 10;
      auto pick = [&](SplitMix64& r, size_t k) {
        L lc;
        for (size_t j = 0; j < k; j++) lc = std::move(lc) + vars[lower + r.next() % cur];   // Rust: `lc = lc + var` moves
        return lc;
      };
      const Variable ci = vars[lower + rc.next() % cur];
      if (i % 2 == 0) {
        const Variable extra = cs.new_lc([&] { return pick(rc, ka); });
        cs.enforce_r1cs_constraint([&] { return pick(ra, ka) + extra; }, [&] { return pick(rb, kb); },
                                   [&] { return L() + ci; });
      } else {
        cs.enforce_r1cs_constraint([&] { return pick(ra, ka); }, [&] { return pick(rb, kb); }, [&] { return L() + ci; });
      }
      for (int j = 0; j < 3; j++) vars.push_back(cs.new_witness_variable([&] { return a; }));
    }
  }
};

static void test_bench_lc_is_satisfied() {
  BenchLc<Fr> c(50);
  auto cs = ConstraintSystemRef<Fr>::new_ref();
  c.generate_constraints(cs);
  CHECK(cs.num_constraints() == 50 && cs.num_instance_variables() == 2 && cs.num_witness_variables() == 60);
  cs.finalize();
  CHECK(cs.is_satisfied());
}

// host synthesis throughput: the reference's examples/bench.rs measurement -- generate_constraints
// in Prove{construct_matrices: true} mode, then finalize -- plus to_matrices, on this machine's host
template <class F>
static int run_synth_bench(size_t n) {
  if (getenv("ARK355_SYNTH_WITNESS_ONLY")) {          // the per-proof path alone, on a fresh heap
    RefBenchCircuit<F> c0(F::from_u64(0x7654321), n);
    auto cs0 = ConstraintSystemRef<F>::new_ref();
    cs0.set_mode(SynthesisMode::prove(false, false));
    auto ta = std::chrono::steady_clock::now();
    c0.generate_constraints(cs0);
    cs0.finalize();
    std::vector<F> z0 = cs0.borrow().full_assignment();
    auto tb = std::chrono::steady_clock::now();
    printf("witness_only_ms=%.3f\nwitness_only_z_len=%zu\n", std::chrono::duration<double, std::milli>(tb - ta).count(), z0.size());
    return 0;
  }
  RefBenchCircuit<F> circ(F::from_u64(0x1234567), n);
  auto cs = ConstraintSystemRef<F>::new_ref();
  cs.set_optimization_goal(OptimizationGoal::Constraints);
  cs.set_mode(SynthesisMode::prove(true, false));
  auto t0 = std::chrono::steady_clock::now();
  circ.generate_constraints(cs);
  auto t1 = std::chrono::steady_clock::now();
  cs.finalize();
  auto t2 = std::chrono::steady_clock::now();
  const auto all = cs.to_matrices();
  const auto& m = all.at("R1CS");
  auto t3 = std::chrono::steady_clock::now();
  size_t nnz = 0;
  for (int k = 0; k < 3; k++)
    for (const auto& row : m[k]) nnz += row.size();
  auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  printf("synth_constraints=%zu\nsynth_witnesses=%zu\nsynth_nnz=%zu\n", (size_t)cs.num_constraints(),
         (size_t)cs.num_witness_variables(), nnz);
  printf("synth_generate_ms=%.3f\nsynth_finalize_ms=%.3f\nsynth_to_matrices_ms=%.3f\n", ms(t0, t1), ms(t1, t2), ms(t2, t3));
  printf("synth_constraints_per_s=%.0f\n", n / (ms(t0, t2) * 1e-3));
  // witness-only re-synthesis (what every proof after the first does: SynthesisMode::Prove{false, false} records no
  // constraint, constraint_system_ref.rs:241-243) + assembling z
  RefBenchCircuit<F> circ2(F::from_u64(0x7654321), n);
  auto cs2 = ConstraintSystemRef<F>::new_ref();
  cs2.set_mode(SynthesisMode::prove(false, false));
  auto t4 = std::chrono::steady_clock::now();
  circ2.generate_constraints(cs2);
  cs2.finalize();
  std::vector<F> z = cs2.borrow().full_assignment();
  auto t5 = std::chrono::steady_clock::now();
  printf("witness_only_ms=%.3f\nwitness_only_constraints_per_s=%.0f\nwitness_only_z_len=%zu\n", ms(t4, t5),
         n / (ms(t4, t5) * 1e-3), z.size());
  return 0;
}

static void test_circuit2_matrices() {
  // gr1cs/tests/mod.rs:136-147 against circuit2.rs:19-43
  Fr one = Fr::one(), two = one + one;
  Circuit2<Fr> c(one, one, two);
  auto cs = ConstraintSystemRef<Fr>::new_ref();
  c.generate_constraints(cs);
  cs.finalize();
  auto m = cs.to_matrices().at("R1CS");
  using Row = std::vector<std::pair<Fr, size_t>>;
  std::vector<Matrix<Fr>> golden = {
      {Row{{one, 1}}, Row{{one, 1}}, Row{{one, 0}}},
      {Row{{two, 2}}, Row{{one, 1}, {one, 2}}, Row{{two, 1}, {two, 2}}},
      {Row{{one, 3}}, Row{{one, 1}, {one, 2}}, Row{{two, 1}, {two, 2}}},
  };
  CHECK(m.size() == 3);
  for (int k = 0; k < 3; k++) {
    CHECK(m[k].size() == golden[k].size());
    for (size_t i = 0; i < m[k].size(); i++) {
      CHECK(m[k][i].size() == golden[k][i].size());
      for (size_t t = 0; t < m[k][i].size(); t++)
        CHECK(m[k][i][t].first == golden[k][i][t].first && m[k][i][t].second == golden[k][i][t].second);
    }
  }
  CHECK(cs.is_satisfied());
}

static void test_variable_ordering() {
  // utils/variable.rs:206-266
  CHECK(Variable::Zero() < Variable::One());
  CHECK(Variable::One() < Variable::instance(0));
  CHECK(Variable::instance(0) < Variable::instance(1));
  CHECK(Variable::instance(1000) < Variable::witness(0));
  CHECK(Variable::witness(0) < Variable::witness(1));
  CHECK(Variable::witness(1000) < Variable::symbolic_lc(0));
  CHECK(Variable::symbolic_lc(0) < Variable::symbolic_lc(1));
  size_t idx;
  CHECK(Variable::One().get_variable_index(5, &idx) && idx == 0);
  CHECK(Variable::instance(3).get_variable_index(5, &idx) && idx == 3);
  CHECK(Variable::witness(3).get_variable_index(5, &idx) && idx == 8);
  CHECK(!Variable::symbolic_lc(3).get_variable_index(5, &idx));
}

static void test_dummy_circuit_synthesizes() {
  // sr1cs/mod.rs:320-330
  DummyCircuit<Fr> c(Fr::from_u64(3), Fr::from_u64(5), 128, 128);
  auto cs = ConstraintSystemRef<Fr>::new_ref();
  c.generate_constraints(cs);
  CHECK(cs.num_constraints() == 128 && cs.num_instance_variables() == 2 && cs.num_witness_variables() == 127);
  cs.finalize();
  CHECK(cs.is_satisfied());
}

static void test_modes_and_quirks() {
  // witness-only mode records nothing (constraint_system_ref.rs:241-243)
  auto cs = ConstraintSystemRef<Fr>::new_ref();
  cs.set_mode(SynthesisMode::prove(false, false));
  Variable a = cs.new_witness_variable([] { return Fr::from_u64(3); });
  cs.enforce_r1cs_constraint([&] { return LC() + a; }, [&] { return LC() + a; }, [&] { return LC() + a; });
  CHECK(cs.num_constraints() == 0);
  // trivial LCs are not stored (constraint_system.rs:480-485)
  auto cs2 = ConstraintSystemRef<Fr>::new_ref();
  Variable w = cs2.new_witness_variable([] { return Fr::from_u64(7); });
  CHECK(cs2.new_lc([] { return LC(); }) == Variable::symbolic_lc(0));
  CHECK(cs2.new_lc([&] { return LC() + w; }) == w);
  CHECK(cs2.new_lc([&] { return LC() + std::make_pair(Fr::from_u64(2), w); }) == Variable::symbolic_lc(1));
  // unsatisfied constraint is reported by index
  auto cs3 = ConstraintSystemRef<Fr>::new_ref();
  Variable x = cs3.new_witness_variable([] { return Fr::from_u64(2); });
  Variable y = cs3.new_witness_variable([] { return Fr::from_u64(5); });
  cs3.enforce_r1cs_constraint([&] { return LC() + x; }, [&] { return LC() + x; }, [&] { return LC() + x + x; });
  cs3.enforce_r1cs_constraint([&] { return LC() + x; }, [&] { return LC() + x; }, [&] { return LC() + y; });
  CHECK(cs3.which_is_unsatisfied() == "R1CS - 1");
  // setup mode stores no assignments and is_satisfied errors with AssignmentMissing (:598-600, :661-663)
  auto cs4 = ConstraintSystemRef<Fr>::new_ref();
  cs4.set_mode(SynthesisMode::setup());
  cs4.new_witness_variable([]() -> Fr { throw std::logic_error("closure must not run in setup mode"); });
  bool threw = false;
  try {
    cs4.is_satisfied();
  } catch (const SynthesisError& e) {
    threw = e.kind == SynthesisErrorKind::AssignmentMissing;
  }
  CHECK(threw);
}

// relations/examples/satisfiable.rs:29 (`assert!(cs.is_satisfied().unwrap())`), non_satisfiable.rs:41 (`which_is_unsatisfied`)
static void test_reference_examples() {
  auto cs = ConstraintSystemRef<Fr>::new_ref();
  ExampleCircuit<Fr>(true).generate_constraints(cs);
  CHECK(cs.num_constraints() == 8 && cs.num_instance_variables() == 6 && cs.num_witness_variables() == 11);
  CHECK(cs.is_satisfied());
  auto bad = ConstraintSystemRef<Fr>::new_ref();
  ExampleCircuit<Fr>(false).generate_constraints(bad);
  CHECK(!bad.is_satisfied());
  CHECK(bad.which_is_unsatisfied() == "R1CS - 1");
}

static void hex(const char* name, const std::vector<uint8_t>& b) {
  printf("%s=", name);
  for (uint8_t x : b) printf("%02x", x);
  printf("\n");
}

template <class C>
static int run_prove(const std::string& circuit, size_t n) {
  using G = ark_snark::Groth16<C>;
  using F = typename G::Fr;
  auto be = std::make_shared<ark_snark::Backend>(0);
  G groth(be);
  // scripted rng: setup draws tau, alpha, beta, gamma, delta; prove draws r then s
  uint64_t seq[] = {0x1234567, 11, 22, 33, 44, 0xabcdef01, 0x13579bdf};
  size_t pos = 0;
  typename G::Rng rng = [&]() { return F::from_u64(seq[pos++]); };
  std::unique_ptr<ConstraintSynthesizer<F>> circ;
  if (circuit == "dummy") circ.reset(new DummyCircuit<F>(F::from_u64(3), F::from_u64(5), n, n));
  else if (circuit == "benchlc") circ.reset(new BenchLc<F>(n));
  else if (circuit == "example") circ.reset(new ExampleCircuit<F>(true));
  else circ.reset(new MulChain<F>(F::from_u64(0x355), F::from_u64(0x356), n));
  auto keys = groth.circuit_specific_setup(*circ, rng);
  auto proof = groth.prove(keys.first, *circ, rng);
  hex("proof_a", proof.a);
  hex("proof_b", proof.b);
  hex("proof_c", proof.c);
  hex("vk_alpha_g1", keys.second.alpha_g1);
  hex("vk_gamma_abc_g1", keys.second.gamma_abc_g1);
  // second proof of the same circuit runs witness-only synthesis against the resident CSR
  uint64_t seq2[] = {0x777, 0x888};
  size_t p2 = 0;
  typename G::Rng rng2 = [&]() { return F::from_u64(seq2[p2++]); };
  auto proof2 = groth.prove(keys.first, *circ, rng2);
  hex("proof2_a", proof2.a);
  hex("proof2_b", proof2.b);
  hex("proof2_c", proof2.c);
  // ark355_prove_batch through the mirror: same circuit twice; the first pair of randomisers repeats proof2's
  uint64_t seq3[] = {0x777, 0x888, 0x999, 0xaaa};
  size_t p3 = 0;
  typename G::Rng rng3 = [&]() { return F::from_u64(seq3[p3++]); };
  auto batch = groth.prove_batch(keys.first, {circ.get(), circ.get()}, rng3, 2);
  if (batch.size() != 2 || batch[0].a != proof2.a || batch[0].b != proof2.b || batch[0].c != proof2.c) {
    fprintf(stderr, "prove_batch[0] differs from the single proof with the same randomisers\n");
    return 3;
  }
  if (batch[1].a == proof2.a) {
    fprintf(stderr, "prove_batch[1] ignored its randomisers\n");
    return 3;
  }
  printf("batch_ok 1\n");
  // SNARK::verify through the same ABI: the proofs of this run verify, a proof against the wrong input does not
  {
    typename G::CSRef cs = G::CSRef::new_ref();
    circ->generate_constraints(cs);
    cs.finalize();
    const auto& inner = cs.borrow();
    std::vector<F> x(inner.instance_assignment.begin() + 1, inner.instance_assignment.end());
    std::vector<F> wrong = x;
    if (!wrong.empty()) wrong[0] = wrong[0] + F::one();
    uint64_t seq4[] = {0x5151, 0x6262, 0x7373};
    size_t p4 = 0;
    typename G::Rng rng4 = [&]() { return F::from_u64(seq4[p4++ % 3]); };
    const bool v1 = groth.verify(keys.second, x, proof);
    const bool v2 = groth.verify_batch(keys.second, {x, x, x}, {proof, proof2, batch[1]}, rng4);
    const bool v3 = wrong.empty() ? false : groth.verify(keys.second, wrong, proof);
    if (!v1 || !v2 || v3) {
      fprintf(stderr, "verify: single %d batch %d wrong-input %d\n", (int)v1, (int)v2, (int)v3);
      return 4;
    }
    printf("verify_ok 1\n");
  }
  return 0;
}

// End-to-end SNARK::prove including synthesis: `count` instances of the S2 mulchain circuit with
// different seeds, K synthesis threads feeding `inflight` in-flight GPU proofs (Groth16::prove_pipelined), next to the
// device-only figure (the same assignments already synthesised, ark355_prove_batch).  Prints constraints/s for both
// and checks every pipelined proof against the batch proof with the same randomisers.
// circuit "benchlc": the S3 shape (the reference's own benchmark circuit, relations/examples/bench.rs:22-83, made satisfiable:
// BenchLc above; every instance synthesises the same system, as the reference's benchmark does) instead of the S2 mulchain.
template <class C>
static int run_e2e(size_t n, size_t count, uint32_t synth_threads, uint32_t inflight, bool benchlc = false, bool e2e_only = false) {
  using G = ark_snark::Groth16<C>;
  using F = typename G::Fr;
  auto be = std::make_shared<ark_snark::Backend>(0);
  G groth(be);
  uint64_t seq[] = {0x1234567, 11, 22, 33, 44};
  size_t pos = 0;
  typename G::Rng rng = [&]() { return F::from_u64(seq[pos++ % 5]); };
  auto make = [n, benchlc](size_t i) -> std::unique_ptr<ConstraintSynthesizer<F>> {
    if (benchlc) return std::unique_ptr<ConstraintSynthesizer<F>>(new BenchLc<F>(n, 0x355));
    return std::unique_ptr<ConstraintSynthesizer<F>>(new MulChain<F>(F::from_u64(0x355 + 2 * i), F::from_u64(0x356 + 2 * i), n));
  };
  auto c0 = make(0);
  auto t0 = std::chrono::steady_clock::now();
  auto keys = groth.circuit_specific_setup(*c0, rng);
  auto secs = [](auto a, auto b) { return std::chrono::duration<double>(b - a).count(); };
  printf("e2e_setup_s=%.3f\n", secs(t0, std::chrono::steady_clock::now()));
  std::vector<std::pair<F, F>> rs;
  for (size_t i = 0; i < count; i++) rs.emplace_back(F::from_u64(0x1000 + i), F::from_u64(0x2000 + i));
  // warm-up: loads key + matrices, touches every prover context
  typename G::PipelineStats st;
  groth.prove_pipelined(keys.first, std::min<size_t>(count, 2 * inflight), make, rs, synth_threads, inflight, &st);
  // optional sweep over the number of synthesis threads (ARK355_E2E_SWEEP="4,6,12"): one line each
  if (const char* sw = getenv("ARK355_E2E_SWEEP")) {
    for (const char* q = sw; *q;) {
      const uint32_t k = (uint32_t)strtoul(q, const_cast<char**>(&q), 10);
      if (*q == ',') q++;
      if (!k) break;
      typename G::PipelineStats s2;
      groth.prove_pipelined(keys.first, count, make, rs, k, inflight, &s2);
      printf("sweep synth_threads=%u inflight=%u wall_s=%.4f constraints_per_s=%.0f synth_cpu_s=%.3f prove_call_s=%.3f\n", k,
             inflight, s2.wall_s, (double)n * count / s2.wall_s, s2.synth_s, s2.prove_s);
    }
  }
  auto proofs = groth.prove_pipelined(keys.first, count, make, rs, synth_threads, inflight, &st);
  printf("e2e_synth_threads=%u\ne2e_inflight=%u\n", synth_threads, inflight);
  printf("e2e_wall_s=%.4f\ne2e_constraints_per_s=%.0f\ne2e_synth_cpu_s=%.4f\ne2e_prove_call_s=%.4f\n", st.wall_s,
         (double)n * count / st.wall_s, st.synth_s, st.prove_s);
  if (e2e_only) {           // (bench.py: the serial up-front synthesis of the comparison passes below costs minutes for S3 at 2^20)
    printf("e2e_ok 1\n");
    return 0;
  }
  // device-only: the same circuits, synthesised up front on this thread, then ark355_prove_batch
  std::vector<std::unique_ptr<ConstraintSynthesizer<F>>> owned;
  std::vector<ConstraintSynthesizer<F>*> ptrs;
  for (size_t i = 0; i < count; i++) {
    owned.push_back(make(i));
    ptrs.push_back(owned.back().get());
  }
  size_t p2 = 0;
  typename G::Rng rng2 = [&]() {
    const size_t i = p2++;
    return (i & 1) ? rs[i / 2].second : rs[i / 2].first;
  };
  double dev_s = 0;
  auto batch = groth.prove_batch(keys.first, ptrs, rng2, inflight, &dev_s);
  printf("device_only_s=%.4f\ndevice_only_constraints_per_s=%.0f\n", dev_s, (double)n * count / dev_s);
  for (size_t i = 0; i < count; i++) {
    if (proofs[i].a != batch[i].a || proofs[i].b != batch[i].b || proofs[i].c != batch[i].c) {
      fprintf(stderr, "pipelined proof %zu differs from the batch proof\n", i);
      return 3;
    }
  }
  printf("e2e_ratio=%.3f\n", dev_s / st.wall_s);
  // the same with the assignments in page-locked buffers (what prove_pipelined hands to the device): the fair
  // device-side ceiling of the pipeline
  {
    namespace g = ark_relations::gr1cs;
    std::vector<std::pair<void*, size_t>> bufs;
    std::vector<const F*> zs;
    uint64_t z_len = 0;
    for (size_t i = 0; i < count; i++) {
      auto cs = G::CSRef::new_ref();
      cs.set_optimization_goal(g::OptimizationGoal::Constraints);
      cs.set_mode(g::SynthesisMode::prove(false, false));
      owned[i]->generate_constraints(cs);
      cs.finalize();
      const std::vector<F> z = cs.borrow().full_assignment();
      auto b = be->take_pinned(z.size() * sizeof(F));
      std::memcpy(b.first, z.data(), z.size() * sizeof(F));
      bufs.push_back(b);
      zs.push_back(static_cast<const F*>(b.first));
      z_len = z.size();
    }
    double pin_s = 0;
    auto pinned = groth.prove_assignments(keys.first, zs, z_len, rs, inflight, &pin_s);
    for (auto& b : bufs) be->give_pinned(b);
    for (size_t i = 0; i < count; i++) {
      if (pinned[i].a != batch[i].a || pinned[i].b != batch[i].b || pinned[i].c != batch[i].c) {
        fprintf(stderr, "proof %zu from a page-locked assignment differs from the batch proof\n", i);
        return 3;
      }
    }
    printf("device_only_pinned_s=%.4f\ndevice_only_pinned_constraints_per_s=%.0f\ne2e_ratio_pinned=%.3f\n", pin_s,
           (double)n * count / pin_s, pin_s / st.wall_s);
  }
  printf("e2e_ok 1\n");
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 5 && std::string(argv[1]) == "--prove") {
    std::string curve = argv[2], circuit = argv[3];
    size_t n = strtoull(argv[4], nullptr, 10);
    try {
      if (curve == "bn254") return run_prove<ark_snark::BnCurveTag>(circuit, n);
      return run_prove<ark_snark::BlsCurveTag>(circuit, n);
    } catch (const std::exception& e) {
      fprintf(stderr, "error: %s\n", e.what());
      return 2;
    }
  }
  if (argc >= 7 && std::string(argv[1]) == "--e2e") {
    // --e2e <curve> <n> <count> <synth_threads> <inflight> [mulchain | benchlc] [e2e-only]
    const size_t n = strtoull(argv[3], nullptr, 10), count = strtoull(argv[4], nullptr, 10);
    const uint32_t k = (uint32_t)atoi(argv[5]), inflight = (uint32_t)atoi(argv[6]);
    const bool benchlc = argc >= 8 && std::string(argv[7]) == "benchlc";
    const bool e2e_only = argc >= 9 && std::string(argv[8]) == "e2e-only";
    try {
      if (std::string(argv[2]) == "bn254") return run_e2e<ark_snark::BnCurveTag>(n, count, k, inflight, benchlc, e2e_only);
      return run_e2e<ark_snark::BlsCurveTag>(n, count, k, inflight, benchlc, e2e_only);
    } catch (const std::exception& e) {
      fprintf(stderr, "error: %s\n", e.what());
      return 2;
    }
  }
  if (argc >= 4 && std::string(argv[1]) == "--synth-bench") {
    const size_t n = strtoull(argv[3], nullptr, 10);
    if (std::string(argv[2]) == "bn254") return run_synth_bench<ark_snark::Field<ark355::BnFr>>(n);
    return run_synth_bench<Fr>(n);
  }
  test_circuit2_matrices();
  test_bench_lc_is_satisfied();
  test_variable_ordering();
  test_dummy_circuit_synthesizes();
  test_modes_and_quirks();
  test_reference_examples();
  printf("host mirror: all CPU checks passed\n");
  return 0;
}
