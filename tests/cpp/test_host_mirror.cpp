// Tests of the C++ host mirror, written to read like the reference's own tests
// (/root/reference/relations/src/gr1cs/tests/mod.rs, utils/variable.rs:206-266, sr1cs/mod.rs:320-330).
//
//   ./test_host_mirror                 CPU-only checks (no device call)
//   ./test_host_mirror --prove <curve> <circuit> <n>
//        GPU: Groth16 circuit_specific_setup + prove through libark355.so with a scripted rng; prints the
//        proof / vk bytes so that the pytest side can compare them with the oracle.
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../snark_amd/host/snark.hpp"

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
  test_circuit2_matrices();
  test_variable_ordering();
  test_dummy_circuit_synthesizes();
  test_modes_and_quirks();
  printf("host mirror: all CPU checks passed\n");
  return 0;
}
