// C++ host-side mirror of `ark_snark::SNARK` (/root/reference/snark/src/lib.rs:22-81) with ONE implementor,
// Groth16<Curve>, whose `prove` calls the MI355X backend through the C ABI (include/ark355.h).
//
//   SNARK::circuit_specific_setup(circuit, rng) -> (ProvingKey, VerifyingKey)     lib.rs:43-46, :87-92
//   SNARK::prove(&pk, circuit, rng) -> Proof                                       lib.rs:50-54
//   SNARK::verify / verify_with_processed_vk (+ verify_batch)                      lib.rs:59-80: ark355_verify_batch
//       (random linear combination on the device MSM, Miller loops + final exponentiation on host threads);
//       proofs stay byte-compatible with the arkworks CPU verifier.
//
// `prove` follows the un-vendored ark-groth16 `create_random_proof_with_reduction`: new constraint system,
// OptimizationGoal::Constraints, generate_constraints, finalize, matrices + assignment, then r and s drawn
// from the rng IN THAT ORDER, then the device call.  After the first proof of a circuit the CSR matrices are
// resident and synthesis runs in the witness-only mode of SURVEY.md 3.2
// (SynthesisMode::Prove{construct_matrices: false, generate_lc_assignments: false}).
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <vector>

#include "../include/ark355.h"
#include "../snark_amd/csrc/curve.cuh"
#include "relations.hpp"

namespace ark355 {
template <class FP>
inline FP fr_from_params_host(uint32_t (*f)(int)) {
  FP r;
  for (int i = 0; i < FP::N; i++) r.l[i] = f(i);
  return r;
}
}  // namespace ark355

namespace ark_snark {

// Field element with value semantics over the library's Montgomery arithmetic (host instantiation).
template <class FP>
struct Field {
  FP v;
  static Field zero() { return Field{FP::zero()}; }
  static Field one() { return Field{FP::one()}; }
  static Field from_u64(uint64_t x) {
    FP c = FP::zero();
    c.l[0] = (uint32_t)x;
    c.l[1] = (uint32_t)(x >> 32);
    return Field{FP::to_mont(c)};
  }
  // canonical little-endian bytes -> element (value must be < modulus)
  static Field from_canonical_bytes(const uint8_t* b) {
    FP c;
    memcpy(c.l, b, sizeof(FP));
    return Field{FP::to_mont(c)};
  }
  void to_canonical_bytes(uint8_t* out) const {
    FP c = FP::from_mont(v);
    memcpy(out, c.l, sizeof(FP));
  }
  Field operator+(const Field& o) const { return Field{FP::add(v, o.v)}; }
  Field operator-(const Field& o) const { return Field{FP::sub(v, o.v)}; }
  Field operator-() const { return Field{FP::neg(v)}; }
  Field operator*(const Field& o) const { return Field{FP::mul(v, o.v)}; }
  bool operator==(const Field& o) const { return v == o.v; }
  bool operator!=(const Field& o) const { return !(v == o.v); }
  Field inverse() const { return Field{FP::inv(v)}; }
  Field pow_u64(uint64_t e) const {
    Field r = one(), b = *this;
    while (e) {
      if (e & 1) r = r * b;
      b = b * b;
      e >>= 1;
    }
    return r;
  }
};

struct BlsCurveTag {
  static constexpr int ID = ARK355_BLS12_381;
  using FrP = ark355::BlsFr;
  using FqP = ark355::BlsFq;
  using Consts = ark355::BlsCurveConsts;
  using FrParams = ark355::BlsFrParams;
};
struct BnCurveTag {
  static constexpr int ID = ARK355_BN254;
  using FrP = ark355::BnFr;
  using FqP = ark355::BnFq;
  using Consts = ark355::BnCurveConsts;
  using FrParams = ark355::BnFrParams;
};

struct BackendError : std::runtime_error {
  int code;
  BackendError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

// RAII device context shared by keys
class Backend {
 public:
  explicit Backend(int device = 0) : device_(device) {
    int rc = ark355_ctx_create(device, &ctx_);
    if (rc != ARK355_OK) throw BackendError(rc, "ark355_ctx_create failed (no GPU? the backend has no CPU fallback)");
  }
  ~Backend() {
    for (ark355_ctx* c : workers_) ark355_ctx_destroy(c);
    for (auto& b : pinned_) ark355_host_free(b.first);
    ark355_ctx_destroy(ctx_);
  }
  Backend(const Backend&) = delete;
  ark355_ctx* ctx() const { return ctx_; }
  int device() const { return device_; }
  void check(int rc) const {
    if (rc == ARK355_OK) return;
    if (rc == ARK355_E_ASSIGNMENT_MISSING) throw ark_relations::SynthesisError(ark_relations::SynthesisErrorKind::AssignmentMissing);
    if (rc == ARK355_E_POLY_DEGREE_TOO_LARGE) throw ark_relations::SynthesisError(ark_relations::SynthesisErrorKind::PolynomialDegreeTooLarge);
    if (rc == ARK355_E_UNSATISFIABLE) throw ark_relations::SynthesisError(ark_relations::SynthesisErrorKind::Unsatisfiable);
    throw BackendError(rc, ark355_last_error(ctx_));
  }

  // Worker context k (k = 0, 1, ...) of the pipelined prover: same device, own streams and scratch.  Created on first
  // use and kept for the backend's lifetime -- a context's scratch (bucket sets, NTT ping-pong buffers, events) is
  // gigabytes of hipMalloc, far too slow to repeat per call.  nullptr when the device refuses another context.
  ark355_ctx* worker(size_t k) const {
    std::lock_guard<std::mutex> lk(mu_);
    while (workers_.size() <= k) {
      ark355_ctx* c = nullptr;
      if (ark355_ctx_create(device_, &c) != ARK355_OK) return nullptr;
      workers_.push_back(c);
    }
    return workers_[k];
  }
  // Page-locked host buffers (ark355_host_alloc), recycled across calls: pinning 32 MB costs milliseconds.
  std::pair<void*, size_t> take_pinned(size_t bytes) const {
    {
      std::lock_guard<std::mutex> lk(mu_);
      for (size_t k = 0; k < pinned_.size(); k++)
        if (pinned_[k].second >= bytes) {
          auto b = pinned_[k];
          pinned_.erase(pinned_.begin() + k);
          return b;
        }
    }
    void* raw = nullptr;
    if (ark355_host_alloc(bytes, &raw) != ARK355_OK) throw std::bad_alloc();
    return {raw, bytes};
  }
  void give_pinned(std::pair<void*, size_t> b) const {
    std::lock_guard<std::mutex> lk(mu_);
    pinned_.push_back(b);
  }

 private:
  ark355_ctx* ctx_ = nullptr;
  int device_ = 0;
  mutable std::mutex mu_;
  mutable std::vector<ark355_ctx*> workers_;
  mutable std::vector<std::pair<void*, size_t>> pinned_;
};

template <class C>
class Groth16 {
 public:
  using Fr = Field<typename C::FrP>;
  using Rng = std::function<Fr()>;                      // uniform field elements (`Fr::rand(rng)`)
  using Circuit = ark_relations::gr1cs::ConstraintSynthesizer<Fr>;
  using CSRef = ark_relations::gr1cs::ConstraintSystemRef<Fr>;
  static constexpr size_t FR = sizeof(typename C::FrP), G1 = 2 * sizeof(typename C::FqP), G2 = 4 * sizeof(typename C::FqP);

  struct VerifyingKey {
    std::vector<uint8_t> alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1;
  };
  struct ProvingKey {
    VerifyingKey vk;
    std::vector<uint8_t> beta_g1, delta_g1, a_query, b_g1_query, b_g2_query, h_query, l_query;
    uint64_t ell = 0, w = 0, n = 0, N = 0;
    // device residency (created on first prove; shared_ptr so that keys stay copyable like the Rust type)
    struct Resident {
      ark355_pk* pk = nullptr;
      ark355_r1cs* r1cs = nullptr;
      ~Resident() {
        ark355_pk_free(pk);
        ark355_r1cs_free(r1cs);
      }
    };
    std::shared_ptr<Resident> resident;
  };
  struct Proof {
    std::vector<uint8_t> a, b, c;      // raw affine points (Montgomery images)
  };

  explicit Groth16(std::shared_ptr<Backend> be) : be_(std::move(be)) {}

  // ---- CSR of the R1CS matrices ---------------------------------------------------------------------------
  struct Csr {
    std::vector<uint64_t> row_ptr[3];
    std::vector<uint32_t> col[3];
    std::vector<uint8_t> coeff[3];
  };
  static Csr to_csr(const std::vector<ark_relations::Matrix<Fr>>& mats) {
    Csr c;
    for (int k = 0; k < 3; k++) {
      c.row_ptr[k].push_back(0);
      for (const auto& row : mats[k]) {
        for (const auto& cj : row) {
          c.col[k].push_back((uint32_t)cj.second);
          const uint8_t* p = reinterpret_cast<const uint8_t*>(cj.first.v.l);
          c.coeff[k].insert(c.coeff[k].end(), p, p + FR);
        }
        c.row_ptr[k].push_back(c.col[k].size());
      }
      if (c.col[k].empty()) c.col[k].push_back(0);
      if (c.coeff[k].empty()) c.coeff[k].resize(FR);
    }
    return c;
  }

  // ---- SNARK::circuit_specific_setup ------------------------------------------------------------------------
  std::pair<ProvingKey, VerifyingKey> circuit_specific_setup(Circuit& circuit, const Rng& rng) const {
    namespace g = ark_relations::gr1cs;
    CSRef cs = CSRef::new_ref();
    cs.set_optimization_goal(g::OptimizationGoal::Constraints);
    cs.set_mode(g::SynthesisMode::setup());
    circuit.generate_constraints(cs);
    cs.finalize();
    auto mats = cs.to_matrices().at(g::R1CS_PREDICATE_LABEL);
    const uint64_t n = cs.num_constraints(), ell = cs.num_instance_variables(), w = cs.num_witness_variables(), m = ell + w;
    uint32_t lg = 0;
    while ((1ull << lg) < n + ell) lg++;
    if (lg > (uint32_t)C::FrParams::TWO_ADICITY)
      throw ark_relations::SynthesisError(ark_relations::SynthesisErrorKind::PolynomialDegreeTooLarge);
    const uint64_t N = 1ull << lg;
    const Fr tau = rng(), alpha = rng(), beta = rng(), gamma = rng(), delta = rng();
    // Lagrange coefficients L_k(tau) = Z(tau)/N * w^k / (tau - w^k)
    Fr root{ark355::fr_from_params_host<typename C::FrP>(&C::FrParams::root)};
    Fr omega = root;
    for (uint32_t i = 0; i < (uint32_t)C::FrParams::TWO_ADICITY - lg; i++) omega = omega * omega;
    Fr zt = tau.pow_u64(N) - Fr::one();
    std::vector<Fr> wk(N), den(N), pref(N);
    Fr cur = Fr::one(), acc = Fr::one();
    for (uint64_t k = 0; k < N; k++) {
      wk[k] = cur;
      den[k] = tau - cur;
      acc = acc * den[k];
      pref[k] = acc;
      cur = cur * omega;
    }
    Fr inv = acc.inverse();
    Fr cN = zt * Fr::from_u64(N).inverse();
    std::vector<Fr> L(N);
    for (uint64_t k = N; k-- > 0;) {
      Fr dinv = inv * (k ? pref[k - 1] : Fr::one());
      inv = inv * den[k];
      L[k] = cN * wk[k] * dinv;
    }
    std::vector<Fr> u(m, Fr::zero()), v(m, Fr::zero()), ww(m, Fr::zero());
    for (uint64_t i = 0; i < ell; i++) u[i] = L[n + i];
    std::vector<Fr>* tgt[3] = {&u, &v, &ww};
    for (int k = 0; k < 3; k++)
      for (uint64_t i = 0; i < n; i++)
        for (const auto& cj : mats[k][i]) (*tgt[k])[cj.second] = (*tgt[k])[cj.second] + L[i] * cj.first;
    const Fr gi = gamma.inverse(), di = delta.inverse();
    std::vector<Fr> abc(m), gabc(ell), ls(w), hs(N ? N - 1 : 0);
    for (uint64_t i = 0; i < m; i++) abc[i] = beta * u[i] + alpha * v[i] + ww[i];
    for (uint64_t i = 0; i < ell; i++) gabc[i] = abc[i] * gi;
    for (uint64_t i = 0; i < w; i++) ls[i] = abc[ell + i] * di;
    Fr t = zt * di;
    for (uint64_t i = 0; i + 1 < N; i++) {
      hs[i] = t;
      t = t * tau;
    }
    ProvingKey pk;
    pk.ell = ell;
    pk.w = w;
    pk.n = n;
    pk.N = N;
    auto g1 = g1_generator(), g2 = g2_generator();
    auto one1 = fixed_base(1, g1, {alpha, beta, delta});
    auto one2 = fixed_base(2, g2, {beta, gamma, delta});
    pk.vk.alpha_g1.assign(one1.begin(), one1.begin() + G1);
    pk.beta_g1.assign(one1.begin() + G1, one1.begin() + 2 * G1);
    pk.delta_g1.assign(one1.begin() + 2 * G1, one1.end());
    pk.vk.beta_g2.assign(one2.begin(), one2.begin() + G2);
    pk.vk.gamma_g2.assign(one2.begin() + G2, one2.begin() + 2 * G2);
    pk.vk.delta_g2.assign(one2.begin() + 2 * G2, one2.end());
    pk.vk.gamma_abc_g1 = fixed_base(1, g1, gabc);
    pk.a_query = fixed_base(1, g1, u);
    pk.b_g1_query = fixed_base(1, g1, v);
    pk.b_g2_query = fixed_base(2, g2, v);
    pk.h_query = fixed_base(1, g1, hs);
    pk.l_query = fixed_base(1, g1, ls);
    return {pk, pk.vk};
  }

  // ---- SNARK::prove -----------------------------------------------------------------------------------------------
  Proof prove(ProvingKey& pk, Circuit& circuit, const Rng& rng) const {
    namespace g = ark_relations::gr1cs;
    CSRef cs = CSRef::new_ref();
    cs.set_optimization_goal(g::OptimizationGoal::Constraints);
    const bool resident = pk.resident && pk.resident->r1cs;
    if (resident) cs.set_mode(g::SynthesisMode::prove(false, false));      // witness-only synthesis
    circuit.generate_constraints(cs);
    cs.finalize();
    if (!resident) load(pk, cs);
    std::vector<Fr> z = cs.borrow().full_assignment();
    const Fr r = rng(), s = rng();
    return create_proof_with_assignment(pk, z, r, s);
  }

  // upstream `create_proof_with_reduction_and_matrices` counterpart: explicit assignment and randomisers
  Proof create_proof_with_assignment(ProvingKey& pk, const std::vector<Fr>& z, const Fr& r, const Fr& s) const {
    if (!pk.resident || !pk.resident->pk || !pk.resident->r1cs) throw std::logic_error("proving key is not resident");
    uint8_t rc[32] = {0}, sc[32] = {0};
    r.to_canonical_bytes(rc);
    s.to_canonical_bytes(sc);
    ark355_proof_raw raw;
    be_->check(ark355_prove(be_->ctx(), pk.resident->pk, pk.resident->r1cs, reinterpret_cast<const uint8_t*>(z.data()),
                            z.size(), rc, sc, &raw));
    Proof p;
    p.a.assign(raw.a, raw.a + G1);
    p.b.assign(raw.b, raw.b + G2);
    p.c.assign(raw.c, raw.c + G1);
    return p;
  }

  // Many proofs of one circuit (ark355_prove_batch): each circuit instance is synthesised on the calling thread (the
  // constraint system is single-threaded by construction, constraint_system_ref.rs:33), then all assignments are
  // proved with up to `inflight` proofs sharing the GPU.  Randomisers are drawn as r_0, s_0, r_1, s_1, ...
  std::vector<Proof> prove_batch(ProvingKey& pk, const std::vector<Circuit*>& circuits, const Rng& rng,
                                 uint32_t inflight = 3, double* device_seconds = nullptr) const {
    namespace g = ark_relations::gr1cs;
    std::vector<std::vector<Fr>> zs;
    std::vector<uint8_t> rc(32 * circuits.size(), 0), sc(32 * circuits.size(), 0);
    for (size_t i = 0; i < circuits.size(); i++) {
      CSRef cs = CSRef::new_ref();
      cs.set_optimization_goal(g::OptimizationGoal::Constraints);
      const bool resident = pk.resident && pk.resident->r1cs;
      if (resident) cs.set_mode(g::SynthesisMode::prove(false, false));
      circuits[i]->generate_constraints(cs);
      cs.finalize();
      if (!resident) load(pk, cs);
      zs.push_back(cs.borrow().full_assignment());
      rng().to_canonical_bytes(&rc[32 * i]);
      rng().to_canonical_bytes(&sc[32 * i]);
    }
    std::vector<const uint8_t*> zp;
    uint64_t z_len = zs.empty() ? 0 : zs[0].size();
    for (const auto& z : zs) {
      zp.push_back(reinterpret_cast<const uint8_t*>(z.data()));
      if (z.size() < z_len) z_len = z.size();
    }
    std::vector<ark355_proof_raw> raw(circuits.size());
    if (circuits.empty()) return {};
    const auto t_dev = std::chrono::steady_clock::now();
    be_->check(ark355_prove_batch(be_->ctx(), pk.resident->pk, pk.resident->r1cs, zp.data(), z_len, rc.data(), sc.data(),
                                  circuits.size(), inflight, raw.data()));
    if (device_seconds) *device_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_dev).count();
    std::vector<Proof> out(circuits.size());
    for (size_t i = 0; i < circuits.size(); i++) {
      out[i].a.assign(raw[i].a, raw[i].a + G1);
      out[i].b.assign(raw[i].b, raw[i].b + G2);
      out[i].c.assign(raw[i].c, raw[i].c + G1);
    }
    return out;
  }

  // Proofs for assignments that are already synthesised (z_i = instance || witness, Montgomery images, z_len
  // elements each; page-locked buffers copy at PCIe rate): ark355_prove_batch over the resident key.
  std::vector<Proof> prove_assignments(ProvingKey& pk, const std::vector<const Fr*>& z, uint64_t z_len,
                                       const std::vector<std::pair<Fr, Fr>>& randomisers, uint32_t inflight = 3,
                                       double* device_seconds = nullptr) const {
    if (!(pk.resident && pk.resident->r1cs)) throw std::logic_error("prove_assignments needs a resident key (prove once first)");
    if (randomisers.size() < z.size()) throw std::logic_error("one (r, s) pair per proof");
    if (z.empty()) return {};
    std::vector<uint8_t> rc(32 * z.size(), 0), sc(32 * z.size(), 0);
    std::vector<const uint8_t*> zp;
    for (size_t i = 0; i < z.size(); i++) {
      randomisers[i].first.to_canonical_bytes(&rc[32 * i]);
      randomisers[i].second.to_canonical_bytes(&sc[32 * i]);
      zp.push_back(reinterpret_cast<const uint8_t*>(z[i]));
    }
    std::vector<ark355_proof_raw> raw(z.size());
    const auto t_dev = std::chrono::steady_clock::now();
    be_->check(ark355_prove_batch(be_->ctx(), pk.resident->pk, pk.resident->r1cs, zp.data(), z_len, rc.data(), sc.data(),
                                  z.size(), inflight, raw.data()));
    if (device_seconds) *device_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_dev).count();
    std::vector<Proof> out(z.size());
    for (size_t i = 0; i < z.size(); i++) {
      out[i].a.assign(raw[i].a, raw[i].a + G1);
      out[i].b.assign(raw[i].b, raw[i].b + G2);
      out[i].c.assign(raw[i].c, raw[i].c + G1);
    }
    return out;
  }

  // End-to-end SNARK::prove for MANY instances of one circuit, synthesis included and overlapped with the GPU:
  // the reference's parallel unit is one OS thread per proof because ConstraintSystemRef is Rc<RefCell<..>>
  // (relations/src/gr1cs/constraint_system_ref.rs:33), so `synth_threads` host threads each run
  // generate_constraints on their OWN constraint system (witness-only mode once the matrices are resident,
  // constraint_system_ref.rs:241-243) and hand the finished assignment to one of `inflight` prover threads, each with
  // its own device context over the shared resident key (the same arrangement ark355_prove_batch uses inside the
  // library).  A bounded queue keeps at most 2 * inflight assignments waiting.  make_circuit(i) builds instance i on
  // the synthesis thread; randomisers[i] = (r_i, s_i).  Returns the proofs in order; stats (optional) receives the
  // wall time and the summed synthesis / device times.
  struct PipelineStats {
    double wall_s = 0, synth_s = 0, prove_s = 0;
  };
  std::vector<Proof> prove_pipelined(ProvingKey& pk, size_t count,
                                     const std::function<std::unique_ptr<Circuit>(size_t)>& make_circuit,
                                     const std::vector<std::pair<Fr, Fr>>& randomisers, uint32_t synth_threads,
                                     uint32_t inflight, PipelineStats* stats = nullptr) const {
    namespace g = ark_relations::gr1cs;
    if (count == 0) return {};
    if (randomisers.size() < count) throw std::logic_error("one (r, s) pair per proof");
    if (!(pk.resident && pk.resident->r1cs)) {          // first instance: matrices + key upload (sequential)
      CSRef cs = CSRef::new_ref();
      cs.set_optimization_goal(g::OptimizationGoal::Constraints);
      auto c0 = make_circuit(0);
      c0->generate_constraints(cs);
      cs.finalize();
      load(pk, cs);
    }
    if (synth_threads < 1) synth_threads = 1;
    if (inflight < 1) inflight = 1;
    // assignments travel in page-locked buffers (ark355_host_alloc, recycled by the backend): the H2D copy inside
    // ark355_prove then runs at PCIe rate instead of being staged through the runtime's bounce buffers
    struct Pinned {
      Fr* p = nullptr;
      size_t n = 0, cap = 0;        // elements in use / bytes owned
    };
    struct Item {
      size_t index;
      Pinned z;
    };
    auto take_buf = [&](size_t n) {
      auto b = be_->take_pinned(n * sizeof(Fr));
      return Pinned{static_cast<Fr*>(b.first), n, b.second};
    };
    auto give_buf = [&](Pinned b) { be_->give_pinned({b.p, b.cap}); };
    std::deque<Item> queue;
    std::mutex mu;
    std::condition_variable cv_not_empty, cv_not_full;
    const size_t cap = 2 * (size_t)inflight;
    std::atomic<size_t> next{0};
    std::atomic<uint32_t> producers_left{synth_threads};
    std::vector<Proof> out(count);
    std::exception_ptr first_error;
    std::atomic<bool> failed{false};
    std::atomic<uint64_t> synth_ns{0}, prove_ns{0};
    auto fail = [&](std::exception_ptr e) {
      std::lock_guard<std::mutex> lk(mu);
      if (!first_error) first_error = e;
      failed = true;
      cv_not_empty.notify_all();
      cv_not_full.notify_all();
    };
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto producer = [&] {
      try {
        for (;;) {
          const size_t i = next.fetch_add(1);
          if (i >= count || failed) break;
          const auto t0 = now();
          CSRef cs = CSRef::new_ref();
          cs.set_optimization_goal(g::OptimizationGoal::Constraints);
          cs.set_mode(g::SynthesisMode::prove(false, false));
          auto c = make_circuit(i);
          c->generate_constraints(cs);
          cs.finalize();
          const auto& inner = cs.borrow();
          const size_t zi = inner.instance_assignment.size(), zw = inner.witness_assignment.size();
          Item it{i, take_buf(zi + zw)};
          // the page-locked buffer goes back to the backend's pool unless the queue takes it over (a failing batch must
          // not leak pinned host memory)
          struct Return {
            decltype(give_buf)& give;
            Item* it;
            ~Return() {
              if (it) give(it->z);
            }
          } guard{give_buf, &it};
          std::memcpy(it.z.p, inner.instance_assignment.data(), zi * sizeof(Fr));       // z = instance || witness
          std::memcpy(it.z.p + zi, inner.witness_assignment.data(), zw * sizeof(Fr));
          it.z.n = zi + zw;
          synth_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(now() - t0).count();
          std::unique_lock<std::mutex> lk(mu);
          cv_not_full.wait(lk, [&] { return queue.size() < cap || failed; });
          if (failed) break;
          guard.it = nullptr;
          queue.push_back(std::move(it));
          cv_not_empty.notify_one();
        }
      } catch (...) {
        fail(std::current_exception());
      }
      if (--producers_left == 0) {
        std::lock_guard<std::mutex> lk(mu);
        cv_not_empty.notify_all();
      }
    };
    auto consumer = [&](ark355_ctx* ctx) {
      try {
        for (;;) {
          Item it;
          {
            std::unique_lock<std::mutex> lk(mu);
            cv_not_empty.wait(lk, [&] { return !queue.empty() || producers_left == 0 || failed; });
            if (failed || queue.empty()) break;
            it = std::move(queue.front());
            queue.pop_front();
            cv_not_full.notify_one();
          }
          uint8_t rc[32] = {0}, sc[32] = {0};
          randomisers[it.index].first.to_canonical_bytes(rc);
          randomisers[it.index].second.to_canonical_bytes(sc);
          ark355_proof_raw raw;
          const auto t0 = now();
          const int e = ark355_prove(ctx, pk.resident->pk, pk.resident->r1cs, reinterpret_cast<const uint8_t*>(it.z.p),
                                     it.z.n, rc, sc, &raw);
          prove_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(now() - t0).count();
          give_buf(it.z);
          if (e != ARK355_OK) throw BackendError(e, ark355_last_error(ctx));
          Proof& p = out[it.index];
          p.a.assign(raw.a, raw.a + G1);
          p.b.assign(raw.b, raw.b + G2);
          p.c.assign(raw.c, raw.c + G1);
        }
      } catch (...) {
        fail(std::current_exception());
      }
    };
    // prover contexts: the backend's own plus inflight - 1 of its persistent workers
    std::vector<ark355_ctx*> extra_ctx;
    for (uint32_t k = 1; k < inflight; k++) {
      ark355_ctx* c = be_->worker(k - 1);
      if (!c) break;
      extra_ctx.push_back(c);
    }
    const auto t_start = now();
    std::vector<std::thread> th;
    for (uint32_t k = 0; k < synth_threads; k++) th.emplace_back(producer);
    for (ark355_ctx* c : extra_ctx) th.emplace_back(consumer, c);
    consumer(be_->ctx());
    for (auto& t : th) t.join();
    const double wall = std::chrono::duration<double>(now() - t_start).count();
    for (auto& it : queue) give_buf(it.z);
    if (first_error) std::rethrow_exception(first_error);
    if (stats) {
      stats->wall_s = wall;
      stats->synth_s = synth_ns.load() * 1e-9;
      stats->prove_s = prove_ns.load() * 1e-9;
    }
    return out;
  }

  void load(ProvingKey& pk, const CSRef& cs) const {
    namespace g = ark_relations::gr1cs;
    auto res = std::make_shared<typename ProvingKey::Resident>();
    Csr csr = to_csr(cs.to_matrices().at(g::R1CS_PREDICATE_LABEL));
    const uint64_t* rp[3] = {csr.row_ptr[0].data(), csr.row_ptr[1].data(), csr.row_ptr[2].data()};
    const uint32_t* cl[3] = {csr.col[0].data(), csr.col[1].data(), csr.col[2].data()};
    const uint8_t* cf[3] = {csr.coeff[0].data(), csr.coeff[1].data(), csr.coeff[2].data()};
    be_->check(ark355_r1cs_load(be_->ctx(), C::ID, cs.num_constraints(), cs.num_instance_variables(),
                                cs.num_witness_variables(), rp, cl, cf, &res->r1cs));
    ark355_pk_desc d;
    d.num_instance = pk.ell;
    d.num_witness = pk.w;
    d.domain_size = pk.N;
    d.a_query = pk.a_query.data();
    d.b_g1_query = pk.b_g1_query.data();
    d.b_g2_query = pk.b_g2_query.data();
    d.h_query = pk.h_query.data();
    d.l_query = pk.l_query.data();
    d.alpha_g1 = pk.vk.alpha_g1.data();
    d.beta_g1 = pk.beta_g1.data();
    d.delta_g1 = pk.delta_g1.data();
    d.beta_g2 = pk.vk.beta_g2.data();
    d.delta_g2 = pk.vk.delta_g2.data();
    be_->check(ark355_pk_load(be_->ctx(), C::ID, &d, &res->pk));
    pk.resident = res;
  }

  // ---- SNARK::verify (lib.rs:59-80) and batch verification over the C ABI (ark355_verify_batch) ------------------------
  bool verify(const VerifyingKey& vk, const std::vector<Fr>& public_inputs, const Proof& proof) const {
    return verify_batch(vk, {public_inputs}, {proof}, Rng());
  }
  // every proof of ONE verifying key with a random linear combination; rng yields the coefficients (count > 1)
  bool verify_batch(const VerifyingKey& vk, const std::vector<std::vector<Fr>>& public_inputs,
                    const std::vector<Proof>& proofs, const Rng& rng) const {
    const size_t count = proofs.size(), ell = vk.gamma_abc_g1.size() / G1;
    if (count == 0) return true;
    if (public_inputs.size() != count) return false;
    for (const auto& x : public_inputs)
      if (x.size() + 1 != ell) return false;
    ark355_vk_desc d;
    d.num_instance = ell;
    d.alpha_g1 = vk.alpha_g1.data();
    d.beta_g2 = vk.beta_g2.data();
    d.gamma_g2 = vk.gamma_g2.data();
    d.delta_g2 = vk.delta_g2.data();
    d.gamma_abc_g1 = vk.gamma_abc_g1.data();
    std::vector<ark355_proof_raw> raw(count);
    std::vector<Fr> xs;
    std::vector<uint8_t> rho(32 * count, 0);
    for (size_t j = 0; j < count; j++) {
      memset(&raw[j], 0, sizeof(raw[j]));
      memcpy(raw[j].a, proofs[j].a.data(), G1);
      memcpy(raw[j].b, proofs[j].b.data(), G2);
      memcpy(raw[j].c, proofs[j].c.data(), G1);
      xs.insert(xs.end(), public_inputs[j].begin(), public_inputs[j].end());
      if (count > 1) {
        uint8_t k[32];
        rng().to_canonical_bytes(k);
        memcpy(&rho[32 * j], k, 16);             // 128-bit coefficients
        rho[32 * j] |= 1;
      }
    }
    int32_t ok = 0;
    be_->check(ark355_verify_batch(be_->ctx(), C::ID, &d, raw.data(), reinterpret_cast<const uint8_t*>(xs.data()),
                                   count > 1 ? rho.data() : nullptr, count, &ok));
    return ok == 1;
  }

  static std::vector<uint8_t> g1_generator() {
    using Fq = typename C::FqP;
    std::vector<uint8_t> out(G1);
    Fq x, y;
    for (int i = 0; i < Fq::N; i++) {
      x.l[i] = C::Consts::g1_gen_x(i);
      y.l[i] = C::Consts::g1_gen_y(i);
    }
    memcpy(out.data(), x.l, sizeof(Fq));
    memcpy(out.data() + sizeof(Fq), y.l, sizeof(Fq));
    return out;
  }
  static std::vector<uint8_t> g2_generator() {
    using Fq = typename C::FqP;
    std::vector<uint8_t> out(G2);
    Fq c[4];
    for (int i = 0; i < Fq::N; i++) {
      c[0].l[i] = C::Consts::g2_gen_x0(i);
      c[1].l[i] = C::Consts::g2_gen_x1(i);
      c[2].l[i] = C::Consts::g2_gen_y0(i);
      c[3].l[i] = C::Consts::g2_gen_y1(i);
    }
    for (int k = 0; k < 4; k++) memcpy(out.data() + k * sizeof(Fq), c[k].l, sizeof(Fq));
    return out;
  }

 private:
  std::vector<uint8_t> fixed_base(int group, const std::vector<uint8_t>& base, const std::vector<Fr>& scalars) const {
    const size_t psz = group == 1 ? G1 : G2;
    std::vector<uint8_t> sc(scalars.size() * FR), out(scalars.size() * psz);
    for (size_t i = 0; i < scalars.size(); i++) scalars[i].to_canonical_bytes(sc.data() + i * FR);
    be_->check(ark355_fixed_base_mul(be_->ctx(), C::ID, group, base.data(), sc.data(), scalars.size(), out.data()));
    return out;
  }
  std::shared_ptr<Backend> be_;
};

}  // namespace ark_snark
