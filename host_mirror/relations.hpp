// C++ host-side mirror of the part of `ark_relations::gr1cs` that the Groth16 path consumes.
//
// The reference is Rust (/root/reference/relations/src/gr1cs/) and no Rust toolchain exists in this
// environment, so the host side above the C ABI is C++ with the SAME names, argument meaning and error
// behaviour, for this path only (R1CS predicate).  With a Rust toolchain the
// real `ark-relations` is used unchanged and only INTEGRATION.md's `extern "C"` block is needed.
//
//   Variable                      utils/variable.rs:4-18,52-113   (3-bit tag | 61-bit index, same Ord)
//   LinearCombination<F>          utils/linear_combination.rs:15,53-82,174-212
//   FieldInterner<F>, LcMap<F>    gr1cs/field_interner.rs:16-72, gr1cs/lc_map.rs:52-135 (interned coefficients, flat LC storage)
//   ConstraintSystem<F>           gr1cs/constraint_system.rs:44-97,109-139,...
//   ConstraintSystemRef<F>        gr1cs/constraint_system_ref.rs:26-34 (Rc<RefCell<..>> -> shared_ptr)
//   ConstraintSynthesizer<F>      gr1cs/mod.rs:54-61
//   SynthesisMode / OptimizationGoal   gr1cs/mod.rs:74-106
//   SynthesisError                utils/error.rs:5-21
//   Matrix<F>, mat_vec_mul        utils/matrix.rs:4,26-36
#pragma once
#include <stdint.h>
#include <algorithm>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <utility>
#include <vector>

namespace ark_relations {

enum class SynthesisErrorKind {
  MissingCS,
  AssignmentMissing,
  DivisionByZero,
  Unsatisfiable,
  PolynomialDegreeTooLarge,
  UnexpectedIdentity,
  MalformedVerifyingKey,
  UnconstrainedVariable,
  PredicateNotFound,
  ArityMismatch,
};

struct SynthesisError : std::runtime_error {
  SynthesisErrorKind kind;
  explicit SynthesisError(SynthesisErrorKind k) : std::runtime_error(name(k)), kind(k) {}
  static const char* name(SynthesisErrorKind k) {
    switch (k) {
      case SynthesisErrorKind::MissingCS: return "MissingCS";
      case SynthesisErrorKind::AssignmentMissing: return "AssignmentMissing";
      case SynthesisErrorKind::DivisionByZero: return "DivisionByZero";
      case SynthesisErrorKind::Unsatisfiable: return "Unsatisfiable";
      case SynthesisErrorKind::PolynomialDegreeTooLarge: return "PolynomialDegreeTooLarge";
      case SynthesisErrorKind::UnexpectedIdentity: return "UnexpectedIdentity";
      case SynthesisErrorKind::MalformedVerifyingKey: return "MalformedVerifyingKey";
      case SynthesisErrorKind::UnconstrainedVariable: return "UnconstrainedVariable";
      case SynthesisErrorKind::PredicateNotFound: return "PredicateNotFound";
      case SynthesisErrorKind::ArityMismatch: return "ArityMismatch";
    }
    return "?";
  }
};

// ---- Variable: [ tag: 3 bits | payload: 61 bits ]  (utils/variable.rs:4-18) -------------------------------
class Variable {
 public:
  enum Kind : uint8_t { KZero = 0, KOne = 1, KInstance = 2, KWitness = 3, KSymbolicLc = 4 };
  static constexpr uint64_t TAG_SHIFT = 61;
  static constexpr uint64_t PAYLOAD_MASK = (1ull << TAG_SHIFT) - 1;
  constexpr Variable() : v_(0) {}
  static constexpr Variable Zero() { return Variable(0, 0); }
  static constexpr Variable One() { return Variable(KOne, 0); }
  static constexpr Variable instance(size_t i) { return Variable(KInstance, i); }
  static constexpr Variable witness(size_t i) { return Variable(KWitness, i); }
  static constexpr Variable symbolic_lc(size_t i) { return Variable(KSymbolicLc, i); }
  constexpr Kind kind() const { return (Kind)(v_ >> TAG_SHIFT); }
  constexpr uint64_t payload() const { return v_ & PAYLOAD_MASK; }
  constexpr bool is_zero() const { return v_ == 0; }
  constexpr bool is_one() const { return kind() == KOne; }
  constexpr bool is_instance() const { return kind() == KInstance; }
  constexpr bool is_witness() const { return kind() == KWitness; }
  constexpr bool is_lc() const { return kind() == KSymbolicLc; }
  // utils/variable.rs:105-113
  bool get_variable_index(size_t witness_offset, size_t* out) const {
    switch (kind()) {
      case KOne: *out = 0; return true;
      case KInstance: *out = (size_t)payload(); return true;
      case KWitness: *out = (size_t)payload() + witness_offset; return true;
      default: return false;
    }
  }
  bool get_lc_index(size_t* out) const {
    if (!is_lc()) return false;
    *out = (size_t)payload();
    return true;
  }
  // derived Ord on the packed u64: Zero < One < Instance < Witness < SymbolicLc, then by index
  constexpr bool operator<(const Variable& o) const { return v_ < o.v_; }
  constexpr bool operator==(const Variable& o) const { return v_ == o.v_; }
  constexpr bool operator!=(const Variable& o) const { return v_ != o.v_; }
  constexpr bool operator>=(const Variable& o) const { return v_ >= o.v_; }
  constexpr uint64_t raw() const { return v_; }

 private:
  constexpr Variable(uint64_t tag, uint64_t idx) : v_((tag << TAG_SHIFT) | (idx & PAYLOAD_MASK)) {}
  uint64_t v_;
};

// ---- LinearCombination<F>(Vec<(F, Variable)>) -------------------------------------------------------------------
template <class F>
class LinearCombination {
 public:
  std::vector<std::pair<F, Variable>> terms;
  LinearCombination() = default;
  static LinearCombination zero() { return LinearCombination(); }
  // lc![a, b, ...]  (utils/linear_combination.rs:28)
  static LinearCombination sum_vars(std::initializer_list<Variable> vs) {
    LinearCombination l;
    for (auto v : vs) l.terms.emplace_back(F::one(), v);
    return l;
  }
  size_t len() const { return terms.size(); }
  // utils/linear_combination.rs:174-191: below 6 terms the scan never reports a hit
  bool get_var_loc(const Variable& var, size_t* idx) const {
    if (terms.size() < 6) {
      size_t found = 0;
      for (size_t i = 0; i < terms.size(); i++) {
        if (terms[i].second >= var) {
          found = i;
          break;
        }
        found += 1;
      }
      *idx = found;
      return false;
    }
    size_t lo = 0, hi = terms.size();
    while (lo < hi) {
      size_t mid = (lo + hi) / 2;
      if (terms[mid].second < var) lo = mid + 1;
      else if (var < terms[mid].second) hi = mid;
      else {
        *idx = mid;
        return true;
      }
    }
    *idx = lo;
    return false;
  }
  // AddAssign<(F, Variable)>  (:204-212)
  LinearCombination& operator+=(const std::pair<F, Variable>& cv) {
    size_t i;
    if (get_var_loc(cv.second, &i)) terms[i].first = terms[i].first + cv.first;
    else terms.insert(terms.begin() + i, cv);
    return *this;
  }
  LinearCombination operator+(const std::pair<F, Variable>& cv) const& {
    LinearCombination r = *this;
    r += cv;
    return r;
  }
  // the reference's Add takes self by value (utils/linear_combination.rs:194-202): `lc = lc + x` moves, it never copies
  // the terms.  Here that is `lc = std::move(lc) + x` (or `lc += x`).
  LinearCombination operator+(const std::pair<F, Variable>& cv) && {
    *this += cv;
    return std::move(*this);
  }
  LinearCombination& operator+=(Variable v) { return *this += std::make_pair(F::one(), v); }
  LinearCombination operator+(Variable v) const& { return *this + std::make_pair(F::one(), v); }
  LinearCombination operator+(Variable v) && { return std::move(*this) + std::make_pair(F::one(), v); }
  LinearCombination operator-(const std::pair<F, Variable>& cv) const& { return *this + std::make_pair(-cv.first, cv.second); }
  LinearCombination operator-(const std::pair<F, Variable>& cv) && { return std::move(*this) + std::make_pair(-cv.first, cv.second); }
  LinearCombination operator-(Variable v) const& { return *this - std::make_pair(F::one(), v); }
  LinearCombination operator-(Variable v) && { return std::move(*this) - std::make_pair(F::one(), v); }
  // compactify (:53-82): sort by Variable, merge equal keys
  void compactify() {
    if (terms.size() <= 1) return;
    std::sort(terms.begin(), terms.end(), [](const auto& a, const auto& b) { return a.second < b.second; });
    size_t w = 0;
    for (size_t r = 1; r < terms.size(); r++) {
      if (terms[w].second == terms[r].second) terms[w].first = terms[w].first + terms[r].first;
      else terms[++w] = terms[r];
    }
    terms.resize(w + 1);
  }
};

template <class F>
inline LinearCombination<F> lc() { return LinearCombination<F>(); }

template <class F>
using Matrix = std::vector<std::vector<std::pair<F, size_t>>>;   // utils/matrix.rs:4

// utils/matrix.rs:26-36
template <class F>
std::vector<F> mat_vec_mul(const Matrix<F>& m, const std::vector<F>& z) {
  std::vector<F> out;
  out.reserve(m.size());
  for (const auto& row : m) {
    F acc = F::zero();
    for (const auto& cj : row) acc = acc + cj.first * z[cj.second];
    out.push_back(acc);
  }
  return out;
}

namespace gr1cs {

static const char* const R1CS_PREDICATE_LABEL = "R1CS";   // predicate/polynomial_constraint.rs:69

// ---- FieldInterner<F>  (gr1cs/field_interner.rs:16-72) --------------------------------------------------------------
// Coefficients are stored once and referred to by a 32-bit id; One and -One are interned up front (ids 0 and 1) and One
// never goes through the map -- almost every coefficient of a real circuit is one of the two.
template <class F>
class FieldInterner {
  static_assert(std::is_trivially_copyable<F>::value && sizeof(F) % 8 == 0, "the interner hashes the limbs of F");

 public:
  FieldInterner() {
    intern(F::one());
    intern(-F::one());
  }
  uint32_t get_or_intern(const F& value) {
    if (value == vec_[0]) return 0;
    auto it = map_.find(value);
    return it != map_.end() ? it->second : intern(value);
  }
  const F& value(uint32_t id) const { return vec_[id]; }       // ids only ever come from get_or_intern
  size_t len() const { return vec_.size(); }

 private:
  struct Hash {
    size_t operator()(const F& f) const {
      uint64_t w[sizeof(F) / 8];
      std::memcpy(w, &f, sizeof(F));
      uint64_t h = 0x9e3779b97f4a7c15ull;
      for (uint64_t x : w) {
        h ^= x;
        h *= 0xff51afd7ed558ccdull;
        h ^= h >> 32;
      }
      return (size_t)h;
    }
  };
  uint32_t intern(const F& value) {
    const uint32_t id = (uint32_t)vec_.size();
    map_.emplace(value, id);
    vec_.push_back(value);
    return id;
  }
  std::unordered_map<F, uint32_t, Hash> map_;
  std::vector<F> vec_;
};

// ---- LcMap<F>  (gr1cs/lc_map.rs:52-135) -----------------------------------------------------------------------------
// Every linear combination of the constraint system in ONE pair of flat arrays (variables, interned coefficients) with
// an offsets array: LC i occupies [offsets[i], offsets[i + 1]).  12 bytes per term and no allocation per LC, against a
// heap vector of 40-byte (F, Variable) pairs each.
template <class F>
class LcMap {
 public:
  LcMap() : offsets_{0} {}
  void reserve(size_t lcs, size_t terms) {
    vars_.reserve(terms);
    coeffs_.reserve(terms);
    offsets_.reserve(lcs + 1);
  }
  // push (lc_map.rs:92-109): appends one LC
  template <class It>
  void push(It first, It last, FieldInterner<F>& interner) {
    for (; first != last; ++first) {
      coeffs_.push_back(interner.get_or_intern(first->first));
      vars_.push_back(first->second);
    }
    offsets_.push_back(vars_.size());
  }
  size_t num_lcs() const { return offsets_.size() - 1; }
  size_t total_lc_size() const { return vars_.size(); }
  // term range of LC i (bounds-checked like the reference's `get`)
  size_t begin(size_t i) const { return offsets_.at(i); }
  size_t end(size_t i) const { return offsets_.at(i + 1); }
  Variable var(size_t k) const { return vars_[k]; }
  uint32_t coeff(size_t k) const { return coeffs_[k]; }
  bool any_lc_var() const {                                       // any_lcs_used (constraint_system.rs:762-764)
    for (const Variable& v : vars_)
      if (v.is_lc()) return true;
    return false;
  }

 private:
  std::vector<Variable> vars_;
  std::vector<uint32_t> coeffs_;
  std::vector<size_t> offsets_;
};

// gr1cs/mod.rs:74-90
struct SynthesisMode {
  enum Tag { Setup, Prove } tag = Prove;
  bool construct_matrices = true;
  bool generate_lc_assignments = true;
  static SynthesisMode setup() { return SynthesisMode{Setup, true, false}; }
  static SynthesisMode prove(bool construct_matrices, bool generate_lc_assignments) {
    return SynthesisMode{Prove, construct_matrices, generate_lc_assignments};
  }
};
enum class OptimizationGoal { None, Constraints, Weight };   // gr1cs/mod.rs:95-106

template <class F>
class ConstraintSystem {
 public:
  using LC = LinearCombination<F>;
  // constraint_system.rs:109-139
  ConstraintSystem() {
    instance_assignment.push_back(F::one());
    lc_assignment.push_back(F::zero());
    lc_map.push(static_cast<const std::pair<F, Variable>*>(nullptr), static_cast<const std::pair<F, Variable>*>(nullptr),
                field_interner);               // the zero LC (:111)
    r1cs_args.resize(3);                       // R1CS predicate registered by default (:136-137)
  }

  void set_mode(SynthesisMode m) { mode = m; }                                  // :535-537
  bool is_in_setup_mode() const { return mode.tag == SynthesisMode::Setup; }
  bool should_construct_matrices() const { return is_in_setup_mode() || mode.construct_matrices; }
  bool should_generate_lc_assignments() const { return !is_in_setup_mode() && mode.generate_lc_assignments; }
  void set_optimization_goal(OptimizationGoal g) {                               // :563-566
    if (!is_new()) throw std::logic_error("set_optimization_goal: constraint system is not new");
    optimization_goal = g;
  }
  bool is_new() const { return num_instance_variables == 1 && num_witness_variables == 0 && num_r1cs_constraints == 0 && num_linear_combinations == 1; }

  // :591-617
  // (the value / LC closures are template parameters, as the reference's `FnOnce() -> ..` generics are: no type erasure,
  // no allocation for a closure that captures more than two pointers)
  template <class Fn>
  Variable new_input_variable(Fn&& f) {
    size_t i = num_instance_variables++;
    if (!is_in_setup_mode()) instance_assignment.push_back(f());
    return Variable::instance(i);
  }
  template <class Fn>
  Variable new_witness_variable(Fn&& f) {
    size_t i = num_witness_variables++;
    if (!is_in_setup_mode()) witness_assignment.push_back(f());
    return Variable::witness(i);
  }

  // :523-532
  template <class Fn>
  Variable new_lc(Fn&& f) { return new_lc_helper(f); }

  // :431-438 -> :323-353
  template <class FA, class FB, class FC>
  void enforce_r1cs_constraint(FA&& a, FB&& b, FC&& c) {
    if (should_construct_matrices()) {
      Variable va = new_constraint_lc(a), vb = new_constraint_lc(b), vc = new_constraint_lc(c);
      r1cs_args[0].push_back(va);
      r1cs_args[1].push_back(vb);
      r1cs_args[2].push_back(vc);
      num_r1cs_constraints++;
    }
  }

  size_t num_constraints() const { return num_r1cs_constraints; }              // :210-215 (R1CS only here)

  // :691-707 (instance outlining is not used by Groth16)
  void finalize() { inline_all_lcs(); }

  // :717-758
  void inline_all_lcs() {
    if (!should_construct_matrices()) return;
    if (!lc_map.any_lc_var()) return;
    LcMap<F> inlined;
    inlined.reserve(lc_map.num_lcs(), lc_map.total_lc_size());
    LC out;
    out.terms.reserve(10);
    const F zero = F::zero();
    for (size_t i = 0; i < lc_map.num_lcs(); i++) {
      for (size_t k = lc_map.begin(i); k < lc_map.end(i); k++) {
        const uint32_t cid = lc_map.coeff(k);
        const Variable var = lc_map.var(k);
        size_t idx;
        if (var.get_lc_index(&idx)) {
          // already transformed: LCs only refer to earlier ones
          const size_t b = inlined.begin(idx), e = inlined.end(idx);
          if (cid == 0) {                                        // coefficient One
            for (size_t j = b; j < e; j++) out.terms.emplace_back(field_interner.value(inlined.coeff(j)), inlined.var(j));
          } else {
            const F coeff = field_interner.value(cid);
            for (size_t j = b; j < e; j++) {
              const F& c2 = field_interner.value(inlined.coeff(j));
              const Variable v2 = inlined.var(j);
              if (!v2.is_zero() && !(c2 == zero)) out.terms.emplace_back(coeff * c2, v2);
            }
          }
        } else {
          out.terms.emplace_back(field_interner.value(cid), var);
        }
      }
      out.compactify();
      inlined.push(out.terms.begin(), out.terms.end(), field_interner);
      out.terms.clear();
    }
    lc_map = std::move(inlined);
  }

  // :777-788
  LC get_lc(Variable v) const {
    LC l;
    if (v.is_zero()) return l;
    size_t idx;
    if (v.get_lc_index(&idx)) {
      const size_t b = lc_map.begin(idx), e = lc_map.end(idx);
      l.terms.reserve(e - b);
      for (size_t k = b; k < e; k++) l.terms.emplace_back(field_interner.value(lc_map.coeff(k)), lc_map.var(k));
    } else {
      l.terms.emplace_back(F::one(), v);
    }
    return l;
  }
  // :792-804
  std::vector<std::pair<F, size_t>> make_row(const LC& l) const {
    std::vector<std::pair<F, size_t>> row;
    for (const auto& cv : l.terms) {
      if (cv.first == F::zero() || cv.second.is_zero()) continue;
      size_t idx;
      if (!cv.second.get_variable_index(num_instance_variables, &idx))
        throw std::logic_error("make_row: un-inlined symbolic LC (the reference panics here, constraint_system.rs:800)");
      row.emplace_back(cv.first, idx);
    }
    return row;
  }
  // :768-774 -> predicate/mod.rs:207-217.  BTreeMap<Label, Vec<Matrix<F>>> with the single label "R1CS".
  std::map<std::string, std::vector<Matrix<F>>> to_matrices() const {
    std::map<std::string, std::vector<Matrix<F>>> out;
    std::vector<Matrix<F>>& mats = out[R1CS_PREDICATE_LABEL];
    mats.resize(3);
    for (int k = 0; k < 3; k++) mats[k].reserve(num_r1cs_constraints);
    for (size_t i = 0; i < num_r1cs_constraints; i++)
      for (int k = 0; k < 3; k++) mats[k].push_back(make_row_of(r1cs_args[k][i]));
    return out;
  }
  // make_row(get_lc(v)) without materialising the intermediate LinearCombination (same rows, same order)
  std::vector<std::pair<F, size_t>> make_row_of(Variable v) const {
    std::vector<std::pair<F, size_t>> row;
    if (v.is_zero()) return row;
    size_t lc_idx;
    auto push = [&](const F& c, const Variable& var) {
      if (c == F::zero() || var.is_zero()) return;
      size_t idx;
      if (!var.get_variable_index(num_instance_variables, &idx))
        throw std::logic_error("make_row: un-inlined symbolic LC (the reference panics here, constraint_system.rs:800)");
      row.emplace_back(c, idx);
    };
    if (v.get_lc_index(&lc_idx)) {
      const size_t b = lc_map.begin(lc_idx), e = lc_map.end(lc_idx);
      row.reserve(e - b);
      for (size_t k = b; k < e; k++) push(field_interner.value(lc_map.coeff(k)), lc_map.var(k));
    } else {
      push(F::one(), v);
    }
    return row;
  }

  // assignment.rs:26-35
  bool assigned_value(Variable v, F* out) const {
    switch (v.kind()) {
      case Variable::KZero: *out = F::zero(); return true;
      case Variable::KOne: *out = F::one(); return true;
      case Variable::KInstance: if (v.payload() < instance_assignment.size()) { *out = instance_assignment[v.payload()]; return true; } return false;
      case Variable::KWitness: if (v.payload() < witness_assignment.size()) { *out = witness_assignment[v.payload()]; return true; } return false;
      case Variable::KSymbolicLc: if (v.payload() < lc_assignment.size()) { *out = lc_assignment[v.payload()]; return true; } return false;
    }
    return false;
  }

  // :661-687 -> predicate/mod.rs:185-204: "R1CS - i" of the first failing constraint, or empty
  std::string which_is_unsatisfied() const {
    if (is_in_setup_mode()) throw SynthesisError(SynthesisErrorKind::AssignmentMissing);
    for (size_t i = 0; i < num_r1cs_constraints; i++) {
      F v[3];
      for (int k = 0; k < 3; k++) {
        if (!assigned_value(r1cs_args[k][i], &v[k])) {
          F acc = F::zero();
          for (const auto& cv : get_lc(r1cs_args[k][i]).terms) {
            F x;
            if (!assigned_value(cv.second, &x)) throw std::logic_error("variable is not assigned; did you run finalize()?");
            acc = acc + cv.first * x;
          }
          v[k] = acc;
        }
      }
      if (!(v[0] * v[1] == v[2])) return std::string(R1CS_PREDICATE_LABEL) + " - " + std::to_string(i);
    }
    return "";
  }
  bool is_satisfied() const { return which_is_unsatisfied().empty(); }

  // the prover's view (:193-206)
  std::vector<F> full_assignment() const {
    std::vector<F> z = instance_assignment;
    z.insert(z.end(), witness_assignment.begin(), witness_assignment.end());
    return z;
  }

  size_t num_instance_variables = 1;
  size_t num_witness_variables = 0;
  size_t num_linear_combinations = 1;
  size_t num_r1cs_constraints = 0;
  std::vector<F> instance_assignment, witness_assignment, lc_assignment;
  SynthesisMode mode;
  OptimizationGoal optimization_goal = OptimizationGoal::None;

 private:
  // :455-461
  template <class Fn>
  Variable new_constraint_lc(Fn&& f) {
    if (should_construct_matrices()) return new_lc_helper(f);
    return Variable::symbolic_lc(num_linear_combinations++);
  }
  // :503-519
  template <class Fn>
  Variable new_lc_helper(Fn&& f) {
    if (should_construct_matrices() || should_generate_lc_assignments()) return new_lc_add_helper(f());
    return Variable::symbolic_lc(num_linear_combinations++);
  }
  // :472-499 (the closure's LC is consumed: its terms go into the flat map, coefficients interned)
  Variable new_lc_add_helper(LC&& l) {
    if (l.terms.empty() || (l.terms.size() == 1 && l.terms[0].second.is_zero())) return Variable::symbolic_lc(0);
    if (l.terms.size() == 1 && l.terms[0].first == F::one()) return l.terms[0].second;
    size_t index = num_linear_combinations++;
    lc_map.push(l.terms.begin(), l.terms.end(), field_interner);             // LcMap::push (lc_map.rs:92-109)
    const auto& stored = l.terms;
    if (should_generate_lc_assignments()) {
      F acc = F::zero();
      for (const auto& cv : stored) {
        F x;
        if (!assigned_value(cv.second, &x)) throw SynthesisError(SynthesisErrorKind::AssignmentMissing);
        acc = acc + cv.first * x;
      }
      lc_assignment.push_back(acc);
    }
    return Variable::symbolic_lc(index);
  }

  LcMap<F> lc_map;
  FieldInterner<F> field_interner;
  std::vector<std::vector<Variable>> r1cs_args;   // column-wise argument_lcs (predicate/mod.rs:81-94)
};

// enum ConstraintSystemRef { None, CS(Rc<RefCell<ConstraintSystem>>) }  (constraint_system_ref.rs:26-34)
template <class F>
class ConstraintSystemRef {
 public:
  using LC = LinearCombination<F>;
  ConstraintSystemRef() = default;                                    // None
  explicit ConstraintSystemRef(std::shared_ptr<ConstraintSystem<F>> p) : p_(std::move(p)) {}
  static ConstraintSystemRef new_ref() { return ConstraintSystemRef(std::make_shared<ConstraintSystem<F>>()); }   // :142-144
  bool is_none() const { return !p_; }
  ConstraintSystem<F>& borrow() const {
    if (!p_) throw SynthesisError(SynthesisErrorKind::MissingCS);
    return *p_;
  }
  void set_mode(SynthesisMode m) const { borrow().set_mode(m); }
  void set_optimization_goal(OptimizationGoal g) const { borrow().set_optimization_goal(g); }
  template <class Fn>
  Variable new_input_variable(Fn&& f) const { return borrow().new_input_variable(f); }
  template <class Fn>
  Variable new_witness_variable(Fn&& f) const { return borrow().new_witness_variable(f); }
  template <class Fn>
  Variable new_lc(Fn&& f) const { return borrow().new_lc(f); }
  // constraint_system_ref.rs:235-250: returns Ok(()) without recording when matrices are off (:241-243)
  template <class FA, class FB, class FC>
  void enforce_r1cs_constraint(FA&& a, FB&& b, FC&& c) const {
    auto& cs = borrow();
    if (!cs.should_construct_matrices()) return;
    cs.enforce_r1cs_constraint(a, b, c);
  }
  // constraint_system_ref.rs `assigned_value`: None when the cs is None or the variable has no assignment
  bool assigned_value(Variable v, F* out) const { return p_ ? p_->assigned_value(v, out) : false; }
  F assigned_value(Variable v) const {                  // the `.unwrap()` form the reference's examples use
    F x;
    if (!assigned_value(v, &x)) throw SynthesisError(SynthesisErrorKind::AssignmentMissing);
    return x;
  }
  void finalize() const { borrow().finalize(); }
  bool is_satisfied() const { return borrow().is_satisfied(); }
  std::string which_is_unsatisfied() const { return borrow().which_is_unsatisfied(); }
  std::map<std::string, std::vector<Matrix<F>>> to_matrices() const { return borrow().to_matrices(); }
  size_t num_constraints() const { return borrow().num_constraints(); }
  size_t num_instance_variables() const { return borrow().num_instance_variables; }
  size_t num_witness_variables() const { return borrow().num_witness_variables; }

 private:
  std::shared_ptr<ConstraintSystem<F>> p_;
};

// gr1cs/mod.rs:54-61
template <class F>
struct ConstraintSynthesizer {
  virtual ~ConstraintSynthesizer() = default;
  virtual void generate_constraints(ConstraintSystemRef<F> cs) = 0;
};

}  // namespace gr1cs
}  // namespace ark_relations
